// gemm_big.hip -- second-generation tuned bf16 GEMM for gfx950: one wavefront per SIMD, 128 x 128 outputs per wavefront.
//
//   out[M,N] = epilogue(A[M,K] . W[N,K]^T), fp32 accumulate, for the DiT block linears (QKV+RoPE, attention out, FF1+GELU, FF2).
//
// Why a second kernel: the 8-wave kernel of gemm_fast.hip is bound by LDS traffic, not by the matrix pipe -- a 128 x 64 wave
// tile re-reads (128 + 64) rows of fragments for 128 x 64 outputs per K-step.  With 256 threads per workgroup every wave may
// use the whole unified register file of its SIMD (256 arch VGPRs + 256 AccVGPRs): the 128 x 128 fp32 accumulator lives in the
// AccVGPRs, the fragments of the *next* K-step are read into a second register buffer while the 64 MFMAs of the current one
// issue (software pipelining inside one instruction stream instead of across two waves), LDS fragment traffic per FLOP drops
// by a third and there is one s_barrier per K-step instead of two.
//
//   * 256 x 256 output tile per workgroup, 4 waves as 2 x 2, v_mfma_f32_16x16x32_bf16 "swapped" (weight rows on the MFMA row
//     index: a lane ends with 4 consecutive features of one token, as in gemm_fast.hip);
//   * operands HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) through a 5-slot ring of 32-deep K-steps (160 KiB), 4 K-steps
//     in flight, counted s_waitcnt vmcnt; same lane-linear 16 rows x 64 B pieces and XOR swizzle as gemm_fast.hip;
//   * XCD-aware tile order (gemm_fast.hip);
//   * lean epilogues only: STORE_T (+GELU-tanh / Mish), GATE_T (per-column gate, row mask as GemmParams::rowbits), ROPE_T.
//     Rows >= M are clamped on load and dropped on store; everything else (N % 256, other epilogues) stays on gemm_fast.hip.
#include "gemm_tile.h"

namespace {

// In-place MFMA with the accumulator pinned to the AccVGPR file.  (Left to the register allocator, a 256-register accumulator
// plus the double-buffered fragments makes LLVM shuttle accumulator tiles between the two files: ~350 v_accvgpr moves per
// pair of K-steps.)  The compiler does not see an MFMA here, so the hazard padding before the first accumulator read is
// explicit (mfma_drain()).
__device__ __forceinline__ void mfma_acc(f32x4& c, const bf16x8& w, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
}
// LDS fragment read as an opaque instruction: it stays exactly where the source puts it (between MFMAs), and it is waited for
// by the explicit lgkmcnt(0) at the end of the K-step -- the compiler does not know that the result arrives later, so nothing
// may touch `dst` before that wait.
template <int OFF> __device__ __forceinline__ void lds_read16(bf16x8& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int EPI, bool RAGGED>
__global__ __launch_bounds__(256, 1) void gemm_big_kernel(GemmParams p, int tiles_n, int nblocks) {
    constexpr int BM = 256, BN = 256, BK = 32, WM = 128, WN = 128, NS = 5, D = NS - 1;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * BK * 2, STAGE = A_BYTES + BN * BK * 2;
    constexpr int PPW = 8;  // DMA pieces (16 rows x 64 B) per wave per K-step: 4 activation + 4 weight
    __shared__ __attribute__((aligned(16))) char smem[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- tile of this workgroup (XCD-aware band + L2 patch order, see gemm_fast.hip)
    int m0, n0;
    {
        const int bid = blockIdx.x;
        const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;
        const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        const int gm = p.tile_group > 0 ? p.tile_group : 1;
        const int tiles_m_all = nblocks / tiles_n;
        const int grp = swz / (gm * tiles_n);
        const int gsz = min(gm, tiles_m_all - grp * gm);
        const int rin = swz - grp * gm * tiles_n;
        const int tile_n = rin / gsz, tile_m = grp * gm + (rin - tile_n * gsz);
        m0 = tile_m * BM;
        n0 = tile_n * BN;
    }
    const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
    const bf16_t* W = reinterpret_cast<const bf16_t*>(p.W);
    const int nk = p.K / BK;
    const int abl = p.ablate;

    // ---- DMA sources: wave w moves rows [64 w, 64 w + 64) of both operand tiles, 4 pieces of 16 rows each
    const int prow = lane >> 2, pchunk = lane & 3;
    const int plc = pchunk ^ ((0 - (prow >> 2)) & 3);  // logical 16-byte chunk kept at this physical slot
    const bf16_t* a_src[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int gm = m0 + wave * 64 + j * 16 + prow;
        if (RAGGED) gm = min(gm, p.M - 1);
        a_src[j] = A + (size_t)gm * p.lda + plc * 8;
    }
    const bf16_t* w_src = W + (size_t)(n0 + wave * 64 + prow) * p.ldw + plc * 8;
    const size_t w_step = (size_t)16 * p.ldw;
    // K-steps are fetched in pairs (kt, kt+1), piece by piece: the two 64-byte halves of every 128-byte line are requested by
    // adjacent instructions, so the second request hits the line the first one brought into the vector L1 instead of fetching the
    // whole line from L2 again a K-step later (the L2 -> L1 over-fetch of half-used lines was the DMA ceiling).
    auto issue_pair = [&](int kt) {
        char* s0 = smem + (kt % NS) * STAGE + wave * 4096;
        char* s1 = smem + ((kt + 1) % NS) * STAGE + wave * 4096;
        const size_t ko = (size_t)kt * BK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dma16(a_src[j] + ko, s0 + j * 1024);
            dma16(a_src[j] + ko + BK, s1 + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            dma16(w_src + j * w_step + ko, s0 + A_BYTES + j * 1024);
            dma16(w_src + j * w_step + ko + BK, s1 + A_BYTES + j * 1024);
        }
    };

    // ---- fragment read offsets
    const int fr = lane & 15, fq = lane >> 4;
    const int c0 = (fq ^ ((0 - (fr >> 2)) & 3)) * 16;
    const int a_off = (wm * WM + fr) * 64 + c0;
    const int w_off = A_BYTES + (wn * WN + fr) * 64 + c0;
    struct Frags {
        bf16x8 w[NI], a[MI];
    };
    auto read_frags = [&](int kt, Frags& f) {
        const char* sb = smem + (kt % NS) * STAGE;
#pragma unroll
        for (int i = 0; i < NI; ++i) f.w[i] = *reinterpret_cast<const bf16x8*>(sb + w_off + i * 1024);
#pragma unroll
        for (int j = 0; j < MI; ++j) f.a[j] = *reinterpret_cast<const bf16x8*>(sb + a_off + j * 1024);
    };

    // ---- epilogue operands per feature tile: fetched during the last K-steps (static code after the steady-state loop), so
    //      they are neither live in the loop nor waited for in the epilogue.  The loads are unconditional (dummy address when an
    //      operand is absent) so that the vmcnt bookkeeping of the last K-steps is static.
    f32x4 bias4[NI];
    [[maybe_unused]] f32x4 gate4[NI];
    [[maybe_unused]] unsigned keepbits = 0xffu;
    constexpr int EPI_LOADS = EPI == EPI_GATE_T ? 2 * NI + 1 : NI;
    auto load_epilogue_operands = [&]() {
        const int n = n0 + wn * WN + 4 * fq;
#pragma unroll
        for (int i = 0; i < NI; ++i) bias4[i] = *reinterpret_cast<const f32x4*>(p.bias + n + i * 16);
        if constexpr (EPI == EPI_GATE_T) {
            const float* g = p.gate ? p.gate : p.bias;
#pragma unroll
            for (int i = 0; i < NI; ++i) gate4[i] = *reinterpret_cast<const f32x4*>(g + n + i * 16);
            const int mw = m0 + wm * WM;  // multiple of 128
            const uint8_t* rb = (p.rowbits && mw < p.M) ? p.rowbits + (mw >> 7) * 16 + fr : reinterpret_cast<const uint8_t*>(p.bias);
            keepbits = *rb;
        }
    };
    auto fix_epilogue_operands = [&]() {
        if constexpr (EPI == EPI_GATE_T) {
            if (!p.gate) {
#pragma unroll
                for (int i = 0; i < NI; ++i) gate4[i] = f32x4{1.f, 1.f, 1.f, 1.f};
            }
            if (!(p.rowbits && m0 + wm * WM < p.M)) keepbits = 0xffu;
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: D K-steps in flight, K-step 0 landed for everybody, its fragments in registers  (nk >= D, nk even)
    issue_pair(0);
    issue_pair(2);
    wait_vm<2 * PPW>();  // pair (2, 3) may still be in flight (and, harmlessly waited for, the last piece of K-step 1)
    __builtin_amdgcn_s_barrier();
    Frags fa, fb;
    read_frags(0, fa);
#pragma unroll
    for (int i = 0; i < NI; ++i) asm volatile("" ::"v"(fa.w[i]), "v"(fa.a[i]));

    // One K-step: wait for K-step kt+1 (everybody's pieces), then issue, woven into the 64 MFMAs of K-step kt (fragments `cur`):
    // the 8 DMA pieces that refill the slot K-step kt-1 lived in (one per 2 MFMAs) and the 16 fragment reads of K-step kt+1 into
    // `nxt` (one per 2 MFMAs).  A wave that issued them in one burst would sit in the LDS / TA queues instead of feeding the
    // matrix pipe (measured: +0.2 us per K-step each).  VMW = DMA pieces of this wave that may still be in flight once K-step
    // kt+1 has landed (vmcnt retires in order); ISSUE: K-step kt+D exists; MORE: K-step kt+1 exists.
    auto kstep = [&](int kt, const Frags& cur, Frags& nxt, auto vmw, auto issue_c, auto more_c) {
        constexpr int VMW = decltype(vmw)::value;
        constexpr bool ISSUE = decltype(issue_c)::value, MORE = decltype(more_c)::value;
        [[maybe_unused]] char *dma_base = nullptr, *dma_base1 = nullptr;
        [[maybe_unused]] size_t ko = 0;
        [[maybe_unused]] unsigned ra = 0, rw = 0;
        if constexpr (MORE) {
            wait_vm<VMW>();
            __builtin_amdgcn_s_barrier();  // K-step kt+1 is in LDS; every wave has finished the MFMAs of K-step kt-1
            const unsigned sb = (unsigned)(size_t)(lptr_t)smem + ((kt + 1) % NS) * STAGE;
            ra = sb + a_off;
            rw = sb + w_off;
            if constexpr (ISSUE) {  // pair (kt+4, kt+5) into the slots K-steps kt-1 and kt lived in (kt's fragments are in registers)
                dma_base = smem + ((kt + 4) % NS) * STAGE + wave * 4096;
                dma_base1 = smem + ((kt + 5) % NS) * STAGE + wave * 4096;
                ko = (size_t)(kt + 4) * BK;
            }
        }
        static_for<NI * MI>([&](auto tc) {
            constexpr int t = decltype(tc)::value, i = t / MI, j = t % MI;
            mfma_acc(acc[i][j], cur.w[i], cur.a[j]);
            if constexpr (ISSUE && t < 32 && (t & 3) == 3) {
                // DMA piece 0..7 (4 activation, 4 weight) of K-steps kt+4 and kt+5: the two halves of each 128-byte line must be
                // requested by ADJACENT instructions -- the vector L1 merges them only then (tools/dma_probe.hip: 0.39 vs 0.58 us
                // per K-step; with 64 cycles between the halves the gain is gone)
                constexpr int pc = t / 4;
                if constexpr (pc < 4) {
                    dma16(a_src[pc] + ko, dma_base + pc * 1024);
                    dma16(a_src[pc] + ko + BK, dma_base1 + pc * 1024);
                } else {
                    dma16(w_src + (pc - 4) * w_step + ko, dma_base + A_BYTES + (pc - 4) * 1024);
                    dma16(w_src + (pc - 4) * w_step + ko + BK, dma_base1 + A_BYTES + (pc - 4) * 1024);
                }
            }
            if constexpr (MORE && t >= 32 && (t & 1) == 1) {
                constexpr int rd = (t - 32) / 2;  // fragment read 0..15: 8 weight, 8 activation
                if constexpr (rd < 8)
                    lds_read16<rd * 1024>(nxt.w[rd], rw);
                else
                    lds_read16<(rd - 8) * 1024>(nxt.a[rd - 8], ra);
            }
        });
        if constexpr (MORE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    using std::integral_constant;
    using T = std::true_type;
    using F = std::false_type;
    int kt = 0;
    for (; kt < nk - D; kt += 2) {  // steady state: the pair behind K-step kt+1 (resp. kt+2) stays in flight
        kstep(kt, fa, fb, integral_constant<int, 2 * PPW>{}, T{}, T{});
        kstep(kt + 1, fb, fa, integral_constant<int, 2 * PPW>{}, F{}, T{});
    }
    kstep(kt, fa, fb, integral_constant<int, 2 * PPW>{}, F{}, T{});                // nk-4: K-steps nk-2, nk-1 in flight
    load_epilogue_operands();                                                        // EPI_LOADS more loads behind them
    kstep(kt + 1, fb, fa, integral_constant<int, EPI_LOADS>{}, F{}, T{});          // nk-3 (first of the last pair: waits for all but one piece of it)
    kstep(kt + 2, fa, fb, integral_constant<int, EPI_LOADS>{}, F{}, T{});          // nk-2
    kstep(kt + 3, fb, fa, integral_constant<int, 0>{}, F{}, F{});                  // nk-1
    fix_epilogue_operands();
    mfma_drain();
    if (abl & 8) return;

    // ---------------------------------------------------------------- epilogue
    const int nwide = n0 + wn * WN + 16 * (fq & 1) + 8 * (fq >> 1);  // after pair_swap: 8 consecutive features, + 32 per tile pair
    const int mrow0 = m0 + wm * WM + fr;
    bf16_t* orow = reinterpret_cast<bf16_t*>(p.out_t) + (size_t)mrow0 * p.ldo + nwide;
    const size_t jstride = (size_t)16 * p.ldo;
    // RoPE (fused QKV projection): each 64-feature half of the wave tile is one head of q, k or v
    [[maybe_unused]] bool rope_half[2] = {false, false};
    [[maybe_unused]] int pos0 = 0;
    [[maybe_unused]] f32x4 rp[2][4];
    auto load_rope = [&](int j, f32x4 (&dst)[4]) {
        int pos = pos0 + 16 * j;
        pos -= pos >= p.rows_per_batch ? p.rows_per_batch : 0;  // rows_per_batch >= 128 (gemm_big_supported)
        const float* t = p.rope + (size_t)pos * 64 + 4 * fq;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const f32x4*>(t + 16 * i);
    };
    if constexpr (EPI == EPI_ROPE_T) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int nw = n0 + wn * WN + hh * 64;
            const int part = nw / p.rope_inner;
            rope_half[hh] = part < 2 && ((nw - part * p.rope_inner) >> 6) < p.rope_heads;
        }
        pos0 = min(mrow0, p.M - 1) % p.rows_per_batch;
        if (rope_half[0] || rope_half[1]) load_rope(0, rp[0]);
    }
    const int act = p.act;
    static_for<MI>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr (EPI == EPI_ROPE_T && j + 1 < MI) {
            if (rope_half[0] || rope_half[1]) load_rope(j + 1, rp[(j + 1) & 1]);
        }
        const bool okm = !RAGGED || mrow0 + 16 * j < p.M;
        bf16_t* o = orow + j * jstride;
        static_for<NI / 2>([&](auto hc) {
            constexpr int i0 = decltype(hc)::value * 2;
            f32x4 vals[2];
            static_for<2>([&](auto ec) {
                constexpr int i = i0 + decltype(ec)::value;
                f32x4 v = acc[i][j] + bias4[i];
                if constexpr (EPI == EPI_STORE_T || EPI == EPI_GATE_T) {
                    if (act == ACT_GELU_TANH) {
                        constexpr float a = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
                        const f32x4 u = v * __builtin_elementwise_fma(v * v, f32x4{a * 0.044715f, a * 0.044715f, a * 0.044715f, a * 0.044715f}, f32x4{a, a, a, a});
                        f32x4 d;
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_exp2f(u[e]);
                        d = d + 1.0f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_rcpf(d[e]);
                        v = v * d;
                    } else if (act == ACT_MISH) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fast_mish(v[e]);
                    }
                }
                if constexpr (EPI == EPI_GATE_T) v = v * gate4[i];
                if constexpr (EPI == EPI_ROPE_T) {
                    if (rope_half[i0 >> 2]) {  // x_transformers apply_rotary_pos_emb: adjacent pairs, fp32 math
                        const f32x4 cs = rp[j & 1][i & 3];
                        v = f32x4{v[0] * cs[0] - v[1] * cs[1], v[1] * cs[0] + v[0] * cs[1], v[2] * cs[2] - v[3] * cs[3], v[3] * cs[2] + v[2] * cs[3]};
                    }
                }
                vals[i - i0] = v;
            });
            u32x4 q = pair_swap(to_bf16x4(vals[0]), to_bf16x4(vals[1]));
            if constexpr (EPI == EPI_GATE_T) {
                const bool keep = (keepbits >> j) & 1u;
                q = keep ? q : u32x4{0u, 0u, 0u, 0u};
            }
            if (okm) *reinterpret_cast<u32x4*>(o + 32 * (i0 / 2)) = q;
        });
    });
}

}  // namespace

int g_gemm_big_ablate = 0;  // tuning knob ("gemm_big_ablate"): timing-only ablation bits, see GemmParams::ablate
int g_gemm_big = 0;  // tuning knob ("gemm_big"): 1 = DiT block linears run on the 4-wave 128x128-per-wave kernel where it applies (measured: on par with / slightly behind gemm_fast.hip)

bool gemm_big_supported(const GemmParams& p, int mode, int epi) {
    if (!g_gemm_big || mode != GEMM_DENSE) return false;
    if (!(epi == EPI_STORE_T || epi == EPI_GATE_T || epi == EPI_ROPE_T)) return false;
    if (p.M <= 0 || p.N <= 0 || p.N % 256 != 0 || p.K < 128 || p.K % 64 != 0) return false;  // nk even and >= ring depth
    if ((p.lda & 7) || (p.ldw & 7) || (p.ldo & 7) || p.a_row_mod > 0 || !p.bias || !p.out_t) return false;
    if (p.act == ACT_GELU_ERF) return false;
    if (epi == EPI_ROPE_T && (p.act != ACT_NONE || p.rows_per_batch < 128 || p.rope_inner % 64 != 0 || !p.rope)) return false;
    if (epi == EPI_GATE_T && ((p.gate && p.gate_bstride != 0) || (p.rowmask && !p.rowbits))) return false;
    return cdiv(p.M, 256) * (p.N / 256) >= 160;  // fewer tiles than that: the narrower tiles of gemm_fast.hip keep the CUs busier
}

extern int g_gemm_group;

int launch_gemm_big(const GemmParams& p0, int epi, hipStream_t stream) {
    GemmParams p = p0;
    p.tile_group = g_gemm_group > 0 ? g_gemm_group : 8;
    p.ablate = g_gemm_big_ablate;
    const int tiles_m = cdiv(p.M, 256), tiles_n = p.N / 256;
    const int nblocks = tiles_m * tiles_n;
    const bool ragged = p.M % 256 != 0;
    dim3 grid(nblocks), block(256);
#define F5_BIG_CASE(E)                                                                                                   \
    case E:                                                                                                              \
        if (ragged)                                                                                                      \
            hipLaunchKernelGGL((gemm_big_kernel<E, true>), grid, block, 0, stream, p, tiles_n, nblocks);                 \
        else                                                                                                             \
            hipLaunchKernelGGL((gemm_big_kernel<E, false>), grid, block, 0, stream, p, tiles_n, nblocks);                \
        break;
    switch (epi) {
        F5_BIG_CASE(EPI_STORE_T)
        F5_BIG_CASE(EPI_GATE_T)
        F5_BIG_CASE(EPI_ROPE_T)
        default:
            return f5_fail(F5_EINVAL, "gemm_big: unsupported epilogue %d", epi);
    }
#undef F5_BIG_CASE
    F5_LAUNCH_CHECK();
    return 0;
}
