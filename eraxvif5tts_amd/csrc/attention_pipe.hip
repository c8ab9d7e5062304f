// attention_pipe.hip -- software-pipelined flash attention for the DiT blocks on gfx950 (bf16 in/out, head dim 64, non-causal,
// key-padding mask; reference model/modules.py:483-497).  32 queries per wavefront, 128 per workgroup: the launcher
// (attention_fast.hip) picks it when the 256-query workgroups of the wide kernel would leave CUs idle (single-utterance serving).
//
//   * each wave owns 32 queries and keeps TWO score tiles: while the softmax of tile t runs on the vector ALU, the QK^T MFMAs of
//     tile t+1 and the PV MFMAs of tile t (each 16-key step as soon as its P is converted) are laid into the same instruction
//     stream by hand: micro-groups of {one MFMA, 7 vector instructions} between scheduling fences;
//   * swapped products (S^T = K.Q^T, O^T = V^T.P^T, v_mfma_f32_32x32x16_bf16): a lane holds 16 keys of ONE query per 32-key block,
//     so max / sum are in-register (one v_permlane32_swap per tile) and the bf16 P registers are directly the B operand of PV;
//   * K/V tiles of 64 keys: K runs one tile ahead of V (two buffers each, 32 KiB per workgroup); the tiles come by LDS-DMA
//     (global_load_lds_dwordx4, swizzle on the source address), issued a whole tile before the barrier that publishes them;
//     ONE barrier per tile;
//   * LDS fragment reads (ds_read_b128 for K, ds_read_b64_tr_b16 for V^T) are inline asm with hand-counted s_waitcnt lgkmcnt, issued a
//     whole slot before their MFMA (reads the compiler can see make it drain the LDS-DMA in front of each of them);
//   * deferred rescale of O (reference moved only when a row maximum outgrew it by more than 2^16);
//   * XCD-aware block order: all query blocks of one (batch, head) share an XCD's L2.
// What bounds it (tools/valu_probe.hip, tools/attn_ablate.hip, profiles/r2_attention_*.txt): per 32 x 64 tile a wave issues 16 MFMAs
// (512 matrix-pipe cycles) and ~180 vector instructions (~700 cycles of SIMD time at two waves per SIMD: v_exp_f32 8.2, v_cvt_pk /
// v_max3 4.5, fma / add 2.5 cycles); on ONE SIMD the two pipes mostly serialise (an MFMA hides ~10 cycles of plain vector work, ~23
// of v_exp), and the 16 KiB of LDS fragments a wave pulls per tile are hidden by neither.
#include <type_traits>

#include "kernels.h"

namespace {

template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)s[base + j];
    return r;
}
// v_permlane32_swap with the same value on both sides: the results are {own half | lower half's value, upper half's value | own half},
// i.e. every lane ends up with its own value and the one lane ^ 32 holds (VALU only, no LDS round trip)
__device__ __forceinline__ float max_halves(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
// v_max3_f32 without the canonicalising v_max the compiler puts in front of fmaxf on MFMA outputs (scores are never signalling NaNs)
__device__ __forceinline__ float max3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float sum_halves(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// WAVES wavefronts x 32 queries per workgroup.
// ABL: timing-only ablation bits (wrong results by construction; only tools/attn_ablate.hip instantiates ABL != 0, the library never does):
//   1 no exp2 (P = the scaled score), 2 no PV MFMAs, 4 no QK^T MFMAs, 8 no K/V tile refresh (no global loads, LDS writes or barriers),
//   16 no row-sum adds, 32 no row max, 64 no LDS fragment reads in the loop, 128 / 256 no K / no V fragment reads, 512 fragments read
//   and waited for but the MFMAs take register operands
// SEG (ragged sampler): the launch covers several utterances of different lengths that sit at row offsets inside a longer concatenation;
// batch index = utterance * nbr + branch, N and the row offset come from `segs`, query blocks past an utterance's end leave at once.  Every
// other block computes exactly what it computes in a launch over its utterance alone (same N, same rows).
template <bool MASKED, int WAVES, int ABL = 0, int WPE = 2, bool SEG = false>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void attn_pipe_kernel(const bf16_t* __restrict__ qkv, int ldq, int inner,
                                                                                 const uint8_t* __restrict__ mask, bf16_t* __restrict__ out,
                                                                                 int ldo, int N, int bs /* rows between batch items */, float c /* scale * log2(e) */,
                                                                                 AttnSegs segs) {
    constexpr int KT = 64;                // keys per tile
    constexpr int TB = KT * 128;          // 64 keys x 64 dims x 2 B
    constexpr int NTH = WAVES * 64, CH = 512 / NTH, QW = 32 * WAVES;
    __shared__ __attribute__((aligned(16))) char smem[4 * TB];  // K0 | K1 | V0 | V1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int QB = gridDim.x, BH = gridDim.y * gridDim.z;
    int qb = blockIdx.x, bh = blockIdx.y + blockIdx.z * gridDim.y;
    if ((BH & 7) == 0) {  // blocks id and id + 8 share an XCD: give all query blocks of one (batch, head) equal id % 8
        const int id = blockIdx.x + QB * bh;
        const int xcd = id & 7, j = id >> 3;
        qb = j % QB;
        bh = (j / QB) * 8 + xcd;
    }
    const int b = bh / gridDim.y, head = bh - b * gridDim.y, q0 = qb * QW + wave * 32;
    const int r = lane & 31, h = lane >> 5;
    size_t row0 = (size_t)b * bs;
    if constexpr (SEG) {
        const int u = b / segs.nbr, br = b - u * segs.nbr;
        N = segs.n[u];
        if (qb * QW >= N) return;  // (the whole workgroup: nothing has been issued yet)
        row0 = (size_t)br * bs + segs.off[u];
    }
    const bf16_t* base = qkv + row0 * ldq + head * 64;
    const bf16_t* kbase = base + inner;
    const bf16_t* vbase = base + 2 * inner;

    // ---- Q fragments (B operand: lane holds Q[query r][d = 16*ds + 8*h .. +7]), kept in registers for the whole kernel
    bf16x8 qf[4];
    {
        int qrow = q0 + r;
        if (qrow >= N) qrow = N - 1;  // clamped rows are computed and dropped
        const bf16_t* qp = base + (size_t)qrow * ldq + 8 * h;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) qf[ds] = *reinterpret_cast<const bf16x8*>(qp + 16 * ds);
    }

    // ---- K/V tile staging by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write): a piece is 8 key rows x 128 B
    //      (1 KiB, one wave-instruction); the LDS image is lane-linear, so the swizzles sit on the SOURCE address.  A tile has 8 K
    //      pieces and 8 V pieces; wave w moves pieces w, w + WAVES, ... of each.
    const int drow = lane >> 3, dchunk = lane & 7;
    uint8_t mreg = 1;
    auto dma_k = [&](int k0, int buf) {
#pragma unroll
        for (int pc = 0; pc < 8 / WAVES + (8 % WAVES != 0); ++pc) {
            const int piece = __builtin_amdgcn_readfirstlane(wave) + pc * WAVES, row = piece * 8 + drow;
            int key = k0 + row;
            if (key >= N) key = N - 1;
            const bf16_t* src = kbase + (size_t)key * ldq + ((dchunk ^ ((row >> 1) & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + buf * TB + piece * 1024), 16, 0, 0);
        }
    };
    auto dma_v = [&](int k0, int buf) {
#pragma unroll
        for (int pc = 0; pc < 8 / WAVES + (8 % WAVES != 0); ++pc) {
            const int piece = __builtin_amdgcn_readfirstlane(wave) + pc * WAVES, row = piece * 8 + drow;
            int key = k0 + row;
            if (key >= N) key = N - 1;
            const bf16_t* src = vbase + (size_t)key * ldq + ((dchunk ^ (((row >> 1) & 1) << 2)) << 3);
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smem + (2 + buf) * TB + piece * 1024), 16, 0, 0);
        }
    };
    auto load_m = [&](int k0) {
        if constexpr (MASKED) {
            const int key = k0 + lane;
            mreg = key < N ? (mask ? mask[(size_t)b * N + key] : (uint8_t)1) : (uint8_t)0;
        }
    };

    // ---- per-lane LDS read offsets
    // K fragment (A operand of S^T): row = 32*kb + r, logical chunk = 2*ds + h
    int k_off[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) k_off[ds] = r * 128 + (((2 * ds + h) ^ ((r >> 1) & 7)) << 4);
    // V^T fragment via ds_read_b64_tr_b16: this lane supplies the address of key row (4*h + ((lane&15)>>2)) [+ 8*g + 16*ks + 32*kb],
    // d columns 16*((lane>>4)&1) + 4*(lane&3) [+ 32*mb]
    const int v_row = 4 * h + ((lane & 15) >> 2);
    const int v_colb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;  // byte column inside the 64-byte half mb

    f32x16 o_acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) o_acc[0][i] = o_acc[1][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;
    const int nt = (N + KT - 1) / KT;

    // ---- prologue: K(0), V(0), K(1) into LDS; S(0)
    dma_k(0, 0);
    dma_v(0, 0);
    dma_k(nt > 1 ? KT : 0, 1);
    load_m(0);
    unsigned long long vm = ~0ull;  // validity of the 64 keys of the tile whose softmax comes next
    if constexpr (MASKED) vm = __ballot(mreg != 0);
    load_m(KT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- LDS fragment reads are inline asm with hand-counted s_waitcnt lgkmcnt: the instruction stream of a tile is fixed, and reads the
    //      compiler can see would make it drain the LDS-DMA (vmcnt(0)) in front of every one of them (it cannot prove that the tile
    //      being read is not the one a DMA in flight is writing).  Order of the LDS operations of one step -- the wait counts below are
    //      "operations issued after the one needed":
    //        block A: K0 K1 K2 K3 V0(4 ops) | slot 0: K4 K5 | slot 1: V1(4), K6 K7 | slot 2: V2(4) | slot 3: V3(4)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned ka[4], va_addr[2];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) ka[ds] = lds0 + k_off[ds];
    {
        const int sw = ((v_row >> 1) & 1) << 6;  // rows 16*s + 8*g + v_row: the swizzle bit of a row does not depend on s, g
        va_addr[0] = lds0 + v_row * 128 + (v_colb ^ sw);
        va_addr[1] = lds0 + v_row * 128 + ((64 + v_colb) ^ sw);
    }
    auto k_frag = [&](auto nc, auto bufc) {  // A operand of QK^T MFMA n = 4*kb + ds of the tile in K buffer `buf`
        constexpr int n = decltype(nc)::value, buf = decltype(bufc)::value;
        bf16x8 d;
        if constexpr (ABL & (64 | 128)) return qf[n & 3];
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(ka[n & 3]), "n"(buf * TB + (n >> 2) * 32 * 128));
        return d;
    };
    auto v_frag = [&](auto sc, auto mbc, auto bufc) {  // A operand (V^T, 32 dims x 16 keys) of PV step s = 2*kb + ks, dim half mb, V buffer `buf`
        constexpr int s = decltype(sc)::value, mb = decltype(mbc)::value, buf = decltype(bufc)::value;
        if constexpr (ABL & (64 | 256)) return qf[mb];
        bf16x4 lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(va_addr[mb]), "n"((2 + buf) * TB + 16 * s * 128));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(va_addr[mb]), "n"((2 + buf) * TB + (16 * s + 8) * 128));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // the fragment is an in/out operand of the wait, so its MFMA cannot be scheduled in front of it (ablation builds drain everything)
    auto lds_wait = [&](auto cntc, bf16x8& frag) {
        constexpr int cnt = ABL ? 0 : decltype(cntc)::value;
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(cnt));
    };
#define F5_IC(x) std::integral_constant<int, (x)> {}
#define F5_FENCE() __builtin_amdgcn_sched_barrier(0)
    auto mma_qk = [&](const bf16x8& a, const bf16x8& bq, f32x16& acc) {
        if constexpr (ABL & 4) {
            asm volatile("" ::"v"(a), "v"(bq));
        } else if constexpr (ABL & 512) {  // fragment read and waited for, MFMA fed from registers
            asm volatile("" ::"v"(a));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bq, bq, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc, 0, 0, 0);
        }
    };
    auto mma_pv = [&](const bf16x8& a, const bf16x8& bp, f32x16& acc) {
        if constexpr (ABL & 2) {
            asm volatile("" ::"v"(a), "v"(bp));
        } else if constexpr (ABL & 512) {
            asm volatile("" ::"v"(a));
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bp, bp, acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bp, acc, 0, 0, 0);
        }
    };
    constexpr float ATTN_DEFER_LOG2 = 16.0f;  // a row's exponent reference is moved when its maximum exceeds it by more than this (log2 units)

    f32x16 sA[2], sB[2];
    static_for<8>([&](auto nc) {  // S(0) = K(0) . Q^T (no overlap to exploit yet)
        constexpr int n = decltype(nc)::value, kb = n >> 2, ds = n & 3;
        if constexpr (ds == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) sA[kb][i] = 0.f;
        }
        bf16x8 kf = k_frag(nc, F5_IC(0));
        lds_wait(F5_IC(0), kf);
        sA[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], sA[kb], 0, 0, 0);
    });
    __syncthreads();  // every wave has read K(0): the first step may overwrite its buffer with K(2)

    // One tile: softmax + PV of tile t (scores in `cur`), QK^T of tile t+1 into `nxt`.  The instruction stream is laid out by hand in
    // micro-groups of {one MFMA, 7 vector instructions} separated by scheduling fences: the vector ALU never waits for the matrix
    // pipe and every MFMA runs in the shadow of vector work.  LDS fragments are read a whole slot (4 micro-groups, ~150 issue cycles)
    // before their MFMA; the last PV step of a tile is carried in registers into the next tile's row-max block.
    bf16x8 pend_p, pend_va, pend_vb;  // P and V^T fragments of the pending PV step (zero before the first tile)
#pragma unroll
    for (int j = 0; j < 8; ++j) pend_p[j] = pend_va[j] = pend_vb[j] = (bf16_t)0.f;
    auto step = [&](auto parc, f32x16 (&cur)[2], f32x16 (&nxt)[2], int t) {
        constexpr int PAR = decltype(parc)::value;  // t & 1: K(t+1) sits in K buffer PAR ^ 1, V(t) in V buffer PAR
        constexpr auto KB = F5_IC(PAR ^ 1);
        constexpr auto VB = F5_IC(PAR);
        // K(t+2) / V(t+1) start their way into the buffers of K(t) / V(t-1), which every wave finished reading before the last barrier;
        // they have this whole tile to land
        if constexpr (!(ABL & 8)) {
            dma_k(t + 2 < nt ? (t + 2) * KT : 0, PAR);
            dma_v(t + 1 < nt ? (t + 1) * KT : 0, PAR ^ 1);
        }
        if constexpr (MASKED) {
            if (vm != ~0ull) {
                const unsigned long long vmh = h ? (vm >> 4) : vm;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int bit = 32 * kb + (i & 3) + 8 * (i >> 2);
                        if (!((vmh >> bit) & 1ull)) cur[kb][i] = -INFINITY;
                    }
            }
        }
        // ---- block A: row max of tile t (this lane: one query, keys 32*kb + (i&3) + 8*(i>>2) + 4*h) with the pending PV step in its shadow
        bf16x8 kq0 = k_frag(F5_IC(0), KB), kq1 = k_frag(F5_IC(1), KB), kq2 = k_frag(F5_IC(2), KB), kq3 = k_frag(F5_IC(3), KB);
        bf16x8 va0 = v_frag(F5_IC(0), F5_IC(0), VB), vb0 = v_frag(F5_IC(0), F5_IC(1), VB), va1, vb1;
        float mx0 = max3(cur[0][0], cur[0][1], cur[0][2]), mx1 = max3(cur[1][0], cur[1][1], cur[1][2]);
#pragma unroll
        for (int i = 3; i < 9; i += 2) {
            mx0 = max3(mx0, cur[0][i], cur[0][i + 1]);
            mx1 = max3(mx1, cur[1][i], cur[1][i + 1]);
        }
        F5_FENCE();
        lds_wait(F5_IC(8), pend_va);  // V3 of the previous tile landed before the barrier (lgkmcnt(0) there); only this block's 8 reads may fly
        mma_pv(pend_va, pend_p, o_acc[0]);
#pragma unroll
        for (int i = 9; i < 15; i += 2) {
            mx0 = max3(mx0, cur[0][i], cur[0][i + 1]);
            mx1 = max3(mx1, cur[1][i], cur[1][i + 1]);
        }
        F5_FENCE();
        lds_wait(F5_IC(8), pend_vb);
        mma_pv(pend_vb, pend_p, o_acc[1]);
        mx0 = max3(mx0, mx1, cur[0][15]);
        const float mt = (ABL & 32) ? cur[0][0] * c : max_halves(max3(mx0, cur[1][15], cur[1][15])) * c;  // (lane ^ 32 holds the other 32 keys of the query)
#pragma unroll
        for (int i = 0; i < 16; ++i) nxt[0][i] = nxt[1][i] = 0.f;
        F5_FENCE();
        // Deferred rescale: the exponent reference m_run of a query only moves when its row maximum has outgrown it by more than 2^THR
        // (softmax is invariant under the choice of reference; the fp32 sums and the bf16 P keep their relative precision at any scale,
        // and P <= 2^THR is nowhere near the range limit).  With random scores the exact running max of SOME query of a wave moves in
        // nearly every tile, and the 32-register rescale of O with it; the deferred form rescales once or twice per query block.
        if (__builtin_amdgcn_ballot_w64(mt > m_run + ATTN_DEFER_LOG2) != 0ull) {
            const float m_new = fmaxf(m_run, mt);  // -inf (a fully masked tile) leaves the reference alone; m_run starts finite
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o_acc[0][i] *= alpha;
                o_acc[1][i] *= alpha;
            }
        }
        // ---- block B: four slots of 16 keys: exp2 -> bf16 P.  Slot s issues the PV MFMAs of slot s-1 and two QK^T MFMAs of tile t+1 and
        //      reads the fragments the NEXT slot's MFMAs need.
        const float nm = -m_run;
        float rs0 = 0.f, rs1 = 0.f;
        bf16x8 pf_prev;
        auto slot = [&](auto sc) {
            constexpr int s = decltype(sc)::value, kb = s >> 1, e0 = 8 * (s & 1);
            float p[8];
            // -- LDS reads for the next slot (K fragments 4..7 reuse the registers of 0..3 once those MFMAs are issued)
            if constexpr (s == 1) {
                va1 = v_frag(F5_IC(1), F5_IC(0), VB);
                vb1 = v_frag(F5_IC(1), F5_IC(1), VB);
            } else if constexpr (s == 2) {
                va0 = v_frag(F5_IC(2), F5_IC(0), VB);
                vb0 = v_frag(F5_IC(2), F5_IC(1), VB);
            } else if constexpr (s == 3) {
                va1 = v_frag(F5_IC(3), F5_IC(0), VB);
                vb1 = v_frag(F5_IC(3), F5_IC(1), VB);
            }
            // -- micro-group 1
            if constexpr (s == 0) { lds_wait(F5_IC(7), kq0); mma_qk(kq0, qf[0], nxt[0]); }
            if constexpr (s == 1) { lds_wait(F5_IC(8), va0); mma_pv(va0, pf_prev, o_acc[0]); }
            if constexpr (s == 2) { lds_wait(F5_IC(8), va1); mma_pv(va1, pf_prev, o_acc[0]); }
            if constexpr (s == 3) { lds_wait(F5_IC(6), va0); mma_pv(va0, pf_prev, o_acc[0]); }
#pragma unroll
            for (int j = 0; j < 4; ++j) p[j] = __builtin_fmaf(cur[kb][e0 + j], c, nm);
#pragma unroll
            for (int j = 0; j < 3; ++j) p[j] = (ABL & 1) ? p[j] : __builtin_amdgcn_exp2f(p[j]);
            F5_FENCE();
            // -- micro-group 2
            if constexpr (s == 0) { lds_wait(F5_IC(6), kq1); mma_qk(kq1, qf[1], nxt[0]); }
            if constexpr (s == 1) { lds_wait(F5_IC(6), vb0); mma_pv(vb0, pf_prev, o_acc[1]); }
            if constexpr (s == 2) { lds_wait(F5_IC(6), vb1); mma_pv(vb1, pf_prev, o_acc[1]); }
            if constexpr (s == 3) { lds_wait(F5_IC(4), vb0); mma_pv(vb0, pf_prev, o_acc[1]); }
            if constexpr (s == 0) {  // K fragments 4, 5 (used in slot 1)
                kq0 = k_frag(F5_IC(4), KB);
                kq1 = k_frag(F5_IC(5), KB);
            }
            p[3] = (ABL & 1) ? p[3] : __builtin_amdgcn_exp2f(p[3]);
#pragma unroll
            for (int j = 4; j < 8; ++j) p[j] = __builtin_fmaf(cur[kb][e0 + j], c, nm);
#pragma unroll
            for (int j = 4; j < 6; ++j) p[j] = (ABL & 1) ? p[j] : __builtin_amdgcn_exp2f(p[j]);
            F5_FENCE();
            // -- micro-group 3
            if constexpr (s == 0) { lds_wait(F5_IC(7), kq2); mma_qk(kq2, qf[2], nxt[0]); }
            if constexpr (s == 1) { lds_wait(F5_IC(5), kq0); mma_qk(kq0, qf[0], nxt[1]); }
            if constexpr (s == 2) { lds_wait(F5_IC(5), kq2); mma_qk(kq2, qf[2], nxt[1]); }
#pragma unroll
            for (int j = 6; j < 8; ++j) p[j] = (ABL & 1) ? p[j] : __builtin_amdgcn_exp2f(p[j]);
            if constexpr (!(ABL & 16)) {
                rs0 += p[0];
                rs1 += p[1];
                rs0 += p[2];
                rs1 += p[3];
                rs0 += p[4];
            }
            F5_FENCE();
            // -- micro-group 4
            if constexpr (s == 0) { lds_wait(F5_IC(6), kq3); mma_qk(kq3, qf[3], nxt[0]); }
            if constexpr (s == 1) { lds_wait(F5_IC(4), kq1); mma_qk(kq1, qf[1], nxt[1]); }
            if constexpr (s == 2) { lds_wait(F5_IC(4), kq3); mma_qk(kq3, qf[3], nxt[1]); }
            if constexpr (s == 1) {  // K fragments 6, 7 (used in slot 2)
                kq2 = k_frag(F5_IC(6), KB);
                kq3 = k_frag(F5_IC(7), KB);
            }
            if constexpr (!(ABL & 16)) {
                rs1 += p[5];
                rs0 += p[6];
                rs1 += p[7];
            }
            bf16x8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)p[j];
            pf_prev = pf;
            F5_FENCE();
        };
        slot(F5_IC(0));
        slot(F5_IC(1));
        slot(F5_IC(2));
        slot(F5_IC(3));
        pend_p = pf_prev;  // PV step 3 of this tile rides in the next tile's block A
        pend_va = va1;
        pend_vb = vb1;
        // ---- tail: the DMA'd tiles have landed and this wave's LDS reads are complete; then the same for every wave
        l_run += rs0 + rs1;
        if constexpr (!(ABL & 8)) {
            if constexpr (MASKED) {
                vm = __ballot(mreg != 0);
                load_m((t + 2) * KT);
            }
            F5_FENCE();
            // Every LDS-DMA piece of this wave has landed.  vmcnt(0) in the MASKED build too: load_m above issues its byte load only for
            // keys inside the sequence and only with a mask array, so "the youngest operation is the mask byte" (vmcnt(1)) does not hold
            // for a ragged single utterance (mask == nullptr, N % 64 != 0) -- the last piece of V(t+1) would be left uncovered.  Free: the
            // compiler waits for the byte right behind its load anyway (it feeds the ballot of the next tail).
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            F5_FENCE();
        }
    };

    for (int t = 0; t < nt; t += 2) {
        step(F5_IC(0), sA, sB, t);
        if (t + 1 < nt) step(F5_IC(1), sB, sA, t + 1);
    }

    lds_wait(F5_IC(0), pend_va);  // the last tile's last PV step
    lds_wait(F5_IC(0), pend_vb);
    mma_pv(pend_va, pend_p, o_acc[0]);
    mma_pv(pend_vb, pend_p, o_acc[1]);
    // ---- normalise and store: lane holds query q0 + r, dims 32*mb + (i&3) + 8*(i>>2) + 4*h
    const float l_tot = sum_halves(l_run);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int qrow = q0 + r;
    if (qrow < N) {
        bf16_t* op = out + (row0 + qrow) * ldo + head * 64 + 4 * h;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(o_acc[mb][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(op + 32 * mb + 8 * g) = v4;
            }
    }
}

}  // namespace

int launch_attention_pipe(int waves, int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out, int ldo, hipStream_t stream, int bstride) {
    if (waves != 4) return f5_fail(F5_EINVAL, "attention_pipe: the library instantiates the 4-wave (128 queries per workgroup) build only");
    const float c = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
    const bool masked = mask != nullptr || (N % 64) != 0;
    const dim3 grid(cdiv(N, 128), H, B);
    if (masked)
        hipLaunchKernelGGL((attn_pipe_kernel<true, 4>), grid, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, bstride, c, AttnSegs{});
    else
        hipLaunchKernelGGL((attn_pipe_kernel<false, 4>), grid, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, bstride, c, AttnSegs{});
    F5_LAUNCH_CHECK();
    return 0;
}

// several utterances (all with N % 64 == 0, or all without) of one ragged batch in ONE launch: grid.z = utterance x branch
int launch_attention_pipe_segs(bool masked, int nbr, const AttnSegs& segs, int maxN, int H, const void* qkv, int ldq, void* out, int ldo, hipStream_t stream,
                               int bstride) {
    const float c = 0.125f * 1.4426950408889634f;
    const dim3 grid(cdiv(maxN, 128), H, nbr * segs.cnt);
    if (masked)
        hipLaunchKernelGGL((attn_pipe_kernel<true, 4, 0, 2, true>), grid, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, (const uint8_t*)nullptr, (bf16_t*)out, ldo, maxN,
                           bstride, c, segs);
    else
        hipLaunchKernelGGL((attn_pipe_kernel<false, 4, 0, 2, true>), grid, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, (const uint8_t*)nullptr, (bf16_t*)out, ldo, maxN,
                           bstride, c, segs);
    F5_LAUNCH_CHECK();
    return 0;
}
