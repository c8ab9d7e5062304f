// attention_fast.hip -- flash attention for the DiT blocks on gfx950 (bf16 in/out, head dim 64, non-causal, key-padding mask;
// reference model/modules.py:483-497): the 64-queries-per-wavefront kernel and the launcher that picks between it and the
// software-pipelined 32-queries-per-wavefront kernel of attention_pipe.hip.
//
// Structure (see /opt/skills/guides/cdna_hip_programming.md, Appendix B "Fused attention prefill"):
//   * one workgroup = 4 wavefronts = 256 queries of one (batch, head); each wave owns TWO 32-query blocks, so every K fragment,
//     V^T fragment, K/V tile refresh and barrier is amortised over 64 queries (profiles/r2_attention_*.txt: on one SIMD the matrix
//     pipe and the vector ALU mostly serialise and the LDS -> VGPR fragment traffic is hidden by neither, so halving the fragment
//     bytes per FLOP is worth more than instruction placement);
//   * swapped QK^T: S^T = K.Q^T with v_mfma_f32_32x32x16_bf16, so a lane holds 16 keys x ONE query per 32-key block: sums (and the
//     rare maximum) are in-register, and the un-normalised P converted to bf16 is directly the B operand of the PV product;
//   * O^T = V^T.P^T: the V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose);
//   * K tile XOR-swizzled for conflict-free ds_read_b128, V tile swizzled for the transposed reads;
//   * exp2 with the softmax scale folded into one fma, fp32 sums and output accumulators, plain (unpacked) fp32 vector code: the file
//     is built with -fno-slp-vectorize (packed fp32 VALU beside MFMAs is slower, MI355X_MICROARCH.md cycle constants).
// K/V staging, fragment reads and the maximum-free softmax are described at the kernel.
#include "kernels.h"

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ float max3_asm(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)s[base + j];
    return r;
}

// Start stagger (tuning knob "attn_stagger"): workgroups of equal length that start together stay in phase for the whole launch -- every one of
// them reads its Q block, and later stores its output block, at the same moment as all the others, and the matrix pipes idle while those
// bursts drain.  The workgroups with linear id in [lo, hi) (the second resident workgroup of every CU in the first round) start `ticks` x 10 ns
// late; their successors inherit the offset.  Speed only: no result depends on it.
__device__ __forceinline__ void stagger_delay(int id, int lo, int hi, int ticks) {
    if (ticks > 0 && id >= lo && id < hi) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(16);
    }
}

// attn_wide_kernel:
//   * K/V tiles by LDS-DMA (global_load_lds_dwordx4, swizzle on the SOURCE address) through a ring of three tile buffers: tile t+2 is
//     requested at the start of tile t, a counted s_waitcnt vmcnt + ONE raw s_barrier per tile publish tile t+1.  No staging registers,
//     no ds_write, no address arithmetic beyond one 64-bit add per piece (the register-staged refresh cost 19 % of the kernel);
//   * LDS fragment reads as inline asm with hand-counted s_waitcnt lgkmcnt (reads the compiler can see make it drain the LDS-DMA with
//     vmcnt(0) in front of each of them);
//   * NO row maximum on the common path: P = exp2(s*c - m_ref) is taken against the reference the query already has; softmax does not
//     depend on the reference, fp32 / bf16 keep their relative precision at any scale, so all that can go wrong is range.  The row sums
//     are checked against 2^64 once per tile (any P >= 2^64, inf or NaN trips it); only then the wave takes the classic step: scores
//     recomputed from the K tile still in LDS, exact running maximum, rescale of O and l.  The first tile of a block sets the reference
//     to its exact maximum (nothing to rescale yet).  With that the 22-deep max3 chain in front of the exponentials is gone from the
//     steady state (-8 % on its own).
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_wide_kernel(const bf16_t* __restrict__ qkv, int ldq, int inner, const uint8_t* __restrict__ mask,
                                                           bf16_t* __restrict__ out, int ldo, int N, int bs /* rows between batch items */, float c, int stagger_lo,
                                                           int stagger_hi, int stagger_ticks) {
    constexpr int KT = 64, QB = 2, TB = KT * 128, NBUF = 3, BUF = 2 * TB, WAVES = 4;
    constexpr int PCS = 8 / WAVES;  // K (and V) pieces per wave per tile
    constexpr int MAXT = 128;  // tiles whose key validity bits fit the LDS table (launcher: N <= 64 * MAXT when masked)
    __shared__ __attribute__((aligned(1024))) char smem[NBUF * BUF + (MASKED ? MAXT * 8 : 0)];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NQ = gridDim.x, BH = gridDim.y * gridDim.z;
    int qblk = blockIdx.x, bh = blockIdx.y + blockIdx.z * gridDim.y;
    stagger_delay(blockIdx.x + NQ * bh, stagger_lo, stagger_hi, stagger_ticks);
    if ((BH & 7) == 0) {  // XCD-aware order: all query blocks of one (batch, head) share an XCD
        const int id = blockIdx.x + NQ * bh;
        const int xcd = id & 7, j = id >> 3;
        qblk = j % NQ;
        bh = (j / NQ) * 8 + xcd;
    }
    const int b = bh / gridDim.y, head = bh - b * gridDim.y, q0 = qblk * (32 * QB * WAVES) + wave * (32 * QB);
    const int r = lane & 31, h = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * bs * ldq + head * 64;
    const bf16_t* kbase = base + inner;
    const bf16_t* vbase = base + 2 * inner;
    const int nt = (N + KT - 1) / KT;

    // ---- key validity bits of every tile (masked build): one 64-bit word per tile in LDS, written before any DMA is in flight
    unsigned long long* mbits = reinterpret_cast<unsigned long long*>(smem + NBUF * BUF);
    if constexpr (MASKED) {
        for (int i = wave; i < nt; i += WAVES) {
            const int key = i * KT + lane;
            const uint8_t m = key < N ? (mask ? mask[(size_t)b * N + key] : (uint8_t)1) : (uint8_t)0;
            const unsigned long long bits = __ballot(m != 0);
            if (lane == 0) mbits[i] = bits;
        }
    }

    bf16x8 qf[QB][4];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        int qrow = q0 + 32 * j + r;
        if (qrow >= N) qrow = N - 1;
        const bf16_t* qp = base + (size_t)qrow * ldq + 8 * h;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) qf[j][ds] = *reinterpret_cast<const bf16x8*>(qp + 16 * ds);
    }

    // ---- K/V tile -> LDS by DMA: a piece is 8 key rows x 128 B (one wave-instruction, 1 KiB); wave w moves pieces w (and w + 4 in the 4-wave build) of K and of V
    const int drow = lane >> 3, dchunk = lane & 7;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const unsigned ldq2 = (unsigned)ldq * 2u;
    unsigned kco[PCS], vco[PCS];  // byte offset of this lane's 16-byte chunk inside the key row, K and V swizzles
#pragma unroll
    for (int pc = 0; pc < PCS; ++pc) {
        const int row = (wv + WAVES * pc) * 8 + drow;
        kco[pc] = (unsigned)((dchunk ^ ((row >> 1) & 7)) << 4);
        vco[pc] = (unsigned)((dchunk ^ (((row >> 1) & 1) << 2)) << 4);
        if constexpr (!MASKED) {  // whole tiles only: the row offset is lane-constant too
            kco[pc] += (unsigned)row * ldq2;
            vco[pc] += (unsigned)row * ldq2;
        }
    }
    auto dma_tile = [&](int k0, int buf) {
        const char* kt = reinterpret_cast<const char*>(kbase) + (MASKED ? (size_t)0 : (size_t)k0 * ldq2);  // wave-uniform
        const char* vt = reinterpret_cast<const char*>(vbase) + (MASKED ? (size_t)0 : (size_t)k0 * ldq2);
#pragma unroll
        for (int pc = 0; pc < PCS; ++pc) {
            const int piece = wv + WAVES * pc;
            unsigned ko = kco[pc], vo = vco[pc];
            if constexpr (MASKED) {  // ragged last tile: rows past the sequence re-read its last key (masked out by the validity bits)
                const unsigned ro = (unsigned)min(k0 + piece * 8 + drow, N - 1) * ldq2;
                ko += ro;
                vo += ro;
            }
            char* dst = smem + buf * BUF + piece * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)(kt + ko), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(vt + vo), (lptr_t)(dst + TB), 16, 0, 0);
        }
    };

    // ---- per-lane LDS read addresses (buffer 0; the tile's buffer offset is added per tile)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    // K fragment of d-slice ds: row r, logical 16-byte chunk 2*ds + h at physical chunk (2*ds + h) ^ ((r >> 1) & 7): the address of slice ds
    // is the address of slice 0 with bits 5..6 XORed by ds (the tile buffers are 128-byte aligned), so one register carries all four
    unsigned ka_t = lds0 + r * 128 + ((h ^ ((r >> 1) & 7)) << 4);  // slice 0, buffer of the current tile
    unsigned va[2];
    {
        const int v_row = 4 * h + ((lane & 15) >> 2);
        const int v_colb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
        const int sw = ((v_row >> 1) & 1) << 6;
        va[0] = lds0 + TB + v_row * 128 + (v_colb ^ sw);
        va[1] = lds0 + TB + v_row * 128 + ((64 + v_colb) ^ sw);
    }
    unsigned ka[4];
#define F5_KREAD(dst, n) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ka[(n) & 3]), "n"(((n) >> 2) * 32 * 128))
#define F5_VREAD(dst, s, mb, g) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(va[mb]), "n"((16 * (s) + 8 * (g)) * 128))
#define F5_LWAIT1(cnt, a) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(cnt))
#define F5_LWAIT2(cnt, a, b2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b2) : "n"(cnt))
#define F5_LWAIT4(cnt, a, b2, c2, d) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b2), "+v"(c2), "+v"(d) : "n"(cnt))

    f32x16 o_acc[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        m_run[j] = -1e30f;
        l_run[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) o_acc[j][0][i] = o_acc[j][1][i] = 0.f;
    }
    // ---- prologue: tiles 0 .. NBUF-2 on their way, tile 0 published
    constexpr int DIST = NBUF - 1;  // tiles requested ahead
#define F5_VMWAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    dma_tile(0, 0);
    if (nt > 1) dma_tile(KT, 1);
    {
        const int ahead = min(nt, DIST) - 1;  // tiles requested after tile 0
        if (ahead == 0) F5_VMWAIT(0);
        else if (ahead == 1) F5_VMWAIT(2 * PCS);
        else F5_VMWAIT(4 * PCS);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the validity words
    __builtin_amdgcn_s_barrier();

    constexpr float RANGE_GUARD = 18446744073709551616.0f;  // 2^64: per-lane row sums of one tile at or above this take the classic path
    int boff = 0, boff2 = DIST * BUF;  // LDS offsets of tile t's buffer and of tile t+DIST's
    for (int t = 0; t < nt; ++t) {
        if (t + DIST < nt) dma_tile((t + DIST) * KT, boff2 / BUF);
        ka[0] = ka_t;
        ka[1] = ka_t ^ 32u;
        ka[2] = ka_t ^ 64u;
        ka[3] = ka_t ^ 96u;

        // ---- S^T = K . Q^T
        unsigned long long vmv = ~0ull;  // validity of this tile's 64 keys (the same word in every lane)
        if constexpr (MASKED) {
            const unsigned ma = lds0 + NBUF * BUF + 8 * t;
            asm volatile("ds_read_b64 %0, %1" : "=v"(vmv) : "v"(ma));
        }
        bf16x8 kf[4];  // fragments 4..7 reuse the registers of 0..3 as soon as those MFMAs are issued
        F5_KREAD(kf[0], 0); F5_KREAD(kf[1], 1); F5_KREAD(kf[2], 2); F5_KREAD(kf[3], 3);
        f32x16 s[QB][2];
        auto qk = [&](auto& kfr, int n) {
#pragma unroll
            for (int j = 0; j < QB; ++j)
                s[j][n >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[j][n & 3], (n & 3) ? s[j][n >> 2] : f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        };
        // scheduling fences pin {wait, two MFMAs, next read} groups: left alone the compiler hoists all eight reads (32 registers) above the MFMAs
#define F5_FENCE() __builtin_amdgcn_sched_barrier(0)
        F5_FENCE();
        F5_LWAIT1(3, kf[0]); qk(kf[0], 0); F5_FENCE(); F5_KREAD(kf[0], 4);
        F5_LWAIT1(3, kf[1]); qk(kf[1], 1); F5_FENCE(); F5_KREAD(kf[1], 5);
        F5_LWAIT1(3, kf[2]); qk(kf[2], 2); F5_FENCE(); F5_KREAD(kf[2], 6);
        F5_LWAIT1(3, kf[3]); qk(kf[3], 3); F5_FENCE(); F5_KREAD(kf[3], 7);
        F5_LWAIT1(3, kf[0]); qk(kf[0], 4); F5_FENCE();
        F5_LWAIT1(2, kf[1]); qk(kf[1], 5); F5_FENCE();
        F5_LWAIT1(1, kf[2]); qk(kf[2], 6); F5_FENCE();
        F5_LWAIT1(0, kf[3]); qk(kf[3], 7); F5_FENCE();
        if constexpr (MASKED) F5_LWAIT1(0, vmv);  // ties the validity word to the waits above (it was the oldest read)
        // V^T fragments of the first PV step fly during the softmax
        bf16x4 vf[2][2];  // [mb][g]; the fragments of step s+1 reuse the registers as soon as the MFMAs of step s are issued
        F5_VREAD(vf[0][0], 0, 0, 0); F5_VREAD(vf[0][1], 0, 0, 1); F5_VREAD(vf[1][0], 0, 1, 0); F5_VREAD(vf[1][1], 0, 1, 1);

        auto apply_mask = [&]() {
            if constexpr (MASKED) {
                if (__builtin_amdgcn_ballot_w64(vmv != ~0ull) != 0ull) {
                    const unsigned long long vmh = h ? (vmv >> 4) : vmv;
#pragma unroll
                    for (int j = 0; j < QB; ++j)
#pragma unroll
                        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const int bit = 32 * kb + (i & 3) + 8 * (i >> 2);
                                if (!((vmh >> bit) & 1ull)) s[j][kb][i] = -INFINITY;
                            }
                }
            }
        };
        apply_mask();

        // ---- softmax numerators.  Common path: against the reference the query already has, no maximum.
        float rs[QB] = {0.f, 0.f};
        auto exps = [&](int j) {
            const float nm = -m_run[j];
            float ra = 0.f, rb = 0.f;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const float x0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][kb][i], c, nm));
                    const float x1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][kb][i + 1], c, nm));
                    s[j][kb][i] = x0;
                    s[j][kb][i + 1] = x1;
                    ra += x0;
                    rb += x1;
                }
            rs[j] = ra + rb;
        };
        auto row_max = [&](int j) {  // scaled maximum of query block j's 64 scores (both half-waves)
            float mx0 = max3_asm(s[j][0][0], s[j][0][1], s[j][0][2]), mx1 = max3_asm(s[j][1][0], s[j][1][1], s[j][1][2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) {
                mx0 = max3_asm(mx0, s[j][0][i], s[j][0][i + 1]);
                mx1 = max3_asm(mx1, s[j][1][i], s[j][1][i + 1]);
            }
            mx0 = max3_asm(mx0, mx1, s[j][0][15]);
            const float mt = max3_asm(mx0, s[j][1][15], s[j][1][15]);
            const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
            return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1])) * c;
        };
        if (t == 0) {  // first tile of the block: the reference is its exact maximum (O and l are still zero: nothing to rescale);
                       // -inf (a fully masked tile) leaves the finite start value alone
            m_run[0] = fmaxf(m_run[0], row_max(0));
            m_run[1] = fmaxf(m_run[1], row_max(1));
        }
        exps(0);
        exps(1);
        if (__builtin_amdgcn_ballot_w64(!(rs[0] < RANGE_GUARD) || !(rs[1] < RANGE_GUARD)) != 0ull) {
            // some numerator left the guarded range (a score 64 log2 units above its query's reference, inf or NaN): classic step.  The
            // scores were overwritten by the numerators: recompute them (the K tile is still in LDS), move the references to the exact
            // running maxima, rescale O and l
            bf16x8 k2;  // cold path: one fragment at a time
#define F5_REDO(n) F5_KREAD(k2, n); F5_LWAIT1(0, k2); qk(k2, n);
            F5_REDO(0) F5_REDO(1) F5_REDO(2) F5_REDO(3) F5_REDO(4) F5_REDO(5) F5_REDO(6) F5_REDO(7)
#undef F5_REDO
            apply_mask();
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const float m_new = fmaxf(m_run[j], row_max(j));
                const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
                m_run[j] = m_new;
                l_run[j] *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    o_acc[j][0][i] *= alpha;
                    o_acc[j][1][i] *= alpha;
                }
                exps(j);
            }
        }
        l_run[0] += rs[0];
        l_run[1] += rs[1];

        // ---- O^T += V^T . P^T: four 16-key steps; each half (32 dims) of the V^T fragment is re-requested for the next step right after
        //      its two MFMAs are issued
        auto pv_half = [&](int st, int mb, const bf16x8 (&pf)[QB]) {
            const bf16x8 vfr = __builtin_shufflevector(vf[mb][0], vf[mb][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int j = 0; j < QB; ++j) o_acc[j][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, pf[j], o_acc[j][mb], 0, 0, 0);
        };
#define F5_PV(st, more)                                                                       \
    {                                                                                         \
        bf16x8 pf[QB];                                                                        \
        pf[0] = pack8(s[0][(st) >> 1], 8 * ((st) & 1));                                       \
        pf[1] = pack8(s[1][(st) >> 1], 8 * ((st) & 1));                                       \
        F5_FENCE();                                                                           \
        F5_LWAIT2(2, vf[0][0], vf[0][1]);                                                     \
        pv_half(st, 0, pf);                                                                   \
        F5_FENCE();                                                                           \
        if constexpr (more) {                                                                 \
            F5_VREAD(vf[0][0], (st) + 1, 0, 0);                                               \
            F5_VREAD(vf[0][1], (st) + 1, 0, 1);                                               \
        }                                                                                     \
        F5_LWAIT2((more) ? 2 : 0, vf[1][0], vf[1][1]);                                        \
        pv_half(st, 1, pf);                                                                   \
        F5_FENCE();                                                                           \
        if constexpr (more) {                                                                 \
            F5_VREAD(vf[1][0], (st) + 1, 1, 0);                                               \
            F5_VREAD(vf[1][1], (st) + 1, 1, 1);                                               \
        }                                                                                     \
    }
        F5_PV(0, true)
        F5_PV(1, true)
        F5_PV(2, true)
        F5_PV(3, false)
#undef F5_PV

        // ---- publish tile t+1 (requested a whole tile ago) and retire this tile's buffer: every LDS read of this wave has been waited for
        {  // tile t+1 must have landed: the tiles requested after it may stay in flight
            const int ahead = min(nt - 1, t + DIST) - (t + 1);
            if (ahead <= 0) F5_VMWAIT(0);
            else if (ahead == 1) F5_VMWAIT(2 * PCS);
            else F5_VMWAIT(4 * PCS);
        }
        __builtin_amdgcn_s_barrier();
        const int step = boff + BUF == NBUF * BUF ? -(NBUF - 1) * BUF : BUF;  // wave-uniform
        boff += step;
        ka_t += step;
        va[0] += step;
        va[1] += step;
        boff2 = boff2 + BUF == NBUF * BUF ? 0 : boff2 + BUF;
    }
#undef F5_KREAD
#undef F5_VREAD
#undef F5_LWAIT1
#undef F5_LWAIT4
#undef F5_VMWAIT
#undef F5_LWAIT2
#undef F5_FENCE

    // ---- normalise; stage the wave's 64 x 64 output block through LDS (the ring is free: the loop ended on a barrier with no DMA in flight)
    //      and store whole 128-byte rows
    int lane_e = lane;  // laundered: keeps the epilogue's lane-constant addresses from being hoisted above the tile loop (and spilled around it)
    asm volatile("" : "+v"(lane_e));
    const int r_e = lane_e & 31, h_e = lane_e >> 5;
    char* stage = smem + wave * (64 * 128);
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(o_acc[j][mb][4 * g + e] * inv);
                const int row = 32 * j + r_e, chunk = (4 * mb + g) ^ (row & 7);
                *reinterpret_cast<bf16x4*>(stage + row * 128 + chunk * 16 + 8 * h_e) = v4;
            }
    }
    {
        bf16_t* obase = out + ((size_t)b * bs + q0) * ldo + head * 64;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = 8 * k + (lane_e >> 3), chunk = lane_e & 7;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(stage + row * 128 + ((chunk ^ (row & 7)) * 16));
            if (q0 + row < N) *reinterpret_cast<u32x4_t*>(obase + (size_t)row * ldo + chunk * 8) = v;
        }
    }
}


// attn_persist_kernel: the 64-queries-per-wave loop above on a PERSISTENT grid (two workgroups per CU walk (batch, head, 256-query block) items).
// What the one-item-per-workgroup form pays per item and this form hides (about 8 us of a workgroup's 39 us at N = 1024):
//   * the K/V ring never drains: the DMA cursor runs two tiles ahead of the tile being consumed ACROSS item boundaries, so an item's first
//     tiles are in LDS when the previous item's epilogue ends (no cold prologue, no workgroup launch per item);
//   * Q arrives by LDS-DMA into a buffer of its own (32 KiB: 80 KiB per workgroup, two per CU) a whole item ahead -- each wave fetches and
//     reads only its own 64 query rows, so only its own counted vmcnt orders it -- instead of a global load waited for at workgroup start;
//   * the normalised output is staged through the ring slot the item's last tile just freed -- each wave through exactly the four 1-KiB
//     pieces IT will refill by DMA next (program order is the only ordering needed) -- and leaves as whole 128-byte rows.
// Unmasked sequences of whole 256-query blocks only (the launcher keeps the kernel above for everything else).  vmcnt bookkeeping: loads,
// stores and LDS-DMA share the counter and retire in order; at the end of a tile the K/V pieces of the NEXT tile must have landed, and what
// this wave issued after them may stay in flight: 4 pieces of the refill of this step, plus -- on the first tile of an item -- the previous
// item's 8 output stores and the 8 Q pieces of the item after this one.
template <bool STAMP>
__global__ __launch_bounds__(256, 2) void attn_persist_kernel(const bf16_t* __restrict__ qkv, int ldq, int inner, bf16_t* __restrict__ out, int ldo,
                                                              int N, float c, int NQ, int H, int BH, int stagger_ticks, unsigned long long* dbg) {
    constexpr int KT = 64, QB = 2, TB = KT * 128, NBUF = 3, BUF = 2 * TB, WAVES = 4, PCS = 2;
    constexpr int QOFF = NBUF * BUF;  // Q buffer: 4 waves x 64 queries x 128 B
    __shared__ __attribute__((aligned(1024))) char smem[NBUF * BUF + WAVES * 64 * 128];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x, wg = blockIdx.x;
    const int nt = N / KT;
    // ---- items of this workgroup.  XCD-aware order (blocks b and b + 8 share an XCD; speed only): the (batch, head) pairs congruent to this
    //      workgroup's XCD label, query block fastest, dealt round-robin over the XCD's workgroups -- the NQ blocks of one pair run together
    const bool by_xcd = (BH & 7) == 0 && (G & 7) == 0;
    const int items_all = BH * NQ;
    const int first = by_xcd ? (wg >> 3) : wg, stride = by_xcd ? (G >> 3) : G, span = by_xcd ? items_all >> 3 : items_all;
    const int n_my = first < span ? (span - first + stride - 1) / stride : 0;
    if (n_my == 0) return;
    if constexpr (STAMP) {  // (shader clock, 100 MHz real-time clock) at workgroup start: in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0) {
            dbg[(size_t)G * 64 + wg * 4 + 0] = t0;
            dbg[(size_t)G * 64 + wg * 4 + 1] = r0;
        }
    }
    stagger_delay(wg, G >> 1, G, stagger_ticks);
    auto item_bh_q = [&](int k, int& bh, int& qblk) {
        const int li = first + k * stride;
        qblk = li % NQ;
        bh = by_xcd ? (li / NQ) * 8 + (wg & 7) : li / NQ;
    };
    const int r = lane & 31, h = lane >> 5;
    const unsigned ldq2 = (unsigned)ldq * 2u;
    const int total = n_my * nt;

    // ---- K/V DMA cursor (wave-uniform): next tile to request and the ring slot it goes to
    const int drow = lane >> 3, dchunk = lane & 7;
    unsigned kco[PCS], vco[PCS];
#pragma unroll
    for (int pc = 0; pc < PCS; ++pc) {
        const int row = (wv + WAVES * pc) * 8 + drow;
        kco[pc] = (unsigned)((dchunk ^ ((row >> 1) & 7)) << 4) + (unsigned)row * ldq2;
        vco[pc] = (unsigned)((dchunk ^ (((row >> 1) & 1) << 2)) << 4) + (unsigned)row * ldq2;
    }
    int cur_item = 0, cur_t = 0, cur_slot = 0, cur_g = 0;
    const char *cur_k, *cur_v;
    auto cursor_item = [&](int k) {
        int bh, qb;
        item_bh_q(k, bh, qb);
        const int b = bh / H, head = bh - b * H;
        const bf16_t* base = qkv + (size_t)b * N * ldq + head * 64;
        cur_k = reinterpret_cast<const char*>(base + inner);
        cur_v = reinterpret_cast<const char*>(base + 2 * inner);
    };
    cursor_item(0);
    auto dma_next = [&]() {  // requests global tile cur_g (caller checks cur_g < total)
        const char* kt = cur_k + (size_t)(cur_t * KT) * ldq2;
        const char* vt = cur_v + (size_t)(cur_t * KT) * ldq2;
#pragma unroll
        for (int pc = 0; pc < PCS; ++pc) {
            char* dst = smem + cur_slot * BUF + (wv + WAVES * pc) * 1024;
            __builtin_amdgcn_global_load_lds((gptr_t)(kt + kco[pc]), (lptr_t)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(vt + vco[pc]), (lptr_t)(dst + TB), 16, 0, 0);
        }
        ++cur_g;
        cur_slot = cur_slot + 1 == NBUF ? 0 : cur_slot + 1;
        if (++cur_t == nt) {
            cur_t = 0;
            if (++cur_item < n_my) cursor_item(cur_item);
        }
    };
    // ---- Q of item k -> this wave's 8 KiB of the Q buffer (8 pieces of 8 query rows, K-style swizzle on the source address)
    auto dma_q = [&](int k) {
        int bh, qb;
        item_bh_q(k, bh, qb);
        const int b = bh / H, head = bh - b * H;
        const char* qsrc = reinterpret_cast<const char*>(qkv + ((size_t)b * N + (size_t)qb * 256 + wv * 64) * ldq + head * 64);
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int row = pc * 8 + drow;
            const unsigned off = (unsigned)((dchunk ^ ((row >> 1) & 7)) << 4) + (unsigned)row * ldq2;
            __builtin_amdgcn_global_load_lds((gptr_t)(qsrc + off), (lptr_t)(smem + QOFF + wv * 8192 + pc * 1024), 16, 0, 0);
        }
    };

    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned ka_t = lds0 + r * 128 + ((h ^ ((r >> 1) & 7)) << 4);
    unsigned va[2];
    {
        const int v_row = 4 * h + ((lane & 15) >> 2);
        const int v_colb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
        const int sw = ((v_row >> 1) & 1) << 6;
        va[0] = lds0 + TB + v_row * 128 + (v_colb ^ sw);
        va[1] = lds0 + TB + v_row * 128 + ((64 + v_colb) ^ sw);
    }
    unsigned ka[4];
#define F5_KREAD(dst, n) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(ka[(n) & 3]), "n"(((n) >> 2) * 32 * 128))
#define F5_VREAD(dst, s, mb, g) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(va[mb]), "n"((16 * (s) + 8 * (g)) * 128))
#define F5_LWAIT1(cnt, a) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(cnt))
#define F5_LWAIT2(cnt, a, b2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b2) : "n"(cnt))
#define F5_FENCE() __builtin_amdgcn_sched_barrier(0)
    // counted vmcnt with a wave-uniform count (multiples of 4 up to 20)
    auto vm_wait = [&](int n) {
        switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        }
    };

    // ---- prologue: Q of the first item, tiles 0 and 1; Q and tile 0 landed, tile 0 published
    dma_q(0);
    dma_next();
    if (total > 1) dma_next();
    vm_wait(total > 1 ? 4 : 0);
    __builtin_amdgcn_s_barrier();

    constexpr float RANGE_GUARD = 18446744073709551616.0f;  // 2^64 (see attn_wide_kernel)
    int boff = 0;  // LDS offset of the slot of the tile being consumed
    bf16x8 qf[QB][4];
    for (int it = 0; it < n_my; ++it) {
        // ---- item start: this wave's Q fragments out of its part of the Q buffer, then the NEXT item's Q on its way into the same bytes
        {
            int lane_q = lane;
            asm volatile("" : "+v"(lane_q));  // (laundered: keeps these addresses out of the registers that live across the tile loop)
            const int rq = lane_q & 31, hq = lane_q >> 5;
            const unsigned qa = lds0 + QOFF + wv * 8192 + rq * 128 + ((hq ^ ((rq >> 1) & 7)) << 4);  // slice 0 of query block 0; slice ds: bits 5..6 ^= ds
            const unsigned qa1 = qa ^ 32u, qa2 = qa ^ 64u, qa3 = qa ^ 96u;
#define F5_QREAD(j, ds, a) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qf[j][ds]) : "v"(a), "n"((j) * 32 * 128))
            F5_QREAD(0, 0, qa); F5_QREAD(0, 1, qa1); F5_QREAD(0, 2, qa2); F5_QREAD(0, 3, qa3);
            F5_QREAD(1, 0, qa); F5_QREAD(1, 1, qa1); F5_QREAD(1, 2, qa2); F5_QREAD(1, 3, qa3);
#undef F5_QREAD
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(qf[0][0]), "+v"(qf[0][1]), "+v"(qf[0][2]), "+v"(qf[0][3]), "+v"(qf[1][0]), "+v"(qf[1][1]), "+v"(qf[1][2]), "+v"(qf[1][3]));
        F5_FENCE();
        auto stamp = [&](int k) {  // diagnostic build only: shader-clock stamps of wave 0 (item start, tiles 0 / 1 / last done, epilogue done)
            if constexpr (STAMP) {
                if (wv == 0 && it < 8) {
                    const unsigned long long tnow = __builtin_amdgcn_s_memtime();
                    if (lane == 0) dbg[((size_t)wg * 8 + it) * 8 + k] = tnow;
                }
            }
        };
        stamp(0);
        const bool q_ahead = it + 1 < n_my;
        if (q_ahead) dma_q(it + 1);
        stamp(1);

        f32x16 o_acc[QB][2];
        float m_run[QB], l_run[QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            m_run[j] = -1e30f;
            l_run[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) o_acc[j][0][i] = o_acc[j][1][i] = 0.f;
        }
        for (int t = 0; t < nt; ++t) {
            const bool refill = cur_g < total;
            if (refill) dma_next();
            ka[0] = ka_t;
            ka[1] = ka_t ^ 32u;
            ka[2] = ka_t ^ 64u;
            ka[3] = ka_t ^ 96u;
            // ---- S^T = K . Q^T
            bf16x8 kf[4];
            F5_KREAD(kf[0], 0); F5_KREAD(kf[1], 1); F5_KREAD(kf[2], 2); F5_KREAD(kf[3], 3);
            f32x16 s[QB][2];
            auto qk = [&](auto& kfr, int n) {
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    s[j][n >> 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr, qf[j][n & 3], (n & 3) ? s[j][n >> 2] : f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            };
            F5_FENCE();
            F5_LWAIT1(3, kf[0]); qk(kf[0], 0); F5_FENCE(); F5_KREAD(kf[0], 4);
            F5_LWAIT1(3, kf[1]); qk(kf[1], 1); F5_FENCE(); F5_KREAD(kf[1], 5);
            F5_LWAIT1(3, kf[2]); qk(kf[2], 2); F5_FENCE(); F5_KREAD(kf[2], 6);
            F5_LWAIT1(3, kf[3]); qk(kf[3], 3); F5_FENCE(); F5_KREAD(kf[3], 7);
            F5_LWAIT1(3, kf[0]); qk(kf[0], 4); F5_FENCE();
            F5_LWAIT1(2, kf[1]); qk(kf[1], 5); F5_FENCE();
            F5_LWAIT1(1, kf[2]); qk(kf[2], 6); F5_FENCE();
            F5_LWAIT1(0, kf[3]); qk(kf[3], 7); F5_FENCE();
            bf16x4 vf[2][2];
            F5_VREAD(vf[0][0], 0, 0, 0); F5_VREAD(vf[0][1], 0, 0, 1); F5_VREAD(vf[1][0], 0, 1, 0); F5_VREAD(vf[1][1], 0, 1, 1);

            // ---- softmax numerators against the reference the query already has (no maximum on the common path; attn_wide_kernel)
            float rs[QB] = {0.f, 0.f};
            auto exps = [&](int j) {
                const float nm = -m_run[j];
                float ra = 0.f, rb = 0.f;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; i += 2) {
                        const float x0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][kb][i], c, nm));
                        const float x1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][kb][i + 1], c, nm));
                        s[j][kb][i] = x0;
                        s[j][kb][i + 1] = x1;
                        ra += x0;
                        rb += x1;
                    }
                rs[j] = ra + rb;
            };
            auto row_max = [&](int j) {
                float mx0 = max3_asm(s[j][0][0], s[j][0][1], s[j][0][2]), mx1 = max3_asm(s[j][1][0], s[j][1][1], s[j][1][2]);
#pragma unroll
                for (int i = 3; i < 15; i += 2) {
                    mx0 = max3_asm(mx0, s[j][0][i], s[j][0][i + 1]);
                    mx1 = max3_asm(mx1, s[j][1][i], s[j][1][i + 1]);
                }
                mx0 = max3_asm(mx0, mx1, s[j][0][15]);
                const float mt = max3_asm(mx0, s[j][1][15], s[j][1][15]);
                const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
                return fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1])) * c;
            };
            if (t == 0) {
                m_run[0] = fmaxf(m_run[0], row_max(0));
                m_run[1] = fmaxf(m_run[1], row_max(1));
            }
            exps(0);
            exps(1);
            if (__builtin_amdgcn_ballot_w64(!(rs[0] < RANGE_GUARD) || !(rs[1] < RANGE_GUARD)) != 0ull) {
                bf16x8 k2;
#define F5_REDO(n) F5_KREAD(k2, n); F5_LWAIT1(0, k2); qk(k2, n);
                F5_REDO(0) F5_REDO(1) F5_REDO(2) F5_REDO(3) F5_REDO(4) F5_REDO(5) F5_REDO(6) F5_REDO(7)
#undef F5_REDO
#pragma unroll
                for (int j = 0; j < QB; ++j) {
                    const float m_new = fmaxf(m_run[j], row_max(j));
                    const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
                    m_run[j] = m_new;
                    l_run[j] *= alpha;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        o_acc[j][0][i] *= alpha;
                        o_acc[j][1][i] *= alpha;
                    }
                    exps(j);
                }
            }
            l_run[0] += rs[0];
            l_run[1] += rs[1];

            // ---- O^T += V^T . P^T
            auto pv_half = [&](int mb, const bf16x8 (&pf)[QB]) {
                const bf16x8 vfr = __builtin_shufflevector(vf[mb][0], vf[mb][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int j = 0; j < QB; ++j) o_acc[j][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr, pf[j], o_acc[j][mb], 0, 0, 0);
            };
#define F5_PV(st, more)                                                                       \
    {                                                                                         \
        bf16x8 pf[QB];                                                                        \
        pf[0] = pack8(s[0][(st) >> 1], 8 * ((st) & 1));                                       \
        pf[1] = pack8(s[1][(st) >> 1], 8 * ((st) & 1));                                       \
        F5_FENCE();                                                                           \
        F5_LWAIT2(2, vf[0][0], vf[0][1]);                                                     \
        pv_half(0, pf);                                                                       \
        F5_FENCE();                                                                           \
        if constexpr (more) {                                                                 \
            F5_VREAD(vf[0][0], (st) + 1, 0, 0);                                               \
            F5_VREAD(vf[0][1], (st) + 1, 0, 1);                                               \
        }                                                                                     \
        F5_LWAIT2((more) ? 2 : 0, vf[1][0], vf[1][1]);                                        \
        pv_half(1, pf);                                                                       \
        F5_FENCE();                                                                           \
        if constexpr (more) {                                                                 \
            F5_VREAD(vf[1][0], (st) + 1, 1, 0);                                               \
            F5_VREAD(vf[1][1], (st) + 1, 1, 1);                                               \
        }                                                                                     \
    }
            F5_PV(0, true)
            F5_PV(1, true)
            F5_PV(2, true)
            F5_PV(3, false)
#undef F5_PV

            // ---- publish the next tile: what this wave issued AFTER its K/V pieces may stay in flight (see the kernel comment)
            vm_wait((refill ? 2 * PCS : 0) + (t == 0 ? (it > 0 ? 8 : 0) + (q_ahead ? 8 : 0) : 0));
            __builtin_amdgcn_s_barrier();
            if constexpr (STAMP) {
                if (t == 0) stamp(2);
                if (t == 1) stamp(3);
                if (t == 2) stamp(4);
                if (t == nt - 1) stamp(5);
            }
            if (t + 1 < nt) {  // (after an item's last tile the freed slot first serves the epilogue)
                const int step = boff + BUF == NBUF * BUF ? -(NBUF - 1) * BUF : BUF;
                boff += step;
                ka_t += step;
                va[0] += step;
                va[1] += step;
            }
        }
        // ---- epilogue of the item: normalise, stage 32 rows at a time through this wave's own four DMA pieces of the slot the last tile
        //      freed (pieces wv, wv + 4 of its K half and of its V half), store whole 128-byte rows
        {
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            const int r_e = lane_e & 31, h_e = lane_e >> 5;
            int bh, qb;
            item_bh_q(it, bh, qb);
            const int b = bh / H, head = bh - b * H;
            bf16_t* obase = out + ((size_t)b * N + (size_t)qb * 256 + wv * 64) * ldo + head * 64;
            const unsigned sbase = lds0 + boff + wv * 1024;
            // staging row rr (0..31) lives in piece rr >> 3: +4096 for odd pieces (wv + 4), +TB for pieces 2, 3 (the V half)
            const unsigned wrow = sbase + ((r_e >> 3) & 1) * 4096 + ((r_e >> 4) & 1) * TB + (r_e & 7) * 128 + 8 * h_e;
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
                const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        bf16x4 v4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(o_acc[j][mb][4 * g + e] * inv);
                        const unsigned wa = wrow + (((4 * mb + g) ^ (r_e & 7)) << 4);
                        asm volatile("ds_write_b64 %0, %1" ::"v"(wa), "v"(v4) : "memory");
                    }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                u32x4_t rv[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {  // piece k holds rows 8k .. 8k+7 of this pass
                    const int rr = lane_e >> 3, chunk = lane_e & 7;
                    const unsigned ra = sbase + (k & 1) * 4096 + (k >> 1) * TB + rr * 128 + ((chunk ^ rr) << 4);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(rv[k]) : "v"(ra));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rv[0]), "+v"(rv[1]), "+v"(rv[2]), "+v"(rv[3]));
                F5_FENCE();
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = 32 * j + 8 * k + (lane_e >> 3), chunk = lane_e & 7;
                    *reinterpret_cast<u32x4_t*>(obase + (size_t)row * ldo + chunk * 8) = rv[k];
                }
            }
        }
        if constexpr (STAMP) {
            if (wv == 0 && it < 8) {
                const unsigned long long tnow = __builtin_amdgcn_s_memtime();
                if (lane == 0) dbg[((size_t)wg * 8 + it) * 8 + 6] = tnow;
            }
        }
        {  // the slot of the next item's first tile
            const int step = boff + BUF == NBUF * BUF ? -(NBUF - 1) * BUF : BUF;
            boff += step;
            ka_t += step;
            va[0] += step;
            va[1] += step;
        }
    }
#undef F5_KREAD
#undef F5_VREAD
#undef F5_LWAIT1
#undef F5_LWAIT2
#undef F5_FENCE
}

int g_attn_variant = 0;  // tuning knob ("attn_variant"): 0 = by grid size, 2 = 64 queries per wave (one item per workgroup), 6 = the same on the persistent grid, 5 = software-pipelined 32 queries per wave

unsigned long long* g_attn_stamp_buf = nullptr;  // diagnostic (f5_debug_attn_stamps): device buffer [workgroups][8 items][8 stamps] of shader-clock stamps
int g_attn_stagger = 0;  // tuning knob ("attn_stagger"): start delay of the second resident workgroup of every CU, in 10-ns ticks per 64-key tile of the sequence (0 = off)
// tuning knob ("attn_persist"): eligible launches of the 64-queries-per-wave kernel run on the persistent grid.  OFF by default: measured in one
// process against the one-item-per-workgroup kernel (profiles/r3_attention_persistent_grid.txt) it is 3-5 % SLOWER at C2 and C4 -- the per-item
// cost it was built to hide is not launch / prologue latency (in-kernel stamps: 6 000 of an item's 57 000 cycles at N = 1024, the same as
// before), and its per-tile cost is 5 % higher (static item split, counted-wait switch).  Kept (bit-identical results, tests, attn_variant 6).
int g_attn_persist = 0;

bool attention_fast_supported(int precision, int N, int H) { return precision == F5_PREC_BF16 && N >= 1 && H >= 1; }

int launch_attention_pipe(int waves, int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out, int ldo, hipStream_t stream, int bstride);  // attention_pipe.hip

int launch_attention_pipe_segs(bool masked, int nbr, const AttnSegs& segs, int maxN, int H, const void* qkv, int ldq, void* out, int ldo, hipStream_t stream,
                               int bstride);  // attention_pipe.hip

static int cu_count_cached() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    return cus;
}
// which of the two tuned kernels launch_attention_fast gives a (B, N, H) problem
static bool picks_wide(int B, int N, int H, bool masked, int ldq, int bstride) {
    // 256 queries per workgroup need at least one workgroup per CU to pay; below that (single-utterance serving) the 128-query
    // workgroups of the pipelined kernel fill the chip better (B = 1: 17 vs 23 us)
    bool wide = g_attn_variant == 2 || g_attn_variant == 6 || (g_attn_variant == 0 && (long)B * H * cdiv(N, 256) >= cu_count_cached());
    if (masked && N > 64 * 128) wide = false;  // the wide kernel's table of key validity bits holds 128 tiles
    if ((size_t)N * (size_t)ldq * 2u >= (1ull << 32)) wide = false;  // its per-lane key offsets are 32-bit
    if ((size_t)bstride * (size_t)ldq * 2u >= (1ull << 32)) wide = false;
    return wide;
}

int launch_attention_ragged(int precision, int attn_kernel_opt, const AttnSegs& segs, int H, const void* qkv, int ldq, void* out, int ldo, hipStream_t stream,
                            int bstride) {
    const size_t es = precision == F5_PREC_BF16 ? 2 : 4;
    AttnSegs grp[2];  // [0] utterances with n % 64 == 0 (unmasked build), [1] the others -- as each one's own launch would pick
    int maxn[2] = {0, 0};
    for (int u = 0; u < segs.cnt; ++u) {
        const int nu = segs.n[u];
        const int kind = (attn_kernel_opt != 0 && attention_fast_supported(precision, nu, H)) ? 1 : 0;
        const bool masked = (nu % 64) != 0;
        if (kind == 1 && (ldq & 7) == 0 && (ldo & 7) == 0 && !picks_wide(segs.nbr, nu, H, masked, ldq, bstride)) {
            AttnSegs& g = grp[masked ? 1 : 0];
            g.nbr = segs.nbr;
            g.off[g.cnt] = segs.off[u];
            g.n[g.cnt++] = nu;
            maxn[masked ? 1 : 0] = std::max(maxn[masked ? 1 : 0], nu);
        } else {  // its own launch: the wide kernel, or the reference kernel
            F5_TRY(launch_attention(precision, kind, segs.nbr, nu, H, (const char*)qkv + (size_t)segs.off[u] * ldq * es, ldq, nullptr,
                                    (char*)out + (size_t)segs.off[u] * ldo * es, ldo, stream, bstride));
        }
    }
    for (int k = 0; k < 2; ++k)
        if (grp[k].cnt > 0) F5_TRY(launch_attention_pipe_segs(k == 1, segs.nbr, grp[k], maxn[k], H, qkv, ldq, out, ldo, stream, bstride));
    return 0;
}

int launch_attention_fast(int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out, int ldo, hipStream_t stream, int bstride) {
    if ((ldq & 7) || (ldo & 7)) return f5_fail(F5_EINVAL, "attention_fast: ldq and ldo must be multiples of 8");
    const int cus = cu_count_cached();
    const bool masked = mask != nullptr || (N % 64) != 0;
    const bool wide = picks_wide(B, N, H, masked, ldq, bstride);
    if (!wide) return launch_attention_pipe(4, B, N, H, qkv, ldq, mask, out, ldo, stream, bstride);
    const float c = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
    // persistent grid (two workgroups per CU walking items, K/V ring and Q prefetch running on across items): unmasked whole 256-query blocks,
    // at least three key tiles per item (the Q prefetch and the output stores are retired by the second tile's counted wait) and at least
    // two items per workgroup
    // start stagger (see stagger_delay): only when every CU holds two workgroups for several rounds
    const long wgs = (long)B * H * cdiv(N, 256);
    const int stagger = (g_attn_stagger > 0 && wgs >= 4L * cus) ? g_attn_stagger * cdiv(N, 64) : 0;
    const long items = (long)B * H * (N / 256);
    const bool persist_shape = !masked && N % 256 == 0 && N / 64 >= 3 && items < (1L << 30) && bstride == N;
    if (persist_shape && (g_attn_variant == 6 || (g_attn_variant == 0 && g_attn_persist && items >= 4L * cus))) {
        const int G = (int)(items < 2L * cus ? items : 2L * cus);  // (items is a multiple of 8 whenever B * H is: the kernel's XCD-aware order applies)
        if (g_attn_stamp_buf)
            hipLaunchKernelGGL(attn_persist_kernel<true>, dim3(G), dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, (bf16_t*)out, ldo, N, c, N / 256, H,
                               B * H, stagger, g_attn_stamp_buf);
        else
            hipLaunchKernelGGL(attn_persist_kernel<false>, dim3(G), dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, (bf16_t*)out, ldo, N, c, N / 256, H,
                               B * H, stagger, (unsigned long long*)nullptr);
        F5_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(cdiv(N, 256), H, B);
    if (masked)
        hipLaunchKernelGGL((attn_wide_kernel<true>), grid, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, bstride, c, cus, 2 * cus, stagger);
    else
        hipLaunchKernelGGL((attn_wide_kernel<false>), grid, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, bstride, c, cus, 2 * cus, stagger);
    F5_LAUNCH_CHECK();
    return 0;
}
