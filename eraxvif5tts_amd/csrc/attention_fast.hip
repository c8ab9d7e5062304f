// attention_fast.hip -- flash attention for the DiT blocks on gfx950 (bf16 in/out, head dim 64, non-causal, key-padding mask).
//
// Structure (see /opt/skills/guides/cdna_hip_programming.md, Appendix B "Fused attention prefill"):
//   * one workgroup = 4 wavefronts = 128 queries of one (batch, head); each wave owns 32 queries; K/V tiles of 64 keys
//     are double-buffered in LDS (register-staged: the global loads of tile t+2 are issued right after the barrier of
//     tile t and written to LDS at the end of tile t+1, so their latency hides under a whole tile of MFMA work);
//   * swapped QK^T: S^T = K.Q^T with v_mfma_f32_32x32x16_bf16, so a lane holds 16 keys x ONE query per 32-key block:
//     the softmax max/sum are in-register (one cross-half exchange per tile), and the un-normalised P converted to bf16
//     is directly the B operand of the PV product (accumulator-as-operand k order);
//   * O^T = V^T.P^T: the V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose);
//   * K tile XOR-swizzled for conflict-free ds_read_b128, V tile swizzled for the transposed reads;
//   * exp2 with the softmax scale folded into one fma; fp32 running max / sum / output accumulators.
#include "kernels.h"

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;

__device__ __forceinline__ bf16x8 pack8(const f32x16& s, int base) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)s[base + j];
    return r;
}

// ABL: timing-only ablations (wrong results): 1 = no softmax math (P = S), 2 = no K/V tile refresh (tile 0 reused, no loads/stores/barriers),
// 3 = no PV MFMAs, 4 = no QK^T MFMAs
// OCC: waves per SIMD the register allocation must allow (3: 152 VGPRs as the compiler likes it; 4: capped at 128, a few spills)
template <bool MASKED, int ABL, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void attn_fast_kernel(const bf16_t* __restrict__ qkv, int ldq, int inner, const uint8_t* __restrict__ mask,
                                                           bf16_t* __restrict__ out, int ldo, int N, float c /* scale * log2(e) */) {
    constexpr int KT = 64;                 // keys per tile
    constexpr int TILE_BYTES = KT * 128;   // 64 keys x 64 dims x 2 B
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][K | V]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block -> (batch*head, query block) map: blocks id and id+8 share an XCD (and its L2), so all query blocks of one
    // (batch, head) are given ids with equal id % 8: its K/V (256 KiB at N = 1024) are then fetched from HBM once per XCD pass
    // instead of once per query block.  Falls back to the plain order when B*H is not a multiple of 8.
    const int QB = gridDim.x, BH = gridDim.y * gridDim.z;
    int qb = blockIdx.x, bh = blockIdx.y + blockIdx.z * gridDim.y;
    if ((BH & 7) == 0) {
        const int id = blockIdx.x + QB * bh;  // linear dispatch order (x fastest)
        const int xcd = id & 7, j = id >> 3;
        qb = j % QB;
        bh = (j / QB) * 8 + xcd;
    }
    const int b = bh / gridDim.y, head = bh - b * gridDim.y, q0 = qb * 128 + wave * 32;
    const int r = lane & 31, h = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * N * ldq + head * 64;
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

    // ---- K/V tile staging: thread t moves chunks t and t+256 (row = chunk>>3, 16-byte column = chunk&7) of K and of V
    const int srow0 = tid >> 3, scol = tid & 7;  // rows srow0 and srow0 + 32
    bf16x8 kreg[2], vreg[2];
    uint8_t mreg = 1;
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = k0 + srow0 + 32 * i;
            if (row >= N) row = N - 1;
            kreg[i] = *reinterpret_cast<const bf16x8*>(kbase + (size_t)row * ldq + scol * 8);
            vreg[i] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)row * ldq + scol * 8);
        }
        if constexpr (MASKED) {
            const int key = k0 + lane;
            mreg = key < N ? (mask ? mask[(size_t)b * N + key] : (uint8_t)1) : (uint8_t)0;
        }
    };
    auto store_tile = [&](int buf) {
        char* kb = smem + buf * 2 * TILE_BYTES;
        char* vb = kb + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow0 + 32 * i;
            *reinterpret_cast<bf16x8*>(kb + row * 128 + ((scol ^ ((row >> 1) & 7)) << 4)) = kreg[i];
            *reinterpret_cast<bf16x8*>(vb + row * 128 + ((scol ^ (((row >> 1) & 1) << 2)) << 4)) = vreg[i];
        }
    };

    // ---- per-lane LDS read offsets
    // K fragment (A operand of S^T): row = 32*kb + r, logical chunk = 2*ds + h
    int k_off[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) k_off[ds] = r * 128 + (((2 * ds + h) ^ ((r >> 1) & 7)) << 4);
    // V^T fragment via ds_read_b64_tr_b16: this lane supplies the address of key row (4*h + ((lane&15)>>2)) [+ 8*g + 16*s + 32*kb],
    // d columns 16*((lane>>4)&1) + 4*(lane&3) [+ 32*mb]
    const int v_row = 4 * h + ((lane & 15) >> 2);
    const int v_colb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;  // byte column inside the 64-byte half mb

    f32x16 o_acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) o_acc[0][i] = o_acc[1][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nt = (N + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    unsigned long long vm = ~0ull;
    if constexpr (MASKED) vm = __ballot(mreg != 0);
    __syncthreads();
    if (ABL != 2 && nt > 1) load_tile(KT);

    for (int t = 0; t < nt; ++t) {
        const char* kb_lds = smem + (ABL == 2 ? 0 : (t & 1)) * 2 * TILE_BYTES;
        const char* vb_lds = kb_lds + TILE_BYTES;

        // ---- S^T = K . Q^T for the two 32-key blocks of the tile
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_lds + kb * 32 * 128 + k_off[ds]);
                if constexpr (ABL != 4) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], s[kb], 0, 0, 0);
                else asm volatile("" ::"v"(kf));
            }
        }
        // ---- online softmax (this lane: one query, keys 32*kb + (i&3) + 8*(i>>2) + 4*h)
        if constexpr (MASKED) {
            if (vm != ~0ull) {
                const unsigned long long vmh = h ? (vm >> 4) : vm;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int bit = 32 * kb + (i & 3) + 8 * (i >> 2);
                        if (!((vmh >> bit) & 1ull)) s[kb][i] = -INFINITY;
                    }
            }
        }
        if constexpr (ABL != 1) {
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mt = fmaxf(mt, s[kb][i]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));  // the other half-wave holds the other 32 keys of the same query
        const float m_new = fmaxf(m_run, mt * c);  // -inf * c stays -inf; m_run starts finite
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32): two scores per VALU instruction; only the exp2 itself is scalar
        const f32x2 c2 = {c, c}, nm2 = {-m_new, -m_new};
        f32x2 rs2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 x = __builtin_elementwise_fma(f32x2{s[kb][i], s[kb][i + 1]}, c2, nm2);
                x[0] = __builtin_amdgcn_exp2f(x[0]);
                x[1] = __builtin_amdgcn_exp2f(x[1]);
                s[kb][i] = x[0];
                s[kb][i + 1] = x[1];
                rs2 += x;
            }
        const float rs = rs2[0] + rs2[1];
        l_run = l_run * alpha + rs;
        // the running max settles after the first tiles: skip the 32-register rescale when no query of this wave moved its max
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o_acc[0][i] *= alpha;
                o_acc[1][i] *= alpha;
            }
        }
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pf = pack8(s[kb], 8 * ks);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    bf16x8 vf;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int row = 32 * kb + 16 * ks + 8 * g + v_row;
                        const int colb = (64 * mb + v_colb) ^ (((row >> 1) & 1) << 6);
                        const bf16x4 part = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vb_lds + row * 128 + colb));
                        vf[4 * g + 0] = part[0];
                        vf[4 * g + 1] = part[1];
                        vf[4 * g + 2] = part[2];
                        vf[4 * g + 3] = part[3];
                    }
                    if constexpr (ABL != 3) o_acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[mb], 0, 0, 0);
                    else asm volatile("" ::"v"(vf), "v"(pf));
                }
            }
        // ---- hand the prefetched tile t+1 to LDS, then start fetching tile t+2
        if constexpr (ABL != 2) {
            if (t + 1 < nt) {
                store_tile((t + 1) & 1);
                if constexpr (MASKED) vm = __ballot(mreg != 0);
            }
            __syncthreads();
            if (t + 2 < nt) load_tile((t + 2) * KT);
        }
    }

    // ---- normalise and store: lane holds query q0 + r, dims 32*mb + (i&3) + 8*(i>>2) + 4*h
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int qrow = q0 + r;
    if (qrow < N) {
        bf16_t* op = out + ((size_t)b * N + qrow) * ldo + head * 64 + 4 * h;
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

// ----------------------------------------------------------------------------- LDS-DMA staged K/V tiles (experiment, attn_variant 3)
// Same algorithm as attn_fast_kernel, but the K/V tiles go L2 -> LDS with global_load_lds_dwordx4 (whole 128-byte rows, swizzle applied
// on the source address) through THREE tile buffers instead of two buffers + 16 staging registers per lane; the key-padding mask of
// the whole sequence is copied into LDS once (N <= 4096), so the loop issues no other vector-memory instruction and the DMA waits
// can be counted exactly.
typedef const __attribute__((address_space(1))) void* attn_gptr_t;
typedef __attribute__((address_space(3))) void* attn_lptr_t;
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_fast3_kernel(const bf16_t* __restrict__ qkv, int ldq, int inner, const uint8_t* __restrict__ mask,
                                                           bf16_t* __restrict__ out, int ldo, int N, float c /* scale * log2(e) */) {
    constexpr int KT = 64;                 // keys per tile
    constexpr int TILE_BYTES = KT * 128;   // 64 keys x 64 dims x 2 B
    constexpr int MASK_BYTES = MASKED ? 4096 : 16;
    __shared__ __attribute__((aligned(16))) char smem[6 * TILE_BYTES + MASK_BYTES];  // [buf 0..2][K | V], mask bytes
    constexpr int ABL = 0;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware block -> (batch*head, query block) map: blocks id and id+8 share an XCD (and its L2), so all query blocks of one
    // (batch, head) are given ids with equal id % 8: its K/V (256 KiB at N = 1024) are then fetched from HBM once per XCD pass
    // instead of once per query block.  Falls back to the plain order when B*H is not a multiple of 8.
    const int QB = gridDim.x, BH = gridDim.y * gridDim.z;
    int qb = blockIdx.x, bh = blockIdx.y + blockIdx.z * gridDim.y;
    if ((BH & 7) == 0) {
        const int id = blockIdx.x + QB * bh;  // linear dispatch order (x fastest)
        const int xcd = id & 7, j = id >> 3;
        qb = j % QB;
        bh = (j / QB) * 8 + xcd;
    }
    const int b = bh / gridDim.y, head = bh - b * gridDim.y, q0 = qb * 128 + wave * 32;
    const int r = lane & 31, h = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * N * ldq + head * 64;
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

    // ---- K/V tile staging by LDS-DMA: a piece is 8 key rows x 128 B; wave w moves pieces 2w and 2w+1 of K and of V
    const int drow = lane >> 3, dchunk = lane & 7;
    auto issue_tile = [&](int k0, int buf) {
        char* kb = smem + buf * 2 * TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pc = wave * 2 + j, row = pc * 8 + drow;
            int key = k0 + row;
            if (key >= N) key = N - 1;
            const bf16_t* kp = kbase + (size_t)key * ldq + ((dchunk ^ ((row >> 1) & 7)) << 3);
            const bf16_t* vp = vbase + (size_t)key * ldq + ((dchunk ^ (((row >> 1) & 1) << 2)) << 3);
            __builtin_amdgcn_global_load_lds((attn_gptr_t)kp, (attn_lptr_t)(kb + pc * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((attn_gptr_t)vp, (attn_lptr_t)(kb + TILE_BYTES + pc * 1024), 16, 0, 0);
        }
    };
    [[maybe_unused]] uint8_t* mask_lds = reinterpret_cast<uint8_t*>(smem + 6 * TILE_BYTES);
    if constexpr (MASKED) {
        const int nround = (N + 63) & ~63;
        for (int i = tid; i < nround; i += 256) mask_lds[i] = i < N ? (mask ? mask[(size_t)b * N + i] : (uint8_t)1) : (uint8_t)0;
    }

    // ---- per-lane LDS read offsets
    // K fragment (A operand of S^T): row = 32*kb + r, logical chunk = 2*ds + h
    int k_off[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) k_off[ds] = r * 128 + (((2 * ds + h) ^ ((r >> 1) & 7)) << 4);
    // V^T fragment via ds_read_b64_tr_b16: this lane supplies the address of key row (4*h + ((lane&15)>>2)) [+ 8*g + 16*s + 32*kb],
    // d columns 16*((lane>>4)&1) + 4*(lane&3) [+ 32*mb]
    const int v_row = 4 * h + ((lane & 15) >> 2);
    const int v_colb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;  // byte column inside the 64-byte half mb

    f32x16 o_acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) o_acc[0][i] = o_acc[1][i] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nt = (N + KT - 1) / KT;
    issue_tile(0, 0);
    if (nt > 1) issue_tile(KT, 1);
    if (nt > 1)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // tile 0 landed (my pieces); tile 1's four pieces may be in flight
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // ... everybody's pieces, and the mask bytes
    unsigned long long vm = ~0ull;

    for (int t = 0; t < nt; ++t) {
        const char* kb_lds = smem + (t % 3) * 2 * TILE_BYTES;
        const char* vb_lds = kb_lds + TILE_BYTES;
        if constexpr (MASKED) vm = __ballot(mask_lds[t * KT + lane] != 0);
        if (t + 2 < nt) issue_tile((t + 2) * KT, (t + 2) % 3);  // buffer of tile t-1: everybody passed the barrier that ended it

        // ---- S^T = K . Q^T for the two 32-key blocks of the tile
        f32x16 s[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int i = 0; i < 16; ++i) s[kb][i] = 0.f;
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_lds + kb * 32 * 128 + k_off[ds]);
                if constexpr (ABL != 4) s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ds], s[kb], 0, 0, 0);
                else asm volatile("" ::"v"(kf));
            }
        }
        // ---- online softmax (this lane: one query, keys 32*kb + (i&3) + 8*(i>>2) + 4*h)
        if constexpr (MASKED) {
            if (vm != ~0ull) {
                const unsigned long long vmh = h ? (vm >> 4) : vm;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int bit = 32 * kb + (i & 3) + 8 * (i >> 2);
                        if (!((vmh >> bit) & 1ull)) s[kb][i] = -INFINITY;
                    }
            }
        }
        if constexpr (ABL != 1) {
        float mt = s[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) mt = fmaxf(mt, s[kb][i]);
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));  // the other half-wave holds the other 32 keys of the same query
        const float m_new = fmaxf(m_run, mt * c);  // -inf * c stays -inf; m_run starts finite
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        // packed fp32 math (v_pk_fma_f32 / v_pk_add_f32): two scores per VALU instruction; only the exp2 itself is scalar
        const f32x2 c2 = {c, c}, nm2 = {-m_new, -m_new};
        f32x2 rs2 = {0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                f32x2 x = __builtin_elementwise_fma(f32x2{s[kb][i], s[kb][i + 1]}, c2, nm2);
                x[0] = __builtin_amdgcn_exp2f(x[0]);
                x[1] = __builtin_amdgcn_exp2f(x[1]);
                s[kb][i] = x[0];
                s[kb][i + 1] = x[1];
                rs2 += x;
            }
        const float rs = rs2[0] + rs2[1];
        l_run = l_run * alpha + rs;
        // the running max settles after the first tiles: skip the 32-register rescale when no query of this wave moved its max
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                o_acc[0][i] *= alpha;
                o_acc[1][i] *= alpha;
            }
        }
        }
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 pf = pack8(s[kb], 8 * ks);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    bf16x8 vf;
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int row = 32 * kb + 16 * ks + 8 * g + v_row;
                        const int colb = (64 * mb + v_colb) ^ (((row >> 1) & 1) << 6);
                        const bf16x4 part = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vb_lds + row * 128 + colb));
                        vf[4 * g + 0] = part[0];
                        vf[4 * g + 1] = part[1];
                        vf[4 * g + 2] = part[2];
                        vf[4 * g + 3] = part[3];
                    }
                    if constexpr (ABL != 3) o_acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o_acc[mb], 0, 0, 0);
                    else asm volatile("" ::"v"(vf), "v"(pf));
                }
            }
        // ---- tile t+1 must have landed (my pieces: only tile t+2's four may still be in flight), then for everybody
        if (t + 1 < nt) {
            if (t + 2 < nt)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- normalise and store: lane holds query q0 + r, dims 32*mb + (i&3) + 8*(i>>2) + 4*h
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int qrow = q0 + r;
    if (qrow < N) {
        bf16_t* op = out + ((size_t)b * N + qrow) * ldo + head * 64 + 4 * h;
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

// ----------------------------------------------------------------------------- 64 queries per wave
// Same algorithm with TWO 32-query blocks per wavefront (workgroup = 4 waves = 256 queries): every K fragment, V^T fragment,
// K/V tile refresh and barrier is amortised over twice the queries (the refresh + barrier cost 27 % of the 32-query kernel,
// tools/attn_ablate.py), at the price of 2 instead of 3 resident waves per SIMD.
template <bool MASKED>
__global__ __launch_bounds__(256, 2) void attn_fast2_kernel(const bf16_t* __restrict__ qkv, int ldq, int inner, const uint8_t* __restrict__ mask,
                                                            bf16_t* __restrict__ out, int ldo, int N, float c) {
    constexpr int KT = 64, QB = 2;
    constexpr int TILE_BYTES = KT * 128;
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NQ = gridDim.x, BH = gridDim.y * gridDim.z;
    int qblk = blockIdx.x, bh = blockIdx.y + blockIdx.z * gridDim.y;
    if ((BH & 7) == 0) {  // XCD-aware order: all query blocks of one (batch, head) share an XCD
        const int id = blockIdx.x + NQ * bh;
        const int xcd = id & 7, j = id >> 3;
        qblk = j % NQ;
        bh = (j / NQ) * 8 + xcd;
    }
    const int b = bh / gridDim.y, head = bh - b * gridDim.y, q0 = qblk * (128 * QB) + wave * (32 * QB);
    const int r = lane & 31, h = lane >> 5;
    const bf16_t* base = qkv + (size_t)b * N * ldq + head * 64;
    const bf16_t* kbase = base + inner;
    const bf16_t* vbase = base + 2 * inner;

    bf16x8 qf[QB][4];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        int qrow = q0 + 32 * j + r;
        if (qrow >= N) qrow = N - 1;
        const bf16_t* qp = base + (size_t)qrow * ldq + 8 * h;
#pragma unroll
        for (int ds = 0; ds < 4; ++ds) qf[j][ds] = *reinterpret_cast<const bf16x8*>(qp + 16 * ds);
    }

    const int srow0 = tid >> 3, scol = tid & 7;
    bf16x8 kreg[2], vreg[2];
    uint8_t mreg = 1;
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int row = k0 + srow0 + 32 * i;
            if (row >= N) row = N - 1;
            kreg[i] = *reinterpret_cast<const bf16x8*>(kbase + (size_t)row * ldq + scol * 8);
            vreg[i] = *reinterpret_cast<const bf16x8*>(vbase + (size_t)row * ldq + scol * 8);
        }
        if constexpr (MASKED) {
            const int key = k0 + lane;
            mreg = key < N ? (mask ? mask[(size_t)b * N + key] : (uint8_t)1) : (uint8_t)0;
        }
    };
    auto store_tile = [&](int buf) {
        char* kb = smem + buf * 2 * TILE_BYTES;
        char* vb = kb + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = srow0 + 32 * i;
            *reinterpret_cast<bf16x8*>(kb + row * 128 + ((scol ^ ((row >> 1) & 7)) << 4)) = kreg[i];
            *reinterpret_cast<bf16x8*>(vb + row * 128 + ((scol ^ (((row >> 1) & 1) << 2)) << 4)) = vreg[i];
        }
    };
    int k_off[4];
#pragma unroll
    for (int ds = 0; ds < 4; ++ds) k_off[ds] = r * 128 + (((2 * ds + h) ^ ((r >> 1) & 7)) << 4);
    const int v_row = 4 * h + ((lane & 15) >> 2);
    const int v_colb = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

    f32x16 o_acc[QB][2];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        m_run[j] = -1e30f;
        l_run[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) o_acc[j][0][i] = o_acc[j][1][i] = 0.f;
    }

    const int nt = (N + KT - 1) / KT;
    load_tile(0);
    store_tile(0);
    unsigned long long vm = ~0ull;
    if constexpr (MASKED) vm = __ballot(mreg != 0);
    __syncthreads();
    if (nt > 1) load_tile(KT);

    for (int t = 0; t < nt; ++t) {
        const char* kb_lds = smem + (t & 1) * 2 * TILE_BYTES;
        const char* vb_lds = kb_lds + TILE_BYTES;
        f32x16 s[QB][2];
#pragma unroll
        for (int j = 0; j < QB; ++j)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) s[j][kb][i] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ds = 0; ds < 4; ++ds) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(kb_lds + kb * 32 * 128 + k_off[ds]);  // one read feeds both query blocks
#pragma unroll
                for (int j = 0; j < QB; ++j) s[j][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[j][ds], s[j][kb], 0, 0, 0);
            }
        if constexpr (MASKED) {
            if (vm != ~0ull) {
                const unsigned long long vmh = h ? (vm >> 4) : vm;
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
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            float mt = s[j][0][0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; ++i) mt = fmaxf(mt, s[j][kb][i]);
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run[j], mt * c);
            const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
            m_run[j] = m_new;
            const f32x2 c2 = {c, c}, nm2 = {-m_new, -m_new};
            f32x2 rs2 = {0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    f32x2 x = __builtin_elementwise_fma(f32x2{s[j][kb][i], s[j][kb][i + 1]}, c2, nm2);
                    x[0] = __builtin_amdgcn_exp2f(x[0]);
                    x[1] = __builtin_amdgcn_exp2f(x[1]);
                    s[j][kb][i] = x[0];
                    s[j][kb][i + 1] = x[1];
                    rs2 += x;
                }
            const float rs = rs2[0] + rs2[1];
            l_run[j] = l_run[j] * alpha + rs;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0ull) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    o_acc[j][0][i] *= alpha;
                    o_acc[j][1][i] *= alpha;
                }
            }
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 pf[QB];
#pragma unroll
                for (int j = 0; j < QB; ++j) pf[j] = pack8(s[j][kb], 8 * ks);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    bf16x8 vf;  // one transposed read pair feeds both query blocks
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const int row = 32 * kb + 16 * ks + 8 * g + v_row;
                        const int colb = (64 * mb + v_colb) ^ (((row >> 1) & 1) << 6);
                        const bf16x4 part = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(vb_lds + row * 128 + colb));
                        vf[4 * g + 0] = part[0];
                        vf[4 * g + 1] = part[1];
                        vf[4 * g + 2] = part[2];
                        vf[4 * g + 3] = part[3];
                    }
#pragma unroll
                    for (int j = 0; j < QB; ++j) o_acc[j][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[j], o_acc[j][mb], 0, 0, 0);
                }
            }
        if (t + 1 < nt) {
            store_tile((t + 1) & 1);
            if constexpr (MASKED) vm = __ballot(mreg != 0);
        }
        __syncthreads();
        if (t + 2 < nt) load_tile((t + 2) * KT);
    }

#pragma unroll
    for (int j = 0; j < QB; ++j) {
        const float l_tot = l_run[j] + __shfl_xor(l_run[j], 32, 64);
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        const int qrow = q0 + 32 * j + r;
        if (qrow < N) {
            bf16_t* op = out + ((size_t)b * N + qrow) * ldo + head * 64 + 4 * h;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = (bf16_t)(o_acc[j][mb][4 * g + e] * inv);
                    *reinterpret_cast<bf16x4*>(op + 32 * mb + 8 * g) = v4;
                }
        }
    }
}

int g_attn_occ = 0;      // tuning knob ("attn_occ"): 4 = build of the 32-query kernel capped at 128 VGPRs (4 waves per SIMD)
int g_attn_variant = 0;  // tuning knob ("attn_variant"): 0 = by sequence length, 1 = 32 queries per wave (128 per workgroup), 2 = 64 queries per wave
int g_attn_ablate = 0;  // tuning knob ("attn_ablate"): timing-only ablations of the unmasked kernel

bool attention_fast_supported(int precision, int N, int H) { return precision == F5_PREC_BF16 && N >= 1 && H >= 1; }

int launch_attention_fast(int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out, int ldo, hipStream_t stream) {
    if ((ldq & 7) || (ldo & 3)) return f5_fail(F5_EINVAL, "attention_fast: ldq must be a multiple of 8 and ldo of 4");
    const float c = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
    const bool masked = mask != nullptr || (N % 64) != 0;
    // 64 queries per wave pays for long sequences (+6 % at N = 4096); at N = 1024 with the key mask live it measured 4 % slower in situ
    if (g_attn_ablate == 0 && (g_attn_variant == 2 || (g_attn_variant == 0 && N >= 2048)) && N >= 256) {
        dim3 grid2(cdiv(N, 256), H, B);
        if (masked)
            hipLaunchKernelGGL((attn_fast2_kernel<true>), grid2, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
        else
            hipLaunchKernelGGL((attn_fast2_kernel<false>), grid2, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
        F5_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(cdiv(N, 128), H, B), block(256);
    if (g_attn_variant == 3 && g_attn_ablate == 0 && N <= 4096) {
        if (masked)
            hipLaunchKernelGGL((attn_fast3_kernel<true>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
        else
            hipLaunchKernelGGL((attn_fast3_kernel<false>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    } else if (g_attn_occ == 4 && g_attn_ablate == 0) {
        if (masked)
            hipLaunchKernelGGL((attn_fast_kernel<true, 0, 4>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
        else
            hipLaunchKernelGGL((attn_fast_kernel<false, 0, 4>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    } else if (masked)
        hipLaunchKernelGGL((attn_fast_kernel<true, 0>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    else if (g_attn_ablate == 1)
        hipLaunchKernelGGL((attn_fast_kernel<false, 1>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    else if (g_attn_ablate == 2)
        hipLaunchKernelGGL((attn_fast_kernel<false, 2>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    else if (g_attn_ablate == 3)
        hipLaunchKernelGGL((attn_fast_kernel<false, 3>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    else if (g_attn_ablate == 4)
        hipLaunchKernelGGL((attn_fast_kernel<false, 4>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    else
        hipLaunchKernelGGL((attn_fast_kernel<false, 0>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    F5_LAUNCH_CHECK();
    return 0;
}
