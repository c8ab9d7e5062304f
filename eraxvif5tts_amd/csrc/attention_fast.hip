// attention_fast.hip -- flash attention for the DiT blocks on gfx950 (bf16 in/out, head dim 64, non-causal, key-padding mask;
// reference model/modules.py:483-497): the 64-queries-per-wavefront kernel and the launcher that picks between it and the
// software-pipelined 32-queries-per-wavefront kernel of attention_pipe.hip.
//
// Structure (see /opt/skills/guides/cdna_hip_programming.md, Appendix B "Fused attention prefill"):
//   * one workgroup = 4 wavefronts = 256 queries of one (batch, head); each wave owns TWO 32-query blocks, so every K fragment,
//     V^T fragment, K/V tile refresh and barrier is amortised over 64 queries.  What the measurements of round 2 say decides
//     (profiles/r2_attention_*.txt): on one SIMD the matrix pipe and the vector ALU mostly serialise, and the LDS -> VGPR fragment
//     traffic is not hidden by either, so halving the fragment bytes per FLOP is worth more than instruction placement;
//   * K/V tiles of 64 keys double-buffered in LDS (register-staged: the global loads of tile t+2 are issued right after the
//     barrier of tile t and written to LDS at the end of tile t+1);
//   * swapped QK^T: S^T = K.Q^T with v_mfma_f32_32x32x16_bf16, so a lane holds 16 keys x ONE query per 32-key block:
//     max / sum are in-register (one v_permlane32_swap per tile), and the un-normalised P converted to bf16 is directly the
//     B operand of the PV product (accumulator-as-operand k order);
//   * O^T = V^T.P^T: the V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose);
//   * K tile XOR-swizzled for conflict-free ds_read_b128, V tile swizzled for the transposed reads;
//   * exp2 with the softmax scale folded into one fma; DEFERRED rescale: a query's exponent reference only moves when its row
//     maximum has outgrown it by more than 2^16 (with random scores the exact running max of SOME query of a wave moves in nearly
//     every tile, and the 64-register rescale of O with it); fp32 sums and output accumulators.
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
            float mx0 = max3_asm(s[j][0][0], s[j][0][1], s[j][0][2]), mx1 = max3_asm(s[j][1][0], s[j][1][1], s[j][1][2]);
#pragma unroll
            for (int i = 3; i < 15; i += 2) {
                mx0 = max3_asm(mx0, s[j][0][i], s[j][0][i + 1]);
                mx1 = max3_asm(mx1, s[j][1][i], s[j][1][i + 1]);
            }
            mx0 = max3_asm(mx0, mx1, s[j][0][15]);
            float mt = max3_asm(mx0, s[j][1][15], s[j][1][15]);
            {
                const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
                mt = fmaxf(__uint_as_float(r2[0]), __uint_as_float(r2[1])) * c;
            }
            // deferred rescale (see attention_pipe.hip): the reference only moves when the row maximum outgrew it by more than 2^16
            float alpha = 1.0f;
            if (__builtin_amdgcn_ballot_w64(mt > m_run[j] + 16.0f) != 0ull) {
                const float m_new = fmaxf(m_run[j], mt);
                alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
                m_run[j] = m_new;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    o_acc[j][0][i] *= alpha;
                    o_acc[j][1][i] *= alpha;
                }
            }
            const float m_new = m_run[j];
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

    // ---- normalise; stage the wave's 64 x 64 output block through LDS (the K/V buffers are free: the loop ended on a barrier) and store
    //      whole 128-byte rows: a wave-instruction then writes 8 full lines instead of 8-byte pieces of 32 different rows (the per-lane
    //      form is store-ISSUE bound: 16 dwordx2 per lane; C2 spends ~15 % of a workgroup's life in that tail, C4 a quarter of that)
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
                // row 32*j + r, dims 32*mb + 8*g + 4*h .. +3; 16-byte chunk index XORed with the row so that the 32 rows of a
                // half-wave spread over the banks
                const int row = 32 * j + r, chunk = (4 * mb + g) ^ (row & 7);
                *reinterpret_cast<bf16x4*>(stage + row * 128 + chunk * 16 + 8 * h) = v4;
            }
    }
    {
        bf16_t* obase = out + ((size_t)b * N + q0) * ldo + head * 64;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = 8 * k + (lane >> 3), chunk = lane & 7;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(stage + row * 128 + ((chunk ^ (row & 7)) * 16));
            if (q0 + row < N) *reinterpret_cast<u32x4_t*>(obase + (size_t)row * ldo + chunk * 8) = v;
        }
    }
}

int g_attn_variant = 0;  // tuning knob ("attn_variant"): 0 = by grid size, 2 = 64 queries per wave, 5 = software-pipelined 32 queries per wave

bool attention_fast_supported(int precision, int N, int H) { return precision == F5_PREC_BF16 && N >= 1 && H >= 1; }

int launch_attention_pipe(int waves, int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out, int ldo, hipStream_t stream);  // attention_pipe.hip

int launch_attention_fast(int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out, int ldo, hipStream_t stream) {
    if ((ldq & 7) || (ldo & 7)) return f5_fail(F5_EINVAL, "attention_fast: ldq and ldo must be multiples of 8");
    // 256 queries per workgroup need at least one workgroup per CU to pay (C2: 753 vs 705 TFLOP/s, C4: 923 vs 867); below that
    // (single-utterance serving) the 128-query workgroups of the pipelined kernel fill the chip better (B = 1: 17 vs 25 us)
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    }
    const bool wide = g_attn_variant == 2 || (g_attn_variant == 0 && (long)B * H * cdiv(N, 256) >= cus);
    if (!wide) return launch_attention_pipe(4, B, N, H, qkv, ldq, mask, out, ldo, stream);
    const float c = 0.125f * 1.4426950408889634f;  // 1/sqrt(64) * log2(e)
    const bool masked = mask != nullptr || (N % 64) != 0;
    dim3 grid2(cdiv(N, 256), H, B);
    if (masked)
        hipLaunchKernelGGL((attn_fast2_kernel<true>), grid2, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    else
        hipLaunchKernelGGL((attn_fast2_kernel<false>), grid2, dim3(256), 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, c);
    F5_LAUNCH_CHECK();
    return 0;
}
