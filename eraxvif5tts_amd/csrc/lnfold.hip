// lnfold.hip -- the two small kernels of the LayerNorm fold (gemm.h):
//
//   fold_weights_kernel     once per time grid: for every evaluation time e and DiT block l the per-time weights of the two projections that
//                           follow an AdaLN LayerNorm (modules.py:301-317: attn_norm -> to_q|to_k|to_v;  modules.py:637-638: ff_norm -> ff.0.0)
//                               W'[e][l][n][k] = fp16( W[l][n][k] * (1 + scale[e][l][k]) )
//                               c1[e][l][n]    = sum_k float(W'[e][l][n][k])            (of the ROUNDED values: what the MFMA multiplies)
//                               c2[e][l][n]    = b[l][n] + sum_k W[l][n][k] * shift[e][l][k]
//                           so that  Linear(LN(x) (1 + scale) + shift) = rstd (x . W'^T - mean c1) + c2  with x the fp16 residual stream itself.
//   stats_finalize_kernel   after every in-place residual GEMM: the per-wave-tile partial sums its epilogue wrote (sum (h - pivot), sum (h - pivot)^2
//                           per token row and 64-feature tile) -> (mean, rstd) per row, eps = 1e-6 (modules.py:308,624), and the fp16 range guard
//                           of the stream (the LayerNorm passes that carried it are gone): a row whose sum of squares reaches 65504^2 -- any
//                           element stored at the saturation value does that -- or is not a number raises the plan's flag words.
#include "common.h"
#include "kernels.h"
#include "lnf_stats_math.h"

typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_sum_fixed(float v) {  // xor butterfly: every lane ends with the same total, fixed order
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// grid (ceil(R_total / 4), evals), 256 threads = 4 output rows.  R = rows per block = 3 * inner + ff; row r of block l: r < 3 * inner belongs to the
// attention norm (shift_msa at mod + l * 6D, scale_msa at + D), else to the FF norm (shift_mlp at + 3D, scale_mlp at + 4D)  (modules.py:312).
__global__ __launch_bounds__(256) void fold_weights_kernel(const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ mod,
                                                           int modrow, int depth, int R, int qkv_rows, int D, _Float16* __restrict__ Wt,
                                                           float* __restrict__ c1, float* __restrict__ c2) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);  // over depth * R
    const int e = blockIdx.y;
    if (row >= depth * R) return;
    const int l = row / R, r = row - l * R;
    const float* ml = mod + (size_t)e * modrow + (size_t)l * 6 * D;
    const float* shift = ml + (r < qkv_rows ? 0 : 3 * D);
    const float* scale = shift + D;
    const float* w = W + (size_t)row * D;
    _Float16* wo = Wt + ((size_t)e * depth * R + row) * D;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + c);
        f16x4_t h;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h[k] = (_Float16)(wv[k] * (1.0f + sc[k]));
            s1 += (float)h[k];
            s2 = __builtin_fmaf(wv[k], sh[k], s2);
        }
        *reinterpret_cast<f16x4_t*>(wo + c) = h;
    }
    s1 = wave_sum_fixed(s1);
    s2 = wave_sum_fixed(s2);
    if (lane == 0) {
        c1[(size_t)e * depth * R + row] = s1;
        c2[(size_t)e * depth * R + row] = bias[row] + s2;
    }
}

int launch_fold_weights(const float* W, const float* bias, const float* mod, int modrow, int evals, int depth, int R, int qkv_rows, int D, void* Wt,
                        float* c1, float* c2, hipStream_t stream) {
    if (evals <= 0 || depth <= 0 || R <= 0 || D % 4 != 0) return f5_fail(F5_EINVAL, "fold_weights: bad shape");
    hipLaunchKernelGGL(fold_weights_kernel, dim3((unsigned)((depth * R + 3) / 4), (unsigned)evals), dim3(256), 0, stream, W, bias, mod, modrow, depth, R,
                       qkv_rows, D, (_Float16*)Wt, c1, c2);
    F5_LAUNCH_CHECK();
    return 0;
}

// one thread per token row: partial[c * ld + row] = (s1, s2) of feature tile c (ncols = D / 64), added in the order c = 0, 1, ... (fixed).
// NC > 0: the column count as a constant, so that all NC loads of a row are in flight together (D = 1024: 16; the runtime loop issued them
// one behind the other, 10 us per launch at 65 536 rows against 3 for the bytes).  Blocks past the row blocks only PREFETCH: small launches
// are bound by the latency of the weights behind them, which every block reads from HBM again; as the LayerNorm passes of round 2 did, one
// dword per 128-byte line of up to four weight ranges is touched here (thread t of prefetch block b: line 256 b + t of every range).
template <int NC>
__global__ __launch_bounds__(256) void stats_finalize_kernel(const f32x2_t* __restrict__ partial, int ld, int ncols, int rows, int D,
                                                             const float* pivot /* [rows][2] or null; may alias `stats` */, float* stats, unsigned* sat,
                                                             int sat_tag, int row_blocks, PrefetchSet pf) {
    if ((int)blockIdx.x >= row_blocks) {
        const unsigned line = (blockIdx.x - (unsigned)row_blocks) * 256u + threadIdx.x;
        unsigned acc = 0u;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (pf.p[r] && (size_t)line * 128u < pf.n[r]) acc ^= *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(pf.p[r]) + (size_t)line * 128);
        if (acc == 0x7fc0dead && line == 0xffffffffu) stats[0] = 0.f;  // (never true: keeps the loads alive)
        return;
    }
    const int row = blockIdx.x * 256 + threadIdx.x;
    bool bad = false;
    float sumsq = 0.f;
    if (row < rows) {
        float s1 = 0.f, s2 = 0.f;
        if constexpr (NC > 0) {
            f32x2_t v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = partial[(size_t)c * ld + row];
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                s1 += v[c][0];
                s2 += v[c][1];
            }
        } else {
            for (int c = 0; c < ncols; ++c) {
                const f32x2_t v = partial[(size_t)c * ld + row];
                s1 += v[0];
                s2 += v[1];
            }
        }
        const float pv = pivot ? pivot[(size_t)row * 2] : 0.0f;
        float mean, rstd;
        lnf_row_stats(s1, s2, pv, D, mean, rstd, sumsq);
        stats[(size_t)row * 2] = mean;
        stats[(size_t)row * 2 + 1] = rstd;
        bad = !(sumsq < 65504.0f * 65504.0f);
    }
    lnf_raise_guard(sat, bad, sumsq, sat_tag, row);
}

int launch_stats_finalize(const float* partial, int ld, int ncols, int rows, int D, const float* pivot, float* stats, unsigned* sat, int sat_tag,
                          hipStream_t stream, const PrefetchSet* prefetch) {
    if (rows <= 0) return 0;
    PrefetchSet pf{{nullptr, nullptr, nullptr, nullptr}, {0u, 0u, 0u, 0u}};
    unsigned pf_bytes = 0;
    if (prefetch) {
        pf = *prefetch;
        for (int r = 0; r < 4; ++r)
            if (pf.p[r] && pf.n[r] > pf_bytes) pf_bytes = pf.n[r];
    }
    const int row_blocks = (rows + 255) / 256, pf_blocks = (int)((pf_bytes / 128u + 255u) / 256u);
    const dim3 grid((unsigned)(row_blocks + pf_blocks)), block(256);
    const f32x2_t* pp = reinterpret_cast<const f32x2_t*>(partial);
    if (ncols == 16)
        hipLaunchKernelGGL(stats_finalize_kernel<16>, grid, block, 0, stream, pp, ld, ncols, rows, D, pivot, stats, sat, sat_tag, row_blocks, pf);
    else
        hipLaunchKernelGGL(stats_finalize_kernel<0>, grid, block, 0, stream, pp, ld, ncols, rows, D, pivot, stats, sat, sat_tag, row_blocks, pf);
    F5_LAUNCH_CHECK();
    return 0;
}
