// lnf_stats_math.h -- (mean, rstd) of a token row from the partial sums of the in-place residual epilogues (gemm.h), ONE definition shared by
// stats_finalize_kernel (lnfold.hip) and the in-kernel form of the LayerNorm-fold GEMM (gemm_fast.hip): both must produce the same bits, because
// which of the two a launch takes depends on the tile its batch size selects and a row's value must not.
#pragma once
#include "common.h"

// s1 = sum (h - pivot), s2 = sum (h - pivot)^2 over the row's D elements (eps 1e-6: modules.py:308,624).  sumsq = the sum of squares of the
// elements themselves (range guard of the fp16 stream: >= 65504^2 whenever an element was stored at the saturation value).
__device__ __forceinline__ void lnf_row_stats(float s1, float s2, float pivot, int D, float& mean, float& rstd, float& sumsq) {
    const float inv = 1.0f / (float)D;
    const float md = s1 * inv;  // mean - pivot
    float var = __builtin_fmaf(-md, md, s2 * inv);
    var = var > 0.f ? var : 0.f;
    mean = pivot + md;
    rstd = 1.0f / sqrtf(var + 1e-6f);
    sumsq = __builtin_fmaf(pivot, __builtin_fmaf(2.0f, s1, (float)D * pivot), s2);
}
// the flag words of the range guard (same layout as res_range_guard in elementwise.hip); call with `bad` per lane, any lane of the wave
__device__ __forceinline__ void lnf_raise_guard(unsigned* sat, bool bad, float sumsq, int sat_tag, int row) {
    if (sat && __builtin_amdgcn_ballot_w64(bad) != 0ull) {
        if (bad) {
            const float bound = sqrtf(sumsq);  // an upper bound of the row's largest |element|
            if (bound == bound && bound < 3.0e38f) atomicMax(sat + 1, __float_as_uint(bound));
            if (!(sumsq == sumsq)) __hip_atomic_store(sat + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(sat + 3, 1u << (sat_tag & 15));
            atomicOr(sat + 4, 1u << ((sat_tag >> 4) & 31));
            atomicMax(sat + 5, 0x7fffffffu - (unsigned)row);
        }
    }
}
