// gemm_tile.h -- device helpers of the tuned GEMM kernel (gemm_fast.hip) and the halo-tile conv kernel (conv31.hip)
#pragma once
#include "gemm.h"
#include <type_traits>

// compile-time loop: keeps accumulator indices static even when the optimizer refuses a "#pragma unroll"
// (a runtime-indexed accumulator array would be demoted to scratch memory)
template <int N, int I = 0, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, I + 1>(f);
    }
}

// exp2/rcp forms of the activations for the bf16 epilogues (the output rounding to bf16 dominates their ~1 ulp error):
//   gelu_tanh(x) = 0.5 x (1 + tanh(u)) = x * sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3)
//   mish(x)      = x tanh(softplus(x)) = x * n / (n + 2),  n = e^x (e^x + 2)
__device__ __forceinline__ float fast_gelu_tanh(float x) {
    const float a = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float u2 = x * (a + (a * 0.044715f) * (x * x));
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u2));
}
__device__ __forceinline__ float fast_mish(float x) {
    const float w = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
    const float n = w * (w + 2.0f);
    return x > 20.0f ? x : x * n * __builtin_amdgcn_rcpf(n + 2.0f);
}
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
// Two adjacent 16-feature tiles of one token tile: lane (row r = lane>>4) holds features 4r..4r+3 of tile A and of tile B.
// v_permlane16_swap exchanges the odd 16-lane rows of A with the even rows of B, after which every lane owns 8 consecutive
// features (rows 0/2: tile A features 8*(r>>1).., rows 1/3: tile B) -> one 16-byte store instead of two 8-byte ones
// (the epilogue is store-issue bound).
__device__ __forceinline__ u32x4 pair_swap(bf16x4 a, bf16x4 b) {
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    const u32x2 s0 = __builtin_amdgcn_permlane16_swap(ua[0], ub[0], false, false);
    const u32x2 s1 = __builtin_amdgcn_permlane16_swap(ua[1], ub[1], false, false);
    return u32x4{s0[0], s1[0], s0[1], s1[1]};
}
__device__ __forceinline__ bf16x4 to_bf16x4(const f32x4& v) { return bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]}; }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16(const void* src, char* lds_wave_base) {
    // wave-uniform LDS base + lane*16 <- 16 bytes from each lane's own global address
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

