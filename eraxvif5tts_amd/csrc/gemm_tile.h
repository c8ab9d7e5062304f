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
// The three epilogue expressions that contain a multiply feeding an add exist ONCE, with explicit fused multiply-adds (epi_* below; the library
// is built with -ffp-contract=off, so nothing else fuses): until round 3 the whole-tile ("lean") epilogue of gemm_fast.hip used fused forms and
// the ragged-tile ("generic") one separate multiplies and adds, i.e. a token row's value depended on whether its 256-row tile was whole.  It
// must not: f5_sample_ragged promises every utterance the bits of its own batch-1 call, where its rows sit in other tiles.
__device__ __forceinline__ float fast_gelu_tanh(float x) {
    const float a = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const float u2 = x * __builtin_fmaf(x * x, a * 0.044715f, a);
    return x * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(u2) + 1.0f);
}
__device__ __forceinline__ f32x4 epi_gelu_tanh4(const f32x4& v) {  // elementwise fast_gelu_tanh, same operations in the same order
    constexpr float a = -2.0f * 0.7978845608028654f * 1.4426950408889634f;
    const f32x4 u = v * __builtin_elementwise_fma(v * v, f32x4{a * 0.044715f, a * 0.044715f, a * 0.044715f, a * 0.044715f}, f32x4{a, a, a, a});
    f32x4 d;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_exp2f(u[e]);
    d = d + 1.0f;
#pragma unroll
    for (int e = 0; e < 4; ++e) d[e] = __builtin_amdgcn_rcpf(d[e]);
    return v * d;
}
// x_transformers apply_rotary_pos_emb on two adjacent feature pairs; cs = (cos0, sin0, cos1, sin1)
__device__ __forceinline__ f32x4 epi_rope4(const f32x4& v, const f32x4& cs) {
    return f32x4{__builtin_fmaf(v[0], cs[0], -(v[1] * cs[1])), __builtin_fmaf(v[1], cs[0], v[0] * cs[1]),
                     __builtin_fmaf(v[2], cs[2], -(v[3] * cs[3])), __builtin_fmaf(v[3], cs[2], v[2] * cs[3])};
}
// x + gate * branch
__device__ __forceinline__ f32x4 epi_axpy4(const f32x4& branch, const f32x4& gate, const f32x4& x) {
    return __builtin_elementwise_fma(branch, gate, x);
}
// LayerNorm fold: rstd (acc - mean c1) + c2, two fused multiply-adds per element (shared by the lean and the generic epilogue)
__device__ __forceinline__ f32x4 epi_lnf4(const f32x4& acc, float mean, float rstd, const f32x4& c1, const f32x4& c2) {
    const f32x4 nm{-mean, -mean, -mean, -mean}, rs{rstd, rstd, rstd, rstd};
    return __builtin_elementwise_fma(rs, __builtin_elementwise_fma(nm, c1, acc), c2);
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

