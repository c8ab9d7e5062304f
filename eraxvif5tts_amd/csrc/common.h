// common.h -- shared device/host helpers for libf5hip (gfx950 only: wave64, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/f5hip.h"

// ----------------------------------------------------------------------------- error plumbing
void f5_set_error(const char* fmt, ...);
int f5_fail(int code, const char* fmt, ...);

#define F5_HIP(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t _e = (expr);                                                                              \
        if (_e != hipSuccess) return f5_fail(F5_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                             __FILE__, __LINE__);                                            \
    } while (0)

#define F5_TRY(expr)          \
    do {                      \
        int _rc = (expr);     \
        if (_rc != 0) return _rc; \
    } while (0)

#define F5_LAUNCH_CHECK()                                                                                     \
    do {                                                                                                      \
        hipError_t _e = hipGetLastError();                                                                    \
        if (_e != hipSuccess) return f5_fail(F5_EHIP, "kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                                             __FILE__, __LINE__);                                             \
    } while (0)

// ----------------------------------------------------------------------------- device types
typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

#if defined(__HIPCC__)
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // v_cvt_pk_bf16_f32 (RNE, NaN-safe)

// wave64 butterfly reductions (DPP/ds_swizzle chosen by the compiler from __shfl_xor)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// activations, written exactly as the reference's torch ops define them
__device__ __forceinline__ float act_gelu_tanh(float x) {  // nn.GELU(approximate="tanh"), modules.py:625
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(k0 * (x + k1 * x * x * x)));
}
__device__ __forceinline__ float act_gelu_erf(float x) {  // nn.GELU(), modules.py:255
    return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}
__device__ __forceinline__ float act_mish(float x) {  // nn.Mish: x * tanh(softplus(x)), softplus threshold 20
    float sp = x > 20.0f ? x : log1pf(expf(x));
    return x * tanhf(sp);
}
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + expf(-x)); }
#endif

enum Act { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_GELU_ERF = 2, ACT_MISH = 3 };
