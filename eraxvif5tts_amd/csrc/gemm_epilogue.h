// gemm_epilogue.h -- fused epilogues shared by the reference tile kernel and the tuned kernel.
// One call handles the 4 consecutive output features n..n+3 that one lane holds for token m
// (C/D layout of v_mfma_f32_16x16x32_bf16 / 16x16x4_f32 with weights on the MFMA row index).
#pragma once
#include "gemm.h"

#if defined(__HIPCC__)
__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_GELU_TANH: return act_gelu_tanh(v);
        case ACT_GELU_ERF: return act_gelu_erf(v);
        case ACT_MISH: return act_mish(v);
        default: return v;
    }
}

template <typename T> __device__ __forceinline__ void store4_t(T* dst, const float v[4], bool vec, int nvalid);
template <> __device__ __forceinline__ void store4_t<float>(float* dst, const float v[4], bool vec, int nvalid) {
    if (vec) {
        *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
    } else {
        for (int r = 0; r < nvalid; ++r) dst[r] = v[r];
    }
}
template <> __device__ __forceinline__ void store4_t<bf16_t>(bf16_t* dst, const float v[4], bool vec, int nvalid) {
    if (vec) {
        *reinterpret_cast<bf16x4*>(dst) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    } else {
        for (int r = 0; r < nvalid; ++r) dst[r] = (bf16_t)v[r];
    }
}

template <typename T, int EPI>
__device__ __forceinline__ void gemm_epilogue4(const GemmParams& p, int m, int n, f32x4 acc) {
    if (m >= p.M || n >= p.N) return;
    const int nvalid = min(4, p.N - n);
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if (p.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nvalid) v[r] += p.bias[n + r];
    }
    if constexpr (EPI == EPI_STORE_T) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
        const bool vec = nvalid == 4 && (p.ldo & 3) == 0;
        store4_t<T>(reinterpret_cast<T*>(p.out_t) + (size_t)m * p.ldo + n, v, vec, nvalid);
    } else if constexpr (EPI == EPI_STORE_F32) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = apply_act(v[r], p.act);
        const bool vec = nvalid == 4 && (p.ldof & 3) == 0;
        store4_t<float>(p.out_f + (size_t)m * p.ldof + n, v, vec, nvalid);
    } else if constexpr (EPI == EPI_RESID) {
        if (p.rowmask && p.rowmask[m] == 0) return;  // attn output of padded query rows is masked_fill'd to 0 (modules.py:499-501)
        const float* g = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.gate_bstride + n : nullptr;
        if (p.add2_f16) {  // the residual stream of the bf16 production mode: fp16 storage (saturating), fp32 arithmetic
            _Float16* d = reinterpret_cast<_Float16*>(p.out_f) + (size_t)m * p.ldof + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nvalid) {
                    float t = apply_act(v[r], p.act);
                    if (g) t *= g[r];
                    d[r] = (_Float16)__builtin_amdgcn_fmed3f((float)d[r] + t, -65504.0f, 65504.0f);
                }
            return;
        }
        float* dst = p.out_f + (size_t)m * p.ldof + n;
        const bool vec = nvalid == 4 && (p.ldof & 3) == 0;
        float o[4];
        if (vec) {
            f32x4 cur = *reinterpret_cast<const f32x4*>(dst);
            o[0] = cur[0]; o[1] = cur[1]; o[2] = cur[2]; o[3] = cur[3];
        } else {
            for (int r = 0; r < nvalid; ++r) o[r] = dst[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (r < nvalid) {
                float t = apply_act(v[r], p.act);
                if (g) t *= g[r];
                o[r] += t;
            }
        }
        store4_t<float>(dst, o, vec, nvalid);
    } else if constexpr (EPI == EPI_GATE_T) {
        const bool keep = !(p.rowmask && p.rowmask[m] == 0);  // padded query rows: attention output masked_fill'd to 0 (modules.py:499-501)
        const float* g = p.gate ? p.gate + (size_t)(m / p.rows_per_batch) * p.gate_bstride + n : nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float t = apply_act(v[r], p.act);
            if (g && r < nvalid) t *= g[r];
            v[r] = keep ? t : 0.f;
        }
        const bool vec = nvalid == 4 && (p.ldo & 3) == 0;
        store4_t<T>(reinterpret_cast<T*>(p.out_t) + (size_t)m * p.ldo + n, v, vec, nvalid);
    } else if constexpr (EPI == EPI_ADD2) {
        if (p.add2_f16) {  // fp16 addend / fp16 stream (bf16 production mode)
            const _Float16* a = reinterpret_cast<const _Float16*>(p.addend) + (size_t)m * p.ldadd + n;
            _Float16* d = reinterpret_cast<_Float16*>(p.out_f) + (size_t)m * p.ldof + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nvalid) {
                    v[r] += (float)a[r];
                    d[r] = (_Float16)__builtin_amdgcn_fmed3f(v[r], -65504.0f, 65504.0f);
                }
            const bool vect = nvalid == 4 && (p.ldo & 3) == 0;
            store4_t<T>(reinterpret_cast<T*>(p.out_t) + (size_t)m * p.ldo + n, v, vect, nvalid);
        } else {
            const float* a = p.addend + (size_t)m * p.ldadd + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nvalid) v[r] += a[r];
            const bool vect = nvalid == 4 && (p.ldo & 3) == 0;
            const bool vecf = nvalid == 4 && (p.ldof & 3) == 0;
            store4_t<T>(reinterpret_cast<T*>(p.out_t) + (size_t)m * p.ldo + n, v, vect, nvalid);
            store4_t<float>(p.out_f + (size_t)m * p.ldof + n, v, vecf, nvalid);
        }
    } else if constexpr (EPI == EPI_ROPE_T) {
        // x_transformers apply_rotary_pos_emb on adjacent pairs, fp32 math, q and k parts, first rope_heads heads
        const int part = n / p.rope_inner;
        const int within = n - part * p.rope_inner;
        if (part < 2 && (within >> 6) < p.rope_heads && nvalid == 4) {
            const int pos = m % p.rows_per_batch;
            const int j = (within & 63) >> 1;
            const float* cs = p.rope + ((size_t)pos * 32 + j) * 2;
            const float c0 = cs[0], s0 = cs[1], c1 = cs[2], s1 = cs[3];
            const float a0 = v[0], a1 = v[1], b0 = v[2], b1 = v[3];
            v[0] = a0 * c0 - a1 * s0;
            v[1] = a1 * c0 + a0 * s0;
            v[2] = b0 * c1 - b1 * s1;
            v[3] = b1 * c1 + b0 * s1;
        }
        const bool vec = nvalid == 4 && (p.ldo & 3) == 0;
        store4_t<T>(reinterpret_cast<T*>(p.out_t) + (size_t)m * p.ldo + n, v, vec, nvalid);
    }
}
#endif
