// attention.hip -- non-causal multi-head attention with a key-padding mask over packed q|k|v projections.
//   * attn_ref_kernel : reference kernel, fp32 VALU math (exact softmax order-independent up to rounding),
//                       any sequence length, both activation dtypes.  Used by the fp32 parity mode and
//                       as the cross-check of the tuned kernel.
//   * tuned bf16 flash kernel: attention_fast.hip.
#include "kernels.h"

template <typename T>
__global__ __launch_bounds__(256) void attn_ref_kernel(const T* __restrict__ qkv, int ldq, int inner, const uint8_t* __restrict__ mask,
                                                       T* __restrict__ out, int ldo, int N, int bs /* rows between batch items (N, or more: ragged sampler) */, float scale) {
    constexpr int D = 64, TQ = 64, TK = 64, LD = 65;
    __shared__ float Qs[TQ * LD], Ks[TK * LD], Vs[TK * LD], Ps[TQ * LD];
    const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * TQ;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const T* base = qkv + (size_t)b * bs * ldq + h * D;

    for (int i = tid; i < TQ * D; i += 256) {
        const int r = i >> 6, d = i & 63;
        Qs[r * LD + d] = (q0 + r < N) ? to_f32(base[(size_t)(q0 + r) * ldq + d]) : 0.f;
    }
    float m_i[4], l_i[4], o[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        m_i[i] = -INFINITY;
        l_i[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[i][j] = 0.f;
    }
    for (int k0 = 0; k0 < N; k0 += TK) {
        __syncthreads();  // previous tile's Ps/Vs reads are done (and Qs visible on the first pass)
        for (int i = tid; i < TK * D; i += 256) {
            const int r = i >> 6, d = i & 63;
            const bool ok = k0 + r < N;
            Ks[r * LD + d] = ok ? to_f32(base[(size_t)(k0 + r) * ldq + inner + d]) : 0.f;
            Vs[r * LD + d] = ok ? to_f32(base[(size_t)(k0 + r) * ldq + 2 * inner + d]) : 0.f;
        }
        __syncthreads();
        float s[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s[i][j] = 0.f;
        for (int d = 0; d < D; ++d) {
            float qv[4], kv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) qv[i] = Qs[(4 * ty + i) * LD + d];
#pragma unroll
            for (int j = 0; j < 4; ++j) kv[j] = Ks[(4 * tx + j) * LD + d];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s[i][j] += qv[i] * kv[j];
        }
        bool kvalid[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int key = k0 + 4 * tx + j;
            kvalid[j] = key < N && (mask == nullptr || mask[(size_t)b * N + key] != 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float mt = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[i][j] = kvalid[j] ? s[i][j] * scale : -INFINITY;
                mt = fmaxf(mt, s[i][j]);
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) mt = fmaxf(mt, __shfl_xor(mt, off, 64));  // the 16 tx lanes of a row are contiguous
            const float m_new = fmaxf(m_i[i], mt);
            const float alpha = (m_i[i] == -INFINITY) ? 0.f : expf(m_i[i] - m_new);
            float rs = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float p = (s[i][j] == -INFINITY) ? 0.f : expf(s[i][j] - m_new);
                Ps[(4 * ty + i) * LD + 4 * tx + j] = p;
                rs += p;
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) rs += __shfl_xor(rs, off, 64);
            l_i[i] = l_i[i] * alpha + rs;
            m_i[i] = m_new;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[i][j] *= alpha;
        }
        __syncthreads();
        for (int key = 0; key < TK; ++key) {
            float pv[4], vv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pv[i] = Ps[(4 * ty + i) * LD + key];
#pragma unroll
            for (int j = 0; j < 4; ++j) vv[j] = Vs[key * LD + 4 * tx + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[i][j] += pv[i] * vv[j];
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = q0 + 4 * ty + i;
        if (q < N) {
            const float inv = l_i[i] > 0.f ? 1.0f / l_i[i] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[((size_t)b * bs + q) * ldo + h * D + 4 * tx + j] = from_f32<T>(o[i][j] * inv);
        }
    }
}

int launch_attention_fast(int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out, int ldo, hipStream_t stream, int bstride);  // attention_fast.hip

int launch_attention(int precision, int kernel_kind, int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out, int ldo,
                     hipStream_t stream, int bstride) {
    if (B <= 0 || N <= 0 || H <= 0) return 0;
    if (bstride <= 0) bstride = N;
    if (bstride < N || (bstride != N && mask)) return f5_fail(F5_EINVAL, "attention: batch stride below N, or a key mask with a batch stride");
    if (kernel_kind == 1) {
        if (!attention_fast_supported(precision, N, H)) return f5_fail(F5_EINVAL, "attention: tuned kernel does not support this problem");
        return launch_attention_fast(B, N, H, qkv, ldq, mask, out, ldo, stream, bstride);
    }
    const float scale = 0.125f;  // 1/sqrt(64) (SDPA default scale, modules.py:490)
    dim3 grid(cdiv(N, 64), H, B), block(256);
    if (precision == F5_PREC_BF16)
        hipLaunchKernelGGL((attn_ref_kernel<bf16_t>), grid, block, 0, stream, (const bf16_t*)qkv, ldq, H * 64, mask, (bf16_t*)out, ldo, N, bstride, scale);
    else
        hipLaunchKernelGGL((attn_ref_kernel<float>), grid, block, 0, stream, (const float*)qkv, ldq, H * 64, mask, (float*)out, ldo, N, bstride, scale);
    F5_LAUNCH_CHECK();
    return 0;
}
