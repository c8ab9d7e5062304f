// duration.hip -- the optional duration predictor (reference model/duration_predictor.py:28-46), fp32 throughout.
//
//   x = Embedding(tokens + 1) * mask                    [b, C, nt]   (0 = filler id, batch pad -1 -> 0)
//   h = GroupNorm1(relu(Conv1d(C -> F, k, pad k/2)(x))) * mask
//   h = GroupNorm1(relu(Conv1d(F -> F, k, pad k/2)(h))) * mask
//   out = (Conv1d(F -> 1, 1)(h)) * mask                 [b, nt]      (log-durations per token)
//
// nn.GroupNorm(1, F) normalises over ALL F x nt values of one utterance (not per position), eps 1e-5, affine per channel; Dropout is
// the identity at inference.  A text is a few hundred tokens and the net a few hundred channels (< 0.1 GFLOP): HBM / latency bound,
// three small kernels per layer, no MFMA.
#include "kernels.h"
#include "runtime.h"

namespace {

// out[b][f][t] = relu(bias[f] + sum_{c, j} w[f][c][j] * in(b, c, t + j - k/2));  in() = embedding gather (GATHER) or a [b, C, nt] buffer,
// multiplied by mask[b][t'] and zero outside [0, nt).  One thread per (f, t); the weight row of f is shared by the block's threads.
// cadd (GATHER only, may be null): the speaker conditioning cond(g) [b][C][g_nt], g_nt = 1 (one vector per utterance) or nt, added to the embedding.
template <bool GATHER>
__global__ __launch_bounds__(256) void dp_conv_relu_kernel(const float* __restrict__ in, const float* __restrict__ emb, const int32_t* __restrict__ tok,
                                                           int add_one, int vocab_rows, const int32_t* __restrict__ mask, const float* __restrict__ w,
                                                           const float* __restrict__ bias, int C, int F, int k, int nt, float* __restrict__ out,
                                                           const float* __restrict__ cadd, int g_nt) {
    const int t = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y, b = blockIdx.z;
    if (t >= nt) return;
    const int pad = k / 2;
    float acc = bias[f];
    for (int j = 0; j < k; ++j) {
        const int ts = t + j - pad;
        if (ts < 0 || ts >= nt) continue;
        if (mask[(size_t)b * nt + ts] == 0) continue;
        const float* wr = w + (size_t)f * C * k + j;
        if constexpr (GATHER) {
            int id = tok[(size_t)b * nt + ts] + add_one;
            id = id < 0 ? 0 : (id >= vocab_rows ? vocab_rows - 1 : id);
            const float* er = emb + (size_t)id * C;
            if (cadd) {
                const float* gr = cadd + (size_t)b * C * g_nt + (g_nt == 1 ? 0 : ts);
                for (int c = 0; c < C; ++c) acc = __builtin_fmaf(wr[(size_t)c * k], er[c] + gr[(size_t)c * g_nt], acc);
            } else {
                for (int c = 0; c < C; ++c) acc = __builtin_fmaf(wr[(size_t)c * k], er[c], acc);
            }
        } else {
            const float* xr = in + (size_t)b * C * nt + ts;
            for (int c = 0; c < C; ++c) acc = __builtin_fmaf(wr[(size_t)c * k], xr[(size_t)c * nt], acc);
        }
    }
    out[((size_t)b * F + f) * nt + t] = fmaxf(acc, 0.f);
}

// GroupNorm(1 group) over the F x nt values of utterance b (two passes, fp32), affine per channel, then * mask.  One block per utterance.
__global__ __launch_bounds__(256) void dp_groupnorm_kernel(float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const int32_t* __restrict__ mask, int F, int nt, float eps) {
    __shared__ float red[8];
    const int b = blockIdx.x, n = F * nt;
    float* xb = x + (size_t)b * n;
    auto block_sum = [&](float v) {
        v = wave_sum(v);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        return red[0] + red[1] + red[2] + red[3];
    };
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += xb[i];
    const float mean = block_sum(s) / (float)n;
    float q = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float d = xb[i] - mean;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(block_sum(q) / (float)n + eps);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int f = i / nt, t = i - f * nt;
        const float v = (xb[i] - mean) * rstd * gamma[f] + beta[f];
        xb[i] = mask[(size_t)b * nt + t] ? v : 0.f;
    }
}

// out[b][t] = (bp + sum_f wp[f] * h[b][f][t]) * mask[b][t]   (h is already masked)
__global__ __launch_bounds__(256) void dp_proj_kernel(const float* __restrict__ h, const float* __restrict__ wp, const float* __restrict__ bp,
                                                      const int32_t* __restrict__ mask, int F, int nt, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t >= nt) return;
    float acc = bp[0];
    const float* hr = h + (size_t)b * F * nt + t;
    for (int f = 0; f < F; ++f) acc = __builtin_fmaf(wp[f], hr[(size_t)f * nt], acc);
    out[(size_t)b * nt + t] = mask[(size_t)b * nt + t] ? acc : 0.f;
}

// cond(g): Conv1d(gin -> C, kernel 1): out[b][c][t] = bias[c] + sum_j w[c][j] * g[b][j][t]   (duration_predictor.py:25-26,33-35)
__global__ __launch_bounds__(256) void dp_cond_kernel(const float* __restrict__ g, const float* __restrict__ w, const float* __restrict__ bias, int gin, int C,
                                                      int g_nt, float* __restrict__ out) {
    const int t = blockIdx.x * 256 + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (t >= g_nt) return;
    float acc = bias[c];
    for (int j = 0; j < gin; ++j) acc = __builtin_fmaf(w[(size_t)c * gin + j], g[((size_t)b * gin + j) * g_nt + t], acc);
    out[((size_t)b * C + c) * g_nt + t] = acc;
}

}  // namespace

extern "C" int f5_duration_predict(const f5_duration_weights* w, int batch, int nt, const int32_t* tokens, int add_one, const int32_t* mask,
                                   float* scratch, float* out, f5_stream_t stream) {
    return f5_duration_predict_g(w, batch, nt, tokens, add_one, mask, nullptr, 0, scratch, out, stream);
}

extern "C" int f5_duration_predict_g(const f5_duration_weights* w, int batch, int nt, const int32_t* tokens, int add_one, const int32_t* mask,
                                     const float* g, int g_nt, float* scratch, float* out, f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (!w || !tokens || !mask || !scratch || !out) return f5_fail(F5_EINVAL, "f5_duration_predict: null argument");
    if (batch <= 0 || nt <= 0 || w->in_channels <= 0 || w->filter_channels <= 0 || w->kernel_size <= 0 || (w->kernel_size & 1) == 0 || w->vocab_rows <= 0)
        return f5_fail(F5_EINVAL, "f5_duration_predict: bad sizes (kernel_size must be odd: Conv1d(padding=k//2) keeps the length only then)");
    if (!w->text_embed || !w->conv1_w || !w->conv1_b || !w->norm1_w || !w->norm1_b || !w->conv2_w || !w->conv2_b || !w->norm2_w || !w->norm2_b ||
        !w->proj_w || !w->proj_b)
        return f5_fail(F5_EINVAL, "f5_duration_predict: missing weight tensor");
    hipStream_t st = (hipStream_t)stream;
    const int C = w->in_channels, F = w->filter_channels, k = w->kernel_size;
    float* h1 = scratch;
    float* h2 = scratch + (size_t)batch * F * nt;
    const float* cadd = nullptr;
    if (g) {
        if (w->gin_channels <= 0 || !w->cond_w || !w->cond_b) return f5_fail(F5_EINVAL, "f5_duration_predict_g: g given but the net has no cond layer (gin_channels = 0)");
        if (g_nt != 1 && g_nt != nt) return f5_fail(F5_EINVAL, "f5_duration_predict_g: g must be [batch, gin, 1] or [batch, gin, nt]");
        float* cbuf = scratch + (size_t)2 * batch * F * nt;
        hipLaunchKernelGGL(dp_cond_kernel, dim3(cdiv(g_nt, 256), C, batch), dim3(256), 0, st, g, w->cond_w, w->cond_b, w->gin_channels, C, g_nt, cbuf);
        F5_LAUNCH_CHECK();
        cadd = cbuf;
    }
    const dim3 cgrid(cdiv(nt, 256), F, batch);
    hipLaunchKernelGGL((dp_conv_relu_kernel<true>), cgrid, dim3(256), 0, st, (const float*)nullptr, w->text_embed, tokens, add_one, w->vocab_rows, mask,
                       w->conv1_w, w->conv1_b, C, F, k, nt, h1, cadd, g_nt);
    F5_LAUNCH_CHECK();
    hipLaunchKernelGGL(dp_groupnorm_kernel, dim3(batch), dim3(256), 0, st, h1, w->norm1_w, w->norm1_b, mask, F, nt, 1e-5f);
    F5_LAUNCH_CHECK();
    hipLaunchKernelGGL((dp_conv_relu_kernel<false>), cgrid, dim3(256), 0, st, (const float*)h1, (const float*)nullptr, (const int32_t*)nullptr, 0, 0, mask,
                       w->conv2_w, w->conv2_b, F, F, k, nt, h2, (const float*)nullptr, 0);
    F5_LAUNCH_CHECK();
    hipLaunchKernelGGL(dp_groupnorm_kernel, dim3(batch), dim3(256), 0, st, h2, w->norm2_w, w->norm2_b, mask, F, nt, 1e-5f);
    F5_LAUNCH_CHECK();
    hipLaunchKernelGGL(dp_proj_kernel, dim3(cdiv(nt, 256), batch), dim3(256), 0, st, (const float*)h2, w->proj_w, w->proj_b, mask, F, nt, out);
    F5_LAUNCH_CHECK();
    return 0;
}
