// vocos.hip -- element kernels of the Vocos vocoder head (plug point B: vocoder.decode, f5tts_wrapper.py:524).
// The ISTFT is computed as  spectrum-rows x (windowed inverse-DFT matrix)  on the fp32-input MFMA GEMM followed by
// an HBM-bound overlap-add; exp/clip/cos/sin are fused into the spectrum packing.
#include "kernels.h"

template <typename TO>
__global__ __launch_bounds__(256) void vocos_im2col_kernel(const float* __restrict__ mel, int B, int C, int T, TO* __restrict__ dst, int Kp) {
    const int row = blockIdx.x;  // b * T + t
    const int b = row / T, t = row % T;
    for (int col = threadIdx.x; col < Kp; col += 256) {
        float v = 0.f;
        if (col < 7 * C) {
            const int tap = col / C, c = col % C;
            const int st = t + tap - 3;  // Conv1d(kernel 7, padding 3)
            if (st >= 0 && st < T) v = mel[((size_t)b * C + c) * T + st];
        }
        dst[(size_t)row * Kp + col] = from_f32<TO>(v);
    }
}
int launch_vocos_im2col(int precision_out, const float* mel, int B, int C, int T, void* dst, int Kp, hipStream_t stream) {
    if (B * T <= 0) return 0;
    if (precision_out == F5_PREC_BF16)
        hipLaunchKernelGGL((vocos_im2col_kernel<bf16_t>), dim3(B * T), dim3(256), 0, stream, mel, B, C, T, (bf16_t*)dst, Kp);
    else
        hipLaunchKernelGGL((vocos_im2col_kernel<float>), dim3(B * T), dim3(256), 0, stream, mel, B, C, T, (float*)dst, Kp);
    F5_LAUNCH_CHECK();
    return 0;
}

// ISTFTHead: mag = clip(exp(m), max=1e2); S = mag * (cos p + i sin p)
template <typename TO>
__global__ __launch_bounds__(256) void vocos_spectrum_kernel(const float* __restrict__ head, int ldh, int rows, int F, TO* __restrict__ dst, int Kp) {
    const int row = blockIdx.x;
    for (int col = threadIdx.x; col < Kp; col += 256) {
        float v = 0.f;
        if (col < 2 * F) {
            const int k = col < F ? col : col - F;
            const float mag = fminf(expf(head[(size_t)row * ldh + k]), 100.0f);
            const float ph = head[(size_t)row * ldh + F + k];
            v = col < F ? mag * cosf(ph) : mag * sinf(ph);
        }
        dst[(size_t)row * Kp + col] = from_f32<TO>(v);
    }
}
int launch_vocos_spectrum(int precision_out, const float* head, int ldh, int rows, int F, void* dst, int Kp, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (precision_out == F5_PREC_BF16)
        hipLaunchKernelGGL((vocos_spectrum_kernel<bf16_t>), dim3(rows), dim3(256), 0, stream, head, ldh, rows, F, (bf16_t*)dst, Kp);
    else
        hipLaunchKernelGGL((vocos_spectrum_kernel<float>), dim3(rows), dim3(256), 0, stream, head, ldh, rows, F, (float*)dst, Kp);
    F5_LAUNCH_CHECK();
    return 0;
}

// torch.istft(center=True): overlap-add the windowed frames, divide by the overlap-added squared window, trim n_fft/2.
__global__ __launch_bounds__(256) void vocos_ola_kernel(const float* __restrict__ frames, int T, int n_fft, int hop, const float* __restrict__ wsq,
                                                        float* __restrict__ wave, int out_len) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= out_len) return;
    const int u = s + n_fft / 2;
    int t_hi = u / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    int t_lo = (u - n_fft + hop) / hop;  // ceil((u - n_fft + 1) / hop) for u - n_fft + 1 >= 0
    if (u - n_fft + 1 <= 0) t_lo = 0;
    float acc = 0.f, env = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) {
        const int j = u - t * hop;
        acc += frames[((size_t)b * T + t) * n_fft + j];
        env += wsq[j];
    }
    wave[(size_t)b * out_len + s] = acc / env;
}
int launch_vocos_ola(const float* frames, int B, int T, int n_fft, int hop, const float* wsq, float* wave, hipStream_t stream) {
    const int out_len = (T - 1) * hop;
    if (B <= 0 || out_len <= 0) return 0;
    hipLaunchKernelGGL(vocos_ola_kernel, dim3(cdiv(out_len, 256), B), dim3(256), 0, stream, frames, T, n_fft, hop, wsq, wave, out_len);
    F5_LAUNCH_CHECK();
    return 0;
}
