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

// ISTFTHead as an FFT: one workgroup per frame.  mag = clip(exp(m), 1e2), X_k = mag (cos p + i sin p) for the 513 one-sided bins, Hermitian
// extension to 1024 bins in LDS, unnormalised inverse transform by five radix-4 Stockham passes (256 threads = one butterfly per thread and
// pass, ping-pong between two 8 KiB LDS buffers, twiddles from a 1024-entry table in LDS), then frame[n] = Re x[n] * window[n] / n_fft.
// 51 kFLOP per frame instead of the 2.1 MFLOP of the dense DFT it replaces: the head becomes HBM-bound (reads 4.1 KB, writes 4 KB per frame).
// Only n_fft = 1024 (= 4^5); other sizes keep the DFT GEMM.
__global__ __launch_bounds__(256) void vocos_ifft1024_kernel(const float* __restrict__ head, int ldh, const float* __restrict__ wscaled /*[1024] window / n_fft*/,
                                                             const float* __restrict__ twiddle /*[1024][2] cos, sin of 2 pi j / 1024*/,
                                                             float* __restrict__ frames) {
    constexpr int N = 1024, F = 513, T4 = 256;
    __shared__ float2 buf[2][N];
    __shared__ float2 tw[N];
    const int row = blockIdx.x, j = threadIdx.x;
    const float* hr = head + (size_t)row * ldh;
#pragma unroll
    for (int q = 0; q < 4; ++q) tw[j + 256 * q] = reinterpret_cast<const float2*>(twiddle)[j + 256 * q];
    // bins k = j, j + 256 (and 512 by thread 0): X_k and its mirror conj(X_k) at 1024 - k
    for (int k = j; k < F; k += 256) {
        const float mag = fminf(expf(hr[k]), 100.0f);
        const float ph = hr[F + k];
        float re = mag * cosf(ph), im = mag * sinf(ph);
        if (k == 0 || k == N / 2) im = 0.f;  // irfft ignores the imaginary part of DC and Nyquist
        buf[0][k] = make_float2(re, im);
        if (k != 0 && k != N / 2) buf[0][N - k] = make_float2(re, -im);
    }
    __syncthreads();
    int cur = 0;
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const int Ns = 1 << (2 * p), k = j & (Ns - 1), tstep = 256 >> (2 * p);  // twiddle index of u[r]: k * r * (256 / Ns)
        float2 u[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 a = buf[cur][j + r * T4];
            const float2 w = tw[(k * r * tstep) & (N - 1)];
            u[r] = make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
        }
        // inverse radix-4 butterfly: v_q = sum_r u_r i^(q r)
        const float2 s02 = make_float2(u[0].x + u[2].x, u[0].y + u[2].y), d02 = make_float2(u[0].x - u[2].x, u[0].y - u[2].y);
        const float2 s13 = make_float2(u[1].x + u[3].x, u[1].y + u[3].y), d13 = make_float2(u[1].x - u[3].x, u[1].y - u[3].y);
        const int j0 = ((j - k) << 2) + k;
        buf[cur ^ 1][j0] = make_float2(s02.x + s13.x, s02.y + s13.y);
        buf[cur ^ 1][j0 + Ns] = make_float2(d02.x - d13.y, d02.y + d13.x);       // d02 + i d13
        buf[cur ^ 1][j0 + 2 * Ns] = make_float2(s02.x - s13.x, s02.y - s13.y);
        buf[cur ^ 1][j0 + 3 * Ns] = make_float2(d02.x + d13.y, d02.y - d13.x);   // d02 - i d13
        cur ^= 1;
        __syncthreads();
    }
    float* fr = frames + (size_t)row * N;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = j + 256 * q;
        fr[n] = buf[cur][n].x * wscaled[n];
    }
}
int launch_vocos_ifft1024(const float* head, int ldh, int rows, const float* wscaled, const float* twiddle, float* frames, hipStream_t stream) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(vocos_ifft1024_kernel, dim3(rows), dim3(256), 0, stream, head, ldh, wscaled, twiddle, frames);
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
