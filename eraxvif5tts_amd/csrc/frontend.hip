// frontend.hip -- reference-audio front-end on the device: log-mel spectrogram and sample-rate conversion.
//
// Reference semantics (paths under /root/reference/src/f5_tts):
//   model/modules.py:75-101   get_vocos_mel_spectrogram: torchaudio MelSpectrogram(n_fft 1024, win 1024, hop 256, n_mels 100, power=1,
//                             center=True (reflect pad), periodic Hann, HTK mel scale, norm=None) -> clamp(min=1e-5).log()
//   infer/f5tts_wrapper.py:338-341  torchaudio.transforms.Resample(sr, 24000): sinc interpolation with a Hann window
//                             (lowpass_filter_width 6, rolloff 0.99), a strided FIR over the zero-padded waveform
// torchaudio is a third-party dependency that is absent from the reference tree; both algorithms are restated from their
// published definitions (oracle: oracle/cpu_ref.py mel_spectrogram, infer/audio.py resample; parity "unpinned").
//
// The STFT is a dense real DFT on the fp32-input MFMA (exact fp32 products): frames [B*T, n_fft] x windowed DFT matrix
// [2F, n_fft] -> (re | im), magnitude, HTK filterbank as a second small GEMM, log.  It runs once per sample() on a few
// hundred frames; the point of having it here is that no arithmetic of the path is left to PyTorch ops.
#include <cmath>
#include <cstring>
#include <map>

#include "gemm.h"
#include "kernels.h"
#include "runtime.h"

struct f5_frontend_s {
    f5_mel_config cfg;
    int F = 0, Fp = 0;  // one-sided bins, padded to a multiple of 32 (K of the filterbank GEMM)
    DevArena arena, work;
    size_t work_rows = 0;
    float *dft = nullptr, *fb = nullptr;  // [2F, n_fft] (Hann window folded in), [n_mels, Fp]
    float *frames = nullptr, *spec = nullptr, *mag = nullptr, *melT = nullptr;
    struct Resampler {
        int orig, neu, width, kw;
        float* kernels;  // [neu][kw]
    };
    std::map<std::pair<int, int>, Resampler> resamplers;
};

// ----------------------------------------------------------------------------- kernels
// frames[(b*T + t)][n] = wave[b][reflect(t*hop + n - pad)]: pad = n_fft / 2 for torch.stft(center=True, pad_mode="reflect") (vocos mel),
// pad = (n_fft - hop) / 2 for the explicit reflect padding + center=False of get_bigvgan_mel_spectrogram (modules.py:51-52)
__global__ __launch_bounds__(256) void mel_frames_kernel(const float* __restrict__ wave, int nw, int T, int n_fft, int hop, int pad, float* __restrict__ frames,
                                                         size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i % n_fft);
    const size_t row = i / n_fft;
    const int t = (int)(row % T), b = (int)(row / T);
    int s = t * hop + n - pad;
    if (s < 0) s = -s;
    if (s >= nw) s = 2 * (nw - 1) - s;
    frames[i] = wave[(size_t)b * nw + s];
}
// mag[r][k] = sqrt(re^2 + im^2 + eps) (eps = 0: torchaudio power = 1; 1e-9: modules.py:67), zero in the padding columns F..Fp-1
__global__ __launch_bounds__(256) void mel_magnitude_kernel(const float* __restrict__ spec, int lds, int F, int Fp, float eps, float* __restrict__ mag,
                                                            size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int k = (int)(i % Fp);
    const size_t r = i / Fp;
    float v = 0.f;
    if (k < F) {
        const float re = spec[r * lds + k], im = spec[r * lds + F + k];
        v = sqrtf(re * re + im * im + eps);
    }
    mag[i] = v;
}
// out[b][m][t] = log(max(melT[b*T + t][m], 1e-5))
__global__ __launch_bounds__(256) void mel_log_transpose_kernel(const float* __restrict__ melT, int T, int n_mels, float* __restrict__ out, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int t = (int)(i % T);
    const size_t bm = i / T;
    const int m = (int)(bm % n_mels), b = (int)(bm / n_mels);
    out[i] = logf(fmaxf(melT[((size_t)b * T + t) * n_mels + m], 1e-5f));
}
// y[b][i*neu + ph] = sum_k kernels[ph][k] * xpad[b][i*orig + k],  xpad = x with `width` zeros in front (and zeros behind)
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ x, int n, int orig, int neu, int width, int kw,
                                                       const float* __restrict__ kernels, float* __restrict__ y, int target, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % target), b = (int)(i / target);
    const int blk = j / neu, ph = j - blk * neu;
    const float* kr = kernels + (size_t)ph * kw;
    const float* xb = x + (size_t)b * n;
    const int s0 = blk * orig - width;
    float acc = 0.f;
    for (int k = 0; k < kw; ++k) {
        const int s = s0 + k;
        if (s >= 0 && s < n) acc += kr[k] * xb[s];
    }
    y[i] = acc;
}

// ----------------------------------------------------------------------------- handle
static double hz_to_mel_htk(double f) { return 2595.0 * log10(1.0 + f / 700.0); }
// librosa.filters.mel (Slaney / Auditory-toolbox scale, htk = False): linear below 1 kHz (200 / 3 Hz per mel), logarithmic above (log(6.4) / 27 per mel)
static double hz_to_mel_slaney(double f) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz_slaney(double m) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
    return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

extern "C" int f5_frontend_create(const f5_mel_config* c, f5_frontend_t* out) {
    if (!c || !out) return f5_fail(F5_EINVAL, "null argument");
    *out = nullptr;
    F5_TRY(f5_check_device());
    if (c->n_fft <= 0 || c->n_fft % 32 != 0 || c->hop <= 0 || c->win <= 0 || c->win > c->n_fft || c->n_mels <= 0 || c->sample_rate <= 0 ||
        (c->mel_type != F5_MEL_VOCOS && c->mel_type != F5_MEL_BIGVGAN) || (c->mel_type == F5_MEL_BIGVGAN && c->hop > c->n_fft))
        return f5_fail(F5_EINVAL, "bad mel config (n_fft a multiple of 32, win <= n_fft, mel_type 0 | 1)");
    f5_frontend_s* h = new f5_frontend_s();
    h->cfg = *c;
    const int N = c->n_fft, F = N / 2 + 1, Fp = (int)round_up(F, 32);
    h->F = F;
    h->Fp = Fp;
    // periodic Hann window of length win, centred inside n_fft (torch.stft pads a shorter window on both sides)
    std::vector<double> w(N, 0.0);
    const int off = (N - c->win) / 2;
    for (int n = 0; n < c->win; ++n) w[off + n] = 0.5 - 0.5 * cos(2.0 * M_PI * n / c->win);
    std::vector<float> dft((size_t)2 * F * N);
    for (int k = 0; k < F; ++k)
        for (int n = 0; n < N; ++n) {
            const double ang = 2.0 * M_PI * (double)((long long)k * n % N) / N;
            dft[(size_t)k * N + n] = (float)(w[n] * cos(ang));
            dft[(size_t)(F + k) * N + n] = (float)(-w[n] * sin(ang));
        }
    // torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk"), fp32 arithmetic as torch does it
    std::vector<float> fb((size_t)c->n_mels * Fp, 0.f);
    if (c->mel_type == F5_MEL_BIGVGAN) {
        // librosa.filters.mel(sr, n_fft, n_mels, fmin = 0, fmax = sr / 2, htk = False, norm = "slaney"), float64 arithmetic, float32 result: n_mels + 2
        // band edges equally spaced on the Slaney mel scale, triangles max(0, min(lower, upper)) over the FFT bin frequencies, every filter
        // scaled by 2 / (its band width in Hz) -- unit area.  (librosa is absent from the reference tree: restated from the published algorithm.)
        const int M = c->n_mels;
        std::vector<double> mel_f(M + 2);
        const double m_min = hz_to_mel_slaney(0.0), m_max = hz_to_mel_slaney(c->sample_rate / 2.0);
        for (int i = 0; i < M + 2; ++i) mel_f[i] = mel_to_hz_slaney(m_min + (m_max - m_min) * (double)i / (double)(M + 1));
        for (int m = 0; m < M; ++m) {
            const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
            for (int k = 0; k < F; ++k) {
                const double freq = (c->sample_rate / 2.0) * (double)k / (double)(F - 1);
                const double lower = (freq - mel_f[m]) / (mel_f[m + 1] - mel_f[m]), upper = (mel_f[m + 2] - freq) / (mel_f[m + 2] - mel_f[m + 1]);
                fb[(size_t)m * Fp + k] = (float)(std::max(0.0, std::min(lower, upper)) * enorm);
            }
        }
    } else {
        const int M = c->n_mels;
        std::vector<float> f_pts(M + 2);
        const float m_min = (float)hz_to_mel_htk(0.0), m_max = (float)hz_to_mel_htk(c->sample_rate / 2.0);
        for (int i = 0; i < M + 2; ++i) {
            const float m = m_min + (m_max - m_min) * (float)i / (float)(M + 1);
            f_pts[i] = 700.0f * (powf(10.0f, m / 2595.0f) - 1.0f);
        }
        for (int k = 0; k < F; ++k) {
            const float freq = (float)(c->sample_rate / 2) * (float)k / (float)(F - 1);
            for (int m = 0; m < M; ++m) {
                const float down = (freq - f_pts[m]) / (f_pts[m + 1] - f_pts[m]);
                const float up = (f_pts[m + 2] - freq) / (f_pts[m + 2] - f_pts[m + 1]);
                fb[(size_t)m * Fp + k] = fmaxf(0.0f, fminf(down, up));
            }
        }
    }
    int rc = f5_upload_f32(h->arena, dft.data(), dft.size(), &h->dft);
    if (!rc) rc = f5_upload_f32(h->arena, fb.data(), fb.size(), &h->fb);
    if (rc) {
        delete h;
        return rc;
    }
    *out = h;
    return 0;
}

extern "C" int f5_frontend_destroy(f5_frontend_t h) {
    delete h;
    return 0;
}

static unsigned blocks_of(size_t total) { return (unsigned)((total + 255) / 256); }

extern "C" int f5_frontend_mel(f5_frontend_t h, int B, int nw, const float* wave, float* mel, f5_stream_t stream) {
    if (!h || !wave || !mel || B <= 0) return f5_fail(F5_EINVAL, "bad argument");
    F5_TRY(f5_check_device());
    const f5_mel_config& c = h->cfg;
    const bool big = c.mel_type == F5_MEL_BIGVGAN;
    const int pad = big ? (c.n_fft - c.hop) / 2 : c.n_fft / 2;
    if (nw <= pad || (big && nw + 2 * pad < c.n_fft)) return f5_fail(F5_EINVAL, "mel: %d samples are too few for reflect padding of %d", nw, pad);
    hipStream_t st = (hipStream_t)stream;
    const int N = c.n_fft, F = h->F, Fp = h->Fp, T = big ? (nw + 2 * pad - c.n_fft) / c.hop + 1 : nw / c.hop + 1, M = c.n_mels;
    const size_t rows = (size_t)B * T;
    if (rows > h->work_rows) {  // grow the workspace (the mel runs once per sample(); not a per-step allocation)
        F5_HIP(hipStreamSynchronize(st));
        h->work.release();
        h->work_rows = 0;
        F5_TRY(h->work.alloc_t(&h->frames, rows * N, false));
        F5_TRY(h->work.alloc_t(&h->spec, rows * (size_t)(2 * F), false));
        F5_TRY(h->work.alloc_t(&h->mag, rows * Fp, false));
        F5_TRY(h->work.alloc_t(&h->melT, rows * M, false));
        h->work_rows = rows;
    }
    hipLaunchKernelGGL(mel_frames_kernel, dim3(blocks_of(rows * N)), dim3(256), 0, st, wave, nw, T, N, c.hop, pad, h->frames, rows * N);
    F5_LAUNCH_CHECK();
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = h->frames; g.lda = N; g.W = h->dft; g.ldw = N; g.M = (int)rows; g.N = 2 * F; g.K = N; g.out_f = h->spec; g.ldof = 2 * F;
    F5_TRY(launch_gemm(g, F5_PREC_FP32, GEMM_DENSE, EPI_STORE_F32, 0, st));
    hipLaunchKernelGGL(mel_magnitude_kernel, dim3(blocks_of(rows * Fp)), dim3(256), 0, st, h->spec, 2 * F, F, Fp, big ? 1e-9f : 0.0f, h->mag, rows * Fp);
    F5_LAUNCH_CHECK();
    memset(&g, 0, sizeof(g));
    g.A = h->mag; g.lda = Fp; g.W = h->fb; g.ldw = Fp; g.M = (int)rows; g.N = M; g.K = Fp; g.out_f = h->melT; g.ldof = M;
    F5_TRY(launch_gemm(g, F5_PREC_FP32, GEMM_DENSE, EPI_STORE_F32, 0, st));
    hipLaunchKernelGGL(mel_log_transpose_kernel, dim3(blocks_of((size_t)B * M * T)), dim3(256), 0, st, h->melT, T, M, mel, (size_t)B * M * T);
    F5_LAUNCH_CHECK();
    return 0;
}

extern "C" int f5_frontend_resample(f5_frontend_t h, int B, int n, int orig_freq, int new_freq, const float* wave, float* out, f5_stream_t stream) {
    if (!h || !wave || !out || B <= 0 || n <= 0 || orig_freq <= 0 || new_freq <= 0) return f5_fail(F5_EINVAL, "bad argument");
    F5_TRY(f5_check_device());
    hipStream_t st = (hipStream_t)stream;
    int a = orig_freq, b = new_freq;
    while (b) {
        const int t = a % b;
        a = b;
        b = t;
    }
    const int orig = orig_freq / a, neu = new_freq / a;
    const int target = (int)(((long long)neu * n + orig - 1) / orig);
    if (orig == neu) {
        F5_HIP(hipMemcpyAsync(out, wave, (size_t)B * n * sizeof(float), hipMemcpyDeviceToDevice, st));
        return 0;
    }
    auto key = std::make_pair(orig, neu);
    auto it = h->resamplers.find(key);
    if (it == h->resamplers.end()) {
        // torchaudio.functional._get_sinc_resample_kernel (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99), float64 like infer/audio.py
        const int lpw = 6;
        const double rolloff = 0.99, base = std::min(orig, neu) * rolloff;
        const int width = (int)ceil(lpw * orig / base), kw = 2 * width + orig;
        std::vector<float> k((size_t)neu * kw);
        for (int ph = 0; ph < neu; ++ph)
            for (int j = 0; j < kw; ++j) {
                double t = (-(double)ph / neu + (double)(j - width) / orig) * base;
                t = std::max(-(double)lpw, std::min((double)lpw, t));
                const double win = pow(cos(t * M_PI / lpw / 2.0), 2.0);
                const double tp = t * M_PI;
                const double sinc = tp == 0.0 ? 1.0 : sin(tp) / tp;
                k[(size_t)ph * kw + j] = (float)(sinc * win * (base / orig));
            }
        f5_frontend_s::Resampler r{orig, neu, width, kw, nullptr};
        F5_HIP(hipStreamSynchronize(st));
        F5_TRY(f5_upload_f32(h->arena, k.data(), k.size(), &r.kernels));
        it = h->resamplers.emplace(key, r).first;
    }
    const f5_frontend_s::Resampler& r = it->second;
    const size_t total = (size_t)B * target;
    hipLaunchKernelGGL(resample_kernel, dim3(blocks_of(total)), dim3(256), 0, st, wave, n, r.orig, r.neu, r.width, r.kw, r.kernels, out, target, total);
    F5_LAUNCH_CHECK();
    return 0;
}
