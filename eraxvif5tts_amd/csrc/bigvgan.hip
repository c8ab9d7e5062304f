// bigvgan.hip -- the BigVGAN-v2 generator (mel -> waveform) behind plug point B of the reference: infer/utils_infer.py:125-138 loads it from a
// third-party checkout (`third_party.BigVGAN.bigvgan.BigVGAN.from_pretrained("nvidia/bigvgan_v2_24khz_100band_256x")`, remove_weight_norm()) and
// infer/f5tts_wrapper.py:526 / eval/eval_infer_batch.py:189 call `vocoder(mel)`.
//
// PARITY UNPINNED: that checkout is ABSENT from the reference tree, no checkpoint exists offline and no reference test covers it.  The architecture
// is restated from the published BigVGAN-v2 source as recalled (oracle: oracle/cpu_ref.py bigvgan_forward, which the tests compare this file with):
//   conv_pre Conv1d(mels -> C0, k 7)
//   per stage i:  ConvTranspose1d(C -> C / 2, k_i, stride u_i, padding (k_i - u_i) / 2), then the MEAN of num_kernels AMPBlock1(C / 2, kernel, dilations):
//       for d in dilations:  xt = AA-snake(x); xt = Conv1d(k, dilation d)(xt); xt = AA-snake(xt); xt = Conv1d(k, dilation 1)(xt); x = xt + x
//   AA-snake (anti-aliased SnakeBeta: alias_free_activation Activation1d): 2x up-sampling by a 12-tap Kaiser-windowed sinc (replicate padding),
//       x + sin^2(alpha x) / (beta + 1e-9) per channel (alpha, beta stored as logarithms when snake_logscale), 2x low-pass down-sampling (12 taps)
//   AA-snake -> conv_post Conv1d(C -> 1, k 7, optional bias) -> clamp(-1, 1) or tanh.
//
// MI355X form: activations are TIME-major [T, C] fp32 (rows = time: every Conv1d is an im2col + GEMM on the fp32-input MFMA tile kernel, exact fp32
// products; the transposed convolution is a polyphase GEMM: k = R u, so every output sample takes R input frames, one GEMM of N = u x C_out);
// the anti-aliased activation is ONE kernel (a 32-step x 32/64-channel tile staged in LDS: up-sampling, snake and down-sampling never touch HBM);
// the residual add of a block's second convolution rides in the GEMM epilogue (EPI_RESID).  One utterance at a time (the wrapper decodes per chunk).
#include <cmath>
#include <cstring>

#include "gemm.h"
#include "kernels.h"
#include "runtime.h"

struct BvConv {
    float *w = nullptr, *b = nullptr;  // [Cout][Kp] (column = tap * Cin + c), [Cout]
    int cin = 0, cout = 0, k = 0, kp = 0;
};
struct BvUp {
    float *w = nullptr, *b = nullptr;  // [(u * Cout)][Kp] (row = phase * Cout + co, column = m * Cin + ci for input frame t0 - m), [Cout]
    int cin = 0, cout = 0, k = 0, u = 0, R = 0, kp = 0, pad = 0;
};
struct BvSnake {
    float *a = nullptr, *invb = nullptr;  // alpha and 1 / (beta + 1e-9), both already exponentiated when snake_logscale
};
struct BvBlock {
    BvConv c1[3], c2[3];
    BvSnake act[6];
    int dil[3] = {1, 3, 5};
};

struct f5_bigvgan_s {
    f5_bigvgan_config cfg;
    SlotMap slots;
    bool finalized = false;
    DevArena arena, work;
    size_t work_T = 0;
    BvConv conv_pre, conv_post;
    std::vector<BvUp> ups;
    std::vector<BvBlock> blocks;
    BvSnake act_post;
    float up_f[12], dn_f[12];
    bool have_filters = false;
    float *x = nullptr, *y = nullptr, *xt = nullptr, *xt2 = nullptr, *xs = nullptr, *col = nullptr, *tmp = nullptr;
};

// ----------------------------------------------------------------------------- kernels
// col[t][tap * C + c] = x[t + (tap - (k - 1) / 2) * dil][c] (0 outside [0, T)); columns k * C .. Kp - 1 are zero.  chan_major: x is [C][T] (the mel input)
__global__ __launch_bounds__(256) void bv_im2col_kernel(const float* __restrict__ x, int T, int C, int k, int dil, int Kp, int chan_major,
                                                        float* __restrict__ col, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % Kp);
    const int t = (int)(i / Kp);
    float v = 0.f;
    if (q < k * C) {
        const int tap = q / C, c = q - tap * C;
        const int s = t + (tap - (k - 1) / 2) * dil;
        if (s >= 0 && s < T) v = chan_major ? x[(size_t)c * T + s] : x[(size_t)s * C + c];
    }
    col[i] = v;
}
// transposed convolution, gather side: col[t0][m * C + c] = x[t0 - m][c] for t0 in [0, T], m in [0, R)
__global__ __launch_bounds__(256) void bv_up_gather_kernel(const float* __restrict__ x, int T, int C, int R, int Kp, float* __restrict__ col, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int q = (int)(i % Kp);
    const int t0 = (int)(i / Kp);
    float v = 0.f;
    if (q < R * C) {
        const int m = q / C, c = q - m * C;
        const int s = t0 - m;
        if (s >= 0 && s < T) v = x[(size_t)s * C + c];
    }
    col[i] = v;
}
// transposed convolution, scatter side: y[p][co] = tmp[t0][r * Cout + co] + bias[co] with q = p + pad, t0 = q / u, r = q % u, p in [0, T u)
__global__ __launch_bounds__(256) void bv_up_scatter_kernel(const float* __restrict__ tmp, int T, int Cout, int u, int pad, const float* __restrict__ bias,
                                                            float* __restrict__ y, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int co = (int)(i % Cout);
    const int p = (int)(i / Cout);
    const int q = p + pad, t0 = q / u, r = q - t0 * u;
    y[i] = tmp[(size_t)t0 * (u * Cout) + r * Cout + co] + bias[co];
}
struct BvFilters {
    float up[12], dn[12];
};
// Anti-aliased SnakeBeta, one kernel.  Output step t needs the activated up-sampled signal z[n], n = 2t - 5 .. 2t + 6 (replicate-clamped to [0, 2T));
// z[n] = snake(up[n]), up[n] = 2 sum_m xpad[m] f_up[n + 15 - 2m] over the six m with 0 <= n + 15 - 2m <= 11, xpad[m] = x[clamp(m - 5, 0, T - 1)].
// Tile: TT = 32 output steps x CW channels; x rows t0 - 6 .. t0 + TT + 5 and z values 2 t0 - 5 .. 2 t0 + 2 TT + 4 live in LDS.
template <int CW>
__global__ __launch_bounds__(256) void bv_aa_snake_kernel(const float* __restrict__ x, int T, int C, const float* __restrict__ alpha, const float* __restrict__ invb,
                                                          BvFilters f, float* __restrict__ out) {
    constexpr int TT = 32, NX = TT + 12, NZ = 2 * TT + 10, TY = 256 / CW;
    __shared__ float xs[NX][CW], zs[NZ][CW];
    const int cl = threadIdx.x % CW, ty = threadIdx.x / CW;
    const int c = blockIdx.x * CW + cl;
    const int t0 = blockIdx.y * TT;
    const bool okc = c < C;
    for (int r = ty; r < NX; r += TY) {
        int s = t0 - 6 + r;
        s = s < 0 ? 0 : (s > T - 1 ? T - 1 : s);
        xs[r][cl] = okc ? x[(size_t)s * C + c] : 0.f;
    }
    __syncthreads();
    const float a = okc ? alpha[c] : 1.f, ib = okc ? invb[c] : 0.f;
    for (int r = ty; r < NZ; r += TY) {
        int n = 2 * t0 - 5 + r;
        n = n < 0 ? 0 : (n > 2 * T - 1 ? 2 * T - 1 : n);
        // m from ceil((n + 4) / 2) to floor((n + 15) / 2); xpad[m] = x[clamp(m - 5)] = xs[clamp(m - 5) - (t0 - 6)]
        const int m_lo = (n + 5) >> 1;  // ceil((n + 4) / 2) for n >= 0
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const int m = m_lo + q;
            const int tap = n + 15 - 2 * m;
            int s = m - 5;
            s = s < 0 ? 0 : (s > T - 1 ? T - 1 : s);
            if (tap >= 0 && tap < 12) acc = __builtin_fmaf(xs[s - (t0 - 6)][cl], f.up[tap], acc);
        }
        const float u = 2.0f * acc;
        const float s1 = sinf(u * a);
        zs[r][cl] = u + ib * (s1 * s1);
    }
    __syncthreads();
    for (int r = ty; r < TT; r += TY) {
        const int t = t0 + r;
        if (t < T && okc) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                int n = 2 * t + j - 5;
                n = n < 0 ? 0 : (n > 2 * T - 1 ? 2 * T - 1 : n);
                acc = __builtin_fmaf(zs[n - (2 * t0 - 5)][cl], f.dn[j], acc);
            }
            out[(size_t)t * C + c] = acc;
        }
    }
}
// dst = dst * ds + src * ss
__global__ __launch_bounds__(256) void bv_axpby_kernel(float* __restrict__ dst, const float* __restrict__ src, float ds, float ss, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = dst[i] * ds + src[i] * ss;
}
__global__ __launch_bounds__(256) void bv_final_kernel(const float* __restrict__ src, int use_tanh, float* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = use_tanh ? tanhf(src[i]) : fminf(1.0f, fmaxf(-1.0f, src[i]));
}

// ----------------------------------------------------------------------------- handle
static void bslot(SlotMap& s, const std::string& n, std::vector<int64_t> shape) { s[n].shape = std::move(shape); }

// modified Bessel function I0 (torch.kaiser_window uses it): power series, converges fast for the beta of a 12-tap filter
static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 64; ++k) {
        term *= q / ((double)k * k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}
// alias_free_activation/torch/filter.py: kaiser_sinc_filter1d(cutoff, half_width, kernel_size), even kernel sizes
static void kaiser_sinc_filter(double cutoff, double half_width, int ks, float* out) {
    const int half = ks / 2;
    const double delta_f = 4.0 * half_width, A = 2.285 * (half - 1) * M_PI * delta_f + 7.95;
    const double beta = A > 50.0 ? 0.1102 * (A - 8.7) : (A >= 21.0 ? 0.5842 * pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0) : 0.0);
    std::vector<double> f(ks);
    double sum = 0.0;
    for (int i = 0; i < ks; ++i) {
        const double r = 2.0 * i / (ks - 1) - 1.0;  // torch.kaiser_window(periodic=False)
        const double win = bessel_i0(beta * sqrt(std::max(0.0, 1.0 - r * r))) / bessel_i0(beta);
        const double t = (i - half) + 0.5, a = 2.0 * cutoff * t;
        const double sinc = a == 0.0 ? 1.0 : sin(M_PI * a) / (M_PI * a);
        f[i] = 2.0 * cutoff * win * sinc;
        sum += f[i];
    }
    for (int i = 0; i < ks; ++i) out[i] = (float)(f[i] / sum);
}

extern "C" int f5_bigvgan_create(const f5_bigvgan_config* c, f5_bigvgan_t* out) {
    if (!c || !out) return f5_fail(F5_EINVAL, "null argument");
    *out = nullptr;
    F5_TRY(f5_check_device());
    if (c->num_mels <= 0 || c->upsample_initial_channel <= 0 || c->num_upsamples <= 0 || c->num_upsamples > 8 || c->num_kernels <= 0 || c->num_kernels > 4)
        return f5_fail(F5_EINVAL, "bad BigVGAN config");
    if (c->upsample_initial_channel % (1 << c->num_upsamples) != 0) return f5_fail(F5_EINVAL, "upsample_initial_channel must be divisible by 2^num_upsamples");
    for (int i = 0; i < c->num_upsamples; ++i) {
        const int u = c->upsample_rates[i], k = c->upsample_kernel_sizes[i];
        if (u <= 0 || k < u || k % u != 0 || ((k - u) & 1)) return f5_fail(F5_ENOTSUP, "upsample stage %d: kernel %d / rate %d (kernel must be a multiple of the rate, k - u even)", i, k, u);
    }
    for (int j = 0; j < c->num_kernels; ++j)
        if (c->resblock_kernel_sizes[j] <= 0 || c->resblock_kernel_sizes[j] % 2 == 0) return f5_fail(F5_EINVAL, "resblock kernel sizes must be odd");
    f5_bigvgan_s* v = new f5_bigvgan_s();
    v->cfg = *c;
    SlotMap& s = v->slots;
    int64_t ch = c->upsample_initial_channel;
    bslot(s, "conv_pre.weight", {ch, c->num_mels, 7});
    bslot(s, "conv_pre.bias", {ch});
    for (int i = 0; i < c->num_upsamples; ++i) {
        const std::string p = "ups." + std::to_string(i) + ".0.";
        bslot(s, p + "weight", {ch, ch / 2, c->upsample_kernel_sizes[i]});
        bslot(s, p + "bias", {ch / 2});
        ch /= 2;
        for (int j = 0; j < c->num_kernels; ++j) {
            const std::string r = "resblocks." + std::to_string(i * c->num_kernels + j) + ".";
            for (int t = 0; t < 3; ++t)
                for (const char* cv : {"convs1.", "convs2."}) {
                    bslot(s, r + cv + std::to_string(t) + ".weight", {ch, ch, c->resblock_kernel_sizes[j]});
                    bslot(s, r + cv + std::to_string(t) + ".bias", {ch});
                }
            for (int a = 0; a < 6; ++a) {
                bslot(s, r + "activations." + std::to_string(a) + ".act.alpha", {ch});
                bslot(s, r + "activations." + std::to_string(a) + ".act.beta", {ch});
            }
        }
    }
    bslot(s, "activation_post.act.alpha", {ch});
    bslot(s, "activation_post.act.beta", {ch});
    bslot(s, "conv_post.weight", {1, ch, 7});
    if (c->use_bias_at_final) bslot(s, "conv_post.bias", {1});
    kaiser_sinc_filter(0.25, 0.3, 12, v->up_f);  // UpSample1d / DownSample1d(ratio 2): cutoff 0.5 / ratio, half_width 0.6 / ratio, 12 taps
    kaiser_sinc_filter(0.25, 0.3, 12, v->dn_f);
    *out = v;
    return 0;
}

extern "C" int f5_bigvgan_has_tensor(f5_bigvgan_t v, const char* name, int64_t* numel) {
    if (!v || !name) return 0;
    if (strcmp(name, "aa_up_filter") == 0 || strcmp(name, "aa_down_filter") == 0) {
        if (numel) *numel = 12;
        return 1;
    }
    auto it = v->slots.find(name);
    if (it == v->slots.end()) return 0;
    if (numel) *numel = it->second.numel();
    return 1;
}

extern "C" int f5_bigvgan_set_tensor(f5_bigvgan_t v, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!v || !name || !host || !shape) return f5_fail(F5_EINVAL, "null argument");
    if (v->finalized) return f5_fail(F5_ESTATE, "vocoder already finalized");
    if (strcmp(name, "aa_up_filter") == 0 || strcmp(name, "aa_down_filter") == 0) {  // the 12-tap filters a checkpoint carries as buffers (optional)
        int64_t n = 1;
        for (int i = 0; i < ndim; ++i) n *= shape[i];
        if (n != 12) return f5_fail(F5_EINVAL, "%s must have 12 taps", name);
        memcpy(name[3] == 'u' ? v->up_f : v->dn_f, host, 12 * sizeof(float));
        return 0;
    }
    return f5_slot_set(v->slots, name, host, shape, ndim);
}

static int bv_upload_conv(f5_bigvgan_s* v, const std::string& name, int cin, int cout, int k, bool bias, BvConv* dst) {
    const std::vector<float>& w = v->slots[name + ".weight"].host;  // [cout][cin][k]
    dst->cin = cin; dst->cout = cout; dst->k = k; dst->kp = (int)round_up((size_t)k * cin, 32);
    std::vector<float> r((size_t)cout * dst->kp, 0.f);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int tap = 0; tap < k; ++tap) r[(size_t)co * dst->kp + (size_t)tap * cin + ci] = w[((size_t)co * cin + ci) * k + tap];
    F5_TRY(f5_upload_f32(v->arena, r.data(), r.size(), &dst->w));
    if (bias) return f5_upload_f32(v->arena, v->slots[name + ".bias"].host.data(), cout, &dst->b);
    std::vector<float> z(cout, 0.f);
    return f5_upload_f32(v->arena, z.data(), cout, &dst->b);
}
static int bv_upload_snake(f5_bigvgan_s* v, const std::string& name, int ch, BvSnake* dst) {
    const std::vector<float>&al = v->slots[name + ".alpha"].host, &be = v->slots[name + ".beta"].host;
    std::vector<float> a(ch), ib(ch);
    for (int c = 0; c < ch; ++c) {
        const float aa = v->cfg.snake_logscale ? expf(al[c]) : al[c], bb = v->cfg.snake_logscale ? expf(be[c]) : be[c];
        a[c] = aa;
        ib[c] = 1.0f / (bb + 1e-9f);
    }
    F5_TRY(f5_upload_f32(v->arena, a.data(), ch, &dst->a));
    return f5_upload_f32(v->arena, ib.data(), ch, &dst->invb);
}

extern "C" int f5_bigvgan_finalize(f5_bigvgan_t v) {
    if (!v) return f5_fail(F5_EINVAL, "null vocoder");
    if (v->finalized) return 0;
    F5_TRY(f5_check_device());
    F5_TRY(f5_slots_all_set(v->slots));
    const f5_bigvgan_config& c = v->cfg;
    int ch = c.upsample_initial_channel;
    F5_TRY(bv_upload_conv(v, "conv_pre", c.num_mels, ch, 7, true, &v->conv_pre));
    v->ups.resize(c.num_upsamples);
    v->blocks.resize((size_t)c.num_upsamples * c.num_kernels);
    for (int i = 0; i < c.num_upsamples; ++i) {
        BvUp& up = v->ups[i];
        const std::string p = "ups." + std::to_string(i) + ".0";
        up.cin = ch; up.cout = ch / 2; up.k = c.upsample_kernel_sizes[i]; up.u = c.upsample_rates[i]; up.R = up.k / up.u; up.pad = (up.k - up.u) / 2;
        up.kp = (int)round_up((size_t)up.R * up.cin, 32);
        const std::vector<float>& w = v->slots[p + ".weight"].host;  // [cin][cout][k]
        std::vector<float> r((size_t)up.u * up.cout * up.kp, 0.f);
        for (int ph = 0; ph < up.u; ++ph)
            for (int co = 0; co < up.cout; ++co)
                for (int m = 0; m < up.R; ++m)
                    for (int ci = 0; ci < up.cin; ++ci)
                        r[((size_t)ph * up.cout + co) * up.kp + (size_t)m * up.cin + ci] = w[((size_t)ci * up.cout + co) * up.k + ph + m * up.u];
        F5_TRY(f5_upload_f32(v->arena, r.data(), r.size(), &up.w));
        F5_TRY(f5_upload_f32(v->arena, v->slots[p + ".bias"].host.data(), up.cout, &up.b));
        ch /= 2;
        for (int j = 0; j < c.num_kernels; ++j) {
            BvBlock& b = v->blocks[(size_t)i * c.num_kernels + j];
            const std::string r2 = "resblocks." + std::to_string(i * c.num_kernels + j) + ".";
            for (int t = 0; t < 3; ++t) {
                b.dil[t] = c.resblock_dilations[j][t];
                F5_TRY(bv_upload_conv(v, r2 + "convs1." + std::to_string(t), ch, ch, c.resblock_kernel_sizes[j], true, &b.c1[t]));
                F5_TRY(bv_upload_conv(v, r2 + "convs2." + std::to_string(t), ch, ch, c.resblock_kernel_sizes[j], true, &b.c2[t]));
            }
            for (int a = 0; a < 6; ++a) F5_TRY(bv_upload_snake(v, r2 + "activations." + std::to_string(a) + ".act", ch, &b.act[a]));
        }
    }
    F5_TRY(bv_upload_snake(v, "activation_post.act", ch, &v->act_post));
    F5_TRY(bv_upload_conv(v, "conv_post", ch, 1, 7, c.use_bias_at_final != 0, &v->conv_post));
    for (auto& kv : v->slots) std::vector<float>().swap(kv.second.host);
    v->finalized = true;
    return 0;
}

extern "C" int f5_bigvgan_destroy(f5_bigvgan_t v) {
    delete v;
    return 0;
}

static unsigned bv_blocks(size_t n) { return (unsigned)((n + 255) / 256); }

static int bv_conv(f5_bigvgan_s* v, const BvConv& cv, const float* x, int T, int dil, int chan_major, float* out, bool residual, hipStream_t st) {
    const size_t total = (size_t)T * cv.kp;
    hipLaunchKernelGGL(bv_im2col_kernel, dim3(bv_blocks(total)), dim3(256), 0, st, x, T, cv.cin, cv.k, dil, cv.kp, chan_major, v->col, total);
    F5_LAUNCH_CHECK();
    GemmParams g;
    memset(&g, 0, sizeof(g));
    g.A = v->col; g.lda = cv.kp; g.W = cv.w; g.ldw = cv.kp; g.M = T; g.N = cv.cout; g.K = cv.kp; g.bias = cv.b; g.out_f = out; g.ldof = cv.cout;
    g.rows_per_batch = T;
    return launch_gemm(g, F5_PREC_FP32, GEMM_DENSE, residual ? EPI_RESID : EPI_STORE_F32, 0, st);
}
static int bv_snake(f5_bigvgan_s* v, const BvSnake& s, const float* x, int T, int C, float* out, hipStream_t st) {
    BvFilters f;
    memcpy(f.up, v->up_f, sizeof(f.up));
    memcpy(f.dn, v->dn_f, sizeof(f.dn));
    if (C % 64 == 0)
        hipLaunchKernelGGL(bv_aa_snake_kernel<64>, dim3((unsigned)(C / 64), (unsigned)((T + 31) / 32)), dim3(256), 0, st, x, T, C, s.a, s.invb, f, out);
    else
        hipLaunchKernelGGL(bv_aa_snake_kernel<32>, dim3((unsigned)((C + 31) / 32), (unsigned)((T + 31) / 32)), dim3(256), 0, st, x, T, C, s.a, s.invb, f, out);
    F5_LAUNCH_CHECK();
    return 0;
}

// mel [B][num_mels][T] f32 -> wave [B][T * prod(upsample_rates)] f32
extern "C" int f5_bigvgan_forward(f5_bigvgan_t v, int B, int T, const float* mel, float* wave, f5_stream_t stream) {
    if (!v || !mel || !wave || B <= 0 || T <= 0) return f5_fail(F5_EINVAL, "bad argument");
    if (!v->finalized) return f5_fail(F5_ESTATE, "f5_bigvgan_finalize must be called first");
    F5_TRY(f5_check_device());
    hipStream_t st = (hipStream_t)stream;
    const f5_bigvgan_config& c = v->cfg;
    size_t total_up = 1;
    for (int i = 0; i < c.num_upsamples; ++i) total_up *= (size_t)c.upsample_rates[i];
    if ((size_t)T > v->work_T) {  // grow the workspace (once per longest chunk; not a per-call allocation)
        F5_HIP(hipStreamSynchronize(st));
        v->work.release();
        v->work_T = 0;
        size_t act = (size_t)T * c.upsample_initial_channel, col = (size_t)T * v->conv_pre.kp, tmp = 0;
        size_t Ti = T;
        int ch = c.upsample_initial_channel;
        for (int i = 0; i < c.num_upsamples; ++i) {
            const BvUp& up = v->ups[i];
            col = std::max(col, (Ti + 1) * (size_t)up.kp);
            tmp = std::max(tmp, (Ti + 1) * (size_t)up.u * up.cout);
            Ti *= (size_t)up.u;
            ch /= 2;
            act = std::max(act, Ti * (size_t)ch);
            for (int j = 0; j < c.num_kernels; ++j) col = std::max(col, Ti * (size_t)v->blocks[(size_t)i * c.num_kernels + j].c1[0].kp);
        }
        col = std::max(col, Ti * (size_t)v->conv_post.kp);
        F5_TRY(v->work.alloc_t(&v->x, act, false));
        F5_TRY(v->work.alloc_t(&v->y, act, false));
        F5_TRY(v->work.alloc_t(&v->xt, act, false));
        F5_TRY(v->work.alloc_t(&v->xt2, act, false));
        F5_TRY(v->work.alloc_t(&v->xs, act, false));
        F5_TRY(v->work.alloc_t(&v->col, col, false));
        F5_TRY(v->work.alloc_t(&v->tmp, tmp, false));
        v->work_T = (size_t)T;
    }
    for (int b = 0; b < B; ++b) {
        const float* m = mel + (size_t)b * c.num_mels * T;
        int Ti = T, ch = c.upsample_initial_channel;
        F5_TRY(bv_conv(v, v->conv_pre, m, Ti, 1, 1, v->x, false, st));
        for (int i = 0; i < c.num_upsamples; ++i) {
            const BvUp& up = v->ups[i];
            {
                const size_t rows = (size_t)Ti + 1, tg = rows * up.kp;
                hipLaunchKernelGGL(bv_up_gather_kernel, dim3(bv_blocks(tg)), dim3(256), 0, st, v->x, Ti, up.cin, up.R, up.kp, v->col, tg);
                F5_LAUNCH_CHECK();
                GemmParams g;
                memset(&g, 0, sizeof(g));
                g.A = v->col; g.lda = up.kp; g.W = up.w; g.ldw = up.kp; g.M = (int)rows; g.N = up.u * up.cout; g.K = up.kp; g.out_f = v->tmp; g.ldof = up.u * up.cout;
                F5_TRY(launch_gemm(g, F5_PREC_FP32, GEMM_DENSE, EPI_STORE_F32, 0, st));
                const size_t ts = (size_t)Ti * up.u * up.cout;
                hipLaunchKernelGGL(bv_up_scatter_kernel, dim3(bv_blocks(ts)), dim3(256), 0, st, v->tmp, Ti, up.cout, up.u, up.pad, up.b, v->x, ts);
                F5_LAUNCH_CHECK();
            }
            Ti *= up.u;
            ch /= 2;
            const size_t n = (size_t)Ti * ch;
            for (int j = 0; j < c.num_kernels; ++j) {
                const BvBlock& blk = v->blocks[(size_t)i * c.num_kernels + j];
                F5_HIP(hipMemcpyAsync(v->y, v->x, n * sizeof(float), hipMemcpyDeviceToDevice, st));
                for (int t = 0; t < 3; ++t) {
                    F5_TRY(bv_snake(v, blk.act[2 * t], v->y, Ti, ch, v->xt, st));
                    F5_TRY(bv_conv(v, blk.c1[t], v->xt, Ti, blk.dil[t], 0, v->xt2, false, st));
                    F5_TRY(bv_snake(v, blk.act[2 * t + 1], v->xt2, Ti, ch, v->xt, st));
                    F5_TRY(bv_conv(v, blk.c2[t], v->xt, Ti, 1, 0, v->y, true, st));  // y += conv2(...)
                }
                if (j == 0)
                    F5_HIP(hipMemcpyAsync(v->xs, v->y, n * sizeof(float), hipMemcpyDeviceToDevice, st));
                else {
                    hipLaunchKernelGGL(bv_axpby_kernel, dim3(bv_blocks(n)), dim3(256), 0, st, v->xs, v->y, 1.0f, 1.0f, n);
                    F5_LAUNCH_CHECK();
                }
            }
            hipLaunchKernelGGL(bv_axpby_kernel, dim3(bv_blocks(n)), dim3(256), 0, st, v->x, v->xs, 0.0f, 1.0f / (float)c.num_kernels, n);
            F5_LAUNCH_CHECK();
        }
        F5_TRY(bv_snake(v, v->act_post, v->x, Ti, ch, v->xt, st));
        F5_TRY(bv_conv(v, v->conv_post, v->xt, Ti, 1, 0, v->y, false, st));
        hipLaunchKernelGGL(bv_final_kernel, dim3(bv_blocks((size_t)Ti)), dim3(256), 0, st, v->y, c.use_tanh_at_final, wave + (size_t)b * T * total_up, (size_t)Ti);
        F5_LAUNCH_CHECK();
    }
    return 0;
}
