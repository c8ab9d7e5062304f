// vocoder.hip -- Vocos (mel -> 24 kHz wave) behind plug point B of the reference (utils_infer.py:101-124 load_vocoder,
// f5tts_wrapper.py:524 vocoder.decode).  The vocos package is absent from the reference tree; the architecture is
// restated from its published definition (parity "unpinned"): Conv1d(100->512,k7) -> LN -> 8 x ConvNeXt
// (dw-conv k7, LN, Linear 512->1536, GELU, Linear 1536->512, layer-scale gamma, residual) -> LN -> Linear(512->1026) ->
// ISTFT head (exp, clip 1e2, cos/sin, inverse rFFT 1024, Hann window, overlap-add hop 256, centre trim).
// All dense math runs on the fp32-input MFMA (the reference runs the vocoder in fp32; phase feeds cos/sin).
#include <cmath>
#include <cstring>

#include "gemm.h"
#include "kernels.h"
#include "runtime.h"

struct VocosBlockW {
    float *dw_wt = nullptr, *dw_b = nullptr, *ln_w = nullptr, *ln_b = nullptr, *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *gamma = nullptr;
};

struct f5_vocoder_s {
    f5_vocos_config cfg;
    SlotMap slots;
    bool finalized = false;
    DevArena arena, work;
    size_t work_rows = 0;
    std::vector<float> window;  // head.istft.window (periodic Hann by default)
    std::vector<VocosBlockW> blocks;
    float *w_embed = nullptr, *b_embed = nullptr, *norm_w = nullptr, *norm_b = nullptr, *fnorm_w = nullptr, *fnorm_b = nullptr;
    float *w_head = nullptr, *b_head = nullptr, *w_dft = nullptr, *wsq = nullptr, *wscaled = nullptr, *twiddle = nullptr;
    int k_embed = 0, ld_head = 0, k_spec = 0, F = 0;
    // workspace
    float *x0 = nullptr, *xres = nullptr, *hT = nullptr, *h2 = nullptr, *head = nullptr, *spec = nullptr, *frames = nullptr;
};

static void vslot(SlotMap& s, const std::string& n, std::vector<int64_t> shape) { s[n].shape = std::move(shape); }

extern "C" int f5_vocoder_create(const f5_vocos_config* c, f5_vocoder_t* out) {
    if (!c || !out) return f5_fail(F5_EINVAL, "null argument");
    *out = nullptr;
    F5_TRY(f5_check_device());
    if (c->n_mels <= 0 || c->dim <= 0 || c->dim % 32 != 0 || c->dim > 1024 || c->inter_dim % 32 != 0 || c->layers < 0 || c->n_fft <= 0 ||
        c->n_fft % 64 != 0 || c->hop <= 0 || c->n_fft % c->hop != 0)
        return f5_fail(F5_EINVAL, "bad vocos config");
    f5_vocoder_s* v = new f5_vocoder_s();
    v->cfg = *c;
    const int64_t C = c->n_mels, D = c->dim, I = c->inter_dim, F = c->n_fft / 2 + 1;
    SlotMap& s = v->slots;
    vslot(s, "backbone.embed.weight", {D, C, 7});
    vslot(s, "backbone.embed.bias", {D});
    vslot(s, "backbone.norm.weight", {D});
    vslot(s, "backbone.norm.bias", {D});
    for (int i = 0; i < c->layers; ++i) {
        const std::string p = "backbone.convnext." + std::to_string(i) + ".";
        vslot(s, p + "dwconv.weight", {D, 1, 7});
        vslot(s, p + "dwconv.bias", {D});
        vslot(s, p + "norm.weight", {D});
        vslot(s, p + "norm.bias", {D});
        vslot(s, p + "pwconv1.weight", {I, D});
        vslot(s, p + "pwconv1.bias", {I});
        vslot(s, p + "pwconv2.weight", {D, I});
        vslot(s, p + "pwconv2.bias", {D});
        vslot(s, p + "gamma", {D});
    }
    vslot(s, "backbone.final_layer_norm.weight", {D});
    vslot(s, "backbone.final_layer_norm.bias", {D});
    vslot(s, "head.out.weight", {2 * F, D});
    vslot(s, "head.out.bias", {2 * F});
    v->window.resize(c->n_fft);
    for (int n = 0; n < c->n_fft; ++n) v->window[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / c->n_fft));  // torch.hann_window (periodic)
    *out = v;
    return 0;
}

extern "C" int f5_vocoder_has_tensor(f5_vocoder_t v, const char* name, int64_t* numel) {
    if (!v || !name) return 0;
    if (strcmp(name, "head.istft.window") == 0) {
        if (numel) *numel = v->cfg.n_fft;
        return 1;
    }
    auto it = v->slots.find(name);
    if (it == v->slots.end()) return 0;
    if (numel) *numel = it->second.numel();
    return 1;
}

extern "C" int f5_vocoder_set_tensor(f5_vocoder_t v, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!v || !name || !host || !shape) return f5_fail(F5_EINVAL, "null argument");
    if (v->finalized) return f5_fail(F5_ESTATE, "vocoder already finalized");
    if (strcmp(name, "head.istft.window") == 0) {
        if (ndim != 1 || shape[0] != v->cfg.n_fft) return f5_fail(F5_EINVAL, "head.istft.window must have n_fft elements");
        v->window.assign(host, host + v->cfg.n_fft);
        return 0;
    }
    return f5_slot_set(v->slots, name, host, shape, ndim);
}

extern "C" int f5_vocoder_finalize(f5_vocoder_t v) {
    if (!v) return f5_fail(F5_EINVAL, "null vocoder");
    if (v->finalized) return 0;
    F5_TRY(f5_check_device());
    F5_TRY(f5_slots_all_set(v->slots));
    const f5_vocos_config& c = v->cfg;
    const size_t C = c.n_mels, D = c.dim, I = c.inter_dim, NF = c.n_fft, F = NF / 2 + 1;
    v->F = (int)F;
    DevArena& A = v->arena;
    auto Hs = [&](const std::string& n) -> const std::vector<float>& { return v->slots[n].host; };
    // embed conv [D, C, 7] -> im2col weight [D, Kp] with column = tap*C + c
    v->k_embed = (int)round_up(7 * C, 32);
    {
        std::vector<float> w(D * v->k_embed, 0.f);
        const std::vector<float>& e = Hs("backbone.embed.weight");
        for (size_t n = 0; n < D; ++n)
            for (size_t ch = 0; ch < C; ++ch)
                for (int tap = 0; tap < 7; ++tap) w[n * v->k_embed + tap * C + ch] = e[(n * C + ch) * 7 + tap];
        F5_TRY(f5_upload_f32(A, w.data(), w.size(), &v->w_embed));
    }
    F5_TRY(f5_upload_f32(A, Hs("backbone.embed.bias").data(), D, &v->b_embed));
    F5_TRY(f5_upload_f32(A, Hs("backbone.norm.weight").data(), D, &v->norm_w));
    F5_TRY(f5_upload_f32(A, Hs("backbone.norm.bias").data(), D, &v->norm_b));
    F5_TRY(f5_upload_f32(A, Hs("backbone.final_layer_norm.weight").data(), D, &v->fnorm_w));
    F5_TRY(f5_upload_f32(A, Hs("backbone.final_layer_norm.bias").data(), D, &v->fnorm_b));
    v->blocks.resize(c.layers);
    for (int i = 0; i < c.layers; ++i) {
        const std::string p = "backbone.convnext." + std::to_string(i) + ".";
        VocosBlockW& b = v->blocks[i];
        std::vector<float> wt(7 * D);
        const std::vector<float>& dw = Hs(p + "dwconv.weight");
        for (size_t ch = 0; ch < D; ++ch)
            for (int tap = 0; tap < 7; ++tap) wt[(size_t)tap * D + ch] = dw[ch * 7 + tap];
        F5_TRY(f5_upload_f32(A, wt.data(), wt.size(), &b.dw_wt));
        F5_TRY(f5_upload_f32(A, Hs(p + "dwconv.bias").data(), D, &b.dw_b));
        F5_TRY(f5_upload_f32(A, Hs(p + "norm.weight").data(), D, &b.ln_w));
        F5_TRY(f5_upload_f32(A, Hs(p + "norm.bias").data(), D, &b.ln_b));
        F5_TRY(f5_upload_f32(A, Hs(p + "pwconv1.weight").data(), I * D, &b.w1));
        F5_TRY(f5_upload_f32(A, Hs(p + "pwconv1.bias").data(), I, &b.b1));
        F5_TRY(f5_upload_f32(A, Hs(p + "pwconv2.weight").data(), D * I, &b.w2));
        F5_TRY(f5_upload_f32(A, Hs(p + "pwconv2.bias").data(), D, &b.b2));
        F5_TRY(f5_upload_f32(A, Hs(p + "gamma").data(), D, &b.gamma));
    }
    F5_TRY(f5_upload_f32(A, Hs("head.out.weight").data(), 2 * F * D, &v->w_head));
    F5_TRY(f5_upload_f32(A, Hs("head.out.bias").data(), 2 * F, &v->b_head));
    v->ld_head = (int)round_up(2 * F, 8);
    v->k_spec = (int)round_up(2 * F, 32);
    {
        // windowed inverse real DFT as a matrix: frame[n] = w[n]/NF * sum_k c_k (Re_k cos(2 pi k n/NF) - Im_k sin(2 pi k n/NF))
        std::vector<float> w(NF * v->k_spec, 0.f), wsq(NF);
        for (size_t n = 0; n < NF; ++n) {
            const double wn = v->window[n] / (double)NF;
            for (size_t k = 0; k < F; ++k) {
                const double ck = (k == 0 || k == NF / 2) ? 1.0 : 2.0;
                const double ang = 2.0 * M_PI * (double)((k * n) % NF) / (double)NF;
                w[n * v->k_spec + k] = (float)(wn * ck * cos(ang));
                w[n * v->k_spec + F + k] = (float)(-wn * ck * sin(ang));
            }
            wsq[n] = v->window[n] * v->window[n];
        }
        F5_TRY(f5_upload_f32(A, w.data(), w.size(), &v->w_dft));
        F5_TRY(f5_upload_f32(A, wsq.data(), wsq.size(), &v->wsq));
        if (NF == 1024) {  // the FFT form of the head (vocos.hip): window / n_fft and the twiddle table
            std::vector<float> ws(NF), tw(2 * NF);
            for (size_t n = 0; n < NF; ++n) {
                ws[n] = (float)(v->window[n] / (double)NF);
                tw[2 * n] = (float)cos(2.0 * M_PI * (double)n / (double)NF);
                tw[2 * n + 1] = (float)sin(2.0 * M_PI * (double)n / (double)NF);
            }
            F5_TRY(f5_upload_f32(A, ws.data(), ws.size(), &v->wscaled));
            F5_TRY(f5_upload_f32(A, tw.data(), tw.size(), &v->twiddle));
        }
    }
    for (auto& kv : v->slots) {
        kv.second.host.clear();
        kv.second.host.shrink_to_fit();
    }
    F5_HIP(hipDeviceSynchronize());
    v->finalized = true;
    return 0;
}

extern "C" int f5_vocoder_destroy(f5_vocoder_t v) {
    delete v;
    return 0;
}

int g_vocos_fft = 1;  // tuning knob ("vocos_fft"): ISTFT head by FFT (1) or by the dense inverse-DFT GEMM (0: the parity cross-check)

static int ensure_work(f5_vocoder_s* v, size_t rows) {
    if (rows <= v->work_rows) return 0;
    F5_HIP(hipDeviceSynchronize());  // growing the workspace is a (rare) blocking event, never on the steady-state path
    v->work.release();
    v->work_rows = 0;
    const f5_vocos_config& c = v->cfg;
    const size_t rp = (size_t)round_up(rows, 256);
    F5_TRY(v->work.alloc_t(&v->x0, rp * v->k_embed));
    F5_TRY(v->work.alloc_t(&v->xres, rp * c.dim));
    F5_TRY(v->work.alloc_t(&v->hT, rp * c.dim));
    F5_TRY(v->work.alloc_t(&v->h2, rp * c.inter_dim));
    F5_TRY(v->work.alloc_t(&v->head, rp * v->ld_head));
    F5_TRY(v->work.alloc_t(&v->spec, rp * v->k_spec));
    F5_TRY(v->work.alloc_t(&v->frames, rp * c.n_fft));
    v->work_rows = rows;
    return 0;
}

static GemmParams vg() {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    return g;
}

static int istft_from_head(f5_vocoder_s* v, int B, int T, const float* head, int ldh, float* wave, hipStream_t st) {
    const f5_vocos_config& c = v->cfg;
    const int rows = B * T;
    if (v->twiddle && g_vocos_fft) {  // n_fft = 1024: inverse FFT in LDS, one workgroup per frame (HBM-bound)
        F5_TRY(launch_vocos_ifft1024(head, ldh, rows, v->wscaled, v->twiddle, v->frames, st));
        return launch_vocos_ola(v->frames, B, T, c.n_fft, c.hop, v->wsq, wave, st);
    }
    F5_TRY(launch_vocos_spectrum(F5_PREC_FP32, head, ldh, rows, v->F, v->spec, v->k_spec, st));
    GemmParams g = vg();
    g.A = v->spec; g.lda = v->k_spec; g.W = v->w_dft; g.ldw = v->k_spec; g.M = rows; g.N = c.n_fft; g.K = v->k_spec;
    g.out_f = v->frames; g.ldof = c.n_fft;
    F5_TRY(launch_gemm(g, F5_PREC_FP32, GEMM_DENSE, EPI_STORE_F32, 0, st));
    return launch_vocos_ola(v->frames, B, T, c.n_fft, c.hop, v->wsq, wave, st);
}

extern "C" int f5_vocoder_istft_head(f5_vocoder_t v, int B, int T, const float* head_out, float* wave, f5_stream_t stream) {
    if (!v || !head_out || !wave) return f5_fail(F5_EINVAL, "null argument");
    if (!v->finalized) return f5_fail(F5_ESTATE, "vocoder not finalized");
    if (B <= 0 || T < 2) return f5_fail(F5_EINVAL, "need B >= 1 and T >= 2 frames");
    F5_TRY(f5_check_device());
    F5_TRY(ensure_work(v, (size_t)B * T));
    return istft_from_head(v, B, T, head_out, 2 * v->F, wave, (hipStream_t)stream);
}

extern "C" int f5_vocoder_decode(f5_vocoder_t v, int B, int T, const float* mel, float* wave, f5_stream_t stream) {
    if (!v || !mel || !wave) return f5_fail(F5_EINVAL, "null argument");
    if (!v->finalized) return f5_fail(F5_ESTATE, "vocoder not finalized");
    if (B <= 0 || T < 2) return f5_fail(F5_EINVAL, "need B >= 1 and T >= 2 frames");
    F5_TRY(f5_check_device());
    F5_TRY(ensure_work(v, (size_t)B * T));
    hipStream_t st = (hipStream_t)stream;
    const f5_vocos_config& c = v->cfg;
    const int rows = B * T, D = c.dim, I = c.inter_dim, P = F5_PREC_FP32;
    F5_TRY(launch_vocos_im2col(P, mel, B, c.n_mels, T, v->x0, v->k_embed, st));
    GemmParams g = vg();
    g.A = v->x0; g.lda = v->k_embed; g.W = v->w_embed; g.ldw = v->k_embed; g.M = rows; g.N = D; g.K = v->k_embed;
    g.bias = v->b_embed; g.out_f = v->xres; g.ldof = D;
    F5_TRY(launch_gemm(g, P, GEMM_DENSE, EPI_STORE_F32, 0, st));
    F5_TRY(launch_layernorm(P, v->xres, D, rows, D, v->norm_w, v->norm_b, 0, rows, 0, v->xres, D, st));  // row-local: in place is safe
    for (int i = 0; i < c.layers; ++i) {
        const VocosBlockW& b = v->blocks[i];
        F5_TRY(launch_dwconv7_ln(P, v->xres, B, T, D, b.dw_wt, b.dw_b, b.ln_w, b.ln_b, v->hT, D, st));
        g = vg();
        g.A = v->hT; g.lda = D; g.W = b.w1; g.ldw = D; g.M = rows; g.N = I; g.K = D; g.bias = b.b1; g.act = ACT_GELU_ERF;
        g.out_t = v->h2; g.ldo = I;
        F5_TRY(launch_gemm(g, P, GEMM_DENSE, EPI_STORE_T, 0, st));
        g = vg();
        g.A = v->h2; g.lda = I; g.W = b.w2; g.ldw = I; g.M = rows; g.N = D; g.K = I; g.bias = b.b2;
        g.out_f = v->xres; g.ldof = D; g.gate = b.gamma; g.gate_bstride = 0; g.rows_per_batch = T;
        F5_TRY(launch_gemm(g, P, GEMM_DENSE, EPI_RESID, 0, st));
    }
    F5_TRY(launch_layernorm(P, v->xres, D, rows, D, v->fnorm_w, v->fnorm_b, 0, rows, 0, v->hT, D, st));
    g = vg();
    g.A = v->hT; g.lda = D; g.W = v->w_head; g.ldw = D; g.M = rows; g.N = 2 * v->F; g.K = D; g.bias = v->b_head;
    g.out_f = v->head; g.ldof = v->ld_head;
    F5_TRY(launch_gemm(g, P, GEMM_DENSE, EPI_STORE_F32, 0, st));
    return istft_from_head(v, B, T, v->head, v->ld_head, wave, st);
}
