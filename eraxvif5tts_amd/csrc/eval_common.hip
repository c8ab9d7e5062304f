// eval_common.hip -- what every backbone evaluation shares: tuning knobs, GEMM dispatch, stage taps, the AdaLN table (modules.py:311,332), the text
// embedding (dit.py:49-79), the hoisted half of the input embedding (dit.py:88-96), and the public single-evaluation entry points.
#include "model_internal.h"

// ----------------------------------------------------------------------------- helpers
int g_w_prefetch = 16384;  // tuning knob ("w_prefetch"): LayerNorm passes prefetch the following GEMMs' weights when the launch has at most this many token rows
                            // (0 = never).  M = 8192: +2.3 %, M = 2048: +2.9 % mel-frames/s; M = 65536: no effect (each weight line serves 256 token tiles there)
int g_res_f16 = 1;   // tuning knob ("residual_f16"): bf16 production mode keeps the residual stream in fp16 from the first block on (0 = fp32)
int g_ln_defer = 1;  // tuning knob ("ln_defer"): write the residual stream once per DiT block (0 = after every LayerNorm pass)
int g_resid_rmw = 1;  // tuning knob ("resid_rmw"): see dit_eval
int g_ln_fold = 1;    // tuning knob ("ln_fold"): LayerNorm fold (dit_eval); 0 = the two LayerNorm passes per block of round 3
int g_ln_fold_inkernel = 0;  // tuning knob ("ln_fold_inkernel"): 1 = folded projections on tiles narrower than 256 take their row statistics from the
                             // partial sums INSIDE the kernel, no statistics launch in front (and the producers prefetch the weights).  Same bits
                             // (tested), but measured slower where it applies -- 1 x 1024: 79.0 - 80.1 against 76.9 - 77.6 ms per sample(), 2 x 1024:
                             // 107.8 - 108.6 against 105.5 - 105.9 ms (gpurun_out/r4h_*): the 16 partial loads per row sit in front of a
                             // latency-bound main loop and cost more than the 44 small launches they replace.  Off.
int g_ln_fold_fin = 0;  // tuning knob ("ln_fold_fin"): in-place residual GEMMs on the non-persistent schedules finish the row statistics inside the launch
                        // (last workgroup of a token block; gemm.h: fin_counter) -- no statistics launch behind them.  Same bits as stats_finalize_kernel
                        // (tested).  Measured (gpurun_out/r4p_ab.log, same box, alternating): the drain + ticket + last-arriver pass costs each residual
                        // GEMM 3 - 4 us, the two statistics launches it replaces 6.5 us each: 1 x 1024 77.4 - 78.0 against 76.7 - 77.1 ms per sample(),
                        // 2 x 1024 102.6 - 103.3 against 103.8, 4 x 1024 equal.  Off.
int g_gemm_pad_rows = 1;  // tuning knob ("gemm_pad_rows"): the block GEMMs of a DiT evaluation run over the token rows rounded up to 256 (dit_eval)
int g_sync_evals = 0;  // diagnostic knob ("sync_evals"): an eager sample() synchronises the stream after every network evaluation, which bounds the
                       // number of dispatches in flight (profiles/r3_rocprof_pmc_sigsegv.md: rocprofv3 --pmc died under ~5 400 queued dispatches)

// fp16 residual storage for this plan's evaluations (bf16 mode without stage taps; the plan option overrides the process-wide knob)
bool plan_res_f16(const f5_plan_s* p) {
    const bool want = p->res_f16 < 0 ? g_res_f16 != 0 : p->res_f16 != 0;
    if (p->m->cfg.backbone != F5_BACKBONE_DIT || p->m->cfg.long_skip) return false;  // (UNetT, MMDiT and the long-skip DiT keep fp32 streams)
    return want && p->taps.empty() && g_ln_defer && p->m->cfg.precision == F5_PREC_BF16 && p->xres16 && p->base16;
}

GemmParams gp_zero() {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    return g;
}
int run_gemm(f5_plan_s* p, const GemmParams& g, int mode, int epi, hipStream_t st) {
    const int prec = p->m->cfg.precision;
    // 1 = tuned kernel wherever it can run; -1 (auto) = tuned kernel from 512 token rows on (narrower tiles keep the CUs busy at small M)
    int kind = 0;
    if (p->gemm_kernel != 0 && gemm_fast_supported(g, prec, mode, epi) && (p->gemm_kernel == 1 || g.M >= 512)) kind = 1;
    return launch_gemm(g, prec, mode, epi, kind, st);
}
float* tap_dst(f5_plan_s* p, const std::string& name) {
    auto it = p->taps.find(name);
    return it == p->taps.end() ? nullptr : it->second;
}
int tap_f32(f5_plan_s* p, const std::string& name, const float* src, int ld, int rows, int cols, hipStream_t st) {
    float* d = tap_dst(p, name);
    if (!d) return 0;
    return launch_convert_back(F5_PREC_FP32, src, ld, rows, cols, d, cols, st);
}
int tap_t(f5_plan_s* p, const std::string& name, const void* src, int ld, int rows, int cols, hipStream_t st) {
    float* d = tap_dst(p, name);
    if (!d) return 0;
    return launch_convert_back(p->m->cfg.precision, src, ld, rows, cols, d, cols, st);
}

// time values (device, n of them) -> modulation rows [n][modrow] (AdaLN of every block + final), t_emb in p->temb
int compute_modulation(f5_plan_s* p, const float* tvals_dev, int n, hipStream_t st) {
    f5_model_s* m = p->m;
    const int D = m->cfg.dim;
    F5_TRY(launch_time_sinus(tvals_dev, n, p->tsin, st));
    F5_TRY(launch_gemv_rows(p->tsin, 256, n, m->w_t0, m->b_t0, D, 256, 0, 1, p->thid, D, st));  // Linear -> SiLU
    F5_TRY(launch_gemv_rows(p->thid, D, n, m->w_t2, m->b_t2, D, D, 0, 0, p->temb, D, st));      // Linear
    F5_TRY(tap_f32(p, "t_emb", p->temb, D, n, D, st));
    // every AdaLN: Linear(SiLU(t_emb))  (modules.py:311,332); UNetT has none: its layers see the time as a token (unett.py:211-213)
    if (m->modrow > 0) F5_TRY(launch_gemv_rows(p->temb, D, n, m->w_adaln, m->b_adaln, m->modrow, D, 1, 0, p->mod, m->modrow, st));
    return 0;
}

// TextEmbedding.forward (dit.py:49-79) -> out f32 [B*N, td]
int compute_text_embed(f5_plan_s* p, const int32_t* text, int nt, int B, int N, int drop_text, float* out, hipStream_t st) {
    f5_model_s* m = p->m;
    const f5_dit_config& c = m->cfg;
    const int td = c.text_dim, P = c.precision;
    if (c.backbone == F5_BACKBONE_MMDIT) {  // mmdit.py:40-61: [B, nt, dim], not padded to the frame count; table of 1024 positions
        F5_TRY(launch_text_gather(text, nt, B, nt, td, m->text_table, m->text_pos, m->text_pos_rows, drop_text, out, p->filler, st));
        if (c.text_mask_padding) F5_TRY(launch_mask_rows(out, B * nt, td, p->filler, st));
        return 0;
    }
    const int rows = B * N;
    const bool extra = c.conv_layers > 0;
    F5_TRY(launch_text_gather(text, nt, B, N, td, m->text_table, extra ? m->text_pos : nullptr, m->text_pos_rows, drop_text, out, p->filler, st));
    if (!extra) return 0;
    const bool mp = c.text_mask_padding != 0;
    if (mp) F5_TRY(launch_mask_rows(out, rows, td, p->filler, st));
    for (int i = 0; i < c.conv_layers; ++i) {
        const TextBlockW& t = m->tblocks[i];
        F5_TRY(launch_dwconv7_ln(P, out, B, N, td, t.dw_wt, t.dw_b, t.ln_w, t.ln_b, p->teT, td, st));
        GemmParams g = gp_zero();
        g.A = p->teT; g.lda = td; g.W = t.w1; g.ldw = td; g.M = rows; g.N = 2 * td; g.K = td;
        g.bias = t.b1; g.act = ACT_GELU_ERF; g.out_t = p->te_h; g.ldo = 2 * td;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_T, st));
        F5_TRY(launch_grn(P, p->te_h, B, N, 2 * td, t.gamma, t.beta, p->grn_scratch, st));
        g = gp_zero();
        g.A = p->te_h; g.lda = 2 * td; g.W = t.w2; g.ldw = 2 * td; g.M = rows; g.N = td; g.K = 2 * td;
        g.bias = t.b2; g.act = ACT_NONE; g.out_f = out; g.ldof = td; g.rows_per_batch = N;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_RESID, st));
        if (mp) F5_TRY(launch_mask_rows(out, rows, td, p->filler, st));
    }
    return 0;
}

// base[rows, D] = b_in + W_cond . cond + W_text . text_embed for `nb` batch rows starting at row offset row0
int compute_base(f5_plan_s* p, const float* cond, const int32_t* lens, const float* te, int nb, int N, int zero_cond, size_t row0,
                        hipStream_t st) {
    f5_model_s* m = p->m;
    const f5_dit_config& c = m->cfg;
    const int D = c.dim, td = m->in_td, P = c.precision, kct = MELP + m->td_pad;  // (columns td .. td_pad stay zero: the arena zero-fills)
    const size_t es = f5_elem_size(P);
    void* ab = (char*)p->abase + row0 * kct * es;
    F5_TRY(launch_pack_base(P, cond, lens, te, nb, N, c.mel_dim, MELP, td, zero_cond, ab, kct, st));
    GemmParams g = gp_zero();
    g.A = ab; g.lda = kct; g.W = m->w_ct; g.ldw = kct; g.M = nb * N; g.N = D; g.K = kct;
    g.bias = m->b_in; g.out_f = p->base + row0 * D; g.ldof = D;
    F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_F32, st));
    if (p->base16) F5_TRY(launch_f32_to_f16(p->base + row0 * D, (char*)p->base16 + row0 * D * 2, (size_t)nb * N * D, st, plan_res_f16(p) ? p->sat_flag : nullptr));
    return 0;
}


// one network evaluation of whichever backbone the model is (plug point A)
int net_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, int time_row, int per_batch_rows, const uint8_t* mask, hipStream_t st) {
    f5_model_s* m = p->m;
    if (m->cfg.backbone == F5_BACKBONE_UNETT)
        return unett_eval(p, x, xrows, nb, N, p->temb + (size_t)time_row * m->cfg.dim, per_batch_rows ? m->cfg.dim : 0, mask, st);
    if (m->cfg.backbone == F5_BACKBONE_MMDIT)
        return mmdit_eval(p, x, xrows, nb, N, p->mod + (size_t)time_row * m->modrow, per_batch_rows ? m->modrow : 0, mask, st);
    p->fold_eval = per_batch_rows ? -1 : time_row;  // (the fold table holds one set of weights per evaluation TIME of the staged grid)
    return dit_eval(p, x, xrows, nb, N, p->mod + (size_t)time_row * m->modrow, per_batch_rows ? m->modrow : 0, mask, st);
}

int check_plan_shape(f5_plan_s* p, int B, int N) {
    if (!p) return f5_fail(F5_EINVAL, "null plan");
    if (B <= 0 || N <= 0 || B > p->maxB || N > p->maxN || (size_t)B * N > (size_t)p->maxB * p->maxN)
        return f5_fail(F5_EINVAL, "shape (B=%d, N=%d) exceeds the plan (B<=%d, N<=%d)", B, N, p->maxB, p->maxN);
    return f5_check_device();
}

// ----------------------------------------------------------------------------- public: text embed / forward
extern "C" int f5_text_embed(f5_plan_t p, int B, int N, const int32_t* text, int nt, int drop_text, float* out, f5_stream_t stream) {
    F5_TRY(check_plan_shape(p, B, N));
    if (!text || !out || nt <= 0) return f5_fail(F5_EINVAL, "null/empty text");
    return compute_text_embed(p, text, nt, B, N, drop_text, out, (hipStream_t)stream);
}

extern "C" int f5_dit_forward(f5_plan_t p, int B, int N, const float* x, const float* cond, const float* text_embed, const float* time,
                              int drop_audio_cond, const uint8_t* mask, float* out, f5_stream_t stream) {
    F5_TRY(check_plan_shape(p, B, N));
    if (!x || !cond || !text_embed || !time || !out) return f5_fail(F5_EINVAL, "null argument");
    if (p->m->cfg.backbone == F5_BACKBONE_MMDIT) return f5_fail(F5_ENOTSUP, "MMDiT: the text stream has its own length, call f5_mmdit_forward");
    F5_TRY(finish_if_pending(p));
    hipStream_t st = (hipStream_t)stream;
    f5_model_s* m = p->m;
    p->mod_tv.clear();  // p->mod is overwritten with per-sample times
    F5_TRY(compute_modulation(p, time, B, st));
    F5_TRY(compute_base(p, cond, nullptr, text_embed, B, N, drop_audio_cond, 0, st));
    F5_TRY(net_eval(p, x, B * N, B, N, 0, 1, mask, st));
    return launch_convert_back(F5_PREC_FP32, p->vout, MELP, B * N, m->cfg.mel_dim, out, m->cfg.mel_dim, st);
}

extern "C" int f5_mmdit_forward(f5_plan_t p, int B, int N, int nt, const float* x, const float* cond, const float* text_embed, const float* time,
                                int drop_audio_cond, const uint8_t* mask, float* out, f5_stream_t stream) {
    F5_TRY(check_plan_shape(p, B, N));
    if (!x || !cond || !text_embed || !time || !out) return f5_fail(F5_EINVAL, "null argument");
    if (p->m->cfg.backbone != F5_BACKBONE_MMDIT) return f5_fail(F5_ENOTSUP, "f5_mmdit_forward needs an F5_BACKBONE_MMDIT model");
    if (nt <= 0 || nt > p->maxN) return f5_fail(F5_EINVAL, "text length %d outside 1 .. the plan's max_seq %d", nt, p->maxN);
    hipStream_t st = (hipStream_t)stream;
    f5_model_s* m = p->m;
    p->mod_tv.clear();  // p->mod is overwritten with per-sample times
    F5_TRY(compute_modulation(p, time, B, st));
    F5_TRY(compute_base(p, cond, nullptr, nullptr, B, N, drop_audio_cond, 0, st));
    p->c_src[0] = text_embed;
    p->c_src[1] = nullptr;
    p->c_nt = nt;
    p->c_rows_each = B * nt;
    F5_TRY(net_eval(p, x, B * N, B, N, 0, 1, mask, st));
    return launch_convert_back(F5_PREC_FP32, p->vout, MELP, B * N, m->cfg.mel_dim, out, m->cfg.mel_dim, st);
}

