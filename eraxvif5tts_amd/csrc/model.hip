// model.hip -- DiT weights in HBM, the per-(batch, seq) plan/workspace, DiT.forward, CFM.sample and hipGraph replay.
//
// Reference semantics followed (paths under /root/reference/src/f5_tts):
//   model/backbones/dit.py:185-233  DiT.forward          model/cfm.py:82-208   CFM.sample
//   model/modules.py:301-336,610-641 AdaLN / DiTBlock     torchdiffeq fixed-grid euler / midpoint
//
// MI355X-first restructuring (algebraically identical):
//   * time is one scalar per evaluation, so every AdaLN modulation vector of every block is computed ONCE per
//     sample() for all evaluation times (fp32 weights, small-M kernel) and never touches the ODE loop again;
//   * the input projection is split: W_cond.cond + W_text.text_embed + b is constant over the ODE loop and is
//     computed once per CFG branch; per step only W_x.x (K = 100 -> 128) is evaluated;
//   * CFG runs cond and uncond branches as ONE 2B batch through every kernel;
//   * the whole loop (precompute + steps x evaluation) is captured into one hipGraph per shape bucket.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gemm.h"
#include "kernels.h"
#include "runtime.h"

static const int MELP = 128;  // mel channels padded to one MFMA k-block multiple

struct BlockW {
    void *w_qkv = nullptr, *w_o = nullptr, *w_ff1 = nullptr, *w_ff2 = nullptr;
    float *b_qkv = nullptr, *b_o = nullptr, *b_ff1 = nullptr, *b_ff2 = nullptr;
    // UNetT layers (unett.py:139-171): skip projection [D, 2D] of the later half (concat type), RMSNorm gains
    void* w_skip = nullptr;
    float *g_attn = nullptr, *g_ff = nullptr;
    float *w_qn = nullptr, *w_kn = nullptr;  // qk_norm = "rms_norm": RMSNorm(dim_head) weights of q and k (modules.py:394-396)
    // MMDiT blocks (modules.py:646-707): the text stream's own projections; absent (null) in the last, context_pre_only block except w_qkv_c
    void *w_qkv_c = nullptr, *w_o_c = nullptr, *w_ff1_c = nullptr, *w_ff2_c = nullptr;
    float *b_qkv_c = nullptr, *b_o_c = nullptr, *b_ff1_c = nullptr, *b_ff2_c = nullptr;
};
struct TextBlockW {
    float *dw_wt = nullptr, *dw_b = nullptr, *ln_w = nullptr, *ln_b = nullptr, *b1 = nullptr, *gamma = nullptr, *beta = nullptr, *b2 = nullptr;
    void *w1 = nullptr, *w2 = nullptr;
};

// per-evaluation-time weights of one time grid (lnfold.hip: fold_weights_kernel), shared by every plan of the model that samples on this grid
struct FoldTable {
    std::vector<float> tv;  // the evaluation times it was built for
    DevArena arena;
    void* Wt = nullptr;     // [evals][depth][R][D] fp16
    float *c1 = nullptr, *c2 = nullptr;  // [evals][depth][R]
    hipEvent_t ready = nullptr;          // recorded behind the build; a plan on another stream waits for it once
    uint64_t id = 0;
    int users = 0;          // plans holding it (their captured graphs bake its addresses)
    ~FoldTable() {
        if (ready) (void)hipEventDestroy(ready);
    }
};
static const size_t F5_FOLD_TABLES = 2;
static const int F5_FOLD_MAX_EVALS = 64;  // 64 evaluation times x 231 MB (F5TTS_Base) = 14.8 GB; longer grids run the unfolded path

struct f5_model_s {
    f5_dit_config cfg;
    SlotMap slots;
    bool finalized = false;
    DevArena arena;
    int inner = 0, modrow = 0, conv_cg = 0, conv_win = 0, rope_heads = 0;
    std::vector<BlockW> blocks;
    std::vector<TextBlockW> tblocks;
    float *w_adaln = nullptr, *b_adaln = nullptr;  // [depth*6D + 2D, D] fp32: every attn_norm.linear then norm_out.linear
    float *w_t0 = nullptr, *b_t0 = nullptr, *w_t2 = nullptr, *b_t2 = nullptr;
    float *text_table = nullptr, *text_pos = nullptr;
    void *w_x = nullptr, *w_ct = nullptr;
    float* b_in = nullptr;
    void* w_conv[2] = {nullptr, nullptr};
    float* b_conv[2] = {nullptr, nullptr};
    void* w_out = nullptr;
    float* b_out = nullptr;
    float* g_out = nullptr;  // UNetT: norm_out.g
    void* w_lskip = nullptr;  // long_skip_connection.weight [D, 2D] (dit.py:153)
    int td_pad = 0;          // text_dim rounded up to the GEMM's K granule (E2-TTS: text_dim = mel_dim = 100)
    int in_td = 0;           // text columns of the input projection (text_dim; 0 for MMDiT, whose text is a stream of its own)
    int text_pos_rows = 4096;  // rows of the sinusoidal table added to the text embedding (dit.py:41; 1024 mmdit.py:37)
    float inv_freq[32];
    // LayerNorm fold (gemm.h; bf16 DiT without qk_norm): fp32 masters of the two projections behind an AdaLN LayerNorm, [depth][R][D] with
    // R = 3 * inner (fused q|k|v rows) + ff (ff.0.0 rows), their biases [depth][R], and the per-time-grid tables built from them
    float *w_fold = nullptr, *b_fold = nullptr;
    int fold_R = 0;
    std::vector<FoldTable*> folds;  // at most F5_FOLD_TABLES time grids, oldest dropped first (never one a plan still points to)
    uint64_t fold_seq = 0;
    ~f5_model_s();
};

f5_model_s::~f5_model_s() {
    for (FoldTable* t : folds) delete t;
}

extern int g_tuning_epoch;  // bumped by every f5_tuning_set: graphs captured under other knob values are dropped (ops.hip)

struct GraphEntry {
    int B, N, nt, steps, method, cfg_on, mask_on;
    float cfg;
    int epoch;
    uint64_t fold_id = 0;  // the FoldTable whose addresses the capture baked (0 = none)
    std::vector<int> rn;   // ragged sample(): the utterances' frame counts (empty: a uniform batch)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct SampleArgs {
    int B, N, nt, steps, method, cfg_on, mask_on;
    float cfg;
};
// f5_sample_ragged: utterances of different frame counts concatenated along the token axis.  One CFG half holds utterance i at rows
// [off[i], off[i] + n[i]) followed by at least RAGGED_GAP rows that are kept ZERO wherever the position conv reads them, so the conv's own
// zero padding (modules.py:167-190, padding = 15) is what every utterance sees on both sides; T rows per half in all.
static const int RAGGED_GAP = 16;
struct Ragged {
    int T = 0;
    std::vector<int> n, off;
};
// a sample() whose range-guard check was deferred (plan option "residual_guard" = 2): what f5_sample_finish needs to repeat it
struct PendingSample {
    bool valid = false;
    SampleArgs a{};
    int use_graph = 0;
    float* out = nullptr;
    float* trajectory = nullptr;
    hipStream_t stream = nullptr;  // the stream the deferred call was enqueued on
};

struct f5_plan_s {
    f5_model_s* m = nullptr;
    int maxB = 0, maxN = 0, maxE = 0;
    size_t rows_cap = 0;
    DevArena arena;
    void* base16 = nullptr;  // fp16 copy of `base` (bf16 production mode: the input embedding adds it and writes the stream as fp16)
    void* xres16 = nullptr;  // residual stream of the bf16 production mode from the first block on: fp16 storage (see dit_eval)
    float *xres = nullptr, *base = nullptr, *vout = nullptr, *mod = nullptr, *temb = nullptr, *tsin = nullptr, *thid = nullptr;
    float *tvals = nullptr, *coefs = nullptr, *te[2] = {nullptr, nullptr}, *grn_scratch = nullptr, *traj = nullptr, *xmid = nullptr;
    float *cond_in = nullptr, *rope = nullptr, *tap_scratch = nullptr;
    void *yA = nullptr;  // attention-branch output when the residual write is deferred (dit_eval)
    void *hT = nullptr, *cT = nullptr, *yT = nullptr, *qkv = nullptr, *ffh = nullptr, *abase = nullptr, *xin = nullptr, *teT = nullptr, *te_h = nullptr;
    uint8_t *filler = nullptr, *mask = nullptr, *rowbits = nullptr;
    const uint8_t* rowbits_src = nullptr;  // the row mask `rowbits` was built from (GemmParams::rowbits)
    int32_t *text_in = nullptr, *lens_in = nullptr, *dur_in = nullptr;
    int rope_n = 0;
    int gemm_kernel = -1, attn_kernel = -1;  // -1 = auto (tuned kernel when it supports the problem)
    // Range guard of the fp16 residual stream (bf16 production mode): the LayerNorm passes raise `sat_flag` (device word) when an element of
    // the stream reaches fp16's largest finite value or is NaN; f5_sample reads it after the loop (the call's one synchronisation) and
    // repeats the loop with fp32 residual storage, which this plan then keeps (`res_f16` = 0).
    unsigned* sat_flag = nullptr;
    unsigned* sat_base = nullptr;
    unsigned* sat_host = nullptr;  // pinned
    int res_f16 = -1;              // plan option "residual_f16": -1 = the process-wide knob, 0 = fp32 storage, 1 = fp16 storage
    int sat_check = 1;             // plan option "residual_guard": 0 = never read the flag (f5_sample stays fully asynchronous)
    int ragged_graph = 0;          // plan option "ragged_graph": f5_sample_ragged replays a hipGraph captured for this exact list of frame counts
    PendingSample pending;
    // UNetT (unett.py:185-253): the stream carries one time token per utterance in front of the frames
    float* xin_res = nullptr;          // input projection + hoisted embedding, before the time token is prepended [B*N, D]
    float* vout_s = nullptr;           // proj_out over all N + 1 tokens [B*(N+1), MELP]
    void* catT = nullptr;              // cat(x, skip) of the concat skip connection [rows, 2D], activation dtype
    uint8_t* mask1 = nullptr;          // key mask with the leading 1 of the time token
    std::vector<float*> skips;         // depth / 2 saved streams
    // ragged sampler: RoPE table expanded per row of a half (position restarts at every utterance), gap-row flags over both halves
    float* rope_exp = nullptr;
    uint8_t* gapflag = nullptr;
    const Ragged* rg = nullptr;        // set while a ragged sample() runs its evaluations
    // MMDiT (mmdit.py:146-190): the text is a second residual stream of nt tokens per utterance; attention runs over [frames | text]
    float* cres = nullptr;             // text stream [2B * nt, D] f32
    void* qkvJ = nullptr;              // q|k|v of the joint sequence [2B * (N + nt), 3 * inner]
    void* attJ = nullptr;              // attention output over the joint sequence [2B * (N + nt), inner]
    uint8_t* maskJ = nullptr;          // key mask with trailing 1s over the text
    const float* c_src[2] = {nullptr, nullptr};  // text embeddings the stream starts from at every evaluation (cond rows, then uncond rows)
    int c_nt = 0, c_rows_each = 0;
    int fallbacks = 0;             // calls repeated with fp32 storage so far (f5_plan_get_option "residual_fallbacks")
    unsigned sat_amax_bits = 0;    // what the last event saw: largest finite |element| (float bits) and whether a NaN was read
    bool sat_nan = false;
    unsigned sat_pass = 0, sat_blocks = 0, sat_row = 0;  // which passes / DiT blocks raised it, smallest offending token row
    // LayerNorm fold: the time grid's table (model-owned, shared), row statistics (mean, rstd) [rows_cap + 256][2], partial sums
    // [D / 64][rows_cap] float2 of the in-place residual epilogues
    FoldTable* fold = nullptr;
    float *lnf_stats = nullptr, *lnf_partial = nullptr;
    int fold_eval = -1;  // evaluation index of the running net_eval (-1: no table row applies, e.g. f5_dit_forward's per-sample times)
    std::map<std::string, float*> taps;
    std::vector<float> mod_tv;  // evaluation times the AdaLN rows in `mod` were computed for (empty = stale); see f5_sample
    hipStream_t mod_stream = nullptr;  // ... and the stream they were computed on (a call on another stream recomputes them)
    std::vector<GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;  // capture happens on a private stream (the caller's may be the legacy null stream)
    // in-situ timing of the block kernels: HIP event pairs around every launch of an eager sample() (f5_plan_timing_*)
    // The pairs live in a BOUNDED ring (F5_EV_RING pairs, created once per plan and reused): when it is full the older half is folded into
    // site_ms -- the host waits for the last event of that half, while the younger half's launches are still queued, so the device never idles.
    bool timing = false;
    std::vector<hipEvent_t> ev;   // 2 * F5_EV_RING events: pair i = ev[2i], ev[2i+1]
    std::vector<int> ev_site;     // call site of pair i (F5_SITE_*)
    size_t ev_head = 0, ev_live = 0;  // oldest unfolded pair, number of unfolded pairs
    double site_ms[F5_SITE_COUNT] = {0};
    int site_n[F5_SITE_COUNT] = {0};
};
static const size_t F5_EV_RING = 512;

// folds the `count` oldest recorded pairs into the per-site sums (blocks until the last of them has completed)
static void timing_fold(f5_plan_s* p, size_t count) {
    count = std::min(count, p->ev_live);
    if (!count) return;
    (void)hipEventSynchronize(p->ev[2 * ((p->ev_head + count - 1) % F5_EV_RING) + 1]);
    for (size_t k = 0; k < count; ++k) {
        const size_t i = (p->ev_head + k) % F5_EV_RING;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]) == hipSuccess) {
            p->site_ms[p->ev_site[i]] += ms;
            ++p->site_n[p->ev_site[i]];
        }
    }
    p->ev_head = (p->ev_head + count) % F5_EV_RING;
    p->ev_live -= count;
}

// runs `launch` (a kernel launcher returning a status) between an event pair tagged with `site` while timing is on
template <typename F> static int timed(f5_plan_s* p, int site, hipStream_t st, F&& launch) {
    if (!p->timing) return launch();
    if (p->ev_live == F5_EV_RING) timing_fold(p, F5_EV_RING / 2);
    const size_t i = (p->ev_head + p->ev_live) % F5_EV_RING;
    (void)hipEventRecord(p->ev[2 * i], st);
    const int rc = launch();
    (void)hipEventRecord(p->ev[2 * i + 1], st);
    p->ev_site[i] = site;
    ++p->ev_live;
    return rc;
}

static int finish_if_pending(f5_plan_s* p);  // completes a deferred sample() before the plan's buffers are reused (defined with f5_sample)

// ----------------------------------------------------------------------------- model
static void add_slot(SlotMap& s, const std::string& name, std::vector<int64_t> shape) { s[name].shape = std::move(shape); }

extern "C" int f5_model_create(const f5_dit_config* c, f5_model_t* out) {
    if (!c || !out) return f5_fail(F5_EINVAL, "null argument");
    *out = nullptr;
    F5_TRY(f5_check_device());
    if ((c->qk_norm || c->long_skip) && c->backbone != F5_BACKBONE_DIT)
        return f5_fail(F5_ENOTSUP, "qk_norm / long_skip_connection are implemented for the DiT backbone only (null / False in every shipped config)");
    if (c->qk_norm != 0 && c->qk_norm != 1) return f5_fail(F5_EINVAL, "qk_norm: 0 (None) or 1 (\"rms_norm\", modules.py:394)");
    if (c->dim_head != 64) return f5_fail(F5_ENOTSUP, "dim_head=%d: only 64 is implemented", c->dim_head);
    if (c->dim <= 0 || c->dim % 128 != 0 || c->dim > 2048) return f5_fail(F5_EINVAL, "dim=%d must be a multiple of 128 (<= 2048)", c->dim);
    if (c->depth <= 0 || c->heads <= 0 || c->ff_inner <= 0 || c->ff_inner % 32 != 0) return f5_fail(F5_EINVAL, "bad depth/heads/ff_inner");
    if (c->mel_dim <= 0 || c->mel_dim > MELP || c->mel_dim % 4 != 0) return f5_fail(F5_EINVAL, "mel_dim=%d unsupported", c->mel_dim);
    if (c->backbone != F5_BACKBONE_DIT && c->backbone != F5_BACKBONE_UNETT && c->backbone != F5_BACKBONE_MMDIT) return f5_fail(F5_EINVAL, "bad backbone");
    if (c->backbone == F5_BACKBONE_MMDIT && (c->text_dim != c->dim || c->conv_layers != 0))
        return f5_fail(F5_EINVAL, "MMDiT: text_dim must equal dim and conv_layers be 0 (mmdit.py:101: TextEmbedding(dim, ...))");
    if (c->backbone == F5_BACKBONE_UNETT && (c->depth % 2 != 0 || c->skip_connect < F5_SKIP_CONCAT || c->skip_connect > F5_SKIP_NONE))
        return f5_fail(F5_EINVAL, "UNetT: depth must be even (unett.py:120) and skip_connect one of F5_SKIP_*");
    if (c->text_dim <= 0 || c->text_dim % 4 != 0 || c->text_dim > 1024 || (c->conv_layers > 0 && c->text_dim % 32 != 0))
        return f5_fail(F5_EINVAL, "text_dim=%d must be a multiple of 4 (of 32 with ConvNeXt text blocks)", c->text_dim);
    if (c->text_num_embeds <= 0 || c->conv_layers < 0) return f5_fail(F5_EINVAL, "bad text config");
    if (c->precision != F5_PREC_BF16 && c->precision != F5_PREC_FP32) return f5_fail(F5_EINVAL, "bad precision");
    if (c->rope_layout != F5_ROPE_ADJACENT && c->rope_layout != F5_ROPE_HALF_SPLIT) return f5_fail(F5_EINVAL, "bad rope_layout");
    f5_model_s* m = new f5_model_s();
    m->cfg = *c;
    const int64_t D = c->dim, td = c->text_dim, inner = (int64_t)c->heads * 64, ff = c->ff_inner, mel = c->mel_dim;
    m->inner = (int)inner;
    const bool mm = c->backbone == F5_BACKBONE_MMDIT;
    // AdaLN rows per evaluation time.  DiT: 6D per block + 2D.  MMDiT: [x 6D | c 6D] per block, [x 6D | c 2D] in the last, + 2D.
    m->modrow = c->backbone == F5_BACKBONE_UNETT ? 0 : mm ? (int)(c->depth * 12 * D - 4 * D + 2 * D) : (int)(c->depth * 6 * D + 2 * D);
    m->in_td = mm ? 0 : c->text_dim;
    m->td_pad = (int)round_up(m->in_td, 32);
    m->text_pos_rows = mm ? 1024 : 4096;
    m->rope_heads = (mm || c->pe_attn_head <= 0 || c->pe_attn_head > c->heads) ? c->heads : c->pe_attn_head;  // (JointAttnProcessor: every head)
    SlotMap& s = m->slots;
    add_slot(s, "time_embed.time_mlp.0.weight", {D, 256});
    add_slot(s, "time_embed.time_mlp.0.bias", {D});
    add_slot(s, "time_embed.time_mlp.2.weight", {D, D});
    add_slot(s, "time_embed.time_mlp.2.bias", {D});
    add_slot(s, "text_embed.text_embed.weight", {c->text_num_embeds + 1, td});
    for (int i = 0; i < c->conv_layers; ++i) {
        const std::string p = "text_embed.text_blocks." + std::to_string(i) + ".";
        add_slot(s, p + "dwconv.weight", {td, 1, 7});
        add_slot(s, p + "dwconv.bias", {td});
        add_slot(s, p + "norm.weight", {td});
        add_slot(s, p + "norm.bias", {td});
        add_slot(s, p + "pwconv1.weight", {2 * td, td});
        add_slot(s, p + "pwconv1.bias", {2 * td});
        add_slot(s, p + "grn.gamma", {1, 1, 2 * td});
        add_slot(s, p + "grn.beta", {1, 1, 2 * td});
        add_slot(s, p + "pwconv2.weight", {td, 2 * td});
        add_slot(s, p + "pwconv2.bias", {td});
    }
    const std::string in_pre = mm ? "audio_embed." : "input_embed.";  // mmdit.py:69-71: AudioEmbedding.linear(2 mel -> dim) + conv_pos_embed
    add_slot(s, in_pre + (mm ? "linear.weight" : "proj.weight"), {D, 2 * mel + m->in_td});
    add_slot(s, in_pre + (mm ? "linear.bias" : "proj.bias"), {D});
    for (int i = 0; i < 4; i += 2) {
        add_slot(s, in_pre + "conv_pos_embed.conv1d." + std::to_string(i) + ".weight", {D, D / 16, 31});
        add_slot(s, in_pre + "conv_pos_embed.conv1d." + std::to_string(i) + ".bias", {D});
    }
    if (mm) {  // mmdit.py:110-124: MMDiTBlock(context_pre_only = last)
        for (int i = 0; i < c->depth; ++i) {
            const std::string p = "transformer_blocks." + std::to_string(i) + ".";
            const bool last = i == c->depth - 1;
            add_slot(s, p + "attn_norm_c.linear.weight", {(last ? 2 : 6) * D, D});
            add_slot(s, p + "attn_norm_c.linear.bias", {(last ? 2 : 6) * D});
            add_slot(s, p + "attn_norm_x.linear.weight", {6 * D, D});
            add_slot(s, p + "attn_norm_x.linear.bias", {6 * D});
            for (const char* nm : {"to_q", "to_k", "to_v", "to_q_c", "to_k_c", "to_v_c"}) {
                add_slot(s, p + "attn." + nm + ".weight", {inner, D});
                add_slot(s, p + "attn." + nm + ".bias", {inner});
            }
            add_slot(s, p + "attn.to_out.0.weight", {D, inner});
            add_slot(s, p + "attn.to_out.0.bias", {D});
            if (!last) {
                add_slot(s, p + "attn.to_out_c.weight", {D, inner});
                add_slot(s, p + "attn.to_out_c.bias", {D});
            }
            for (const char* sx : {"ff_x.", "ff_c."}) {
                if (last && sx[3] == 'c') continue;
                add_slot(s, p + sx + "ff.0.0.weight", {ff, D});
                add_slot(s, p + sx + "ff.0.0.bias", {ff});
                add_slot(s, p + sx + "ff.2.weight", {D, ff});
                add_slot(s, p + sx + "ff.2.bias", {D});
            }
        }
        add_slot(s, "norm_out.linear.weight", {2 * D, D});
        add_slot(s, "norm_out.linear.bias", {2 * D});
    } else if (c->backbone == F5_BACKBONE_UNETT) {  // unett.py:139-175: layers.<i> = [skip_proj | None, attn_norm, attn, ff_norm, ff]
        for (int i = 0; i < c->depth; ++i) {
            const std::string p = "layers." + std::to_string(i) + ".";
            if (i >= c->depth / 2 && c->skip_connect == F5_SKIP_CONCAT) add_slot(s, p + "0.weight", {D, 2 * D});
            add_slot(s, p + "1.g", {D});
            for (const char* nm : {"to_q", "to_k", "to_v"}) {
                add_slot(s, p + "2." + nm + ".weight", {inner, D});
                add_slot(s, p + "2." + nm + ".bias", {inner});
            }
            add_slot(s, p + "2.to_out.0.weight", {D, inner});
            add_slot(s, p + "2.to_out.0.bias", {D});
            add_slot(s, p + "3.g", {D});
            add_slot(s, p + "4.ff.0.0.weight", {ff, D});
            add_slot(s, p + "4.ff.0.0.bias", {ff});
            add_slot(s, p + "4.ff.2.weight", {D, ff});
            add_slot(s, p + "4.ff.2.bias", {D});
        }
        add_slot(s, "norm_out.g", {D});
    } else {
        for (int i = 0; i < c->depth; ++i) {
            const std::string p = "transformer_blocks." + std::to_string(i) + ".";
            add_slot(s, p + "attn_norm.linear.weight", {6 * D, D});
            add_slot(s, p + "attn_norm.linear.bias", {6 * D});
            for (const char* nm : {"to_q", "to_k", "to_v"}) {
                add_slot(s, p + "attn." + nm + ".weight", {inner, D});
                add_slot(s, p + "attn." + nm + ".bias", {inner});
            }
            add_slot(s, p + "attn.to_out.0.weight", {D, inner});
            add_slot(s, p + "attn.to_out.0.bias", {D});
            add_slot(s, p + "ff.ff.0.0.weight", {ff, D});
            add_slot(s, p + "ff.ff.0.0.bias", {ff});
            add_slot(s, p + "ff.ff.2.weight", {D, ff});
            add_slot(s, p + "ff.ff.2.bias", {D});
            if (c->qk_norm) {
                add_slot(s, p + "attn.q_norm.weight", {64});
                add_slot(s, p + "attn.k_norm.weight", {64});
            }
        }
        if (c->long_skip) add_slot(s, "long_skip_connection.weight", {D, 2 * D});
        add_slot(s, "norm_out.linear.weight", {2 * D, D});
        add_slot(s, "norm_out.linear.bias", {2 * D});
    }
    add_slot(s, "proj_out.weight", {mel, D});
    add_slot(s, "proj_out.bias", {mel});
    // x_transformers RotaryEmbedding(64).inv_freq (persistent buffer; optional in checkpoints)
    for (int j = 0; j < 32; ++j) m->inv_freq[j] = 1.0f / powf(10000.0f, (float)(2 * j) / 64.0f);
    *out = m;
    return 0;
}

extern "C" int f5_model_has_tensor(f5_model_t m, const char* name, int64_t* numel) {
    if (!m || !name) return 0;
    if (strcmp(name, "rotary_embed.inv_freq") == 0) {
        if (numel) *numel = 32;
        return 1;
    }
    auto it = m->slots.find(name);
    if (it == m->slots.end()) return 0;
    if (numel) *numel = it->second.numel();
    return 1;
}

extern "C" int f5_model_set_tensor(f5_model_t m, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!m || !name || !host || !shape) return f5_fail(F5_EINVAL, "null argument");
    if (m->finalized) return f5_fail(F5_ESTATE, "model already finalized");
    if (strcmp(name, "rotary_embed.inv_freq") == 0) {
        if (ndim != 1 || shape[0] != 32) return f5_fail(F5_EINVAL, "rotary_embed.inv_freq must have 32 elements");
        memcpy(m->inv_freq, host, 32 * sizeof(float));
        return 0;
    }
    return f5_slot_set(m->slots, name, host, shape, ndim);
}

static const std::vector<float>& H(f5_model_s* m, const std::string& name) { return m->slots[name].host; }

extern "C" int f5_model_finalize(f5_model_t m) {
    if (!m) return f5_fail(F5_EINVAL, "null model");
    if (m->finalized) return 0;
    F5_TRY(f5_check_device());
    F5_TRY(f5_slots_all_set(m->slots));
    const f5_dit_config& c = m->cfg;
    const int P = c.precision;
    const size_t D = c.dim, td = c.text_dim, inner = m->inner, ff = c.ff_inner, mel = c.mel_dim;
    DevArena& A = m->arena;
    // time MLP (fp32)
    F5_TRY(f5_upload_f32(A, H(m, "time_embed.time_mlp.0.weight").data(), D * 256, &m->w_t0));
    F5_TRY(f5_upload_f32(A, H(m, "time_embed.time_mlp.0.bias").data(), D, &m->b_t0));
    F5_TRY(f5_upload_f32(A, H(m, "time_embed.time_mlp.2.weight").data(), D * D, &m->w_t2));
    F5_TRY(f5_upload_f32(A, H(m, "time_embed.time_mlp.2.bias").data(), D, &m->b_t2));
    // AdaLN linears of every block + the final one, concatenated (fp32)
    const bool mm = c.backbone == F5_BACKBONE_MMDIT;
    if (mm) {
        std::vector<float> w((size_t)m->modrow * D), b(m->modrow);
        size_t row = 0;
        auto put = [&](const std::string& pre, size_t n) {
            memcpy(&w[row * D], H(m, pre + "weight").data(), n * D * sizeof(float));
            memcpy(&b[row], H(m, pre + "bias").data(), n * sizeof(float));
            row += n;
        };
        for (int i = 0; i < c.depth; ++i) {
            const std::string p = "transformer_blocks." + std::to_string(i) + ".";
            put(p + "attn_norm_x.linear.", 6 * D);
            put(p + "attn_norm_c.linear.", (i == c.depth - 1 ? 2 : 6) * D);
        }
        put("norm_out.linear.", 2 * D);
        if (row != (size_t)m->modrow) return f5_fail(F5_ESTATE, "MMDiT AdaLN layout");
        F5_TRY(f5_upload_f32(A, w.data(), w.size(), &m->w_adaln));
        F5_TRY(f5_upload_f32(A, b.data(), b.size(), &m->b_adaln));
    }
    if (c.backbone == F5_BACKBONE_DIT) {
        std::vector<float> w((size_t)m->modrow * D), b(m->modrow);
        for (int i = 0; i < c.depth; ++i) {
            const std::string p = "transformer_blocks." + std::to_string(i) + ".attn_norm.linear.";
            memcpy(&w[(size_t)i * 6 * D * D], H(m, p + "weight").data(), 6 * D * D * sizeof(float));
            memcpy(&b[(size_t)i * 6 * D], H(m, p + "bias").data(), 6 * D * sizeof(float));
        }
        memcpy(&w[(size_t)c.depth * 6 * D * D], H(m, "norm_out.linear.weight").data(), 2 * D * D * sizeof(float));
        memcpy(&b[(size_t)c.depth * 6 * D], H(m, "norm_out.linear.bias").data(), 2 * D * sizeof(float));
        F5_TRY(f5_upload_f32(A, w.data(), w.size(), &m->w_adaln));
        F5_TRY(f5_upload_f32(A, b.data(), b.size(), &m->b_adaln));
    }
    // text embedder
    F5_TRY(f5_upload_f32(A, H(m, "text_embed.text_embed.weight").data(), (size_t)(c.text_num_embeds + 1) * td, &m->text_table));
    if (c.conv_layers > 0 || mm) {
        // precompute_freqs_cis(text_dim, 4096 | 1024): [cos | sin], theta_j = 10000^(-2j/dim)  (modules.py:196-207)
        const int npos = m->text_pos_rows;
        std::vector<float> pos((size_t)npos * td);
        const int half = (int)td / 2;
        for (int j = 0; j < half; ++j) {
            const float inv = 1.0f / powf(10000.0f, (float)(2 * j) / (float)td);
            for (int p = 0; p < npos; ++p) {
                const float ang = (float)p * inv;
                pos[(size_t)p * td + j] = cosf(ang);
                pos[(size_t)p * td + half + j] = sinf(ang);
            }
        }
        F5_TRY(f5_upload_f32(A, pos.data(), pos.size(), &m->text_pos));
    }
    m->tblocks.resize(c.conv_layers);
    for (int i = 0; i < c.conv_layers; ++i) {
        const std::string p = "text_embed.text_blocks." + std::to_string(i) + ".";
        TextBlockW& t = m->tblocks[i];
        std::vector<float> wt(7 * td);
        const std::vector<float>& dw = H(m, p + "dwconv.weight");  // [td, 1, 7] -> tap-major [7][td]
        for (size_t ch = 0; ch < td; ++ch)
            for (int tap = 0; tap < 7; ++tap) wt[(size_t)tap * td + ch] = dw[ch * 7 + tap];
        F5_TRY(f5_upload_f32(A, wt.data(), wt.size(), &t.dw_wt));
        F5_TRY(f5_upload_f32(A, H(m, p + "dwconv.bias").data(), td, &t.dw_b));
        F5_TRY(f5_upload_f32(A, H(m, p + "norm.weight").data(), td, &t.ln_w));
        F5_TRY(f5_upload_f32(A, H(m, p + "norm.bias").data(), td, &t.ln_b));
        F5_TRY(f5_upload_t(A, P, H(m, p + "pwconv1.weight").data(), 2 * td * td, &t.w1));
        F5_TRY(f5_upload_f32(A, H(m, p + "pwconv1.bias").data(), 2 * td, &t.b1));
        F5_TRY(f5_upload_f32(A, H(m, p + "grn.gamma").data(), 2 * td, &t.gamma));
        F5_TRY(f5_upload_f32(A, H(m, p + "grn.beta").data(), 2 * td, &t.beta));
        F5_TRY(f5_upload_t(A, P, H(m, p + "pwconv2.weight").data(), 2 * td * td, &t.w2));
        F5_TRY(f5_upload_f32(A, H(m, p + "pwconv2.bias").data(), td, &t.b2));
    }
    // input projection split: columns [x | cond | text]  (dit.py:88,95 concat order)
    {
        const std::vector<float>& w = H(m, mm ? "audio_embed.linear.weight" : "input_embed.proj.weight");
        const size_t itd = m->in_td, kin = 2 * mel + itd, kct = MELP + (size_t)m->td_pad;  // (text columns zero-padded to the GEMM's K granule)
        std::vector<float> wx(D * MELP, 0.f), wct(D * kct, 0.f);
        for (size_t n = 0; n < D; ++n) {
            for (size_t k = 0; k < mel; ++k) wx[n * MELP + k] = w[n * kin + k];
            for (size_t k = 0; k < mel; ++k) wct[n * kct + k] = w[n * kin + mel + k];
            for (size_t k = 0; k < itd; ++k) wct[n * kct + MELP + k] = w[n * kin + 2 * mel + k];
        }
        F5_TRY(f5_upload_t(A, P, wx.data(), wx.size(), &m->w_x));
        F5_TRY(f5_upload_t(A, P, wct.data(), wct.size(), &m->w_ct));
        F5_TRY(f5_upload_f32(A, H(m, mm ? "audio_embed.linear.bias" : "input_embed.proj.bias").data(), D, &m->b_in));
    }
    // grouped conv (k=31, groups=16) -> tap-major [31][D][win], zero outside each output row's own group
    {
        const int cg = (int)D / 16;
        int win = 0;
        for (int n0 = 0; n0 < (int)D; n0 += 64) {
            const int w0 = (n0 / cg) * cg, w1 = ((n0 + 63) / cg + 1) * cg;
            win = std::max(win, w1 - w0);
        }
        win = (int)round_up(win, 64);
        m->conv_cg = cg;
        m->conv_win = win;
        for (int li = 0; li < 2; ++li) {
            const std::string p = std::string(mm ? "audio_embed." : "input_embed.") + "conv_pos_embed.conv1d." + std::to_string(li * 2) + ".";
            const std::vector<float>& w = H(m, p + "weight");  // [D, cg, 31]
            std::vector<float> r((size_t)31 * D * win, 0.f);
            for (int n = 0; n < (int)D; ++n) {
                const int w0 = ((n / 64 * 64) / cg) * cg, g0 = (n / cg) * cg;
                for (int ci = 0; ci < cg; ++ci) {
                    const int j = g0 + ci - w0;  // position of this input channel inside the tile's window
                    for (int tap = 0; tap < 31; ++tap) r[((size_t)tap * D + n) * win + j] = w[((size_t)n * cg + ci) * 31 + tap];
                }
            }
            F5_TRY(f5_upload_t(A, P, r.data(), r.size(), &m->w_conv[li]));
            F5_TRY(f5_upload_f32(A, H(m, p + "bias").data(), D, &m->b_conv[li]));
        }
    }
    // transformer blocks: fused QKV weight [3*inner, D]
    m->blocks.resize(c.depth);
    const bool un = c.backbone == F5_BACKBONE_UNETT;
    // LayerNorm fold: fp32 masters of the fused q|k|v and the first FF projection of every block, as the kernels see them (RoPE row order)
    const bool fold = c.backbone == F5_BACKBONE_DIT && P == F5_PREC_BF16 && !c.qk_norm && !c.long_skip && D % 128 == 0;
    const size_t foldR = 3 * inner + ff;
    std::vector<float> fold_w(fold ? (size_t)c.depth * foldR * D : 0), fold_b(fold ? (size_t)c.depth * foldR : 0);
    for (int i = 0; i < c.depth; ++i) {
        const std::string p = (un ? "layers." : "transformer_blocks.") + std::to_string(i) + ".";
        const std::string pa = p + (un ? "2." : "attn."), pf = p + (un ? "4." : mm ? "ff_x." : "ff.");
        BlockW& b = m->blocks[i];
        // fused [3 * inner, D] projection of one stream (sfx "" = frames, "_c" = MMDiT's text stream)
        auto fuse_qkv = [&](const std::string& sfx, void** wdst, float** bdst) -> int {
            std::vector<float> w(3 * inner * D), bias(3 * inner);
            const char* nm[3] = {"to_q", "to_k", "to_v"};
            for (int j = 0; j < 3; ++j) {
                memcpy(&w[(size_t)j * inner * D], H(m, pa + nm[j] + sfx + ".weight").data(), inner * D * sizeof(float));
                memcpy(&bias[(size_t)j * inner], H(m, pa + nm[j] + sfx + ".bias").data(), inner * sizeof(float));
            }
            if (c.rope_layout == F5_ROPE_HALF_SPLIT) {
                // The kernels rotate ADJACENT feature pairs.  The half-split form turns (j, j + 32) with frequency j; QK^T is invariant under
                // one permutation of the 64 features of a head applied to q and k alike, so the q/k output rows are re-ordered once here
                // (new 2j <- old j, new 2j+1 <- old j + 32) and the adjacent-pair rotation then IS the half-split rotation.  v is untouched.
                // (MMDiT: the same permutation on both streams' q and k, which meet in one QK^T.)
                for (int part = 0; part < 2; ++part)
                    for (int hd = 0; hd < m->rope_heads; ++hd) {
                        float* wb = &w[((size_t)part * inner + (size_t)hd * 64) * D];
                        float* bb = &bias[(size_t)part * inner + (size_t)hd * 64];
                        std::vector<float> wo(wb, wb + 64 * D), bo(bb, bb + 64);
                        for (int j = 0; j < 32; ++j) {
                            memcpy(wb + (size_t)(2 * j) * D, &wo[(size_t)j * D], D * sizeof(float));
                            memcpy(wb + (size_t)(2 * j + 1) * D, &wo[(size_t)(j + 32) * D], D * sizeof(float));
                            bb[2 * j] = bo[j];
                            bb[2 * j + 1] = bo[j + 32];
                        }
                    }
            }
            if (fold && sfx.empty()) {
                memcpy(&fold_w[(size_t)i * foldR * D], w.data(), w.size() * sizeof(float));
                memcpy(&fold_b[(size_t)i * foldR], bias.data(), bias.size() * sizeof(float));
            }
            F5_TRY(f5_upload_t(A, P, w.data(), w.size(), wdst));
            return f5_upload_f32(A, bias.data(), bias.size(), bdst);
        };
        F5_TRY(fuse_qkv("", &b.w_qkv, &b.b_qkv));
        if (mm) {
            F5_TRY(fuse_qkv("_c", &b.w_qkv_c, &b.b_qkv_c));
            if (i != c.depth - 1) {
                F5_TRY(f5_upload_t(A, P, H(m, pa + "to_out_c.weight").data(), D * inner, &b.w_o_c));
                F5_TRY(f5_upload_f32(A, H(m, pa + "to_out_c.bias").data(), D, &b.b_o_c));
                F5_TRY(f5_upload_t(A, P, H(m, p + "ff_c.ff.0.0.weight").data(), ff * D, &b.w_ff1_c));
                F5_TRY(f5_upload_f32(A, H(m, p + "ff_c.ff.0.0.bias").data(), ff, &b.b_ff1_c));
                F5_TRY(f5_upload_t(A, P, H(m, p + "ff_c.ff.2.weight").data(), D * ff, &b.w_ff2_c));
                F5_TRY(f5_upload_f32(A, H(m, p + "ff_c.ff.2.bias").data(), D, &b.b_ff2_c));
            }
        }
        F5_TRY(f5_upload_t(A, P, H(m, pa + "to_out.0.weight").data(), D * inner, &b.w_o));
        F5_TRY(f5_upload_f32(A, H(m, pa + "to_out.0.bias").data(), D, &b.b_o));
        F5_TRY(f5_upload_t(A, P, H(m, pf + "ff.0.0.weight").data(), ff * D, &b.w_ff1));
        F5_TRY(f5_upload_f32(A, H(m, pf + "ff.0.0.bias").data(), ff, &b.b_ff1));
        if (fold) {
            memcpy(&fold_w[((size_t)i * foldR + 3 * inner) * D], H(m, pf + "ff.0.0.weight").data(), ff * D * sizeof(float));
            memcpy(&fold_b[(size_t)i * foldR + 3 * inner], H(m, pf + "ff.0.0.bias").data(), ff * sizeof(float));
        }
        F5_TRY(f5_upload_t(A, P, H(m, pf + "ff.2.weight").data(), D * ff, &b.w_ff2));
        F5_TRY(f5_upload_f32(A, H(m, pf + "ff.2.bias").data(), D, &b.b_ff2));
        if (c.qk_norm) {  // (half-split rotary layout: the features of the rope heads were re-ordered above, their norm weights follow)
            for (int part = 0; part < 2; ++part) {
                std::vector<float> w = H(m, pa + (part == 0 ? "q_norm.weight" : "k_norm.weight"));
                if (c.rope_layout == F5_ROPE_HALF_SPLIT && m->rope_heads < c.heads)
                    return f5_fail(F5_ENOTSUP, "qk_norm with the half-split rotary layout needs RoPE on every head (one weight vector serves all heads)");
                if (c.rope_layout == F5_ROPE_HALF_SPLIT) {
                    std::vector<float> o = w;
                    for (int j = 0; j < 32; ++j) {
                        w[2 * j] = o[j];
                        w[2 * j + 1] = o[j + 32];
                    }
                }
                F5_TRY(f5_upload_f32(A, w.data(), 64, part == 0 ? &b.w_qn : &b.w_kn));
            }
        }
        if (un) {
            F5_TRY(f5_upload_f32(A, H(m, p + "1.g").data(), D, &b.g_attn));
            F5_TRY(f5_upload_f32(A, H(m, p + "3.g").data(), D, &b.g_ff));
            if (i >= c.depth / 2 && c.skip_connect == F5_SKIP_CONCAT) F5_TRY(f5_upload_t(A, P, H(m, p + "0.weight").data(), D * 2 * D, &b.w_skip));
        }
    }
    if (fold) {
        F5_TRY(f5_upload_f32(A, fold_w.data(), fold_w.size(), &m->w_fold));
        F5_TRY(f5_upload_f32(A, fold_b.data(), fold_b.size(), &m->b_fold));
        m->fold_R = (int)foldR;
        std::vector<float>().swap(fold_w);
    }
    if (un) F5_TRY(f5_upload_f32(A, H(m, "norm_out.g").data(), D, &m->g_out));
    if (c.long_skip) F5_TRY(f5_upload_t(A, P, H(m, "long_skip_connection.weight").data(), D * 2 * D, &m->w_lskip));
    {
        // proj_out rows padded to MELP so the tuned kernel can run it too (rows >= mel are zero)
        std::vector<float> w((size_t)MELP * D, 0.f), b(MELP, 0.f);
        memcpy(w.data(), H(m, "proj_out.weight").data(), mel * D * sizeof(float));
        memcpy(b.data(), H(m, "proj_out.bias").data(), mel * sizeof(float));
        F5_TRY(f5_upload_t(A, P, w.data(), w.size(), &m->w_out));
        F5_TRY(f5_upload_f32(A, b.data(), b.size(), &m->b_out));
    }
    for (auto& kv : m->slots) {  // host copies are no longer needed
        kv.second.host.clear();
        kv.second.host.shrink_to_fit();
    }
    F5_HIP(hipDeviceSynchronize());
    m->finalized = true;
    return 0;
}

extern "C" int f5_model_destroy(f5_model_t m) {
    delete m;
    return 0;
}

// ----------------------------------------------------------------------------- plan
extern "C" int f5_plan_create(f5_model_t m, int max_batch, int max_seq, int max_evals, f5_plan_t* out) {
    if (!m || !out) return f5_fail(F5_EINVAL, "null argument");
    *out = nullptr;
    if (!m->finalized) return f5_fail(F5_ESTATE, "f5_model_finalize must be called before f5_plan_create");
    if (max_batch <= 0 || max_seq <= 0 || max_evals <= 0 || max_seq > 4096) return f5_fail(F5_EINVAL, "bad plan sizes (seq <= 4096: cfm.py:93)");
    F5_TRY(f5_check_device());
    const f5_dit_config& c = m->cfg;
    const size_t es = f5_elem_size(c.precision);
    const size_t D = c.dim, td = c.text_dim, inner = m->inner, ff = c.ff_inner, mel = c.mel_dim;
    f5_plan_s* p = new f5_plan_s();
    p->m = m;
    p->maxB = max_batch;
    p->maxN = max_seq;
    p->maxE = max_evals;
    const bool un = c.backbone == F5_BACKBONE_UNETT;
    const size_t bn = (size_t)max_batch * (max_seq + (un ? 1 : 0));  // (UNetT: one time token per utterance rides in front of the frames)
    const size_t rows = (size_t)round_up(2 * bn, 256);  // CFG-doubled, padded to the tuned GEMM's tile height
    p->rows_cap = rows;
    DevArena& A = p->arena;
    const size_t modrows = std::max<size_t>(max_evals, 2 * (size_t)max_batch);
    int rc = 0;
    do {
        if ((rc = A.alloc_t(&p->xres, rows * D))) break;
        if (c.precision == F5_PREC_BF16) {
            uint16_t* h16 = nullptr;
            if ((rc = A.alloc_t(&h16, rows * D))) break;
            p->xres16 = h16;
            if ((rc = A.alloc_t(&h16, rows * D))) break;
            p->base16 = h16;
        }
        if ((rc = A.alloc_t(&p->base, rows * D))) break;
        if ((rc = A.alloc_t(&p->vout, rows * MELP))) break;
        if ((rc = A.alloc(&p->hT, rows * D * es))) break;
        if ((rc = A.alloc(&p->cT, rows * std::max(D, inner) * es))) break;
        if ((rc = A.alloc(&p->yT, rows * D * es))) break;
        if ((rc = A.alloc(&p->yA, rows * D * es))) break;
        if ((rc = A.alloc(&p->qkv, rows * 3 * inner * es))) break;
        if ((rc = A.alloc(&p->ffh, rows * ff * es))) break;
        if ((rc = A.alloc(&p->abase, rows * (MELP + (size_t)m->td_pad) * es))) break;
        if ((rc = A.alloc(&p->xin, (size_t)round_up(bn, 256) * MELP * es))) break;
        if ((rc = A.alloc_t(&p->mod, modrows * (size_t)std::max(m->modrow, 1)))) break;
        if ((rc = A.alloc_t(&p->temb, modrows * D))) break;
        if ((rc = A.alloc_t(&p->thid, modrows * D))) break;
        if ((rc = A.alloc_t(&p->tsin, modrows * 256))) break;
        if ((rc = A.alloc_t(&p->tvals, modrows))) break;
        if ((rc = A.alloc_t(&p->coefs, modrows))) break;
        if ((rc = A.alloc_t(&p->te[0], bn * td))) break;
        if ((rc = A.alloc_t(&p->te[1], bn * td))) break;
        if ((rc = A.alloc(&p->teT, (size_t)round_up(bn, 256) * td * es))) break;
        if ((rc = A.alloc(&p->te_h, (size_t)round_up(bn, 256) * 2 * td * es))) break;
        if ((rc = A.alloc_t(&p->grn_scratch, (size_t)max_batch * 2 * td + max_batch))) break;
        if ((rc = A.alloc_t(&p->filler, bn))) break;
        if ((rc = A.alloc_t(&p->mask, 2 * bn))) break;
        if ((rc = A.alloc_t(&p->rowbits, (size_t)(2 * bn / 128 + 1) * 16))) break;
        if ((rc = A.alloc_t(&p->traj, (size_t)(max_evals + 1) * bn * mel))) break;
        if ((rc = A.alloc_t(&p->xmid, bn * mel))) break;
        if ((rc = A.alloc_t(&p->cond_in, bn * mel))) break;
        if ((rc = A.alloc_t(&p->text_in, bn))) break;
        if ((rc = A.alloc_t(&p->lens_in, (size_t)max_batch))) break;
        if ((rc = A.alloc_t(&p->dur_in, (size_t)max_batch))) break;
        // RoPE table for positions < max_seq: angle = p * inv_freq_j in fp32, as x_transformers computes it
        const int rope_n = max_seq + (un ? 1 : 0);
        std::vector<float> rope((size_t)rope_n * 64);
        for (int pos = 0; pos < rope_n; ++pos)
            for (int j = 0; j < 32; ++j) {
                const float ang = (float)pos * m->inv_freq[j];
                rope[((size_t)pos * 32 + j) * 2] = cosf(ang);
                rope[((size_t)pos * 32 + j) * 2 + 1] = sinf(ang);
            }
        if ((rc = f5_upload_f32(A, rope.data(), rope.size(), &p->rope))) break;
        p->rope_n = rope_n;
        if (c.backbone == F5_BACKBONE_DIT) {
            if ((rc = A.alloc_t(&p->rope_exp, (rows / 2 + 1) * 64))) break;
            if ((rc = A.alloc_t(&p->gapflag, rows))) break;
        }
        if (c.long_skip) {  // the input embedding kept for the end of the evaluation, and cat(x, residual) in the activation dtype
            p->skips.assign(1, nullptr);
            if ((rc = A.alloc_t(&p->skips[0], rows * D))) break;
            if ((rc = A.alloc(&p->catT, rows * 2 * D * es))) break;
        }
        if (c.backbone == F5_BACKBONE_MMDIT) {
            const size_t rowsJ = (size_t)round_up(2 * bn * 2, 256);  // text length <= max_seq (f5_sample / f5_mmdit_forward check it)
            if ((rc = A.alloc_t(&p->cres, rows * D))) break;
            if ((rc = A.alloc(&p->qkvJ, rowsJ * 3 * inner * es))) break;
            if ((rc = A.alloc(&p->attJ, rowsJ * inner * es))) break;
            if ((rc = A.alloc_t(&p->maskJ, rowsJ))) break;
        }
        if (un) {
            if ((rc = A.alloc_t(&p->xin_res, rows * D))) break;
            if ((rc = A.alloc_t(&p->vout_s, rows * MELP))) break;
            if (c.skip_connect == F5_SKIP_CONCAT && (rc = A.alloc(&p->catT, rows * 2 * D * es))) break;
            if ((rc = A.alloc_t(&p->mask1, 2 * bn))) break;
            p->skips.assign(c.depth / 2, nullptr);
            for (auto& sk : p->skips)
                if ((rc = A.alloc_t(&sk, rows * D))) break;
            if (rc) break;
        }
        if (m->w_fold) {  // LayerNorm fold: row statistics (padded: a wave's LDS-DMA fetches 128 rows at a time) and the epilogues' partial sums
            if ((rc = A.alloc_t(&p->lnf_stats, (rows + 256) * 2))) break;
            if ((rc = A.alloc_t(&p->lnf_partial, (D / 64) * rows * 2))) break;
        }
        if ((rc = A.alloc_t(&p->sat_base, 1024))) break;  // the 8 flag words sit in the middle of a 4 KiB block of their own
        p->sat_flag = p->sat_base + 512;
        if (hipHostMalloc((void**)&p->sat_host, 32, hipHostMallocDefault) != hipSuccess) {
            rc = f5_fail(F5_ENOMEM, "hipHostMalloc failed");
            break;
        }
        *p->sat_host = 0u;
    } while (0);
    if (rc) {
        if (p->sat_host) (void)hipHostFree(p->sat_host);
        delete p;
        return rc;
    }
    if (const char* e = getenv("F5HIP_GEMM_KERNEL")) p->gemm_kernel = atoi(e);
    if (const char* e = getenv("F5HIP_ATTN_KERNEL")) p->attn_kernel = atoi(e);
    *out = p;
    return 0;
}

extern "C" int f5_plan_destroy(f5_plan_t p) {
    if (!p) return 0;
    for (auto& g : p->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
    if (p->fold) --p->fold->users;
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    if (p->sat_host) (void)hipHostFree(p->sat_host);
    delete p;
    return 0;
}
extern "C" int64_t f5_plan_workspace_bytes(f5_plan_t p) { return p ? (int64_t)p->arena.total : 0; }

extern "C" int f5_plan_set_option(f5_plan_t p, const char* key, int value) {
    if (!p || !key) return f5_fail(F5_EINVAL, "null argument");
    bool rebake = false;  // captured graphs baked the previous kernel choice
    if (strcmp(key, "gemm_kernel") == 0) {
        rebake = p->gemm_kernel != value;
        p->gemm_kernel = value;
    } else if (strcmp(key, "attn_kernel") == 0) {
        rebake = p->attn_kernel != value;
        p->attn_kernel = value;
    } else if (strcmp(key, "residual_f16") == 0) {
        const int v = value < 0 ? -1 : (value != 0);
        rebake = p->res_f16 != v;
        p->res_f16 = v;
    } else if (strcmp(key, "ragged_graph") == 0) {
        p->ragged_graph = value != 0;  // (host-side only: which path the next f5_sample_ragged takes)
    } else if (strcmp(key, "residual_guard") == 0) {
        p->sat_check = value < 0 ? 0 : (value > 2 ? 2 : value);  // 0 off, 1 checked inside f5_sample, 2 deferred to f5_sample_finish (host-side only)
    } else {
        return f5_fail(F5_EINVAL, "unknown option '%s'", key);
    }
    if (rebake) {
        for (auto& g : p->graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
        }
        p->graphs.clear();
    }
    return 0;
}

static bool plan_res_f16(const f5_plan_s* p);

extern "C" int f5_plan_get_option(f5_plan_t p, const char* key, int* value) {
    if (!p || !key || !value) return f5_fail(F5_EINVAL, "null argument");
    if (strcmp(key, "gemm_kernel") == 0)
        *value = p->gemm_kernel;
    else if (strcmp(key, "attn_kernel") == 0)
        *value = p->attn_kernel;
    else if (strcmp(key, "residual_f16") == 0)
        *value = plan_res_f16(p) ? 1 : 0;  // what the next evaluation will use
    else if (strcmp(key, "residual_guard") == 0)
        *value = p->sat_check;
    else if (strcmp(key, "ragged_graph") == 0)
        *value = p->ragged_graph;
    else if (strcmp(key, "residual_fallbacks") == 0)
        *value = p->fallbacks;
    else if (strcmp(key, "residual_guard_amax_bits") == 0)  // diagnostic: float bits of the largest finite |element| the last event saw
        *value = (int)p->sat_amax_bits;
    else if (strcmp(key, "residual_guard_nan") == 0)
        *value = p->sat_nan ? 1 : 0;
    else if (strcmp(key, "residual_guard_pass") == 0)
        *value = (int)p->sat_pass;
    else if (strcmp(key, "residual_guard_blocks") == 0)
        *value = (int)p->sat_blocks;
    else if (strcmp(key, "residual_guard_row") == 0)
        *value = (int)p->sat_row;
    else
        return f5_fail(F5_EINVAL, "unknown option '%s'", key);
    return 0;
}

extern "C" int f5_plan_timing_begin(f5_plan_t p, int max_launches) {
    // max_launches: kept for ABI compatibility (round 1-3 sized an event pool with it); the ring is bounded whatever the call count
    if (!p || max_launches <= 0) return f5_fail(F5_EINVAL, "bad argument");
    if (p->ev.empty()) {
        p->ev.assign(2 * F5_EV_RING, nullptr);
        p->ev_site.assign(F5_EV_RING, 0);
        for (auto& e : p->ev) F5_HIP(hipEventCreate(&e));
    }
    p->ev_head = p->ev_live = 0;
    for (int s = 0; s < F5_SITE_COUNT; ++s) {
        p->site_ms[s] = 0.0;
        p->site_n[s] = 0;
    }
    p->timing = true;
    return 0;
}

extern "C" int f5_plan_timing_end(f5_plan_t p, float* avg_ms, int* launches, f5_stream_t stream) {
    if (!p || !avg_ms || !launches) return f5_fail(F5_EINVAL, "null argument");
    p->timing = false;
    F5_HIP(hipStreamSynchronize((hipStream_t)stream));
    timing_fold(p, p->ev_live);
    *avg_ms = p->site_n[F5_SITE_QKV] ? (float)(p->site_ms[F5_SITE_QKV] / p->site_n[F5_SITE_QKV]) : 0.f;
    *launches = p->site_n[F5_SITE_QKV];
    return 0;
}

extern "C" int f5_plan_timing_site(f5_plan_t p, int site, float* avg_ms, int* launches) {
    if (!p || !avg_ms || !launches || site < 0 || site >= F5_SITE_COUNT) return f5_fail(F5_EINVAL, "bad argument");
    *avg_ms = p->site_n[site] ? (float)(p->site_ms[site] / p->site_n[site]) : 0.f;
    *launches = p->site_n[site];
    return 0;
}

extern "C" int f5_plan_set_tap(f5_plan_t p, const char* stage, float* dst) {
    if (!p) return f5_fail(F5_EINVAL, "null plan");
    if (!stage || !dst) {
        p->taps.clear();
        return 0;
    }
    p->taps[stage] = dst;
    return 0;
}

// ----------------------------------------------------------------------------- helpers
int g_w_prefetch = 16384;  // tuning knob ("w_prefetch"): LayerNorm passes prefetch the following GEMMs' weights when the launch has at most this many token rows
                            // (0 = never).  M = 8192: +2.3 %, M = 2048: +2.9 % mel-frames/s; M = 65536: no effect (each weight line serves 256 token tiles there)
int g_res_f16 = 1;   // tuning knob ("residual_f16"): bf16 production mode keeps the residual stream in fp16 from the first block on (0 = fp32)
int g_ln_defer = 1;  // tuning knob ("ln_defer"): write the residual stream once per DiT block (0 = after every LayerNorm pass)
int g_resid_rmw = 1;  // tuning knob ("resid_rmw"): see dit_eval
int g_ln_fold = 1;    // tuning knob ("ln_fold"): LayerNorm fold (dit_eval); 0 = the two LayerNorm passes per block of round 3
int g_sync_evals = 0;  // diagnostic knob ("sync_evals"): an eager sample() synchronises the stream after every network evaluation, which bounds the
                       // number of dispatches in flight (profiles/r3_rocprof_pmc_sigsegv.md: rocprofv3 --pmc died under ~5 400 queued dispatches)

// fp16 residual storage for this plan's evaluations (bf16 mode without stage taps; the plan option overrides the process-wide knob)
static bool plan_res_f16(const f5_plan_s* p) {
    const bool want = p->res_f16 < 0 ? g_res_f16 != 0 : p->res_f16 != 0;
    if (p->m->cfg.backbone != F5_BACKBONE_DIT || p->m->cfg.long_skip) return false;  // (UNetT, MMDiT and the long-skip DiT keep fp32 streams)
    return want && p->taps.empty() && g_ln_defer && p->m->cfg.precision == F5_PREC_BF16 && p->xres16 && p->base16;
}

static GemmParams gp_zero() {
    GemmParams g;
    memset(&g, 0, sizeof(g));
    return g;
}
static int run_gemm(f5_plan_s* p, const GemmParams& g, int mode, int epi, hipStream_t st) {
    const int prec = p->m->cfg.precision;
    // 1 = tuned kernel wherever it can run; -1 (auto) = tuned kernel from 512 token rows on (narrower tiles keep the CUs busy at small M)
    int kind = 0;
    if (p->gemm_kernel != 0 && gemm_fast_supported(g, prec, mode, epi) && (p->gemm_kernel == 1 || g.M >= 512)) kind = 1;
    return launch_gemm(g, prec, mode, epi, kind, st);
}
static float* tap_dst(f5_plan_s* p, const std::string& name) {
    auto it = p->taps.find(name);
    return it == p->taps.end() ? nullptr : it->second;
}
static int tap_f32(f5_plan_s* p, const std::string& name, const float* src, int ld, int rows, int cols, hipStream_t st) {
    float* d = tap_dst(p, name);
    if (!d) return 0;
    return launch_convert_back(F5_PREC_FP32, src, ld, rows, cols, d, cols, st);
}
static int tap_t(f5_plan_s* p, const std::string& name, const void* src, int ld, int rows, int cols, hipStream_t st) {
    float* d = tap_dst(p, name);
    if (!d) return 0;
    return launch_convert_back(p->m->cfg.precision, src, ld, rows, cols, d, cols, st);
}

// time values (device, n of them) -> modulation rows [n][modrow] (AdaLN of every block + final), t_emb in p->temb
static int compute_modulation(f5_plan_s* p, const float* tvals_dev, int n, hipStream_t st) {
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
static int compute_text_embed(f5_plan_s* p, const int32_t* text, int nt, int B, int N, int drop_text, float* out, hipStream_t st) {
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
static int compute_base(f5_plan_s* p, const float* cond, const int32_t* lens, const float* te, int nb, int N, int zero_cond, size_t row0,
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

// one network evaluation over `nb` batch rows (rows = nb*N) whose noisy mel rows are x[xrows, mel] (xrows divides rows);
// modulation row for batch b is modp + b * mod_bstride.  Result: p->vout [rows, MELP] f32.
static int dit_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* modp, int mod_bstride, const uint8_t* mask,
                    hipStream_t st) {
    f5_model_s* m = p->m;
    const f5_dit_config& c = m->cfg;
    const int D = c.dim, P = c.precision, inner = m->inner, ff = c.ff_inner, rows = nb * N;
    // input embedding: h = W_x . x + base ; x_res = h + mish(conv(mish(conv(h))))
    F5_TRY(launch_convert_pad(P, x, c.mel_dim, xrows, c.mel_dim, MELP, p->xin, MELP, st));
    GemmParams g = gp_zero();
    g.A = p->xin; g.lda = MELP; g.W = m->w_x; g.ldw = MELP; g.M = rows; g.N = D; g.K = MELP;
    g.a_row_mod = xrows < rows ? xrows : 0;
    g.addend = p->base; g.ldadd = D; g.out_t = p->hT; g.ldo = D; g.out_f = p->xres; g.ldof = D;
    // Residual stream storage.  fp32 mode, stage taps or ln_defer = 0: fp32 throughout.  bf16 production mode: fp16 from here on (the
    // hoisted part of the input embedding included) -- the reference's own GPU path keeps the whole model, residual stream included, in
    // fp16 (utils_infer.py:184-193); arithmetic stays fp32 and the branches stay bf16.  Bytes per block of the two LayerNorm passes:
    // 1 408 -> 1 024 MiB at C2; of the input embedding 656 -> 400 MiB.
    const bool defer = p->taps.empty() && g_ln_defer && !c.long_skip;  // (long skip: the stream after the input embedding is needed as a value)
    const bool r16 = plan_res_f16(p);
    unsigned* const sat = r16 ? p->sat_flag : nullptr;
    if (r16) {
        g.addend = reinterpret_cast<const float*>(p->base16);
        g.out_f = reinterpret_cast<float*>(p->xres16);
        g.add2_f16 = 1;
    }
    F5_TRY(timed(p, F5_SITE_INPUT, st, [&] { return run_gemm(p, g, GEMM_DENSE, EPI_ADD2, st); }));
    const Ragged* rg = p->rg;  // ragged sample(): N = rows of one half, the utterances sit inside it between zero gaps
    const size_t aes = f5_elem_size(P);
    if (rg) F5_TRY(launch_zero_rows(p->hT, (size_t)D * aes, rows, p->gapflag, st));
    // x_res = h + mish(conv(mish(conv(h)))): the second conv only STORES its branch (activation dtype); every fp32 residual
    // add of the network is fused into the LayerNorm pass that follows it (coalesced streaming RMW, store-only GEMM epilogues)
    for (int li = 0; li < 2; ++li) {
        g = gp_zero();
        g.A = li == 0 ? p->hT : p->cT; g.lda = D; g.W = m->w_conv[li]; g.M = rows; g.N = D; g.K = 31 * m->conv_win;
        g.bias = m->b_conv[li]; g.act = ACT_MISH; g.rows_per_batch = N; g.conv_cg = m->conv_cg; g.conv_win = m->conv_win;
        g.out_t = li == 0 ? p->cT : p->yT; g.ldo = D;
        F5_TRY(timed(p, F5_SITE_CONV, st, [&] { return run_gemm(p, g, GEMM_CONV31, li == 0 ? EPI_STORE_T : EPI_GATE_T, st); }));
        if (rg && li == 0) F5_TRY(launch_zero_rows(p->cT, (size_t)D * aes, rows, p->gapflag, st));
    }

    // In-place residual updates (bf16 production mode, one time per evaluation): the fp16 stream is updated by the epilogues of the attention
    // out-projection and of the second FF linear (EPI_RESID on the fp16 stream) and the LayerNorm passes only read it (block 0 first adds the
    // position-conv branch): 1 024 MiB of stream + branch traffic per block instead of 1 280, and the two passes shrink from 384 + 640 MiB to
    // 256 + 256.  Otherwise (fp32 stream, stage taps, per-sample time rows, knob "resid_rmw" = 0): store-only branches, adds fused into the passes.
    const bool rmw = r16 && mod_bstride == 0 && g_resid_rmw;
    // LayerNorm fold (round 4; gemm.h, lnfold.hip): from the second LayerNorm of block 0 on, the two LayerNorm passes of a block are gone.  The
    // in-place residual epilogues (out-projection, FF2) also write partial row sums of the values they store, stats_finalize_kernel turns them
    // into (mean, rstd) per row -- and carries the fp16 range guard the passes carried -- and the QKV / FF1 projections read the fp16 stream
    // itself against this evaluation time's W' = fp16(W (1 + scale)), applying rstd (acc - mean c1) + c2 in their epilogues.  Block 0's first
    // pass stays (it folds the position-conv branch in), and so does the final AdaLN pass in front of proj_out.  Needs the in-place stream
    // (rmw), the time grid's table (stage_time_grid) and the tuned kernel at all four call sites.
    const FoldTable* ft = p->fold;
    const bool lnf = rmw && g_ln_fold && ft && p->lnf_stats && p->fold_eval >= 0 && p->fold_eval < (int)ft->tv.size() && p->gemm_kernel != 0 &&
                     (p->gemm_kernel == 1 || rows >= 512) && D % 64 == 0 && inner % 64 == 0 && ff % 64 == 0;
    const size_t fR = (size_t)m->fold_R, frow0 = lnf ? ((size_t)p->fold_eval * c.depth) * fR : 0;
    for (int l = 0; l < c.depth; ++l) {
        const BlockW& b = m->blocks[l];
        const float* ml = modp + (size_t)l * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp (modules.py:312)
        const std::string tn = "blk" + std::to_string(l);
        const bool in16 = r16;
        const void* xin = in16 ? p->xres16 : (const void*)p->xres;
        void* xout = r16 ? p->xres16 : (void*)p->xres;
        // x += (conv branch | previous block's gated FF output); n1 = LN(x) * (1 + scale_msa) + shift_msa
        // With no stage tap set, the residual stream is written once per block: this pass normalises x + y without storing it,
        // the second LayerNorm of the block repeats the add (same operands, same order: bit-identical) and stores x + y + y_attn.
        // small batches: every block's weights come from HBM again and the GEMMs are bound by operand latency, so the LayerNorm passes pull
        // the weights of the launches behind them towards the caches (one dword per line): this pass the out-projection and FF1, the
        // second one FF2 and the next block's QKV projection
        const bool wpf = r16 && g_w_prefetch && rows <= g_w_prefetch;
        const size_t wes = f5_elem_size(P);
        PrefetchSet pf1{{b.w_o, b.w_ff1, nullptr, nullptr}, {(unsigned)(D * inner * wes), (unsigned)(ff * D * wes), 0u, 0u}};
        PrefetchSet pf2{{b.w_ff2, l + 1 < c.depth ? m->blocks[l + 1].w_qkv : nullptr, nullptr, nullptr},
                        {(unsigned)(D * ff * wes), (unsigned)(3 * inner * D * wes), 0u, 0u}};
        const bool lnf1 = lnf && l > 0;  // this block's first LayerNorm is folded into its QKV projection (statistics: the previous block's FF2)
        const char* fW = lnf ? (const char*)ft->Wt + (frow0 + (size_t)l * fR) * D * 2 : nullptr;  // this block's W' rows: q|k|v, then ff.0.0
        const float *fc1 = lnf ? ft->c1 + frow0 + (size_t)l * fR : nullptr, *fc2 = lnf ? ft->c2 + frow0 + (size_t)l * fR : nullptr;
        if (!lnf1) F5_TRY(timed(p, F5_SITE_LN1, st, [&] {
            if (rmw)
                return launch_layernorm_res(P, xin, 1, xout, 1, D, rows, D, l == 0 ? p->yT : nullptr, D, nullptr, 1, ml + D, ml, mod_bstride, N, 1, p->hT, D, st,
                                            wpf ? &pf1 : nullptr, sat, 1 | (l << 4));
            return launch_layernorm_res(P, xin, in16, xout, r16, D, rows, D, p->yT, D, nullptr, defer ? 2 : 1, ml + D, ml, mod_bstride, N, 1, p->hT, D, st,
                                        wpf ? &pf1 : nullptr, sat, 1 | (l << 4));
        }));
        if (l == 0) F5_TRY(tap_f32(p, "input_embed", p->xres, D, rows, D, st));
        if (l == 0 && c.long_skip)  // residual = x  (dit.py:217-218)
            F5_HIP(hipMemcpyAsync(p->skips[0], p->xres, (size_t)rows * D * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (l > 0) F5_TRY(tap_f32(p, "blk" + std::to_string(l - 1) + ".out", p->xres, D, rows, D, st));
        F5_TRY(tap_t(p, tn + ".n1", p->hT, D, rows, D, st));
        g = gp_zero();
        g.A = p->hT; g.lda = D; g.W = b.w_qkv; g.ldw = D; g.M = rows; g.N = 3 * inner; g.K = D;
        g.bias = b.b_qkv; g.out_t = p->qkv; g.ldo = 3 * inner; g.rows_per_batch = N; g.site = 1;
        g.rope = rg ? p->rope_exp : p->rope; g.rope_inner = inner; g.rope_heads = m->rope_heads;  // (ragged: row r of a half -> its position in its utterance)
        if (lnf1) {
            g.A = p->xres16; g.W = fW; g.bias = nullptr;
            g.lnf_stats = p->lnf_stats; g.lnf_c1 = fc1; g.lnf_c2 = fc2;
        }
        // Tile quantisation at small batches: the fused projection has 12 feature tiles per token tile; when the q|k part alone (8 tiles
        // per token tile) fills the CUs a whole number of times but q|k|v does not (M = 8192, 4 utterances x 1024 frames x CFG: 256 + 128
        // tiles on 256 CUs, the second round half empty), v is projected by its own launch on 256 x 128 tiles: 70 -> 62 us per block.
        // (round 3, late: any token count whose q|k tiles fit one round while q|k|v would need a second -- ragged batches, odd batch sizes:
        //  M = 6144: 288 tiles of 256 x 256 = two rounds, 60 us; 192 + 192 narrower ones: 53 us.  Same sums either way.)
        const int tiles_m = (rows + 255) / 256, ncu = f5_cu_count();
        const bool split_v = P == F5_PREC_BF16 && p->gemm_kernel != 0 && inner % 256 == 0 && tiles_m * (2 * inner / 256) <= ncu &&
                             tiles_m * (3 * inner / 256) > ncu && tiles_m * (2 * inner / 256) >= 160;
        if (c.qk_norm) {  // q, k stored as projected; RMSNorm per head, then RoPE, in place (modules.py:463-475)
            g.rope = nullptr;
            g.rope_inner = g.rope_heads = 0;
            F5_TRY(timed(p, F5_SITE_QKV, st, [&] {
                const int rc = run_gemm(p, g, GEMM_DENSE, EPI_STORE_T, st);
                return rc ? rc : launch_qknorm_rope(P, p->qkv, 3 * inner, rows, inner, c.heads, m->rope_heads, b.w_qn, b.w_kn, rg ? p->rope_exp : p->rope, N, st);
            }));
        } else if (split_v) {
            GemmParams gv = g;
            g.N = 2 * inner;
            gv.N = inner;
            gv.W = (const char*)b.w_qkv + (size_t)2 * inner * D * f5_elem_size(P);
            gv.bias = b.b_qkv + 2 * inner;
            if (lnf1) {
                gv.W = fW + (size_t)2 * inner * D * 2;
                gv.bias = nullptr;
                gv.lnf_c1 = fc1 + 2 * inner;
                gv.lnf_c2 = fc2 + 2 * inner;
            }
            gv.out_t = (char*)p->qkv + (size_t)2 * inner * f5_elem_size(P);
            gv.rope = nullptr;
            gv.rope_inner = gv.rope_heads = 0;
            F5_TRY(timed(p, F5_SITE_QKV, st, [&] {
                const int rc = run_gemm(p, g, GEMM_DENSE, EPI_ROPE_T, st);
                return rc ? rc : run_gemm(p, gv, GEMM_DENSE, EPI_STORE_T, st);
            }));
        } else {
            F5_TRY(timed(p, F5_SITE_QKV, st, [&] { return run_gemm(p, g, GEMM_DENSE, EPI_ROPE_T, st); }));
        }
        if (rg) {  // every utterance gets the computation of the launch its own batch-1 sample() makes, on its rows of both halves; the ones
                   // that launch would give to the pipelined kernel share launches (grid.z = utterance x branch, 12 utterances per table)
            for (size_t u0 = 0; u0 < rg->n.size(); u0 += 12) {
                AttnSegs sg;
                sg.nbr = nb;
                for (size_t u = u0; u < rg->n.size() && u < u0 + 12; ++u) {
                    sg.off[sg.cnt] = rg->off[u];
                    sg.n[sg.cnt++] = rg->n[u];
                }
                F5_TRY(launch_attention_ragged(P, p->attn_kernel, sg, c.heads, p->qkv, 3 * inner, p->cT, inner, st, N));
            }
        } else {
            int kind = 0;
            if (p->attn_kernel != 0 && attention_fast_supported(P, N, c.heads)) kind = 1;
            F5_TRY(timed(p, F5_SITE_ATTN, st, [&] { return launch_attention(P, kind, nb, N, c.heads, p->qkv, 3 * inner, mask, p->cT, inner, st); }));
        }
        if (float* d = tap_dst(p, tn + ".attn")) {  // Attention module output before gating (extra GEMM, debug only)
            g = gp_zero();
            g.A = p->cT; g.lda = inner; g.W = b.w_o; g.ldw = inner; g.M = rows; g.N = D; g.K = inner;
            g.bias = b.b_o; g.out_f = d; g.ldof = D;
            F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_F32, st));
        }
        // y = gate_msa * to_out(attn), 0 on padded query rows (modules.py:499-501, 635)
        g = gp_zero();
        g.A = p->cT; g.lda = inner; g.W = b.w_o; g.ldw = inner; g.M = rows; g.N = D; g.K = inner;
        g.bias = b.b_o; g.out_t = defer ? p->yA : p->yT; g.ldo = D; g.gate = ml + 2 * D; g.gate_bstride = mod_bstride; g.rows_per_batch = N;
        g.rowmask = mask; g.site = 2;
        g.rowbits = (mask && mask == p->rowbits_src) ? p->rowbits : nullptr;
        if (rmw) {
            g.out_t = nullptr;
            g.out_f = reinterpret_cast<float*>(p->xres16);
            g.ldof = D;
            g.add2_f16 = 1;
        }
        if (lnf) {  // partial row sums of the updated stream; pivot = the row's previous mean (none yet in block 0: the table is this evaluation's)
            g.stats_out = p->lnf_partial; g.stats_ld = (int)p->rows_cap; g.stats_pivot = l > 0 ? p->lnf_stats : nullptr;
        }
        F5_TRY(timed(p, F5_SITE_OUT, st, [&] { return run_gemm(p, g, GEMM_DENSE, rmw ? EPI_RESID : EPI_GATE_T, st); }));
        // x += y; n2 = LN(x) * (1 + scale_mlp) + shift_mlp
        if (lnf) F5_TRY(timed(p, F5_SITE_LN2, st, [&] {
            return launch_stats_finalize(p->lnf_partial, (int)p->rows_cap, D / 64, rows, D, l > 0 ? p->lnf_stats : nullptr, p->lnf_stats, sat, 2 | (l << 4), st);
        }));
        if (!lnf) F5_TRY(timed(p, F5_SITE_LN2, st, [&] {
            if (rmw)
                return launch_layernorm_res(P, xin, 1, xout, 1, D, rows, D, nullptr, D, nullptr, 1, ml + 4 * D, ml + 3 * D, mod_bstride, N, 1, p->hT, D, st,
                                            wpf ? &pf2 : nullptr, sat, 2 | (l << 4));
            return launch_layernorm_res(P, xin, in16, xout, r16, D, rows, D, p->yT, D, defer ? p->yA : nullptr, defer ? 3 : 1, ml + 4 * D, ml + 3 * D,
                                        mod_bstride, N, 1, p->hT, D, st, wpf ? &pf2 : nullptr, sat, 2 | (l << 4));
        }));
        g = gp_zero();
        g.A = p->hT; g.lda = D; g.W = b.w_ff1; g.ldw = D; g.M = rows; g.N = ff; g.K = D;
        g.bias = b.b_ff1; g.act = ACT_GELU_TANH; g.out_t = p->ffh; g.ldo = ff; g.site = 3;
        if (lnf) {
            g.A = p->xres16; g.W = fW + (size_t)3 * inner * D * 2; g.bias = nullptr;
            g.lnf_stats = p->lnf_stats; g.lnf_c1 = fc1 + 3 * inner; g.lnf_c2 = fc2 + 3 * inner;
        }
        F5_TRY(timed(p, F5_SITE_FF1, st, [&] { return run_gemm(p, g, GEMM_DENSE, EPI_STORE_T, st); }));
        // y = gate_mlp * ff(n2)  (modules.py:639)
        g = gp_zero();
        g.A = p->ffh; g.lda = ff; g.W = b.w_ff2; g.ldw = ff; g.M = rows; g.N = D; g.K = ff;
        g.bias = b.b_ff2; g.out_t = p->yT; g.ldo = D; g.gate = ml + 5 * D; g.gate_bstride = mod_bstride; g.rows_per_batch = N; g.site = 4;
        if (rmw) {
            g.out_t = nullptr;
            g.out_f = reinterpret_cast<float*>(p->xres16);
            g.ldof = D;
            g.add2_f16 = 1;
        }
        const bool lnf_next = lnf && l + 1 < c.depth;  // (the final AdaLN pass reads the stream itself)
        if (lnf_next) {
            g.stats_out = p->lnf_partial; g.stats_ld = (int)p->rows_cap; g.stats_pivot = p->lnf_stats;
        }
        F5_TRY(timed(p, F5_SITE_FF2, st, [&] { return run_gemm(p, g, GEMM_DENSE, rmw ? EPI_RESID : EPI_GATE_T, st); }));
        if (lnf_next) F5_TRY(timed(p, F5_SITE_LN1, st, [&] {
            return launch_stats_finalize(p->lnf_partial, (int)p->rows_cap, D / 64, rows, D, p->lnf_stats, p->lnf_stats, sat, 1 | ((l + 1) << 4), st);
        }));
    }
    const float* mf = modp + (size_t)c.depth * 6 * D;  // final AdaLN: (scale, shift) (modules.py:333)
    // (no stage tap: the stream itself is not needed any more, so the last add is not written back)
    if (rmw)  // (the stream already holds every branch)
        F5_TRY(launch_layernorm_res(P, p->xres16, 1, p->xres16, 1, D, rows, D, nullptr, D, nullptr, 1, mf, mf + D, mod_bstride, N, 1, p->hT, D, st, nullptr, sat, 3));
    else
        F5_TRY(launch_layernorm_res(P, r16 ? p->xres16 : (const void*)p->xres, r16, r16 ? p->xres16 : (void*)p->xres, r16, D, rows, D, p->yT, D, nullptr,
                                    defer ? 2 : 1, mf, mf + D, mod_bstride, N, 1, p->hT, D, st, nullptr, sat, 3));
    F5_TRY(tap_f32(p, "blk" + std::to_string(c.depth - 1) + ".out", p->xres, D, rows, D, st));
    if (c.long_skip) {  // x = long_skip_connection(cat(x, residual))  (dit.py:227-228), then the final AdaLN on it
        const size_t es = f5_elem_size(P);
        F5_TRY(launch_convert_pad(P, p->xres, D, rows, D, D, p->catT, 2 * D, st));
        F5_TRY(launch_convert_pad(P, p->skips[0], D, rows, D, D, (char*)p->catT + (size_t)D * es, 2 * D, st));
        g = gp_zero();
        g.A = p->catT; g.lda = 2 * D; g.W = m->w_lskip; g.ldw = 2 * D; g.M = rows; g.N = D; g.K = 2 * D;
        g.out_f = p->xres; g.ldof = D;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_F32, st));
        F5_TRY(launch_layernorm(P, p->xres, D, rows, D, mf, mf + D, mod_bstride, N, 1, p->hT, D, st));
    }
    F5_TRY(tap_t(p, "final_norm", p->hT, D, rows, D, st));
    g = gp_zero();
    g.A = p->hT; g.lda = D; g.W = m->w_out; g.ldw = D; g.M = rows; g.N = MELP; g.K = D;
    g.bias = m->b_out; g.out_f = p->vout; g.ldof = MELP;
    return run_gemm(p, g, GEMM_DENSE, EPI_STORE_F32, st);
}

// one evaluation of the UNetT backbone (reference model/backbones/unett.py:185-253) over `nb` batch rows; temb = time embedding of batch row b at
// temb + b * temb_bstride (stride 0: one time for all).  Result: p->vout [nb * N, MELP] f32 (the time token's row dropped, :246).
// The stream is fp32 (`xres`, read-modify-write by the fp32 EPI_RESID epilogues); activations in the precision's dtype.
static int unett_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* temb, int temb_bstride, const uint8_t* mask, hipStream_t st) {
    f5_model_s* m = p->m;
    const f5_dit_config& c = m->cfg;
    const int D = c.dim, P = c.precision, inner = m->inner, ff = c.ff_inner, S = N + 1, rows_in = nb * N, rows = nb * S;
    const size_t es = f5_elem_size(P);
    // InputEmbedding (unett.py:88-98): h = proj(cat(x, cond, text)); x = conv_pos_embed(h) + h
    F5_TRY(launch_convert_pad(P, x, c.mel_dim, xrows, c.mel_dim, MELP, p->xin, MELP, st));
    GemmParams g = gp_zero();
    g.A = p->xin; g.lda = MELP; g.W = m->w_x; g.ldw = MELP; g.M = rows_in; g.N = D; g.K = MELP;
    g.a_row_mod = xrows < rows_in ? xrows : 0;
    g.addend = p->base; g.ldadd = D; g.out_t = p->hT; g.ldo = D; g.out_f = p->xin_res; g.ldof = D;
    F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_ADD2, st));
    for (int li = 0; li < 2; ++li) {
        g = gp_zero();
        g.A = li == 0 ? p->hT : p->cT; g.lda = D; g.W = m->w_conv[li]; g.M = rows_in; g.N = D; g.K = 31 * m->conv_win;
        g.bias = m->b_conv[li]; g.act = ACT_MISH; g.rows_per_batch = N; g.conv_cg = m->conv_cg; g.conv_win = m->conv_win;
        g.out_t = li == 0 ? p->cT : p->yT; g.ldo = D;
        F5_TRY(run_gemm(p, g, GEMM_CONV31, li == 0 ? EPI_STORE_T : EPI_GATE_T, st));
    }
    // x = cat([t, x], dim=1); mask = pad(mask, (1, 0), 1)  (:211-214)
    F5_TRY(launch_pack_time_token(P, p->xin_res, p->yT, temb, temb_bstride, nb, N, D, p->xres, st));
    const uint8_t* mask1 = nullptr;
    if (mask) {
        F5_TRY(launch_pad_mask(mask, nb, N, p->mask1, st));
        mask1 = p->mask1;
    }
    const int half = c.depth / 2;
    for (int l = 0; l < c.depth; ++l) {
        const BlockW& b = m->blocks[l];
        if (l < half) {  // skips.append(x)  (:229-230)
            F5_HIP(hipMemcpyAsync(p->skips[l], p->xres, (size_t)rows * D * sizeof(float), hipMemcpyDeviceToDevice, st));
        } else {         // skip = skips.pop()  (:232-238)
            const float* skip = p->skips[c.depth - 1 - l];
            if (c.skip_connect == F5_SKIP_CONCAT) {
                F5_TRY(launch_convert_pad(P, p->xres, D, rows, D, D, p->catT, 2 * D, st));
                F5_TRY(launch_convert_pad(P, skip, D, rows, D, D, (char*)p->catT + (size_t)D * es, 2 * D, st));
                g = gp_zero();
                g.A = p->catT; g.lda = 2 * D; g.W = b.w_skip; g.ldw = 2 * D; g.M = rows; g.N = D; g.K = 2 * D;
                g.out_f = p->xres; g.ldof = D;
                F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_F32, st));
            } else if (c.skip_connect == F5_SKIP_ADD) {
                F5_TRY(launch_add_f32(p->xres, skip, (size_t)rows * D, st));
            }
        }
        // x = attn(attn_norm(x), rope, mask) + x  (:241)
        F5_TRY(launch_rmsnorm(P, p->xres, D, rows, D, b.g_attn, p->hT, D, st));
        g = gp_zero();
        g.A = p->hT; g.lda = D; g.W = b.w_qkv; g.ldw = D; g.M = rows; g.N = 3 * inner; g.K = D;
        g.bias = b.b_qkv; g.out_t = p->qkv; g.ldo = 3 * inner; g.rows_per_batch = S;
        g.rope = p->rope; g.rope_inner = inner; g.rope_heads = m->rope_heads;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_ROPE_T, st));
        {
            int kind = 0;
            if (p->attn_kernel != 0 && attention_fast_supported(P, S, c.heads)) kind = 1;
            F5_TRY(launch_attention(P, kind, nb, S, c.heads, p->qkv, 3 * inner, mask1, p->cT, inner, st));
        }
        g = gp_zero();
        g.A = p->cT; g.lda = inner; g.W = b.w_o; g.ldw = inner; g.M = rows; g.N = D; g.K = inner;
        g.bias = b.b_o; g.out_f = p->xres; g.ldof = D; g.rows_per_batch = S; g.rowmask = mask1;  // masked query rows: attention output is 0 (modules.py:499-501)
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_RESID, st));
        // x = ff(ff_norm(x)) + x  (:242)
        F5_TRY(launch_rmsnorm(P, p->xres, D, rows, D, b.g_ff, p->hT, D, st));
        g = gp_zero();
        g.A = p->hT; g.lda = D; g.W = b.w_ff1; g.ldw = D; g.M = rows; g.N = ff; g.K = D;
        g.bias = b.b_ff1; g.act = ACT_GELU_TANH; g.out_t = p->ffh; g.ldo = ff;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_T, st));
        g = gp_zero();
        g.A = p->ffh; g.lda = ff; g.W = b.w_ff2; g.ldw = ff; g.M = rows; g.N = D; g.K = ff;
        g.bias = b.b_ff2; g.out_f = p->xres; g.ldof = D; g.rows_per_batch = S;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_RESID, st));
    }
    // x = norm_out(x)[:, 1:, :]; proj_out  (:246-248)
    F5_TRY(launch_rmsnorm(P, p->xres, D, rows, D, m->g_out, p->hT, D, st));
    g = gp_zero();
    g.A = p->hT; g.lda = D; g.W = m->w_out; g.ldw = D; g.M = rows; g.N = MELP; g.K = D;
    g.bias = m->b_out; g.out_f = p->vout_s; g.ldof = MELP;
    F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_F32, st));
    return launch_drop_time_token(p->vout_s, nb, N, MELP, p->vout, st);
}

// one evaluation of the MMDiT backbone (reference model/backbones/mmdit.py:146-190, MMDiTBlock modules.py:646-707, JointAttnProcessor :509-606) over
// `nb` batch rows.  Two fp32 residual streams: frames `xres` [nb * N, D] and text `cres` [nb * nt, D] (restarted from p->c_src at every
// evaluation: unlike DiT's text embedding the text stream passes through the time-conditioned blocks).  Each block projects both streams with
// their own weights (RoPE per stream, positions from 0), gathers q|k|v into the joint [frames | text] sequence of every utterance, runs one
// attention over it, scatters the result back and applies the gated out-projection / FF updates to each stream in place (EPI_RESID).
static int mmdit_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* modp, int mod_bstride, const uint8_t* mask, hipStream_t st) {
    f5_model_s* m = p->m;
    const f5_dit_config& c = m->cfg;
    const int D = c.dim, P = c.precision, inner = m->inner, ff = c.ff_inner, nt = p->c_nt, S = N + nt, rows = nb * N, rows_c = nb * nt;
    const size_t es = f5_elem_size(P);
    if (nt <= 0 || !p->c_src[0]) return f5_fail(F5_ESTATE, "MMDiT: no text stream staged");
    // c = text_embed(text)  (:163-173)
    for (int h = 0; h < 2; ++h) {
        const int r0 = h * p->c_rows_each, nr = std::min(rows_c - r0, p->c_rows_each);
        if (nr <= 0) break;
        if (!p->c_src[h]) return f5_fail(F5_ESTATE, "MMDiT: text stream of the second branch missing");
        F5_HIP(hipMemcpyAsync(p->cres + (size_t)r0 * D, p->c_src[h], (size_t)nr * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    // AudioEmbedding (:69-79): h = Linear(cat(x, cond)); x = conv_pos_embed(h) + h   (the cond half of the linear is hoisted into `base`)
    F5_TRY(launch_convert_pad(P, x, c.mel_dim, xrows, c.mel_dim, MELP, p->xin, MELP, st));
    GemmParams g = gp_zero();
    g.A = p->xin; g.lda = MELP; g.W = m->w_x; g.ldw = MELP; g.M = rows; g.N = D; g.K = MELP;
    g.a_row_mod = xrows < rows ? xrows : 0;
    g.addend = p->base; g.ldadd = D; g.out_t = p->hT; g.ldo = D; g.out_f = p->xres; g.ldof = D;
    F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_ADD2, st));
    for (int li = 0; li < 2; ++li) {
        g = gp_zero();
        g.A = li == 0 ? p->hT : p->cT; g.lda = D; g.W = m->w_conv[li]; g.M = rows; g.N = D; g.K = 31 * m->conv_win;
        g.bias = m->b_conv[li]; g.act = ACT_MISH; g.rows_per_batch = N; g.conv_cg = m->conv_cg; g.conv_win = m->conv_win;
        g.out_t = li == 0 ? p->cT : p->yT; g.ldo = D;
        F5_TRY(run_gemm(p, g, GEMM_CONV31, li == 0 ? EPI_STORE_T : EPI_GATE_T, st));
    }
    const uint8_t* maskJ = nullptr;
    if (mask) {  // modules.py:573: no mask over the text keys
        F5_TRY(launch_joint_mask(mask, nb, N, nt, p->maskJ, st));
        maskJ = p->maskJ;
    }
    const size_t qrow = (size_t)3 * inner * es, arow = (size_t)inner * es;
    // one stream's linear that updates it in place: stream += gate * (A . W^T + b), padded query rows of the frames untouched (:596-599)
    auto resid = [&](const void* A, int lda, const void* W, const float* bias, int K, float* stream, int M, int rpb, const float* gate, const uint8_t* rm) {
        GemmParams q = gp_zero();
        q.A = A; q.lda = lda; q.W = W; q.ldw = K; q.M = M; q.N = D; q.K = K;
        q.bias = bias; q.out_f = stream; q.ldof = D; q.gate = gate; q.gate_bstride = mod_bstride; q.rows_per_batch = rpb; q.rowmask = rm;
        return run_gemm(p, q, GEMM_DENSE, EPI_RESID, st);
    };
    for (int l = 0; l < c.depth; ++l) {
        const BlockW& b = m->blocks[l];
        const bool last = l == c.depth - 1;
        const float* mx = modp + (size_t)l * 12 * D;  // attn_norm_x: shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp (modules.py:312)
        const float* mc = mx + (size_t)6 * D;         // attn_norm_c: the same six, or (scale, shift) of AdaLayerNorm_Final in the last block (:333)
        // frames: norm_x -> q|k|v with RoPE over positions 0 .. N-1
        if (l == 0)  // (x += position-conv branch, written back)
            F5_TRY(launch_layernorm_add(P, p->xres, D, rows, D, p->yT, D, mx + D, mx, mod_bstride, N, 1, p->hT, D, st));
        else
            F5_TRY(launch_layernorm(P, p->xres, D, rows, D, mx + D, mx, mod_bstride, N, 1, p->hT, D, st));
        g = gp_zero();
        g.A = p->hT; g.lda = D; g.W = b.w_qkv; g.ldw = D; g.M = rows; g.N = 3 * inner; g.K = D;
        g.bias = b.b_qkv; g.out_t = p->qkv; g.ldo = 3 * inner; g.rows_per_batch = N;
        g.rope = p->rope; g.rope_inner = inner; g.rope_heads = m->rope_heads;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_ROPE_T, st));
        F5_TRY(launch_copy_segments(p->qkv, (size_t)N * qrow, p->qkvJ, (size_t)S * qrow, (size_t)N * qrow, nb, st));
        // text: norm_c -> q|k|v with RoPE over positions 0 .. nt-1
        if (last)
            F5_TRY(launch_layernorm(P, p->cres, D, rows_c, D, mc, mc + D, mod_bstride, nt, 1, p->hT, D, st));
        else
            F5_TRY(launch_layernorm(P, p->cres, D, rows_c, D, mc + D, mc, mod_bstride, nt, 1, p->hT, D, st));
        g = gp_zero();
        g.A = p->hT; g.lda = D; g.W = b.w_qkv_c; g.ldw = D; g.M = rows_c; g.N = 3 * inner; g.K = D;
        g.bias = b.b_qkv_c; g.out_t = p->qkv; g.ldo = 3 * inner; g.rows_per_batch = nt;
        g.rope = p->rope; g.rope_inner = inner; g.rope_heads = m->rope_heads;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_ROPE_T, st));
        F5_TRY(launch_copy_segments(p->qkv, (size_t)nt * qrow, (char*)p->qkvJ + (size_t)N * qrow, (size_t)S * qrow, (size_t)nt * qrow, nb, st));
        {
            int kind = 0;
            if (p->attn_kernel != 0 && attention_fast_supported(P, S, c.heads)) kind = 1;
            F5_TRY(launch_attention(P, kind, nb, S, c.heads, p->qkvJ, 3 * inner, maskJ, p->attJ, inner, st));
        }
        // text: c += gate_msa * to_out_c(attn_c); c += gate_mlp * ff_c(norm)   (:697-706; nothing in the context_pre_only block)
        if (!last) {
            F5_TRY(launch_copy_segments((const char*)p->attJ + (size_t)N * arow, (size_t)S * arow, p->cT, (size_t)nt * arow, (size_t)nt * arow, nb, st));
            F5_TRY(resid(p->cT, inner, b.w_o_c, b.b_o_c, inner, p->cres, rows_c, nt, mc + 2 * D, nullptr));
            F5_TRY(launch_layernorm(P, p->cres, D, rows_c, D, mc + 4 * D, mc + 3 * D, mod_bstride, nt, 1, p->hT, D, st));
            g = gp_zero();
            g.A = p->hT; g.lda = D; g.W = b.w_ff1_c; g.ldw = D; g.M = rows_c; g.N = ff; g.K = D;
            g.bias = b.b_ff1_c; g.act = ACT_GELU_TANH; g.out_t = p->ffh; g.ldo = ff;
            F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_T, st));
            F5_TRY(resid(p->ffh, ff, b.w_ff2_c, b.b_ff2_c, ff, p->cres, rows_c, nt, mc + 5 * D, nullptr));
        }
        // frames: x += gate_msa * to_out(attn_x) (0 on padded rows); x += gate_mlp * ff_x(norm)   (:709-713)
        F5_TRY(launch_copy_segments(p->attJ, (size_t)S * arow, p->cT, (size_t)N * arow, (size_t)N * arow, nb, st));
        F5_TRY(resid(p->cT, inner, b.w_o, b.b_o, inner, p->xres, rows, N, mx + 2 * D, mask));
        F5_TRY(launch_layernorm(P, p->xres, D, rows, D, mx + 4 * D, mx + 3 * D, mod_bstride, N, 1, p->hT, D, st));
        g = gp_zero();
        g.A = p->hT; g.lda = D; g.W = b.w_ff1; g.ldw = D; g.M = rows; g.N = ff; g.K = D;
        g.bias = b.b_ff1; g.act = ACT_GELU_TANH; g.out_t = p->ffh; g.ldo = ff;
        F5_TRY(run_gemm(p, g, GEMM_DENSE, EPI_STORE_T, st));
        F5_TRY(resid(p->ffh, ff, b.w_ff2, b.b_ff2, ff, p->xres, rows, N, mx + 5 * D, nullptr));
    }
    const float* mf = modp + (size_t)m->modrow - 2 * D;  // norm_out: (scale, shift) (modules.py:333)
    F5_TRY(launch_layernorm(P, p->xres, D, rows, D, mf, mf + D, mod_bstride, N, 1, p->hT, D, st));
    g = gp_zero();
    g.A = p->hT; g.lda = D; g.W = m->w_out; g.ldw = D; g.M = rows; g.N = MELP; g.K = D;
    g.bias = m->b_out; g.out_f = p->vout; g.ldof = MELP;
    return run_gemm(p, g, GEMM_DENSE, EPI_STORE_F32, st);
}

// one network evaluation of whichever backbone the model is (plug point A)
static int net_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, int time_row, int per_batch_rows, const uint8_t* mask, hipStream_t st) {
    f5_model_s* m = p->m;
    if (m->cfg.backbone == F5_BACKBONE_UNETT)
        return unett_eval(p, x, xrows, nb, N, p->temb + (size_t)time_row * m->cfg.dim, per_batch_rows ? m->cfg.dim : 0, mask, st);
    if (m->cfg.backbone == F5_BACKBONE_MMDIT)
        return mmdit_eval(p, x, xrows, nb, N, p->mod + (size_t)time_row * m->modrow, per_batch_rows ? m->modrow : 0, mask, st);
    p->fold_eval = per_batch_rows ? -1 : time_row;  // (the fold table holds one set of weights per evaluation TIME of the staged grid)
    return dit_eval(p, x, xrows, nb, N, p->mod + (size_t)time_row * m->modrow, per_batch_rows ? m->modrow : 0, mask, st);
}

static int check_plan_shape(f5_plan_s* p, int B, int N) {
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

// ----------------------------------------------------------------------------- public: sample

// everything between the staged inputs and the final state traj[steps]; capturable (no syncs, no allocations)
static int sample_body_ragged(f5_plan_s* p, const SampleArgs& a, hipStream_t st);

static int sample_body(f5_plan_s* p, const SampleArgs& a, hipStream_t st) {
    if (p->rg) return sample_body_ragged(p, a, st);
    f5_model_s* m = p->m;
    const f5_dit_config& c = m->cfg;
    const int B = a.B, N = a.N, mel = c.mel_dim, bn = B * N;
    const int nev = a.method == F5_ODE_MIDPOINT ? 2 * a.steps : a.steps;
    const size_t state = (size_t)bn * mel;
    // range guard of the fp16 residual stream: cleared by a KERNEL node of the graph, so every replay starts clean.  (Not hipMemsetAsync: on
    // ROCm 7.2 the 32-byte memset node captured here cleared the words on the first launch of the instantiated graph and filled them with
    // two stale host pointers on every replay -- gpurun_out/r3g: [512..519] = {0xf5dffab8, 0x78fe, 0xea66b000, 0x7909} x 2 -- which read as
    // a raised flag and sent the second sample() of every process through the fp32 fallback.)
    F5_TRY(launch_fill_f32(reinterpret_cast<float*>(p->sat_flag), 8, 0.0f, st));
    // (the AdaLN modulation rows of all evaluation times are already in p->mod: f5_sample keeps them across calls)
    // text embeddings are constants of the whole sample() (the reference caches them per branch, dit.py:202-210)
    F5_TRY(compute_text_embed(p, p->text_in, a.nt, B, N, 0, p->te[0], st));
    F5_TRY(compute_base(p, p->cond_in, p->lens_in, p->te[0], B, N, 0, 0, st));
    if (a.cfg_on) {
        F5_TRY(compute_text_embed(p, p->text_in, a.nt, B, N, 1, p->te[1], st));
        F5_TRY(compute_base(p, p->cond_in, p->lens_in, p->te[1], B, N, 1, (size_t)bn, st));
    }
    const uint8_t* mask = nullptr;
    if (a.mask_on) {
        F5_TRY(launch_len_mask(p->dur_in, B, N, p->mask, st));
        if (a.cfg_on) F5_HIP(hipMemcpyAsync(p->mask + bn, p->mask, bn, hipMemcpyDeviceToDevice, st));
        mask = p->mask;
        F5_TRY(launch_rowbits(mask, (a.cfg_on ? 2 : 1) * (int)bn, p->rowbits, st));
        p->rowbits_src = mask;
    }
    const int nb = a.cfg_on ? 2 * B : B;
    if (c.backbone == F5_BACKBONE_MMDIT) {  // the text stream restarts from these rows at every evaluation (mmdit.py:163-173: the cached embeddings)
        p->c_src[0] = p->te[0];
        p->c_src[1] = a.cfg_on ? p->te[1] : nullptr;
        p->c_nt = a.nt;
        p->c_rows_each = B * a.nt;
    }
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (g_sync_evals) (void)hipStreamIsCapturing(st, &capturing);
    for (int s = 0; s < a.steps; ++s) {
        if (g_sync_evals && capturing == hipStreamCaptureStatusNone && s > 0) F5_HIP(hipStreamSynchronize(st));
        float* xs = p->traj + (size_t)s * state;
        float* xn = p->traj + (size_t)(s + 1) * state;
        const float* vu = a.cfg_on ? p->vout + (size_t)bn * MELP : nullptr;
        if (a.method == F5_ODE_EULER) {
            F5_TRY(net_eval(p, xs, bn, nb, N, s, 0, mask, st));
            F5_TRY(launch_cfg_step(xs, p->vout, vu, MELP, bn, mel, a.cfg, p->coefs + s, xn, nullptr, st));
        } else {
            F5_TRY(net_eval(p, xs, bn, nb, N, 2 * s, 0, mask, st));
            F5_TRY(launch_cfg_step(xs, p->vout, vu, MELP, bn, mel, a.cfg, p->coefs + 2 * s, p->xmid, nullptr, st));
            F5_TRY(net_eval(p, p->xmid, bn, nb, N, 2 * s + 1, 0, mask, st));
            F5_TRY(launch_cfg_step(xs, p->vout, vu, MELP, bn, mel, a.cfg, p->coefs + 2 * s + 1, xn, nullptr, st));
        }
    }
    return 0;
}

static void drop_graphs(f5_plan_s* p) {
    for (auto& g : p->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    p->graphs.clear();
}

// the ODE loop of one sample() on the staged inputs: replay of the hipGraph captured for this exact problem, or eager launches
static int run_sample_loop(f5_plan_s* p, const SampleArgs& a, int use_graph, hipStream_t st) {
    if (use_graph && p->taps.empty() && !p->timing) {
        GraphEntry* ge = nullptr;
        for (size_t i = 0; i < p->graphs.size();) {  // a tuning knob changed since the capture: the graph baked the old kernel choice
            if (p->graphs[i].epoch != g_tuning_epoch) {
                (void)hipGraphExecDestroy(p->graphs[i].exec);
                (void)hipGraphDestroy(p->graphs[i].graph);
                p->graphs.erase(p->graphs.begin() + i);
            } else {
                ++i;
            }
        }
        for (auto& g : p->graphs)
            if (g.B == a.B && g.N == a.N && g.nt == a.nt && g.steps == a.steps && g.method == a.method && g.cfg_on == a.cfg_on &&
                g.mask_on == a.mask_on && g.cfg == a.cfg && g.fold_id == (p->fold && g_ln_fold ? p->fold->id : 0) &&
                g.rn == (p->rg ? p->rg->n : std::vector<int>()))
                ge = &g;
        if (!ge) {
            GraphEntry g{a.B, a.N, a.nt, a.steps, a.method, a.cfg_on, a.mask_on, a.cfg, g_tuning_epoch, (p->fold && g_ln_fold) ? p->fold->id : 0};
            if (p->rg) g.rn = p->rg->n;
            if (!p->cap_stream) F5_HIP(hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking));
            F5_HIP(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed));
            int rc = sample_body(p, a, p->cap_stream);
            hipError_t e = hipStreamEndCapture(p->cap_stream, &g.graph);
            if (rc != 0) {
                if (g.graph) (void)hipGraphDestroy(g.graph);
                return rc;
            }
            if (e != hipSuccess) return f5_fail(F5_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
            e = hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0);
            if (e != hipSuccess) {
                (void)hipGraphDestroy(g.graph);
                return f5_fail(F5_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
            }
            if (p->graphs.size() >= 8) {  // small LRU-less cache: drop the oldest bucket
                (void)hipGraphExecDestroy(p->graphs[0].exec);
                (void)hipGraphDestroy(p->graphs[0].graph);
                p->graphs.erase(p->graphs.begin());
            }
            p->graphs.push_back(g);
            ge = &p->graphs.back();
        }
        F5_HIP(hipGraphLaunch(ge->exec, st));
    } else {
        F5_TRY(sample_body(p, a, st));
    }
    return 0;
}

// The stream was stored as saturating fp16: one small read of the flag words the LayerNorm passes raise (a stream synchronisation).  A
// large-activation checkpoint must not clip silently: the loop is repeated with fp32 residual storage -- y0 is still traj[0], every other
// input is staged -- and the plan keeps fp32 storage from now on.
static int guard_check_and_fallback(f5_plan_s* p, const SampleArgs& a, int use_graph, hipStream_t st) {
    F5_HIP(hipMemcpyAsync(p->sat_host, p->sat_flag, 32, hipMemcpyDeviceToHost, st));
    F5_HIP(hipStreamSynchronize(st));
    if (*p->sat_host != 0u) {
        p->sat_amax_bits = p->sat_host[1];
        p->sat_nan = p->sat_host[2] != 0u;
        p->sat_pass = p->sat_host[3];
        p->sat_blocks = p->sat_host[4];
        p->sat_row = 0x7fffffffu - p->sat_host[5];
        p->res_f16 = 0;
        ++p->fallbacks;
        drop_graphs(p);  // they baked the fp16 kernels
        F5_TRY(run_sample_loop(p, a, use_graph, st));
    }
    return 0;
}

static int finish_outputs(f5_plan_s* p, const SampleArgs& a, float* out, float* trajectory, hipStream_t st) {
    const int mel = p->m->cfg.mel_dim;
    const size_t state = (size_t)a.B * a.N * mel;
    F5_TRY(launch_final_where(p->cond_in, p->traj + (size_t)a.steps * state, p->lens_in, a.B, a.N, mel, out, st));
    if (trajectory) F5_HIP(hipMemcpyAsync(trajectory, p->traj, (size_t)(a.steps + 1) * state * sizeof(float), hipMemcpyDeviceToDevice, st));
    return 0;
}

// LayerNorm fold: point the plan at the model's table for the evaluation times `tv` (p->mod holds their AdaLN rows, computed on `st`), building it
// when no plan has sampled on this grid yet.  Never an error: without a table (knob off, grid too long, allocation refused) the unfolded path runs.
static void drop_graphs(f5_plan_s* p);
static int acquire_fold(f5_plan_s* p, const std::vector<float>& tv, hipStream_t st) {
    f5_model_s* m = p->m;
    const int nev = (int)tv.size();
    FoldTable* want = nullptr;
    if (m->w_fold && g_ln_fold && p->lnf_stats && nev > 0 && nev <= F5_FOLD_MAX_EVALS) {
        for (FoldTable* t : m->folds)
            if (t->tv == tv) want = t;
        if (!want) {
            for (size_t i = 0; i < m->folds.size() && m->folds.size() >= F5_FOLD_TABLES;) {  // oldest first, never one a plan still points to
                if (m->folds[i]->users == 0 && m->folds[i] != p->fold) {
                    (void)hipDeviceSynchronize();  // (launches of other streams may still read it)
                    delete m->folds[i];
                    m->folds.erase(m->folds.begin() + i);
                } else {
                    ++i;
                }
            }
            const f5_dit_config& c = m->cfg;
            const size_t n = (size_t)nev * c.depth * m->fold_R;
            FoldTable* t = new FoldTable();
            bool ok = t->arena.alloc(&t->Wt, n * c.dim * 2, false) == 0 && t->arena.alloc_t(&t->c1, n, false) == 0 && t->arena.alloc_t(&t->c2, n, false) == 0 &&
                      hipEventCreateWithFlags(&t->ready, hipEventDisableTiming) == hipSuccess;
            ok = ok && launch_fold_weights(m->w_fold, m->b_fold, p->mod, m->modrow, nev, c.depth, m->fold_R, 3 * m->inner, c.dim, t->Wt, t->c1, t->c2, st) == 0 &&
                 hipEventRecord(t->ready, st) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                delete t;
            } else {
                t->tv = tv;
                t->id = ++m->fold_seq;
                m->folds.push_back(t);
                want = t;
            }
        } else if (want != p->fold) {
            F5_HIP(hipStreamWaitEvent(st, want->ready, 0));  // built on another plan's stream
        }
    }
    if (want != p->fold) {
        if (p->fold) --p->fold->users;
        p->fold = want;
        if (want) ++want->users;
        for (size_t i = 0; i < p->graphs.size();) {  // captures that baked another table's addresses
            if (p->graphs[i].fold_id != (want ? want->id : 0)) {
                (void)hipGraphExecDestroy(p->graphs[i].exec);
                (void)hipGraphDestroy(p->graphs[i].graph);
                p->graphs.erase(p->graphs.begin() + i);
            } else {
                ++i;
            }
        }
    }
    return 0;
}

// evaluation times / step coefficients of a fixed grid (fp32 op order of torchdiffeq's fixed-grid solvers) -> p->tvals / p->coefs, and the
// AdaLN rows of every evaluation time -> p->mod (kept across calls with the same grid on the same stream)
static int stage_time_grid(f5_plan_s* p, const float* tgrid_host, int steps, int ode_method, hipStream_t st) {
    const int nev = ode_method == F5_ODE_MIDPOINT ? 2 * steps : steps;
    std::vector<float> tv(nev), cf(nev);
    for (int s = 0; s < steps; ++s) {
        const float t0 = tgrid_host[s], t1 = tgrid_host[s + 1];
        const float dt = t1 - t0;
        if (ode_method == F5_ODE_EULER) {
            tv[s] = t0;
            cf[s] = dt;
        } else {
            const float half = 0.5f * dt;
            tv[2 * s] = t0;
            cf[2 * s] = half;
            tv[2 * s + 1] = t0 + half;
            cf[2 * s + 1] = dt;
        }
    }
    // The time MLP and every AdaLN row depend only on the evaluation times: a server calls sample() with the same grid every time,
    // so the 0.56 GB weight pass is done once per grid and kept (1.7 ms per call; 3 % of a single-utterance sample()).
    if (p->mod_tv != tv || p->mod_stream != st) {  // the rows are ordered only behind the stream they were computed on
        p->mod_tv.clear();
        F5_TRY(launch_set_floats(p->tvals, tv.data(), nev, st));
        F5_TRY(compute_modulation(p, p->tvals, nev, st));
        p->mod_tv = tv;
        p->mod_stream = st;
    }
    F5_TRY(acquire_fold(p, tv, st));
    return launch_set_floats(p->coefs, cf.data(), nev, st);
}

// A plan holds ONE deferred sample(): its staged inputs and the device flag words would be overwritten by the next call.  Every entry point
// that reuses the plan's buffers completes the deferred call first (on the stream it was enqueued on), so a raised range-guard flag is never
// lost when a caller round-robins more chunks than it has streams.
extern "C" int f5_sample_finish(f5_plan_t p, f5_stream_t stream);
static int finish_if_pending(f5_plan_s* p) {
    if (p && p->pending.valid) return f5_sample_finish(p, (f5_stream_t)p->pending.stream);
    return 0;
}

extern "C" int f5_sample(f5_plan_t p, int B, int N, const float* cond, const int32_t* text, int nt, const int32_t* lens,
                         const int32_t* durations, const float* y0, const float* tgrid_host, int steps, float cfg_strength, int ode_method,
                         float* out, float* trajectory, int use_graph, f5_stream_t stream) {
    F5_TRY(check_plan_shape(p, B, N));
    if (!cond || !text || !lens || !y0 || !tgrid_host || !out) return f5_fail(F5_EINVAL, "null argument");
    if (steps <= 0 || nt <= 0) return f5_fail(F5_EINVAL, "steps and nt must be positive");
    if (ode_method != F5_ODE_EULER && ode_method != F5_ODE_MIDPOINT) return f5_fail(F5_EINVAL, "bad ode_method");
    const int nev = ode_method == F5_ODE_MIDPOINT ? 2 * steps : steps;
    if (nev > p->maxE) return f5_fail(F5_EINVAL, "%d evaluations exceed the plan's max_evals=%d", nev, p->maxE);
    F5_TRY(finish_if_pending(p));
    hipStream_t st = (hipStream_t)stream;
    f5_model_s* m = p->m;
    const int mel = m->cfg.mel_dim, bn = B * N;
    const size_t state = (size_t)bn * mel;
    const bool mmdit = m->cfg.backbone == F5_BACKBONE_MMDIT;
    if (mmdit && nt > p->maxN) return f5_fail(F5_EINVAL, "MMDiT: text length %d exceeds the plan's max_seq %d", nt, p->maxN);
    const int nt_eff = (mmdit || nt < N) ? nt : N;  // tokens beyond the frame count are curtailed (dit.py:51; MMDiT keeps them all, mmdit.py:40)

    F5_TRY(stage_time_grid(p, tgrid_host, steps, ode_method, st));
    F5_HIP(hipMemcpyAsync(p->cond_in, cond, state * sizeof(float), hipMemcpyDeviceToDevice, st));
    F5_HIP(hipMemcpyAsync(p->traj, y0, state * sizeof(float), hipMemcpyDeviceToDevice, st));
    F5_HIP(hipMemcpy2DAsync(p->text_in, (size_t)nt_eff * 4, text, (size_t)nt * 4, (size_t)nt_eff * 4, B, hipMemcpyDeviceToDevice, st));
    F5_HIP(hipMemcpyAsync(p->lens_in, lens, B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (durations) F5_HIP(hipMemcpyAsync(p->dur_in, durations, B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));

    SampleArgs a{B, N, nt_eff, steps, ode_method, cfg_strength >= 1e-5f ? 1 : 0, durations ? 1 : 0, cfg_strength};  // cfm.py:167
    const bool guarded = plan_res_f16(p) && p->sat_check && !p->timing;
    F5_TRY(run_sample_loop(p, a, use_graph, st));
    p->pending = PendingSample{};
    if (guarded && p->sat_check == 2) {
        // deferred guard (plan option "residual_guard" = 2): nothing synchronises here, so several plans can be fed on several streams from
        // one host thread; f5_sample_finish reads the flag later and repeats the loop if it must (the inputs stay staged in the plan)
        p->pending = PendingSample{true, a, use_graph, out, trajectory, st};
    } else if (guarded) {
        F5_TRY(guard_check_and_fallback(p, a, use_graph, st));
    }
    return finish_outputs(p, a, out, trajectory, st);
}

extern "C" int f5_sample_finish(f5_plan_t p, f5_stream_t stream) {
    if (!p) return f5_fail(F5_EINVAL, "null plan");
    if (!p->pending.valid) return 0;
    const PendingSample ps = p->pending;
    p->pending = PendingSample{};
    hipStream_t st = (hipStream_t)stream;
    const int before = p->fallbacks;
    F5_TRY(guard_check_and_fallback(p, ps.a, ps.use_graph, st));
    if (p->fallbacks != before) return finish_outputs(p, ps.a, ps.out, ps.trajectory, st);  // the loop ran again: write the outputs again
    return 0;
}

// ----------------------------------------------------------------------------- public: ragged sample
// The ODE loop of f5_sample over utterances of DIFFERENT frame counts in one set of launches, with no padding to a common length and no key
// mask: what F5TTSWrapper.generate needs for the text chunks of one call (the reference runs them one after the other at batch 1,
// infer/f5tts_wrapper.py:476-533; a batch-1 sample() has mask = None, cfm.py:152-155).  Every per-row kernel (GEMMs, LayerNorm, CFG step) runs
// over the concatenation; the three places where a token sees its neighbours are handled so that each utterance gets exactly the arithmetic
// of its own batch-1 call: the position conv reads zero gap rows where it would read its zero padding, RoPE takes a per-row position table,
// attention is launched per utterance on its rows.  Text embedding and the hoisted half of the input embedding are computed per utterance.
static int sample_body_ragged(f5_plan_s* p, const SampleArgs& a, hipStream_t st) {
    f5_model_s* m = p->m;
    const f5_dit_config& c = m->cfg;
    const Ragged& rg = *p->rg;
    const int B = a.B, T = rg.T, mel = c.mel_dim, td = c.text_dim;
    const size_t state = (size_t)T * mel;
    F5_TRY(launch_fill_f32(reinterpret_cast<float*>(p->sat_flag), 8, 0.0f, st));
    for (int br = 0; br < (a.cfg_on ? 2 : 1); ++br)
        for (int u = 0; u < B; ++u) {
            const int nu = rg.n[u], nt_eff = a.nt < nu ? a.nt : nu;  // tokens beyond the frame count are curtailed (dit.py:51)
            float* te = p->te[br] + (size_t)rg.off[u] * td;
            F5_HIP(hipMemcpy2DAsync(p->text_in + (size_t)B * a.nt, (size_t)nt_eff * 4, p->text_in + (size_t)u * a.nt, (size_t)a.nt * 4, (size_t)nt_eff * 4, 1,
                                    hipMemcpyDeviceToDevice, st));
            F5_TRY(compute_text_embed(p, p->text_in + (size_t)B * a.nt, nt_eff, 1, nu, br, te, st));
            F5_TRY(compute_base(p, p->cond_in + (size_t)rg.off[u] * mel, p->lens_in + u, te, 1, nu, br, (size_t)br * T + rg.off[u], st));
        }
    const int nb = a.cfg_on ? 2 : 1;
    for (int s = 0; s < a.steps; ++s) {
        float* xs = p->traj + (size_t)s * state;
        float* xn = p->traj + (size_t)(s + 1) * state;
        const float* vu = a.cfg_on ? p->vout + (size_t)T * MELP : nullptr;
        if (a.method == F5_ODE_EULER) {
            F5_TRY(net_eval(p, xs, T, nb, T, s, 0, nullptr, st));
            F5_TRY(launch_cfg_step(xs, p->vout, vu, MELP, T, mel, a.cfg, p->coefs + s, xn, nullptr, st));
        } else {
            F5_TRY(net_eval(p, xs, T, nb, T, 2 * s, 0, nullptr, st));
            F5_TRY(launch_cfg_step(xs, p->vout, vu, MELP, T, mel, a.cfg, p->coefs + 2 * s, p->xmid, nullptr, st));
            F5_TRY(net_eval(p, p->xmid, T, nb, T, 2 * s + 1, 0, nullptr, st));
            F5_TRY(launch_cfg_step(xs, p->vout, vu, MELP, T, mel, a.cfg, p->coefs + 2 * s + 1, xn, nullptr, st));
        }
    }
    return 0;
}

extern "C" int f5_sample_ragged(f5_plan_t p, int B, const int32_t* frames_host, const float* cond, const int32_t* text, int nt, const int32_t* lens,
                                const float* y0, const float* tgrid_host, int steps, float cfg_strength, int ode_method, float* out,
                                f5_stream_t stream) {
    if (!p) return f5_fail(F5_EINVAL, "null plan");
    F5_TRY(f5_check_device());
    if (!frames_host || !cond || !text || !lens || !y0 || !tgrid_host || !out) return f5_fail(F5_EINVAL, "null argument");
    f5_model_s* m = p->m;
    if (m->cfg.backbone != F5_BACKBONE_DIT) return f5_fail(F5_ENOTSUP, "f5_sample_ragged: DiT backbone only");
    if (B <= 0 || B > p->maxB || steps <= 0 || nt <= 0 || nt > p->maxN) return f5_fail(F5_EINVAL, "bad B / steps / nt for this plan");
    if (ode_method != F5_ODE_EULER && ode_method != F5_ODE_MIDPOINT) return f5_fail(F5_EINVAL, "bad ode_method");
    const int nev = ode_method == F5_ODE_MIDPOINT ? 2 * steps : steps;
    if (nev > p->maxE) return f5_fail(F5_EINVAL, "%d evaluations exceed the plan's max_evals=%d", nev, p->maxE);
    if (!p->taps.empty() || p->timing) return f5_fail(F5_ESTATE, "f5_sample_ragged: stage taps / in-situ timing are not available here");
    Ragged rg;
    size_t total = 0;
    for (int u = 0; u < B; ++u) {
        if (frames_host[u] <= 0 || frames_host[u] > p->maxN) return f5_fail(F5_EINVAL, "utterance %d: %d frames outside 1 .. %d", u, frames_host[u], p->maxN);
        rg.off.push_back(rg.T);
        rg.n.push_back(frames_host[u]);
        total += (size_t)frames_host[u];
        rg.T = (int)round_up((size_t)rg.T + frames_host[u] + RAGGED_GAP, 16);
    }
    const size_t bn_cap = (size_t)p->maxB * p->maxN;
    if ((size_t)rg.T > bn_cap || 2 * (size_t)rg.T > p->rows_cap || (size_t)(B + 1) * nt > bn_cap)
        return f5_fail(F5_EINVAL, "%d rows (frames + gaps) exceed the plan (max_batch x max_seq = %zu)", rg.T, bn_cap);
    F5_TRY(finish_if_pending(p));
    hipStream_t st = (hipStream_t)stream;
    const int mel = m->cfg.mel_dim, D = m->cfg.dim, td = m->cfg.text_dim, T = rg.T;
    F5_TRY(stage_time_grid(p, tgrid_host, steps, ode_method, st));
    // stage the inputs at their row offsets; everything between the utterances is zero
    F5_HIP(hipMemsetAsync(p->cond_in, 0, (size_t)T * mel * sizeof(float), st));
    F5_HIP(hipMemsetAsync(p->traj, 0, (size_t)T * mel * sizeof(float), st));
    F5_HIP(hipMemsetAsync(p->te[0], 0, (size_t)T * td * sizeof(float), st));
    F5_HIP(hipMemsetAsync(p->te[1], 0, (size_t)T * td * sizeof(float), st));
    F5_HIP(hipMemsetAsync(p->base, 0, (size_t)2 * T * D * sizeof(float), st));
    if (p->base16) F5_HIP(hipMemsetAsync(p->base16, 0, (size_t)2 * T * D * 2, st));
    F5_HIP(hipMemsetAsync(p->rope_exp, 0, (size_t)T * 64 * sizeof(float), st));
    F5_HIP(hipMemsetAsync(p->gapflag, 1, (size_t)2 * T, st));
    size_t src = 0;
    for (int u = 0; u < B; ++u) {
        const size_t nu = rg.n[u], off = rg.off[u];
        F5_HIP(hipMemcpyAsync(p->cond_in + off * mel, cond + src * mel, nu * mel * sizeof(float), hipMemcpyDeviceToDevice, st));
        F5_HIP(hipMemcpyAsync(p->traj + off * mel, y0 + src * mel, nu * mel * sizeof(float), hipMemcpyDeviceToDevice, st));
        F5_HIP(hipMemcpyAsync(p->rope_exp + off * 64, p->rope, nu * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
        F5_HIP(hipMemsetAsync(p->gapflag + off, 0, nu, st));
        F5_HIP(hipMemsetAsync(p->gapflag + T + off, 0, nu, st));
        src += nu;
    }
    F5_HIP(hipMemcpyAsync(p->text_in, text, (size_t)B * nt * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    F5_HIP(hipMemcpyAsync(p->lens_in, lens, B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));

    SampleArgs a{B, T, nt, steps, ode_method, cfg_strength >= 1e-5f ? 1 : 0, 0, cfg_strength};  // cfm.py:167
    p->pending = PendingSample{};
    p->rg = &rg;
    const int use_graph = p->ragged_graph;  // a bucket shape that recurs (batch inference over fixed buckets) replays its capture
    int rc = run_sample_loop(p, a, use_graph, st);
    if (rc == 0 && plan_res_f16(p) && p->sat_check) rc = guard_check_and_fallback(p, a, use_graph, st);  // (always checked inside the call)
    p->rg = nullptr;
    F5_TRY(rc);
    const float* xf = p->traj + (size_t)steps * T * mel;
    src = 0;
    for (int u = 0; u < B; ++u) {
        const size_t off = rg.off[u];
        F5_TRY(launch_final_where(p->cond_in + off * mel, xf + off * mel, p->lens_in + u, 1, rg.n[u], mel, out + src * mel, st));
        src += (size_t)rg.n[u];
    }
    (void)total;
    return 0;
}
