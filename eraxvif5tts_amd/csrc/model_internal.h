// model_internal.h -- shared by the translation units of the model path (round 4: csrc/model.hip split per concern, no launch changed):
//   model.hip        weights in HBM (f5_model_*), plans / workspaces / options / in-situ timing (f5_plan_*)
//   eval_common.hip  tuning knobs, GEMM dispatch, stage taps, AdaLN table, text embedding, hoisted input embedding, net_eval, f5_dit_forward
//   dit_eval.hip     one DiT evaluation (dit.py:185-233, modules.py:301-336,610-641)
//   unett_eval.hip   one UNetT evaluation (backbones/unett.py:185-253)
//   mmdit_eval.hip   one MMDiT evaluation (backbones/mmdit.py:146-190)
//   sampler.hip      CFM.sample: ODE loop, hipGraph replay, fp16 range guard, LayerNorm-fold tables, ragged sampler (cfm.py:82-208)
#pragma once
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
    unsigned* fin_counter = nullptr;  // one ticket word per 128 token rows: in-launch statistics of the non-persistent residual GEMMs (gemm.h); zero between launches
    float *lnf_stats = nullptr, *lnf_stats2 = nullptr, *lnf_partial = nullptr;  // (two statistics tables: a site's pivots are the previous site's means)
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

int finish_if_pending(f5_plan_s* p);  // completes a deferred sample() before the plan's buffers are reused (sampler.hip)


// ---- eval_common.hip
extern int g_w_prefetch, g_res_f16, g_ln_defer, g_resid_rmw, g_ln_fold, g_ln_fold_inkernel, g_ln_fold_fin, g_gemm_pad_rows, g_sync_evals, g_gemm_w4;
bool plan_res_f16(const f5_plan_s* p);
GemmParams gp_zero();
int run_gemm(f5_plan_s* p, const GemmParams& g, int mode, int epi, hipStream_t st);
float* tap_dst(f5_plan_s* p, const std::string& name);
int tap_f32(f5_plan_s* p, const std::string& name, const float* src, int ld, int rows, int cols, hipStream_t st);
int tap_t(f5_plan_s* p, const std::string& name, const void* src, int ld, int rows, int cols, hipStream_t st);
int compute_modulation(f5_plan_s* p, const float* tvals_dev, int n, hipStream_t st);
int compute_text_embed(f5_plan_s* p, const int32_t* text, int nt, int B, int N, int drop_text, float* out, hipStream_t st);
int compute_base(f5_plan_s* p, const float* cond, const int32_t* lens, const float* te, int nb, int N, int zero_cond, size_t row0, hipStream_t st);
int net_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, int time_row, int per_batch_rows, const uint8_t* mask, hipStream_t st);
int check_plan_shape(f5_plan_s* p, int B, int N);
// ---- one evaluation per backbone
int dit_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* modp, int mod_bstride, const uint8_t* mask, hipStream_t st);
int unett_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* temb, int temb_bstride, const uint8_t* mask, hipStream_t st);
int mmdit_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* modp, int mod_bstride, const uint8_t* mask, hipStream_t st);
