/*
 * f5hip.h -- C ABI of libf5hip.so: the MI355X (gfx950) implementation of the F5-TTS flow-matching
 * inference hot path (CFM.sample -> DiT ODE loop -> Vocos), written from scratch in HIP.
 *
 * The reference (hungkq-1724/EraXviF5TTS) is 100 % Python and has no native ABI; the entry points below
 * are what a binding for the reference's two plug points would call:
 *
 *   plug point A  "backbone class"   f5_tts/infer/f5tts_wrapper.py:134,145  model_cls(**arch, ...)
 *                                    f5_tts/model/cfm.py:164-172            transformer(x, cond, text, time, ...)
 *   plug point B  "vocoder object"   f5_tts/infer/utils_infer.py:101-124    load_vocoder(...)
 *                                    f5_tts/infer/f5tts_wrapper.py:524      vocoder.decode(mel)
 *   sampler                          f5_tts/model/cfm.py:82-208             CFM.sample(...)
 *
 * Conventions
 *   - every function returns 0 on success, a negative F5_E* code on failure; f5_last_error() returns a
 *     thread-local human readable message.  Nothing throws across the boundary.
 *   - "dev" pointers are device (HBM) pointers owned by the caller (e.g. torch tensor data_ptr());
 *     "host" pointers are ordinary host memory.  The library never frees or reallocates caller memory.
 *   - all work is enqueued on the caller's hipStream_t (f5_stream_t); the library only synchronises
 *     inside f5_*_create / f5_model_finalize (uploads) and never on the hot path.
 *   - handles are not thread-safe; distinct handles may be used from distinct threads/processes.
 *   - there is NO CPU fallback: every entry point fails with F5_ENODEVICE when no gfx950 device is usable.
 */
#ifndef F5HIP_H
#define F5HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#define F5_API __attribute__((visibility("default")))

#define F5HIP_VERSION 400 /* 0.4.0 (round 4): + the BigVGAN entry points and mel front-end, f5_op_ln_fold, bounded timing ring; plan options "ln_fold_active", "gemm_w4"; tuning keys "ln_fold", "ln_fold_fin", "gemm_w4"; 0.3.2: + f5_sample_ragged, f5_mmdit_forward, f5_duration_predict_g */

/* error codes */
#define F5_OK 0
#define F5_EINVAL -1    /* bad argument / shape */
#define F5_ENODEVICE -2 /* no usable HIP device */
#define F5_EHIP -3      /* a HIP runtime call failed (message has the hipError string) */
#define F5_ESTATE -4    /* call out of order (e.g. forward before finalize) */
#define F5_ENOTSUP -5   /* configuration the kernels do not implement (qk_norm, long_skip_connection) */
#define F5_ENOMEM -6

/* arithmetic of the dense kernels */
#define F5_PREC_BF16 0 /* bf16 MFMA inputs, fp32 accumulate and arithmetic, fp32 ODE state, residual stream stored as fp16 (production) */
#define F5_PREC_FP32 1 /* fp32-input MFMA everywhere (debug / parity mode, ~1/16 of the bf16 rate) */

/* ODE solvers of torchdiffeq's fixed-grid family used by cfm.py:197 */
#define F5_ODE_EULER 0
#define F5_ODE_MIDPOINT 1

typedef void* f5_stream_t; /* hipStream_t */
typedef struct f5_model_s* f5_model_t;
typedef struct f5_plan_s* f5_plan_t;
typedef struct f5_vocoder_s* f5_vocoder_t;

/* mirrors the kwargs of DiT.__init__ (f5_tts/model/backbones/dit.py:104-122) */
typedef struct f5_dit_config {
    int32_t dim;               /* model width (1024) */
    int32_t depth;             /* number of DiT blocks (22) */
    int32_t heads;             /* attention heads (16) */
    int32_t dim_head;          /* 64 (only 64 is implemented) */
    int32_t ff_inner;          /* int(dim * ff_mult) (2048) */
    int32_t mel_dim;           /* 100 */
    int32_t text_num_embeds;   /* vocab size V (embedding table has V+1 rows) */
    int32_t text_dim;          /* 512 */
    int32_t conv_layers;       /* ConvNeXtV2 blocks in the text embedder (4) */
    int32_t text_mask_padding; /* 0 (F5TTS_Base) / 1 (F5TTS_v1_Base) */
    int32_t pe_attn_head;      /* heads that receive RoPE; <= 0 means all heads (None) */
    int32_t qk_norm;           /* must be 0 (null in every shipped config) */
    int32_t long_skip;         /* must be 0 */
    int32_t precision;         /* F5_PREC_* */
    int32_t rope_layout;       /* F5_ROPE_* : which feature pairs of a head one rotary frequency turns (x_transformers is not vendored
                                * in the reference tree; the adjacent-pair form is the one the pinned >=1.31 releases publish) */
    int32_t backbone;          /* F5_BACKBONE_*: which class of plug point A (0 = DiT, the default of a zero-initialised struct) */
    int32_t skip_connect;      /* UNetT only: F5_SKIP_* (reference unett.py:118 skip_connect_type) */
} f5_dit_config;

/* backbones behind plug point A (reference infer/f5tts_wrapper.py:134: model_cls = f5_tts.model.<cfg.model.backbone>) */
#define F5_BACKBONE_DIT 0   /* model/backbones/dit.py:103   (F5-TTS) */
#define F5_BACKBONE_UNETT 1 /* model/backbones/unett.py:103 (E2-TTS: flat U-Net transformer; tensor names = UNetT.state_dict(): layers.<i>.{0.weight,
                             * 1.g, 2.to_q|to_k|to_v|to_out.0.{weight,bias}, 3.g, 4.ff.0.0.*, 4.ff.2.*}, norm_out.g; depth must be even) */
#define F5_BACKBONE_MMDIT 2 /* model/backbones/mmdit.py:85  (SD3-style joint text / audio blocks; tensor names = MMDiT.state_dict(): audio_embed.*,
                             * transformer_blocks.<i>.{attn_norm_x, attn_norm_c}.linear.*, .attn.{to_q,to_k,to_v}[_c].*, .attn.to_out.0.*, .attn.to_out_c.*,
                             * .ff_x.ff.*, .ff_c.ff.* (no to_out_c / ff_c in the last, context_pre_only block); text_dim must equal dim, conv_layers 0) */
#define F5_SKIP_CONCAT 0 /* x = Linear(2 dim -> dim, no bias)(cat(x, skip)) in the second half of the layers (default) */
#define F5_SKIP_ADD 1
#define F5_SKIP_NONE 2

/* rotary layouts of the q/k head features (dim_head = 64, 32 frequencies), reference model/modules.py:452-461 via x_transformers */
#define F5_ROPE_ADJACENT 0  /* frequency j turns features (2j, 2j+1): rotate_half on '... (d r) -> ... d r', r = 2 (default) */
#define F5_ROPE_HALF_SPLIT 1 /* frequency j turns features (j, j+32): rotate_half on the two halves of the head (GPT-NeoX form) */

F5_API const char* f5_last_error(void);
F5_API int f5_version(void);
/* number of usable gfx950 devices; name_out (>= 64 bytes, may be NULL) receives the arch string of device 0 */
F5_API int f5_device_count(char* name_out);

/* ------------------------------------------------------------------ model (weights resident in HBM) */
F5_API int f5_model_create(const f5_dit_config* cfg, f5_model_t* out);
/* name = DiT.state_dict() key, data = contiguous host fp32, shape as in the checkpoint.
 * Unknown names return F5_EINVAL (callers pass strict=False semantics by skipping names themselves). */
F5_API int f5_model_set_tensor(f5_model_t m, const char* name, const float* host_data, const int64_t* shape, int ndim);
/* returns 1/0 whether `name` is a tensor the model expects (and its element count in *numel if non-NULL) */
F5_API int f5_model_has_tensor(f5_model_t m, const char* name, int64_t* numel);
/* builds the fused device layouts (QKV concat, split input projection, rearranged grouped-conv taps, bf16 copies) */
F5_API int f5_model_finalize(f5_model_t m);
/* Destroy every plan of a model BEFORE the model: a plan may hold one of the model's LayerNorm-fold tables (round 4: per-time-grid weights
 * W' = fp16(W (1 + scale)), up to two grids per model, 231 MB per evaluation time at F5TTS_Base; built by the first f5_sample on a grid). */
F5_API int f5_model_destroy(f5_model_t m);

/* ------------------------------------------------------------------ plan (workspace for one (batch, seq) bucket) */
/* max_batch = utterances per call (CFG doubling is internal), max_seq = frames N, max_evals = network
 * evaluations times held at once (steps for euler, 2*steps for midpoint). */
F5_API int f5_plan_create(f5_model_t m, int max_batch, int max_seq, int max_evals, f5_plan_t* out);
F5_API int f5_plan_destroy(f5_plan_t p);
F5_API int64_t f5_plan_workspace_bytes(f5_plan_t p);

/* CFM.sample (cfm.py:82-208) after text->ids and duration resolution:
 *   cond      dev f32 [B, N, mel]  prompt mel zero-padded to N (NOT yet masked by lens: the kernel applies cfm.py:148-150)
 *   text      dev i32 [B, nt]      token ids, -1 padded (utils.py:88-95)
 *   lens      dev i32 [B]          prompt lengths in frames (cond_mask = frame < lens)
 *   durations dev i32 [B] or NULL  total lengths; NULL = no key-padding mask (the batch==1 path, cfm.py:152-155)
 *   y0        dev f32 [B, N, mel]  initial noise, rows >= duration zero (cfm.py:178-183)
 *   tgrid     host f32 [steps+1]   time grid after sway sampling (cfm.py:193-195)
 *   out       dev f32 [B, N, mel]  where(cond_mask, cond, y(1)) (cfm.py:200-202)
 *   trajectory dev f32 [steps+1, B, N, mel] or NULL
 * use_graph != 0 replays a hipGraph captured for this exact (B, N, nt, steps, method, cfg on/off, mask on/off). */
F5_API int f5_sample(f5_plan_t p, int B, int N, const float* cond, const int32_t* text, int nt, const int32_t* lens,
              const int32_t* durations, const float* y0, const float* tgrid_host, int steps, float cfg_strength,
              int ode_method, float* out, float* trajectory, int use_graph, f5_stream_t stream);

/* Deferred range-guard check (plan option "residual_guard" = 2): f5_sample then enqueues everything and returns without synchronising, so one
 * host thread can feed several plans on several streams (F5TTSWrapper.generate runs the text chunks of a call concurrently that way: they
 * are independent, reference infer/f5tts_wrapper.py:476-533).  f5_sample_finish(plan, stream) synchronises the stream, reads the flag and, if
 * it was raised, repeats the loop with fp32 residual storage and rewrites the outputs; it is a no-op when nothing is pending. */
F5_API int f5_sample_finish(f5_plan_t p, f5_stream_t stream);

/* The same ODE loop over B utterances of DIFFERENT frame counts in one set of launches -- no padding to a common length, no key mask:
 * what F5TTSWrapper.generate needs for the text chunks of one call (the reference samples them one after the other at batch 1,
 * infer/f5tts_wrapper.py:476-533; a batch-1 sample() has mask = None, model/cfm.py:152-155).  Each utterance gets the arithmetic of its own
 * batch-1 f5_sample call: bit-identical output whenever both calls take the tuned kernels (every frames_host[i] >= 256, bf16 mode) or both the
 * fp32 mode's.  frames_host: host int32 [B]; cond, y0, out: dev f32 [sum(frames), mel], the utterances one after the other; text dev int32
 * [B, nt] (-1 padded); lens dev int32 [B] (prompt frames); no trajectory, no hipGraph (shapes rarely recur); DiT backbone only.
 * Plan capacity: sum(frames_i + 16, each rounded up to 16) <= max_batch * max_seq, every frames_i and nt <= max_seq, B <= max_batch.  * Plan option "ragged_graph" = 1 (round 4): the call replays a hipGraph captured for this exact list of frame counts (+ nt, steps, method, CFG)
 * -- batch inference over fixed length buckets (eval/prompts.py) meets the same bucket shapes again; 0 (default): eager launches. */
F5_API int f5_sample_ragged(f5_plan_t p, int B, const int32_t* frames_host, const float* cond, const int32_t* text, int nt, const int32_t* lens,
                     const float* y0, const float* tgrid_host, int steps, float cfg_strength, int ode_method, float* out, f5_stream_t stream);

/* TextEmbedding.forward (dit.py:49-79): text ids [B, nt] (-1 padded) -> dev f32 [B, N, text_dim] */
F5_API int f5_text_embed(f5_plan_t p, int B, int N, const int32_t* text, int nt, int drop_text, float* out, f5_stream_t stream);

/* DiT.forward (dit.py:185-233) for one branch:
 *   x, cond dev f32 [B, N, mel]; text_embed dev f32 [B, N, text_dim] (from f5_text_embed; the Python module owns the
 *   cond/uncond cache of dit.py:202-210); time dev f32 [B]; mask dev u8 [B, N] or NULL; out dev f32 [B, N, mel]. */
F5_API int f5_dit_forward(f5_plan_t p, int B, int N, const float* x, const float* cond, const float* text_embed,
                   const float* time, int drop_audio_cond, const uint8_t* mask, float* out, f5_stream_t stream);

/* MMDiT.forward (mmdit.py:146-190) for one branch: as f5_dit_forward, with the text stream at its own length --
 *   text_embed dev f32 [B, nt, dim] (f5_text_embed on an F5_BACKBONE_MMDIT plan writes [B, nt, dim] and ignores N), nt <= the plan's max_seq. */
F5_API int f5_mmdit_forward(f5_plan_t p, int B, int N, int nt, const float* x, const float* cond, const float* text_embed,
                     const float* time, int drop_audio_cond, const uint8_t* mask, float* out, f5_stream_t stream);

/* debug/parity taps: after the next f5_dit_forward / f5_sample evaluation, copy the named internal stage
 * (converted to f32, row-major [rows, cols]) into `dst` (dev f32).  Names: "t_emb", "input_embed",
 * "blk<i>.n1", "blk<i>.attn", "blk<i>.out", "final_norm".  Pass dst = NULL to clear all taps. */
F5_API int f5_plan_set_tap(f5_plan_t p, const char* stage, float* dst);
/* in-situ timing of the dominant kernel: between begin and end, every fused-QKV GEMM launch of (eager) f5_sample /
 * f5_dit_forward calls on this plan is bracketed by a HIP event pair on the caller's stream; end synchronises the stream and
 * returns the mean device time per launch.  Used by bench.py for roofline.achieved. */
F5_API int f5_plan_timing_begin(f5_plan_t p, int max_launches);
F5_API int f5_plan_timing_end(f5_plan_t p, float* avg_ms, int* launches, f5_stream_t stream);
/* the same measurement for every kernel of a DiT evaluation: after f5_plan_timing_end, mean device time per launch of call site
 * `site`.  The event pairs live in a bounded ring of 512 pairs created once per plan (round 4; `max_launches` is accepted and ignored): when
 * the ring is full its older half is folded into the per-site sums, so any number of launches is covered with 1 024 live events. */
#define F5_SITE_QKV 0   /* fused QKV projection + RoPE            modules.py:452-461 */
#define F5_SITE_ATTN 1  /* scaled-dot-product attention           modules.py:483-497 */
#define F5_SITE_OUT 2   /* attention out-projection x gate_msa    modules.py:499-501,635 */
#define F5_SITE_FF1 3   /* FeedForward first linear + GELU(tanh)  modules.py:258-264 */
#define F5_SITE_FF2 4   /* FeedForward second linear x gate_mlp   modules.py:639 */
#define F5_SITE_LN1 5   /* residual add + AdaLN before attention  modules.py:301-317,632 */
#define F5_SITE_LN2 6   /* residual add + AdaLN before the FF     modules.py:637-638 */
#define F5_SITE_CONV 7  /* grouped Conv1d(k=31) + Mish (x2)       modules.py:167-190 */
#define F5_SITE_INPUT 8 /* input projection of the noisy mel      dit.py:88-96 */
#define F5_SITE_COUNT 9
F5_API int f5_plan_timing_site(f5_plan_t p, int site, float* avg_ms, int* launches);
/* kernel selection for A/B runs: key "gemm_kernel" / "attn_kernel"; value 0 = reference tile kernels, 1 or -1 = tuned
 * kernels wherever they support the problem (default).  Drops any captured graphs.
 * Residual-stream storage of the bf16 mode: key "residual_f16": 1 = fp16 (saturating), 0 = fp32, -1 = the process-wide
 * knob (default: fp16); key "residual_guard" (default 1): f5_sample reads, after the ODE loop, the flag word the LayerNorm passes raise when
 * an element of the fp16 stream reaches +-65504 or is NaN, repeats the loop with fp32 storage and keeps fp32 storage for this plan
 * (f5_sample then synchronises the stream once per call; 0 = no read, fully asynchronous, clipping goes unnoticed; 2 = the read is
 * deferred to f5_sample_finish). */
F5_API int f5_plan_set_option(f5_plan_t p, const char* key, int value);
/* reads an option back; besides the keys above: "residual_fallbacks" = f5_sample calls of this plan that were repeated with fp32 residual
 * storage because the range guard fired ("residual_f16" then reads 0). */
F5_API int f5_plan_get_option(f5_plan_t p, const char* key, int* value);

/* ------------------------------------------------------------------ duration predictor
 * Replaces DurationPredictor.forward / .phoneme_forward (reference model/duration_predictor.py:28-46 / :48-68) as called from
 * F5TTSWrapper.calculate_duration_with_predictor (infer/f5tts_wrapper.py:381-406).  All tensors f32 on the device, PyTorch layouts:
 *   text_embed [vocab_rows, in_channels]; conv1_w [filter, in_channels, k]; conv2_w [filter, filter, k]; norm*_w/b [filter];
 *   proj_w [filter] (the [1, filter, 1] Conv1d weight); proj_b [1]; cond_w / cond_b of the optional speaker conditioning (f5_duration_predict_g). */
typedef struct f5_duration_weights {
    const float *text_embed, *conv1_w, *conv1_b, *norm1_w, *norm1_b, *conv2_w, *conv2_b, *norm2_w, *norm2_b, *proj_w, *proj_b;
    int32_t vocab_rows, in_channels, filter_channels, kernel_size;
    const float *cond_w, *cond_b; /* speaker conditioning Conv1d(gin -> in_channels, 1): [in_channels, gin] and [in_channels]; NULL when gin_channels = 0 */
    int32_t gin_channels;
} f5_duration_weights;
/* tokens i32 [batch, nt] (pad -1), add_one = 1 for forward() (ids shifted so that 0 is the filler), 0 for phoneme_forward();
 * mask i32 [batch, nt] (1 = real token); scratch f32 [2 * batch * filter_channels * nt]; out f32 [batch, nt] = log-durations * mask. */
F5_API int f5_duration_predict(const f5_duration_weights* w, int batch, int nt, const int32_t* tokens, int add_one, const int32_t* mask,
                        float* scratch, float* out, f5_stream_t stream);
/* the same with the speaker conditioning of duration_predictor.py:33-35: x = embedding + cond(g), g dev f32 [batch, gin_channels, g_nt] with
 * g_nt = 1 (one vector per utterance, broadcast over the tokens) or nt; scratch f32 [2 * batch * filter_channels * nt + batch * in_channels * g_nt];
 * g = NULL is f5_duration_predict. */
F5_API int f5_duration_predict_g(const f5_duration_weights* w, int batch, int nt, const int32_t* tokens, int add_one, const int32_t* mask,
                          const float* g, int g_nt, float* scratch, float* out, f5_stream_t stream);

/* ------------------------------------------------------------------ per-op entry points (parity tests, micro-benchmarks) */
/* out[M,N] = A[M,K] @ W[N,K]^T + bias ; A/W/out f32 dev; computed through the precision's GEMM kernel
 * (bf16: inputs rounded to bf16 on device, MFMA, f32 accumulate).  act: 0 none, 1 gelu-tanh, 2 gelu-erf, 3 mish.
 * kernel: 0 = reference tile kernel, 1 = tuned 256x256 LDS-DMA kernel (bf16 only; shapes must be tile multiples). */
F5_API int f5_op_linear(int precision, int kernel, int M, int N, int K, const float* A, const float* W, const float* bias, int act,
                 float* out, f5_stream_t stream);
/* One DiT block linear with the fused store epilogue the sampler uses for it (bf16 path; output converted back to f32):
 *   epi 0: out = act(A W^T + b)                                      FeedForward first linear, reference model/modules.py:258-264
 *   epi 5: out = gate[n] * act(A W^T + b), rows with rowmask[m]==0 -> 0  attention to_out / FF second linear times the AdaLN gate,
 *                                                                    modules.py:499-501,635,639 (gate f32 [N] or NULL, rowmask u8 [M] or NULL)
 *   epi 2: out = out + gate[n] * (A W^T + b) IN PLACE on the fp16 residual stream (`out` is read, rounded to fp16, updated, returned as f32;
 *          rows with rowmask[m]==0 keep their value; M * N % 4 == 0): how the attention to_out and the FF second linear update the stream
 *          in the bf16 production mode, modules.py:635,639
 *   epi 4: out = rope(A W^T + b): fused QKV projection (N = 3*inner), x_transformers rotary on adjacent pairs of the q and k columns
 *          of the first rope_heads heads, modules.py:452-461; rope f32 [seq][32][2] (cos, sin), token position = m % seq
 * kernel: 0 = reference tile kernel, 1 = tuned kernels.  All pointers are device pointers. */
F5_API int f5_op_linear_fused(int kernel, int epi, int M, int N, int K, const float* A, const float* W, const float* bias, int act,
                       const float* gate, const uint8_t* rowmask, const float* rope, int rope_heads, int seq, float* out,
                       f5_stream_t stream);
/* The LayerNorm fold of one call site end to end (round 4; parity tests).  x [M, D]: the fp16 residual stream, handed over and returned as f32.
 * (1) x += gate * (A . Wo^T + bo) in place by the tuned GEMM's EPI_RESID epilogue, which also writes partial row sums of the values it stores
 * (pivot = column 0 of `pivot` [M][2], or NULL); (2) stats [M][2] = (mean, rstd) of every row, eps 1e-6 (modules.py:308,624); (3) W' = fp16(W (1 +
 * scale)), c1 = rowsum W', c2 = bias + W . shift; (4) out [M, N] = epilogue(rstd (x . W'^T - mean c1) + c2) on v_mfma_f32_16x16x32_f16 --
 * epi 0: store with `act` (FF1 + GELU: modules.py:258-264, 637-638), epi 4: RoPE on the q | k columns (fused QKV: modules.py:301-317, 452-480). */
F5_API int f5_op_ln_fold(int epi, int M, int D, int N, int Kb, float* x, const float* A, const float* Wo, const float* bo, const float* gate,
                         const float* pivot, const float* W, const float* bias, const float* scale, const float* shift, int act, const float* rope,
                         int rope_heads, int seq, float* stats, float* out, f5_stream_t stream);
/* LayerNorm(eps 1e-6, no affine) * (1 + scale) + shift ; x f32 [rows, dim]; scale/shift f32 [dim] */
F5_API int f5_op_layernorm_modulate(int rows, int dim, const float* x, const float* scale, const float* shift, float* out,
                             f5_stream_t stream);
/* multi-head attention on packed projections: qkv f32 [B, N, 3, H, 64] (already RoPE'd), mask u8 [B,N] or NULL
 * -> out f32 [B, N, H*64].  kernel: 0 = reference kernel, 1 = tuned flash kernel (bf16). */
F5_API int f5_op_attention(int precision, int kernel, int B, int N, int H, const float* qkv, const uint8_t* mask, float* out,
                    f5_stream_t stream);
/* ConvPositionEmbedding (modules.py:167-190): x f32 [B, N, dim] -> mish(conv(mish(conv(x)))) ; weights f32
 * [dim, dim/16, 31] + bias [dim] (two layers) */
F5_API int f5_op_conv_pos_embed(int precision, int B, int N, int dim, const float* x, const float* w0, const float* b0,
                         const float* w1, const float* b1, float* out, f5_stream_t stream);

/* in-process kernel timing for the roofline leg of bench.py: `iters` back-to-back launches of ONE kernel bracketed by HIP
 * events on `stream`, random bf16 operands; *ms_avg = mean device time per launch.
 * site: 0 fused QKV projection + RoPE, 1 FF1 + GELU-tanh, 2 FF2 + gated residual, 3 attention out-projection + gated residual. */
F5_API int f5_bench_gemm_site(int kernel, int site, int rows, int seq, int dim, int heads, int ff_inner, int iters, float* ms_avg,
                       f5_stream_t stream);
F5_API int f5_bench_attention(int kernel, int B, int N, int H, int iters, float* ms_avg, f5_stream_t stream);
/* Sustained rate of a register-resident v_mfma_f32_16x16x32_bf16 stream on every CU (no memory traffic): random_operands = 0 zeros
 * (clock-limited), 1 pseudo-random bf16 values (power-limited: what a dense bf16 GEMM can approach on this device). */
F5_API int f5_bench_mfma_rate(int random_operands, float* tflops, f5_stream_t stream);
/* diagnostic build of the persistent attention kernel: while dev_buf (dev u64 [workgroups * 64]) is non-NULL every launch writes shader-clock
 * stamps of wave 0 (first 8 items of each workgroup: item start, Q prefetch issued, key tiles 0 / 1 / 2 / last done, epilogue done) */
F5_API int f5_debug_attn_stamps(void* dev_buf);
/* the clock the chip holds under the tuned GEMM: while dev_buf (dev u64 [workgroups * 4]) is non-NULL every workgroup writes (s_memtime,
 * s_memrealtime) at its start and end; clock = d(s_memtime) / d(s_memrealtime) x 100 MHz */
F5_API int f5_debug_gemm_clock(void* dev_buf);
/* process-wide kernel tuning knobs for A/B measurements ("gemm_variant": main-loop schedule of the tuned GEMM) */
F5_API int f5_tuning_set(const char* key, int value);

/* ------------------------------------------------------------------ Vocos vocoder (plug point B) */
typedef struct f5_vocos_config {
    int32_t n_mels;    /* 100 */
    int32_t dim;       /* 512 */
    int32_t inter_dim; /* 1536 */
    int32_t layers;    /* 8 */
    int32_t n_fft;     /* 1024 */
    int32_t hop;       /* 256 */
} f5_vocos_config;
F5_API int f5_vocoder_create(const f5_vocos_config* cfg, f5_vocoder_t* out);
F5_API int f5_vocoder_set_tensor(f5_vocoder_t v, const char* name, const float* host_data, const int64_t* shape, int ndim);
F5_API int f5_vocoder_has_tensor(f5_vocoder_t v, const char* name, int64_t* numel);
F5_API int f5_vocoder_finalize(f5_vocoder_t v);
F5_API int f5_vocoder_destroy(f5_vocoder_t v);
/* Vocos.decode: mel dev f32 [B, n_mels, T] -> wave dev f32 [B, (T-1)*hop] */
F5_API int f5_vocoder_decode(f5_vocoder_t v, int B, int T, const float* mel, float* wave, f5_stream_t stream);
/* ISTFT head alone (for the roofline measurement): spec dev f32 [B, T, n_fft+2] (head.out activations:
 * log-magnitude | phase) -> wave dev f32 [B, (T-1)*hop] */
F5_API int f5_vocoder_istft_head(f5_vocoder_t v, int B, int T, const float* head_out, float* wave, f5_stream_t stream);

/* ------------------------------------------------------------------ BigVGAN-v2 generator (round 4; plug point B, PARITY UNPINNED)
 * Replaces `third_party.BigVGAN.bigvgan.BigVGAN` as the reference uses it: infer/utils_infer.py:125-138 (from_pretrained + remove_weight_norm) and the
 * call `vocoder(mel)` of infer/f5tts_wrapper.py:526 / eval/eval_infer_batch.py:189.  That checkout is absent from the reference tree: the generator is
 * restated from the published BigVGAN-v2 source (conv_pre, per stage ConvTranspose1d + the mean of three AMPBlock1 with anti-aliased SnakeBeta
 * activations, conv_post, clamp / tanh); tensor names are the checkpoint's with weight norm REMOVED (`<module>.weight`, `.bias`, `....act.alpha`,
 * `....act.beta`), plus the optional 12-tap buffers `aa_up_filter` / `aa_down_filter` (default: the Kaiser-windowed sinc of the published filter). */
typedef struct f5_bigvgan_s* f5_bigvgan_t;
typedef struct f5_bigvgan_config {
    int32_t num_mels;                 /* 100 */
    int32_t upsample_initial_channel; /* 1536 */
    int32_t num_upsamples;            /* 6 (<= 8) */
    int32_t upsample_rates[8];        /* 4 4 2 2 2 2 */
    int32_t upsample_kernel_sizes[8]; /* 8 8 4 4 4 4 (a multiple of the rate, k - u even) */
    int32_t num_kernels;              /* 3 (<= 4) AMP blocks per stage */
    int32_t resblock_kernel_sizes[4]; /* 3 7 11 */
    int32_t resblock_dilations[4][3]; /* 1 3 5 each */
    int32_t snake_logscale;           /* 1: alpha / beta are stored as logarithms */
    int32_t use_tanh_at_final;        /* 0: clamp(-1, 1) */
    int32_t use_bias_at_final;        /* 0: conv_post has no bias */
} f5_bigvgan_config;
F5_API int f5_bigvgan_create(const f5_bigvgan_config* cfg, f5_bigvgan_t* out);
F5_API int f5_bigvgan_set_tensor(f5_bigvgan_t v, const char* name, const float* host_data, const int64_t* shape, int ndim);
F5_API int f5_bigvgan_has_tensor(f5_bigvgan_t v, const char* name, int64_t* numel);
F5_API int f5_bigvgan_finalize(f5_bigvgan_t v);
F5_API int f5_bigvgan_destroy(f5_bigvgan_t v);
/* mel dev f32 [B][num_mels][T] -> wave dev f32 [B][T * prod(upsample_rates)]  (BigVGAN.forward, the [B, 1, samples] result without its unit axis) */
F5_API int f5_bigvgan_forward(f5_bigvgan_t v, int B, int T, const float* mel, float* wave, f5_stream_t stream);

/* ------------------------------------------------------------------ reference-audio front-end on the device
 * Replaces the two torchaudio transforms of the path:
 *   f5_frontend_mel       MelSpec / get_vocos_mel_spectrogram, reference model/modules.py:75-143 (as called from cfm.py:103-105):
 *                         wave dev f32 [B, nw] -> log-mel dev f32 [B, n_mels, nw / hop + 1]
 *   f5_frontend_resample  torchaudio.transforms.Resample(sr, target) of F5TTSWrapper.preprocess_reference (infer/f5tts_wrapper.py:338-341) and
 *                         infer_batch_process (infer/utils_infer.py:443-445): wave dev f32 [B, n] -> dev f32 [B, ceil(new * n / orig)] */
typedef struct f5_frontend_s* f5_frontend_t;
typedef struct f5_mel_config {
    int32_t n_fft;       /* 1024 */
    int32_t hop;         /* 256 */
    int32_t win;         /* 1024 (<= n_fft; centred) */
    int32_t n_mels;      /* 100 */
    int32_t sample_rate; /* 24000: the filterbank spans 0 .. sample_rate / 2 */
    int32_t mel_type;    /* F5_MEL_VOCOS (0): torchaudio MelSpectrogram, HTK scale, center = True -> nw / hop + 1 frames (modules.py:75-101);
                            F5_MEL_BIGVGAN (1, round 4): get_bigvgan_mel_spectrogram (modules.py:29-72): reflect padding of (n_fft - hop) / 2,
                            center = False -> (nw + 2 pad - n_fft) / hop + 1 frames, sqrt(re^2 + im^2 + 1e-9), librosa's Slaney-scale
                            area-normalised filterbank */
} f5_mel_config;
#define F5_MEL_VOCOS 0
#define F5_MEL_BIGVGAN 1
F5_API int f5_frontend_create(const f5_mel_config* cfg, f5_frontend_t* out);
F5_API int f5_frontend_destroy(f5_frontend_t h);
F5_API int f5_frontend_mel(f5_frontend_t h, int B, int nw, const float* wave, float* mel, f5_stream_t stream);
F5_API int f5_frontend_resample(f5_frontend_t h, int B, int n, int orig_freq, int new_freq, const float* wave, float* out, f5_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* F5HIP_H */
