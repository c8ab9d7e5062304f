// kernels.h -- launchers of the non-GEMM kernels (all enqueue on `stream`, never synchronise).
#pragma once
#include "common.h"

// ---- attention.hip
// qkv: [B*N, 3*H*64] (q | k | v, token-major), mask u8 [B, N] or null, out [B*N, H*64]; activation dtype by precision.
// kernel_kind 0: reference kernel (fp32 VALU math, any N); 1: tuned bf16 flash kernel (MFMA, in-register softmax).
// bstride: token rows between consecutive batch items (0 = N; larger when the items sit in a longer concatenation, mask must then be null)
int launch_attention(int precision, int kernel_kind, int B, int N, int H, const void* qkv, int ldq, const uint8_t* mask, void* out,
                     int ldo, hipStream_t stream, int bstride = 0);
bool attention_fast_supported(int precision, int N, int H);
// utterances of a ragged batch (f5_sample_ragged): row offset and length of each inside one half of the concatenation
struct AttnSegs {
    int nbr = 1, cnt = 0;  // batch items per utterance (CFG branches), utterances in this table
    int off[12] = {0}, n[12] = {0};
};
// attention over `cnt` utterances of different lengths, each the computation of launch_attention(precision, kind_u, nbr, n[u], ...) on its
// own rows (bstride rows between the branches): utterances whose own launch would take the pipelined kernel share launches, the rest get theirs
int launch_attention_ragged(int precision, int attn_kernel_opt, const AttnSegs& segs, int H, const void* qkv, int ldq, void* out, int ldo, hipStream_t stream,
                            int bstride);

// ---- elementwise.hip
// out[r][c] = LN(x[r][:])[c] * (add_one + mul[b(r)][c]) + add[b(r)][c]; b(r) = r / rows_per_batch; eps 1e-6
int launch_layernorm(int precision_out /*F5_PREC_* of `out`*/, const float* x, int ldx, int rows, int dim, const float* mul,
                     const float* add, int mod_bstride, int rows_per_batch, int add_one, void* out, int ldo, hipStream_t stream);
// same with the fp32 residual add fused in: x[r] += y[r] (y in the activation dtype, may be null) is written back first
int launch_layernorm_add(int precision_out, float* x, int ldx, int rows, int dim, const void* y, int ldy, const float* mul, const float* add,
                         int mod_bstride, int rows_per_batch, int add_one, void* out, int ldo, hipStream_t stream);
// ymode 1: x += y (written back); 2: normalise x + y, x untouched; 3: x = (x + y) + y2 (written back)
int launch_layernorm_add2(int precision_out, float* x, int ldx, int rows, int dim, const void* y, int ldy, const void* y2, int ymode,
                          const float* mul, const float* add, int mod_bstride, int rows_per_batch, int add_one, void* out, int ldo,
                          hipStream_t stream);
// byte ranges a LayerNorm pass pulls towards the caches on its way out (weights of the GEMMs that follow: at small batches every block's
// weights come from HBM again, and the GEMMs there are bound by operand latency)
struct PrefetchSet {
    const void* p[4];
    unsigned n[4];  // bytes (multiples of 128)
};
// saturating, n % 4 == 0; `sat` (device word or null) is set to 1 when an element is at or beyond fp16's range, or NaN
int launch_f32_to_f16(const float* src, void* dst, size_t n, hipStream_t stream, unsigned* sat = nullptr);
int launch_f16_to_f32(const void* src, float* dst, size_t n, hipStream_t stream);  // n % 4 == 0
// residual stream read from xin (fp32, or fp16 when xin_f16) and, ymode 1 / 3, written to xout (fp32 / fp16): elementwise.hip
int launch_layernorm_res(int precision_out, const void* xin, int xin_f16, void* xout, int xout_f16, int ldx, int rows, int dim, const void* y, int ldy,
                         const void* y2, int ymode, const float* mul, const float* add, int mod_bstride, int rows_per_batch, int add_one, void* out,
                         int ldo, hipStream_t stream, const PrefetchSet* prefetch = nullptr, unsigned* sat = nullptr /* range guard of the fp16 stream */,
                         int sat_tag = 0 /* diagnostics: pass kind | DiT block << 4 */);
// ---- UNetT (reference model/backbones/unett.py)
// x_transformers.RMSNorm: out[r] = x[r] / max(||x[r]||, 1e-12) * sqrt(dim) * g    (x f32; out in the activation dtype of precision_out)
int launch_rmsnorm(int precision_out, const float* x, int ldx, int rows, int dim, const float* g, void* out, int ldo, hipStream_t stream);
// dst [B * (N + 1), D] f32: row b * (N + 1) = temb[b * temb_bstride ..], rows after it = h[b] + branch[b] (branch in the activation dtype)
int launch_pack_time_token(int precision, const float* h, const void* branch, const float* temb, int temb_bstride, int B, int N, int D, float* dst,
                           hipStream_t stream);
int launch_pad_mask(const uint8_t* mask, int B, int N, uint8_t* dst, hipStream_t stream);                      // [B, N] -> [B, N + 1], leading 1
int launch_drop_time_token(const float* src, int B, int N, int cols, float* dst, hipStream_t stream);          // [B * (N + 1), cols] -> [B * N, cols]
int launch_add_f32(float* x, const float* y, size_t n, hipStream_t stream);                                    // x += y, n % 4 == 0
// qk_norm "rms_norm" (modules.py:275-294) + RoPE in place on stored q|k|v rows [rows, ldq]; wq / wk f32 [64]; rope [pos][32][cos, sin]
int launch_qknorm_rope(int precision, void* qkv, int ldq, int rows, int inner, int heads, int rope_heads, const float* wq, const float* wk,
                       const float* rope, int rows_per_batch, hipStream_t stream);
// rows whose flag byte is 1 are zeroed (row_bytes a multiple of 16): the gaps between the utterances of a ragged sample()
int launch_zero_rows(void* x, size_t row_bytes, int rows, const uint8_t* flags, hipStream_t stream);
// ---- MMDiT (reference model/backbones/mmdit.py)
// nb byte segments src + b * src_bstride -> dst + b * dst_bstride (everything a multiple of 16 bytes)
int launch_copy_segments(const void* src, size_t src_bstride_bytes, void* dst, size_t dst_bstride_bytes, size_t seg_bytes, int nb, hipStream_t stream);
int launch_joint_mask(const uint8_t* mask, int B, int N, int nt, uint8_t* dst, hipStream_t stream);  // [B, N] -> [B, N + nt], trailing 1s
// depthwise Conv1d(k=7, pad=3) along the sequence (+bias) then LayerNorm(eps 1e-6, affine) -> activation dtype
// x f32 [B*N, C]; wt f32 [7][C] (tap-major); out [B*N, C]
int launch_dwconv7_ln(int precision_out, const float* x, int B, int N, int C, const float* wt, const float* cbias, const float* ln_w,
                      const float* ln_b, void* out, int ldo, hipStream_t stream);
// GRN (modules.py:225-234) in place on h [B*N, C] (activation dtype): h = gamma * (h * Nx) + beta + h
int launch_grn(int precision, void* h, int B, int N, int C, const float* gamma, const float* beta, float* scratch /*[B*C + B]*/,
               hipStream_t stream);
// text ids -> embedding rows (+ abs-pos table) ; also writes filler flags [B*N] (1 where the id is the filler 0)
int launch_text_gather(const int32_t* text, int nt, int B, int N, int td, const float* table, const float* pos_table /*or null*/, int pos_rows,
                       int drop_text, float* out, uint8_t* filler, hipStream_t stream);
int launch_mask_rows(float* x, int rows, int cols, const uint8_t* zero_flags, hipStream_t stream);
// sinusoidal timestep embedding (modules.py:149-161): t [n] -> out [n, 256]
int launch_time_sinus(const float* t, int n, float* out, hipStream_t stream);
// small-M fp32 linear: out[r][n] = post( sum_k pre(in[r][k]) * W[n][k] + b[n] ), pre/post in {none, silu}
int launch_gemv_rows(const float* in, int ldi, int rows, const float* W, const float* b, int N, int K, int pre_silu, int post_silu,
                     float* out, int ldo, hipStream_t stream);
// f32 [rows, cols] -> activation dtype [rows, ldo] with zero padding of columns cols..padcols-1
int launch_convert_pad(int precision_out, const float* src, int lds, int rows, int cols, int padcols, void* dst, int ldo, hipStream_t stream);
int launch_convert_back(int precision_in, const void* src, int lds, int rows, int cols, float* dst, int ldd, hipStream_t stream);
// A_base rows: [cond(mel, padded to melp) | text_embed(td)] per branch (see model.cpp)
int launch_pack_base(int precision_out, const float* cond, const int32_t* lens, const float* text_embed, int B, int N, int mel, int melp,
                     int td, int zero_cond, void* dst, int ldd, hipStream_t stream);
// x_out = x_base + coef[0] * (vc + (vc - vu) * cfg)   (cfm.py:173 + torchdiffeq step); vu == null -> x_base + coef * vc
int launch_cfg_step(const float* x_base, const float* vc, const float* vu, int ldv, int rows, int mel, float cfg, const float* coef,
                    float* x_out, float* x_out2 /*or null*/, hipStream_t stream);
// out = frame < lens[b] ? cond : x   (cfm.py:200-202)
int launch_final_where(const float* cond, const float* x, const int32_t* lens, int B, int N, int mel, float* out, hipStream_t stream);
// mask[b][n] = n < durations[b]
int launch_len_mask(const int32_t* durations, int B, int N, uint8_t* mask, hipStream_t stream);
int launch_rowbits(const uint8_t* mask, int rows, uint8_t* bits, hipStream_t stream);
int launch_fill_f32(float* dst, size_t n, float v, hipStream_t stream);
// dst[0..n) = host_vals[0..n), passed through kernel arguments (no async-memcpy from pageable host memory)
int launch_set_floats(float* dst, const float* host_vals, int n, hipStream_t stream);

// ---- vocos.hip
// im2col for Conv1d(k=7, pad=3): mel f32 [B, C, T] -> rows [B*T, Kp] (col = tap*C + c, zero padded to Kp)
int launch_vocos_im2col(int precision_out, const float* mel, int B, int C, int T, void* dst, int Kp, hipStream_t stream);
// head.out activations [B*T, ld] (log-mag | phase) -> spectrum rows [B*T, Kp]: (re_0..re_{F-1}, im_0..im_{F-1}), F = n_fft/2+1
int launch_vocos_spectrum(int precision_out, const float* head, int ldh, int rows, int F, void* dst, int Kp, hipStream_t stream);
// ISTFT head for n_fft = 1024 as an FFT: head [rows, ldh] (log-mag | phase) -> windowed frames [rows, 1024]; wscaled = window / n_fft,
// twiddle = (cos, sin)(2 pi j / 1024)
int launch_vocos_ifft1024(const float* head, int ldh, int rows, const float* wscaled, const float* twiddle, float* frames, hipStream_t stream);
// overlap-add of windowed frames [B*T, n_fft] (hop), divide by the window-square envelope, trim n_fft/2 each side
int launch_vocos_ola(const float* frames, int B, int T, int n_fft, int hop, const float* wsq /*[n_fft]*/, float* wave, hipStream_t stream);

// ---- LayerNorm fold (lnfold.hip; gemm.h)
// per-evaluation-time projection weights W' = fp16(W (1 + scale)) and column constants c1 = rowsum W', c2 = b + W . shift for `evals` times x `depth`
// blocks x R rows (R = 3 * inner + ff; rows < qkv_rows take the attention norm's shift / scale, the others the FF norm's)
int launch_fold_weights(const float* W, const float* bias, const float* mod, int modrow, int evals, int depth, int R, int qkv_rows, int D, void* Wt,
                        float* c1, float* c2, hipStream_t stream);
// partial row sums of an in-place residual GEMM ([ncols][ld] float2 planes) -> stats[row] = (mean, rstd); also the fp16 range guard of the stream
int launch_stats_finalize(const float* partial, int ld, int ncols, int rows, int D, const float* pivot, float* stats, unsigned* sat, int sat_tag,
                          hipStream_t stream, const PrefetchSet* prefetch = nullptr /* weights of the launches behind it (small launches) */);

