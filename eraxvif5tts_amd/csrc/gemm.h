// gemm.h -- the dense contraction engine of libf5hip:  out[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Both operands are K-contiguous (activations row-major, weights in PyTorch's [out, in] layout), which
// is exactly the MFMA fragment order on CDNA4 (8 consecutive k per lane).  The MFMA is issued "swapped"
// (weight rows on the MFMA row index, tokens on the lane/column index) so that every lane ends up with
// 4 consecutive output features of one token: RoPE pairs are lane-local and stores are 8/16-byte vectors.
#pragma once
#include "common.h"

enum GemmMode {
    GEMM_DENSE = 0,  // plain A[M,K]
    GEMM_CONV31 = 1  // grouped Conv1d(k=31, groups=dim/64, pad=15) as an implicit GEMM: K = 31 taps x 64 channels
};

enum GemmEpi {
    EPI_STORE_T = 0,    // out_t[m][n]  = act(acc + bias[n])                                (activation dtype)
    EPI_STORE_F32 = 1,  // out_f[m][n]  = act(acc + bias[n])                                (f32)
    EPI_RESID = 2,      // out_f[m][n] += gate[b(m)][n] * act(acc + bias[n]), skipped where rowmask[m]==0 (out_f holds fp16 when add2_f16:
                        // the residual stream of the bf16 mode, updated in place by the attention out-projection and the second FF linear)
    EPI_ADD2 = 3,       // v = acc + bias[n] + addend[m][n];  out_t[m][n] = v;  out_f[m][n] = v
    EPI_ROPE_T = 4,     // out_t = rope(acc + bias) on the q/k columns of the first pe heads (fused QKV projection)
    EPI_GATE_T = 5      // out_t[m][n] = gate[b(m)][n] * act(acc + bias[n]), 0 where rowmask[m]==0  (store-only residual branch;
                        // the fp32 residual add itself is fused into the next LayerNorm pass)
};

// LayerNorm fold (round 4; bf16 production mode of the DiT): the QKV / FF1 projections read the fp16 residual stream ITSELF as their A operand
// against per-evaluation-time weights W' = fp16(W . diag(1 + scale)) on v_mfma_f32_16x16x32_f16 and apply the rest of
//   LN(x) (1 + scale) + shift  ->  Linear      =      rstd[m] (acc[m][n] - mean[m] c1[n]) + c2[n],   c1 = rowsum W',  c2 = b + W . shift
// in their epilogue (then RoPE / GELU as before); the row statistics come from the in-place residual epilogues of the launch in front
// (out-projection / FF2: `stats_out` partial sums per 64-feature wave tile, folded by stats_finalize_kernel).  modules.py:301-317,610-641.
struct GemmParams {
    const void* A;  // [M(or a_row_mod), K] activation dtype
    const void* W;  // DENSE: [N, K];  CONV31: [31][N][conv_win] (tap-major, zero outside the row's own group)
    int lda, ldw;
    int M, N, K;
    int a_row_mod;  // > 0: A row = m % a_row_mod (both CFG branches read the same noisy mel rows)
    const float* bias;  // [N] or null
    int act;            // Act
    void* out_t;        // activation dtype output
    int ldo;
    float* out_f;  // f32 output / residual stream (read-modify-write for EPI_RESID)
    int ldof;
    const float* addend;  // EPI_ADD2
    int ldadd;
    int add2_f16;  // EPI_ADD2 / EPI_RESID: `addend` and `out_f` hold fp16 elements (same leading dimensions, in elements): the bf16 mode's residual stream
    const float* gate;  // EPI_RESID: gate[b * gate_bstride + n] or null (=1)
    int gate_bstride;
    int rows_per_batch;      // sequence length N_seq: b(m) = m / rows_per_batch, position = m % rows_per_batch
    int row0;                // tuned kernel: global row index of this launch's row 0 (a launch split into a whole-tile part and a ragged tail: the
                             // tail's positions / batch indices continue where the first part ended; set by the launcher)
    const uint8_t* rowmask;  // [M] or null
    const uint8_t* rowbits;  // optional transposed form of rowmask for the tuned kernel's epilogue (launch_rowbits): byte [m/128][m%16], bit (m%128)/16
    const float* rope;       // [N_seq][32][2] (cos, sin)
    int rope_inner;          // heads * 64
    int rope_heads;          // heads that receive RoPE
    // CONV31: channels per group (dim/16) and the padded input-channel window one 64-channel output tile reads
    int conv_cg, conv_win;
    int site;        // block call site for the per-site tile-walk knobs: 0 = none, 1 = qkv, 2 = out-projection, 3 = FF1, 4 = FF2 (set by dit_eval; the
                     // launcher no longer guesses it from N / K ratios, which misread ff_mult = 4 backbones and the long-skip linear)
    // LayerNorm fold, consumer side (tuned kernel only; template flag LNF): A and W hold fp16, the accumulator starts from 0, `bias` is unused
    const float* lnf_stats;  // [M][2] = (mean, rstd) of every token row of A, or null (no fold)
    const float* lnf_c1;     // [N] row sums of W'
    const float* lnf_c2;     // [N] bias + W . shift
    // ... or, for every tile but the 256-wide one, the statistics are taken from the producer's partial sums INSIDE this kernel (no statistics
    // launch in front: small launches are bound by launch count): lnf_partial[c * lnf_partial_ld + m] = float2 partial of feature tile c,
    // lnf_ncols of them, added in the order c = 0, 1, ... exactly as stats_finalize_kernel adds them (lnf_stats_math.h: same bits); lnf_pivot
    // [M][2] (column 0) or null; the workgroups of feature tile 0 also store (mean, rstd) to lnf_stats_out [M][2] (the next producer's pivots)
    // and carry the fp16 range guard (lnf_sat / lnf_sat_tag).  gemm_fast_lnf_inkernel() tells the caller which form a launch can take.
    const float* lnf_partial;
    int lnf_partial_ld, lnf_ncols;
    const float* lnf_pivot;
    float* lnf_stats_out;
    unsigned* lnf_sat;
    int lnf_sat_tag;
    // weights of the launches BEHIND this one, touched one dword per 128-byte line by the first threads of the grid (small launches are bound
    // by the latency of weights that every block reads from HBM again; until round 4 the LayerNorm passes did this)
    const void* pf_p[2];
    unsigned pf_n[2];
    // LayerNorm fold, producer side (EPI_RESID on the fp16 stream, tuned kernel only): per token row and 64-feature wave tile the partial sums
    // (sum (h - pivot), sum (h - pivot)^2) over the fp16-ROUNDED values just stored: stats_out[(n / 64) * stats_ld + m] as float2, or null
    float* stats_out;
    int stats_ld;                // rows per 64-feature plane of stats_out
    const float* stats_pivot;    // [M][2]: element 0 of row m is the pivot (the row's previous mean), or null (pivot 0)
    // ... and, on the non-persistent schedules, the statistics are FINISHED inside the same launch: the workgroup that completes a block of
    // token rows last (fin_counter[m0 / BM], one ticket per feature tile) adds the row's partial sums in stats_finalize_kernel's order and stores
    // (mean, rstd) to fin_stats [M][2] -- no statistics launch behind the GEMM.  Hand-off as /opt/skills/guides/cdna_hip_programming.md section 6,
    // guideline 16 prescribes for a fan-in: write-through (agent-scope) partial stores, every wave drained, workgroup barrier, one relaxed agent-scope
    // ticket; the last arriver acquires once and reads the partials with agent-scope loads; it also resets the ticket word.  Range guard: lnf_sat / lnf_sat_tag.
    unsigned* fin_counter;
    float* fin_stats;
    int tile_group;  // tuned kernel: token tiles per L2 patch (set by the launcher)
    int tile_reverse;  // tuned kernel: walk the tiles in the opposite order (producer / consumer cache experiments)
    int lean_epi;    // tuned kernel: whole tiles take the lean epilogue (set by the launcher; 0 = always the generic one)
    unsigned long long* clk;  // diagnostic (f5_debug_gemm_clock): [workgroups][4] = (s_memtime, s_memrealtime) at workgroup start and end, or null
};

// kernel_kind: 0 = reference tile kernel (any shape), 1 = tuned 256x256 LDS-DMA bf16 kernel
int launch_gemm(const GemmParams& p, int precision, int mode, int epi, int kernel_kind, hipStream_t stream);
// true when the tuned kernel can run this problem (bf16, tile-multiple shapes, supported epilogue)
bool gemm_fast_supported(const GemmParams& p, int precision, int mode, int epi);
// the token x feature tile launch_gemm_fast would pick for this dense problem (by occupancy and the tuning knobs)
void gemm_fast_tile(const GemmParams& p, int* bm, int* bn);
// LayerNorm fold: true when a launch of this shape can take its row statistics from the partial sums inside the kernel (every tile but 256 x 256)
bool gemm_fast_lnf_inkernel(const GemmParams& p);
// LayerNorm fold: true when an in-place residual launch with these parameters can finish its row statistics inside the launch
// (GemmParams::fin_counter): every schedule but the PERSISTENT one, which leaves them to stats_finalize_kernel
bool gemm_fast_resid_finishes(const GemmParams& p);
// true when a dense launch with these parameters runs on the one-wave-per-SIMD kernel (gemm_w4.hip): whole tiles, at least one per CU, lean operand forms
bool gemm_w4_ok(const GemmParams& p, int mode, int epi);
// LayerNorm fold: a folded projection of this shape finishes the row statistics inside the one-wave-per-SIMD kernel (128-row tiles)
int gemm_w4_lnf_inkernel(int M, int N, int K);  // 0, or the tile height (128) that finishes them
// dedicated kernel for the dim-1024 grouped Conv1d(k = 31) of ConvPositionEmbedding (conv31.hip); GemmParams as for GEMM_CONV31
bool conv31_supported(const GemmParams& p, int precision, int epi);
int launch_conv31(const GemmParams& p, hipStream_t stream);
