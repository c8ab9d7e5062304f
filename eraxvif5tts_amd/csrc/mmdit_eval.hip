// mmdit_eval.hip -- one evaluation of the MMDiT backbone (reference model/backbones/mmdit.py:146-190, model/modules.py:646-707).
#include "model_internal.h"

// one evaluation of the MMDiT backbone (reference model/backbones/mmdit.py:146-190, MMDiTBlock modules.py:646-707, JointAttnProcessor :509-606) over
// `nb` batch rows.  Two fp32 residual streams: frames `xres` [nb * N, D] and text `cres` [nb * nt, D] (restarted from p->c_src at every
// evaluation: unlike DiT's text embedding the text stream passes through the time-conditioned blocks).  Each block projects both streams with
// their own weights (RoPE per stream, positions from 0), gathers q|k|v into the joint [frames | text] sequence of every utterance, runs one
// attention over it, scatters the result back and applies the gated out-projection / FF updates to each stream in place (EPI_RESID).
int mmdit_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* modp, int mod_bstride, const uint8_t* mask, hipStream_t st) {
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

