// unett_eval.hip -- one evaluation of the UNetT backbone (reference model/backbones/unett.py:185-253).
#include "model_internal.h"

// one evaluation of the UNetT backbone (reference model/backbones/unett.py:185-253) over `nb` batch rows; temb = time embedding of batch row b at
// temb + b * temb_bstride (stride 0: one time for all).  Result: p->vout [nb * N, MELP] f32 (the time token's row dropped, :246).
// The stream is fp32 (`xres`, read-modify-write by the fp32 EPI_RESID epilogues); activations in the precision's dtype.
int unett_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* temb, int temb_bstride, const uint8_t* mask, hipStream_t st) {
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

