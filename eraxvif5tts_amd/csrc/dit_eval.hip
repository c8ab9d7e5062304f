// dit_eval.hip -- one evaluation of the DiT backbone (reference model/backbones/dit.py:185-233, model/modules.py:301-336,610-641).
#include "model_internal.h"

// one network evaluation over `nb` batch rows (rows = nb*N) whose noisy mel rows are x[xrows, mel] (xrows divides rows);
// modulation row for batch b is modp + b * mod_bstride.  Result: p->vout [rows, MELP] f32.
int dit_eval(f5_plan_s* p, const float* x, int xrows, int nb, int N, const float* modp, int mod_bstride, const uint8_t* mask,
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
    // Token counts that are not a multiple of the 256-row tile (8 x 1001 frames, every ragged batch): the four block GEMMs run over the rows
    // ROUNDED UP to 256 -- the workspace is padded to that anyway -- so that they stay on the persistent schedule with whole tiles only.  The rows
    // past the last token compute on whatever the padding holds (finite: zero-filled arena, saturating fp16 stores) and nobody reads them:
    // every other kernel works on `rows`, attention on the utterances' own rows.  (Round 4 first split such launches into whole tiles + a tail
    // launch: 8 x 1001 346 against 288 ms per sample(), the ragged 4-chunk batch 210 against 183 ms -- the tail launches are pure latency.)
    const int rows_g = (rmw && g_gemm_pad_rows && rows >= 3584 /* from here on the fused projection takes the 256-wide persistent tile */ && rows % 256 != 0 && (size_t)((rows + 255) / 256 * 256) <= p->rows_cap) ? (rows + 255) / 256 * 256 : rows;
    const FoldTable* ft = p->fold;
    const bool lnf = rmw && g_ln_fold && ft && p->lnf_stats && p->fold_eval >= 0 && p->fold_eval < (int)ft->tv.size() && p->gemm_kernel != 0 &&
                     (p->gemm_kernel == 1 || rows >= 512) && D % 64 == 0 && inner % 64 == 0 && ff % 64 == 0;
    const size_t fR = (size_t)m->fold_R, frow0 = lnf ? ((size_t)p->fold_eval * c.depth) * fR : 0;
    // LayerNorm sites of the folded evaluation, in order: k = 2l is block l's second LayerNorm (statistics: its out-projection), k = 2l - 1 its
    // first (statistics: FF2 of block l - 1).  Site k's (mean, rstd) go to table k & 1; its pivots -- the rows' previous means -- are site k - 1's,
    // in the other table (two tables: with the statistics taken inside the consumer kernel, sibling workgroups must still find the OLD means
    // while feature tile 0's workgroups store the new ones).  A consumer launch on tiles narrower than 256 takes the statistics from the
    // partial sums inside the kernel; the 256-wide tile needs them finalized by stats_finalize_kernel (same bits: lnf_stats_math.h).
    float* const lnfS[2] = {p->lnf_stats, p->lnf_stats2};
    PrefetchSet pf1f_prev{{nullptr, nullptr, nullptr, nullptr}, {0u, 0u, 0u, 0u}};
    bool ink_qkv = false, ink_ff1 = false;
    if (lnf && g_ln_fold_inkernel && p->lnf_stats2) {
        GemmParams t = gp_zero();
        t.M = rows; t.K = D; t.lda = D; t.ldw = D;
        const int tiles_m = (rows + 255) / 256, ncu = f5_cu_count();
        const bool sv = inner % 256 == 0 && tiles_m * (2 * inner / 256) <= ncu && tiles_m * (3 * inner / 256) > ncu && tiles_m * (2 * inner / 256) >= 160;
        t.N = sv ? 2 * inner : 3 * inner;
        ink_qkv = gemm_fast_lnf_inkernel(t);
        if (sv) {
            t.N = inner;
            ink_qkv = ink_qkv && gemm_fast_lnf_inkernel(t);
        }
        t.N = ff;
        ink_ff1 = gemm_fast_lnf_inkernel(t);
    }
    // ... and on the one-wave-per-SIMD kernel's 128-row tiles the consumer finishes the statistics by default (gemm_w4.hip: finish_stats; small
    // batches, where the two statistics launches of a block were 13 of its 102 us).  Only where the 8-wave kernel could take the launch in the
    // same form, should the other kernel refuse it.
    if (lnf && p->lnf_stats2 && !c.qk_norm) {
        GemmParams t = gp_zero();
        t.M = rows_g; t.K = D; t.lda = D; t.ldw = D;
        t.N = 3 * inner;
        if (!ink_qkv && gemm_w4_lnf_inkernel(rows_g, 3 * inner, D) != 0 && gemm_fast_lnf_inkernel(t)) ink_qkv = true;
        t.N = ff;
        if (!ink_ff1 && gemm_w4_lnf_inkernel(rows_g, ff, D) != 0 && gemm_fast_lnf_inkernel(t)) ink_ff1 = true;
    }
    // producer side of site k (out-projection, FF2): the non-persistent schedules (small batches) finish the statistics inside the launch -- the
    // workgroup that completes a block of token rows last turns the partial sums into (mean, rstd), carries the range guard and leaves nothing
    // for a statistics launch to do (gemm.h: fin_counter; same bits as stats_finalize_kernel)
    bool fin_site[2] = {false, false};  // [k & 1]: site k's statistics were finished by its producer
    auto lnf_producer = [&](GemmParams& g, int k, int tag, const PrefetchSet& pf, bool consumer_inkernel) {
        g.stats_out = p->lnf_partial; g.stats_ld = (int)p->rows_cap; g.stats_pivot = k > 0 ? lnfS[(k - 1) & 1] : nullptr;
        const bool fin = g_ln_fold_fin && !consumer_inkernel && p->fin_counter && gemm_fast_resid_finishes(g);
        fin_site[k & 1] = fin;
        if (fin) {
            g.fin_counter = p->fin_counter; g.fin_stats = lnfS[k & 1]; g.lnf_sat = sat; g.lnf_sat_tag = tag;
        }
        if (rows <= g_w_prefetch && r16 && g_w_prefetch && (fin || consumer_inkernel)) {  // no statistics launch behind this one: the GEMM itself touches the weights
            g.pf_p[0] = pf.p[0]; g.pf_n[0] = pf.n[0]; g.pf_p[1] = pf.p[1]; g.pf_n[1] = pf.n[1];
        }
    };
    // consumer side of site k: finalized statistics (by the producer, or one more launch, which also prefetches `pf`) or the in-kernel form
    auto lnf_consumer = [&](GemmParams& g, int k, bool inkernel, int site, int tag, const PrefetchSet* pf) -> int {
        const float* pivots = k > 0 ? lnfS[(k - 1) & 1] : nullptr;
        if (!inkernel && fin_site[k & 1]) {
            g.lnf_stats = lnfS[k & 1];
        } else if (!inkernel) {
            F5_TRY(timed(p, site, st, [&] {
                return launch_stats_finalize(p->lnf_partial, (int)p->rows_cap, D / 64, rows, D, pivots, lnfS[k & 1], sat, tag, st, pf);
            }));
            g.lnf_stats = lnfS[k & 1];
        } else {
            g.lnf_stats = lnfS[k & 1];  // (marks the launch as folded; read only by the 256-wide tile)
            g.lnf_partial = p->lnf_partial; g.lnf_partial_ld = (int)p->rows_cap; g.lnf_ncols = D / 64;
            g.lnf_pivot = pivots; g.lnf_stats_out = lnfS[k & 1]; g.lnf_sat = sat; g.lnf_sat_tag = tag;
        }
        return 0;
    };
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
        // (LayerNorm fold: the statistics kernels sit where the passes sat and prefetch what runs behind THEM -- the one behind the out-projection
        //  this block's W' rows of ff.0.0 and the FF2 weight, the one behind FF2 the next block's W' rows of q|k|v and its out-projection weight)
        PrefetchSet pf2f{{nullptr, nullptr, nullptr, nullptr}, {0u, 0u, 0u, 0u}}, pf1f = pf2f;
        if (lnf) {
            pf2f = PrefetchSet{{(const char*)ft->Wt + (frow0 + (size_t)l * fR + 3 * inner) * D * 2, b.w_ff2, nullptr, nullptr},
                               {(unsigned)((size_t)ff * D * 2), (unsigned)(D * ff * wes), 0u, 0u}};
            if (l + 1 < c.depth)
                pf1f = PrefetchSet{{(const char*)ft->Wt + (frow0 + (size_t)(l + 1) * fR) * D * 2, m->blocks[l + 1].w_o, nullptr, nullptr},
                                   {(unsigned)((size_t)3 * inner * D * 2), (unsigned)(D * inner * wes), 0u, 0u}};
        }
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
        g.A = p->hT; g.lda = D; g.W = b.w_qkv; g.ldw = D; g.M = rows_g; g.N = 3 * inner; g.K = D;
        g.bias = b.b_qkv; g.out_t = p->qkv; g.ldo = 3 * inner; g.rows_per_batch = N; g.site = 1;
        g.rope = rg ? p->rope_exp : p->rope; g.rope_inner = inner; g.rope_heads = m->rope_heads;  // (ragged: row r of a half -> its position in its utterance)
        if (lnf1) {
            g.A = p->xres16; g.W = fW; g.bias = nullptr;
            g.lnf_c1 = fc1; g.lnf_c2 = fc2;
            F5_TRY(lnf_consumer(g, 2 * l - 1, ink_qkv, F5_SITE_LN1, 1 | (l << 4), wpf ? &pf1f_prev : nullptr));
        }
        // Tile quantisation at small batches: the fused projection has 12 feature tiles per token tile; when the q|k part alone (8 tiles
        // per token tile) fills the CUs a whole number of times but q|k|v does not (M = 8192, 4 utterances x 1024 frames x CFG: 256 + 128
        // tiles on 256 CUs, the second round half empty), v is projected by its own launch on 256 x 128 tiles: 70 -> 62 us per block.
        // (round 3, late: any token count whose q|k tiles fit one round while q|k|v would need a second -- ragged batches, odd batch sizes:
        //  M = 6144: 288 tiles of 256 x 256 = two rounds, 60 us; 192 + 192 narrower ones: 53 us.  Same sums either way.)
        const int tiles_m = (rows + 255) / 256, ncu = f5_cu_count();
        // (not when the one-wave-per-SIMD kernel takes the whole fused projection: its 128-row tiles give every CU a whole number of tiles there)
        const bool split_v = P == F5_PREC_BF16 && p->gemm_kernel != 0 && inner % 256 == 0 && tiles_m * (2 * inner / 256) <= ncu &&
                             tiles_m * (3 * inner / 256) > ncu && tiles_m * (2 * inner / 256) >= 160 && !(!c.qk_norm && gemm_w4_ok(g, GEMM_DENSE, EPI_ROPE_T));
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
                gv.lnf_stats_out = nullptr;  // (the q|k launch's feature tile 0 stores the new means)
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
        g.A = p->cT; g.lda = inner; g.W = b.w_o; g.ldw = inner; g.M = rows_g; g.N = D; g.K = inner;
        g.bias = b.b_o; g.out_t = defer ? p->yA : p->yT; g.ldo = D; g.gate = ml + 2 * D; g.gate_bstride = mod_bstride; g.rows_per_batch = N;
        g.rowmask = mask; g.site = 2;
        g.rowbits = (mask && mask == p->rowbits_src) ? p->rowbits : nullptr;
        if (rmw) {
            g.out_t = nullptr;
            g.out_f = reinterpret_cast<float*>(p->xres16);
            g.ldof = D;
            g.add2_f16 = 1;
        }
        // partial row sums of the updated stream (site 2l); pivot = the row's previous mean (none yet at site 0: the tables are this evaluation's)
        if (lnf) lnf_producer(g, 2 * l, 2 | (l << 4), pf2f, ink_ff1);  // (prefetch: the weights of FF1 and FF2)
        F5_TRY(timed(p, F5_SITE_OUT, st, [&] { return run_gemm(p, g, GEMM_DENSE, rmw ? EPI_RESID : EPI_GATE_T, st); }));
        // x += y; n2 = LN(x) * (1 + scale_mlp) + shift_mlp
        if (!lnf) F5_TRY(timed(p, F5_SITE_LN2, st, [&] {
            if (rmw)
                return launch_layernorm_res(P, xin, 1, xout, 1, D, rows, D, nullptr, D, nullptr, 1, ml + 4 * D, ml + 3 * D, mod_bstride, N, 1, p->hT, D, st,
                                            wpf ? &pf2 : nullptr, sat, 2 | (l << 4));
            return launch_layernorm_res(P, xin, in16, xout, r16, D, rows, D, p->yT, D, defer ? p->yA : nullptr, defer ? 3 : 1, ml + 4 * D, ml + 3 * D,
                                        mod_bstride, N, 1, p->hT, D, st, wpf ? &pf2 : nullptr, sat, 2 | (l << 4));
        }));
        g = gp_zero();
        g.A = p->hT; g.lda = D; g.W = b.w_ff1; g.ldw = D; g.M = rows_g; g.N = ff; g.K = D;
        g.bias = b.b_ff1; g.act = ACT_GELU_TANH; g.out_t = p->ffh; g.ldo = ff; g.site = 3;
        if (lnf) {
            g.A = p->xres16; g.W = fW + (size_t)3 * inner * D * 2; g.bias = nullptr;
            g.lnf_c1 = fc1 + 3 * inner; g.lnf_c2 = fc2 + 3 * inner;
            F5_TRY(lnf_consumer(g, 2 * l, ink_ff1, F5_SITE_LN2, 2 | (l << 4), wpf ? &pf2f : nullptr));
        }
        F5_TRY(timed(p, F5_SITE_FF1, st, [&] { return run_gemm(p, g, GEMM_DENSE, EPI_STORE_T, st); }));
        // y = gate_mlp * ff(n2)  (modules.py:639)
        g = gp_zero();
        g.A = p->ffh; g.lda = ff; g.W = b.w_ff2; g.ldw = ff; g.M = rows_g; g.N = D; g.K = ff;
        g.bias = b.b_ff2; g.out_t = p->yT; g.ldo = D; g.gate = ml + 5 * D; g.gate_bstride = mod_bstride; g.rows_per_batch = N; g.site = 4;
        if (rmw) {
            g.out_t = nullptr;
            g.out_f = reinterpret_cast<float*>(p->xres16);
            g.ldof = D;
            g.add2_f16 = 1;
        }
        const bool lnf_next = lnf && l + 1 < c.depth;  // (the final AdaLN pass reads the stream itself)
        if (lnf_next) lnf_producer(g, 2 * l + 1, 1 | ((l + 1) << 4), pf1f, ink_qkv);  // site 2l + 1 (prefetch: the next block's q|k|v weights and its out-projection)
        F5_TRY(timed(p, F5_SITE_FF2, st, [&] { return run_gemm(p, g, GEMM_DENSE, rmw ? EPI_RESID : EPI_GATE_T, st); }));
        pf1f_prev = pf1f;  // (what a statistics launch in front of the next block's QKV projection prefetches)
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

