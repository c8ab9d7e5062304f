// model.hip -- DiT / UNetT / MMDiT weights in HBM and the per-(batch, seq) plan / workspace (the evaluations and the sampler: model_internal.h).
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
#include "model_internal.h"

f5_model_s::~f5_model_s() {
    for (FoldTable* t : folds) delete t;
}

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
        if ((rc = A.alloc_t(&p->rowbits, (size_t)(2 * bn / 128 + 4) * 16))) break;  // (+ the padded rows of a launch rounded up to 256)
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
            if ((rc = A.alloc_t(&p->lnf_stats2, (rows + 256) * 2))) break;
            if ((rc = A.alloc_t(&p->lnf_partial, (D / 64) * rows * 2))) break;
            if ((rc = A.alloc_t(&p->fin_counter, rows / 128 + 64))) break;  // (the arena is zero-filled; the last arriver of a launch resets its word)
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
    else if (strcmp(key, "ln_fold_active") == 0)  // the staged time grid has a LayerNorm-fold table and the knob is on (what the next sample() runs)
        *value = (p->fold && g_ln_fold) ? 1 : 0;
    else if (strcmp(key, "gemm_w4") == 0)  // the one-wave-per-SIMD kernel's knob (which launches take it: gemm_w4_ok, by shape)
        *value = g_gemm_w4;
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

