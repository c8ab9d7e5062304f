// sampler.hip -- CFM.sample on the device (reference model/cfm.py:82-208): the fixed-grid ODE loop over network evaluations with CFG doubling,
// hipGraph capture / replay, the fp16 range guard and its fp32 fallback, the LayerNorm-fold tables of a time grid, and the ragged sampler.
#include "model_internal.h"

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
            if (p->graphs.size() >= 32) {  // small LRU-less cache: drop the oldest bucket (32: batch inference over length buckets meets a few dozen shapes)
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
int finish_if_pending(f5_plan_s* p) {
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

