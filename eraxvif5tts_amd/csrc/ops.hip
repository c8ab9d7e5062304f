// ops.hip -- per-op C entry points used by the parity tests and micro-benchmarks (f32 device tensors in and out;
// conversion to the precision's activation dtype happens on the device, exactly as inside the model path).
#include <cstring>

#include "gemm.h"
#include "kernels.h"
#include "runtime.h"

static int sync_and_release(DevArena& a, hipStream_t st, int rc) {
    hipError_t e = hipStreamSynchronize(st);
    a.release();
    if (rc != 0) return rc;
    if (e != hipSuccess) return f5_fail(F5_EHIP, "stream synchronize failed: %s", hipGetErrorString(e));
    return 0;
}

extern "C" int f5_op_linear(int precision, int kernel, int M, int N, int K, const float* A, const float* W, const float* bias, int act,
                            float* out, f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !out) return f5_fail(F5_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int Kp = (int)round_up(K, 64), Mp = (int)round_up(M, 256), Np = (int)round_up(N, 256);
    const size_t es = f5_elem_size(precision);
    DevArena a;
    void *At = nullptr, *Wt = nullptr;
    int rc = 0;
    do {
        if ((rc = a.alloc(&At, (size_t)Mp * Kp * es))) break;
        if ((rc = a.alloc(&Wt, (size_t)Np * Kp * es))) break;
        if ((rc = launch_convert_pad(precision, A, K, M, K, Kp, At, Kp, st))) break;
        if ((rc = launch_convert_pad(precision, W, K, N, K, Kp, Wt, Kp, st))) break;
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = At; g.lda = Kp; g.W = Wt; g.ldw = Kp; g.M = M; g.N = N; g.K = Kp;
        g.bias = bias; g.act = act; g.out_f = out; g.ldof = N;
        rc = launch_gemm(g, precision, GEMM_DENSE, EPI_STORE_F32, kernel, st);
    } while (0);
    return sync_and_release(a, st, rc);
}

// One DiT block linear with its fused store epilogue, as the sampler launches it (test hook; bf16 output converted back to fp32):
//   epi 0 (EPI_STORE_T): out = act(A W^T + b)                         FF1 (modules.py:258-264)
//   epi 5 (EPI_GATE_T):  out = gate[n] * act(A W^T + b), 0 where rowmask[m] == 0   attention out / FF2 with the AdaLN gate (modules.py:499-501,635,639)
//   epi 4 (EPI_ROPE_T):  out = rope(A W^T + b) on the q/k columns of the first rope_heads heads (N = 3 * inner, modules.py:452-461)
extern "C" int f5_op_linear_fused(int kernel, int epi, int M, int N, int K, const float* A, const float* W, const float* bias, int act,
                                  const float* gate, const uint8_t* rowmask, const float* rope, int rope_heads, int seq, float* out,
                                  f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !out) return f5_fail(F5_EINVAL, "bad argument");
    if (!(epi == EPI_STORE_T || epi == EPI_GATE_T || epi == EPI_ROPE_T || epi == EPI_RESID)) return f5_fail(F5_EINVAL, "f5_op_linear_fused: epilogue %d", epi);
    if (epi == EPI_RESID && ((size_t)M * N) % 4 != 0) return f5_fail(F5_EINVAL, "f5_op_linear_fused: M * N must be a multiple of 4 for the in-place form");
    if (epi == EPI_ROPE_T && (!rope || seq <= 0 || M % seq != 0 || N % 3 != 0 || (N / 3) % 64 != 0)) return f5_fail(F5_EINVAL, "bad RoPE arguments");
    hipStream_t st = (hipStream_t)stream;
    const int Kp = (int)round_up(K, 64), Mp = (int)round_up(M, 256), Np = (int)round_up(N, 256);
    DevArena a;
    void *At = nullptr, *Wt = nullptr, *Ot = nullptr;
    uint8_t* bits = nullptr;
    int rc = 0;
    do {
        if ((rc = a.alloc(&At, (size_t)Mp * Kp * 2))) break;
        if ((rc = a.alloc(&Wt, (size_t)Np * Kp * 2))) break;
        if ((rc = a.alloc(&Ot, (size_t)Mp * N * 2))) break;  // bf16 output, or the fp16 stream of the in-place form
        if ((rc = a.alloc_t(&bits, (size_t)(Mp / 128 + 1) * 16))) break;
        if ((rc = launch_convert_pad(F5_PREC_BF16, A, K, M, K, Kp, At, Kp, st))) break;
        if ((rc = launch_convert_pad(F5_PREC_BF16, W, K, N, K, Kp, Wt, Kp, st))) break;
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = At; g.lda = Kp; g.W = Wt; g.ldw = Kp; g.M = M; g.N = N; g.K = Kp;
        g.bias = bias; g.act = act; g.out_t = Ot; g.ldo = N; g.rows_per_batch = seq > 0 ? seq : M;
        if (epi == EPI_RESID) {  // in-place update of the fp16 residual stream: `out` is read (rounded to fp16), updated and written back
            if ((rc = launch_f32_to_f16(out, Ot, (size_t)M * N, st))) break;
            g.out_t = nullptr;
            g.out_f = reinterpret_cast<float*>(Ot);
            g.ldof = N;
            g.add2_f16 = 1;
        }
        if (epi == EPI_GATE_T || epi == EPI_RESID) {
            g.gate = gate;
            g.rowmask = rowmask;
            if (rowmask) {
                if ((rc = launch_rowbits(rowmask, M, bits, st))) break;
                g.rowbits = bits;
            }
        }
        if (epi == EPI_ROPE_T) { g.rope = rope; g.rope_inner = N / 3; g.rope_heads = rope_heads; }
        if ((rc = launch_gemm(g, F5_PREC_BF16, GEMM_DENSE, epi, kernel, st))) break;
        rc = epi == EPI_RESID ? launch_f16_to_f32(Ot, out, (size_t)M * N, st) : launch_convert_back(F5_PREC_BF16, Ot, N, M, N, out, N, st);
    } while (0);
    return sync_and_release(a, st, rc);
}

// The LayerNorm fold of one call site end to end (parity tests; gemm.h, lnfold.hip).  `x` [M, D] is the fp16 residual stream, handed over and
// returned as f32: (1) x += gate * (A . Wo^T + bo) in place with partial row statistics (EPI_RESID + stats_out; pivots = column 0 of `pivot`
// [M][2] or none), (2) stats_finalize -> `stats` [M][2] = (mean, rstd), (3) W' / c1 / c2 from W [N, D], bias, scale, shift (fold_weights_kernel),
// (4) out [M, N] = epilogue(rstd (x . W'^T - mean c1) + c2): epi 0 = store with `act` (FF1: GELU tanh), epi 4 = RoPE (fused QKV, N = 3 * inner).
extern "C" int f5_op_ln_fold(int epi, int M, int D, int N, int Kb, float* x, const float* A, const float* Wo, const float* bo, const float* gate,
                             const float* pivot, const float* W, const float* bias, const float* scale, const float* shift, int act,
                             const float* rope, int rope_heads, int seq, float* stats, float* out, f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (M <= 0 || D <= 0 || N <= 0 || Kb <= 0 || !x || !A || !Wo || !bo || !W || !bias || !scale || !shift || !stats || !out) return f5_fail(F5_EINVAL, "bad argument");
    if (D % 128 != 0 || Kb % 32 != 0 || N % 64 != 0 || (epi != EPI_STORE_T && epi != EPI_ROPE_T)) return f5_fail(F5_EINVAL, "f5_op_ln_fold: D % 128, Kb % 32, N % 64, epi 0 | 4");
    if (epi == EPI_ROPE_T && (!rope || seq <= 0 || M % seq != 0 || N % 3 != 0 || (N / 3) % 64 != 0)) return f5_fail(F5_EINVAL, "bad RoPE arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t Mp = round_up(M, 256) + 256;
    DevArena a;
    void *At = nullptr, *Wot = nullptr, *xs = nullptr, *Wt = nullptr, *Ot = nullptr;
    float *partial = nullptr, *st2 = nullptr, *mod = nullptr, *c1 = nullptr, *c2 = nullptr;
    int rc = 0;
    do {
        if ((rc = a.alloc(&At, Mp * Kb * 2))) break;
        if ((rc = a.alloc(&Wot, (size_t)round_up(D, 256) * Kb * 2))) break;
        if ((rc = a.alloc(&xs, Mp * D * 2))) break;
        if ((rc = a.alloc(&Wt, (size_t)round_up(N, 256) * D * 2))) break;
        if ((rc = a.alloc(&Ot, Mp * N * 2))) break;
        if ((rc = a.alloc_t(&partial, (size_t)(D / 64) * Mp * 2))) break;
        if ((rc = a.alloc_t(&st2, Mp * 2))) break;
        if ((rc = a.alloc_t(&mod, (size_t)6 * D))) break;
        if ((rc = a.alloc_t(&c1, (size_t)N))) break;
        if ((rc = a.alloc_t(&c2, (size_t)N))) break;
        if ((rc = launch_convert_pad(F5_PREC_BF16, A, Kb, M, Kb, Kb, At, Kb, st))) break;
        if ((rc = launch_convert_pad(F5_PREC_BF16, Wo, Kb, D, Kb, Kb, Wot, Kb, st))) break;
        if ((rc = launch_f32_to_f16(x, xs, (size_t)M * D, st))) break;
        if (pivot) F5_HIP(hipMemcpyAsync(st2, pivot, (size_t)M * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
        F5_HIP(hipMemcpyAsync(mod, shift, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, st));      // (shift_msa, scale_msa) slots of one block
        F5_HIP(hipMemcpyAsync(mod + D, scale, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, st));
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = At; g.lda = Kb; g.W = Wot; g.ldw = Kb; g.M = M; g.N = D; g.K = Kb; g.bias = bo; g.rows_per_batch = seq > 0 ? seq : M;
        g.out_f = reinterpret_cast<float*>(xs); g.ldof = D; g.add2_f16 = 1; g.gate = gate;
        g.stats_out = partial; g.stats_ld = (int)Mp; g.stats_pivot = pivot ? st2 : nullptr;
        if ((rc = launch_gemm(g, F5_PREC_BF16, GEMM_DENSE, EPI_RESID, 1, st))) break;
        if ((rc = launch_stats_finalize(partial, (int)Mp, D / 64, M, D, pivot ? st2 : nullptr, st2, nullptr, 0, st))) break;
        if ((rc = launch_fold_weights(W, bias, mod, 6 * D, 1, 1, N, N, D, Wt, c1, c2, st))) break;
        memset(&g, 0, sizeof(g));
        g.A = xs; g.lda = D; g.W = Wt; g.ldw = D; g.M = M; g.N = N; g.K = D; g.act = act; g.out_t = Ot; g.ldo = N; g.rows_per_batch = seq > 0 ? seq : M;
        g.lnf_stats = st2; g.lnf_c1 = c1; g.lnf_c2 = c2;
        if (epi == EPI_ROPE_T) { g.rope = rope; g.rope_inner = N / 3; g.rope_heads = rope_heads; }
        if ((rc = launch_gemm(g, F5_PREC_BF16, GEMM_DENSE, epi, 1, st))) break;
        if ((rc = launch_f16_to_f32(xs, x, (size_t)M * D, st))) break;
        F5_HIP(hipMemcpyAsync(stats, st2, (size_t)M * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
        rc = launch_convert_back(F5_PREC_BF16, Ot, N, M, N, out, N, st);
    } while (0);
    return sync_and_release(a, st, rc);
}

extern "C" int f5_op_layernorm_modulate(int rows, int dim, const float* x, const float* scale, const float* shift, float* out,
                                        f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (!x || !scale || !shift || !out) return f5_fail(F5_EINVAL, "null argument");
    return launch_layernorm(F5_PREC_FP32, x, dim, rows, dim, scale, shift, 0, rows, 1, out, dim, (hipStream_t)stream);
}

extern "C" int f5_op_attention(int precision, int kernel, int B, int N, int H, const float* qkv, const uint8_t* mask, float* out,
                               f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (B <= 0 || N <= 0 || H <= 0 || !qkv || !out) return f5_fail(F5_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int inner = H * 64, rows = B * N;
    const size_t es = f5_elem_size(precision);
    DevArena a;
    void *q = nullptr, *o = nullptr;
    int rc = 0;
    do {
        if ((rc = a.alloc(&q, (size_t)rows * 3 * inner * es))) break;
        if ((rc = a.alloc(&o, (size_t)rows * inner * es))) break;
        if ((rc = launch_convert_pad(precision, qkv, 3 * inner, rows, 3 * inner, 3 * inner, q, 3 * inner, st))) break;
        if ((rc = launch_attention(precision, kernel, B, N, H, q, 3 * inner, mask, o, inner, st))) break;
        rc = launch_convert_back(precision, o, inner, rows, inner, out, inner, st);
    } while (0);
    return sync_and_release(a, st, rc);
}

int g_op_conv_kernel = 0;  // tuning knob ("op_conv_kernel"): f5_op_conv_pos_embed runs the tuned conv kernels (bf16)

extern "C" int f5_op_conv_pos_embed(int precision, int B, int N, int dim, const float* x, const float* w0, const float* b0, const float* w1,
                                    const float* b1, float* out, f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (B <= 0 || N <= 0 || dim <= 0 || dim % 128 != 0 || !x || !w0 || !b0 || !w1 || !b1 || !out) return f5_fail(F5_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int rows = B * N, cg = dim / 16;
    int win = 0;
    for (int n0 = 0; n0 < dim; n0 += 64) win = std::max(win, ((n0 + 63) / cg + 1) * cg - (n0 / cg) * cg);
    win = (int)round_up(win, 64);
    const size_t es = f5_elem_size(precision);
    DevArena a;
    int rc = 0;
    do {
        // rearrange the two weights on the host (weights come from the device: copy back first)
        std::vector<float> hw((size_t)dim * cg * 31), r((size_t)31 * dim * win);
        void* wr[2] = {nullptr, nullptr};
        const float* ws[2] = {w0, w1};
        for (int li = 0; li < 2 && rc == 0; ++li) {
            if (hipMemcpy(hw.data(), ws[li], hw.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) {
                rc = f5_fail(F5_EHIP, "weight copy failed");
                break;
            }
            std::fill(r.begin(), r.end(), 0.f);
            for (int n = 0; n < dim; ++n) {
                const int w0c = ((n / 64 * 64) / cg) * cg, g0 = (n / cg) * cg;
                for (int ci = 0; ci < cg; ++ci)
                    for (int tap = 0; tap < 31; ++tap) r[((size_t)tap * dim + n) * win + (g0 + ci - w0c)] = hw[((size_t)n * cg + ci) * 31 + tap];
            }
            rc = f5_upload_t(a, precision, r.data(), r.size(), &wr[li]);
        }
        if (rc) break;
        void *xt = nullptr, *c1 = nullptr;
        float* acc = nullptr;
        if ((rc = a.alloc(&xt, (size_t)rows * dim * es))) break;
        if ((rc = a.alloc(&c1, (size_t)rows * dim * es))) break;
        if ((rc = a.alloc_t(&acc, (size_t)rows * dim))) break;  // zero-initialised accumulator for the RESID epilogue
        if ((rc = launch_convert_pad(precision, x, dim, rows, dim, dim, xt, dim, st))) break;
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = xt; g.lda = dim; g.W = wr[0]; g.M = rows; g.N = dim; g.K = 31 * win; g.bias = b0; g.act = ACT_MISH;
        g.rows_per_batch = N; g.conv_cg = cg; g.conv_win = win; g.out_t = c1; g.ldo = dim;
        // tuning knob "op_conv_kernel" = 1: the tuned kernels exactly as dit_eval launches them (store-only second conv, bf16 only)
        const int kind = (g_op_conv_kernel && precision == F5_PREC_BF16 && gemm_fast_supported(g, precision, GEMM_CONV31, EPI_STORE_T)) ? 1 : 0;
        if ((rc = launch_gemm(g, precision, GEMM_CONV31, EPI_STORE_T, kind, st))) break;
        if (kind) {
            void* c2 = nullptr;
            if ((rc = a.alloc(&c2, (size_t)rows * dim * es))) break;
            g.A = c1; g.W = wr[1]; g.bias = b1; g.out_t = c2;
            if ((rc = launch_gemm(g, precision, GEMM_CONV31, EPI_GATE_T, 1, st))) break;
            rc = launch_convert_back(precision, c2, dim, rows, dim, out, dim, st);
            break;
        }
        g.A = c1; g.W = wr[1]; g.bias = b1; g.out_t = nullptr; g.out_f = acc; g.ldof = dim;
        if ((rc = launch_gemm(g, precision, GEMM_CONV31, EPI_RESID, 0, st))) break;
        rc = launch_convert_back(F5_PREC_FP32, acc, dim, rows, dim, out, dim, st);
    } while (0);
    return sync_and_release(a, st, rc);
}

// ----------------------------------------------------------------------------- in-process kernel timing (bench.py roofline leg)
__global__ __launch_bounds__(256) void fill_random_kernel(uint16_t* dst, size_t n, uint32_t seed, float scale) {
    // counter-based hash -> uniform [-scale, scale) in bf16: random operands (zero-filled ones raise the clock and flatter MFMA rates)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    const float u = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
    dst[i] = (uint16_t)(__float_as_uint(u) >> 16);
}
__global__ __launch_bounds__(256) void fill_random_f32_kernel(float* dst, size_t n, uint32_t seed, float scale) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    dst[i] = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
}

static int g_bench_pad_a = 0, g_bench_pad_w = 0;

// Times `iters` back-to-back launches of ONE GEMM kernel (bf16, the epilogue/shape of the DiT call site `site`) with HIP
// events on `stream`; *ms_avg = average device time of one launch.  site: 0 = fused QKV projection + RoPE (N = 3*inner),
// 1 = FF1 + GELU-tanh, 2 = FF2 + gate, 3 = attention out-projection + gate (store-only residual branches).
extern "C" int f5_bench_gemm_site(int kernel, int site, int rows, int seq, int dim, int heads, int ff_inner, int iters, float* ms_avg,
                                  f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (!ms_avg || rows <= 0 || iters <= 0 || seq <= 0 || rows % seq != 0) return f5_fail(F5_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int inner = heads * 64;
    int N, K, epi;
    switch (site) {
        case 0: N = 3 * inner; K = dim; epi = EPI_ROPE_T; break;
        case 1: N = ff_inner; K = dim; epi = EPI_STORE_T; break;
        case 2: N = dim; K = ff_inner; epi = EPI_GATE_T; break;
        case 3: N = dim; K = inner; epi = EPI_GATE_T; break;
        default: return f5_fail(F5_EINVAL, "bad site");
    }
    const size_t Mp = (size_t)round_up(rows, 256);
    const int lda = K + g_bench_pad_a, ldw = K + g_bench_pad_w;  // leading-dimension padding experiments (tuning knobs)
    DevArena a;
    void *A = nullptr, *W = nullptr, *out = nullptr;
    float *bias = nullptr, *resid = nullptr, *gate = nullptr, *rope = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    do {
        if ((rc = a.alloc(&A, Mp * lda * 2))) break;
        if ((rc = a.alloc(&W, (size_t)round_up(N, 256) * ldw * 2))) break;
        if ((rc = a.alloc(&out, Mp * N * 2))) break;
        if ((rc = a.alloc_t(&bias, (size_t)N))) break;
        if ((rc = a.alloc_t(&resid, Mp * (size_t)dim))) break;
        if ((rc = a.alloc_t(&gate, (size_t)dim))) break;
        if ((rc = a.alloc_t(&rope, (size_t)seq * 64))) break;
        hipLaunchKernelGGL(fill_random_kernel, dim3((unsigned)((Mp * lda + 255) / 256)), dim3(256), 0, st, (uint16_t*)A, Mp * lda, 1u, 1.0f);
        hipLaunchKernelGGL(fill_random_kernel, dim3((unsigned)(((size_t)N * ldw + 255) / 256)), dim3(256), 0, st, (uint16_t*)W, (size_t)N * ldw, 2u, 0.05f);
        hipLaunchKernelGGL(fill_random_f32_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, bias, (size_t)N, 3u, 0.1f);
        hipLaunchKernelGGL(fill_random_f32_kernel, dim3((unsigned)((dim + 255) / 256)), dim3(256), 0, st, gate, (size_t)dim, 4u, 0.01f);
        hipLaunchKernelGGL(fill_random_f32_kernel, dim3((unsigned)((seq * 64 + 255) / 256)), dim3(256), 0, st, rope, (size_t)seq * 64, 5u, 0.7f);
        GemmParams g;
        memset(&g, 0, sizeof(g));
        g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.M = rows; g.N = N; g.K = K; g.bias = bias; g.rows_per_batch = seq;
        g.site = N == 3 * K ? 1 : (N == K ? 2 : (N == 2 * K ? 3 : (K == 2 * N ? 4 : 0)));  // (this bench entry times the ff_mult = 2 DiT call sites)
        if (epi == EPI_ROPE_T) { g.out_t = out; g.ldo = N; g.rope = rope; g.rope_inner = inner; g.rope_heads = 1; }
        if (epi == EPI_STORE_T) { g.out_t = out; g.ldo = N; g.act = ACT_GELU_TANH; }
        if (epi == EPI_GATE_T) { g.out_t = out; g.ldo = N; g.gate = gate; g.gate_bstride = 0; }
        if (kernel == 1 && !gemm_fast_supported(g, F5_PREC_BF16, GEMM_DENSE, epi)) { rc = f5_fail(F5_ENOTSUP, "tuned kernel cannot run this site"); break; }
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = f5_fail(F5_EHIP, "hipEventCreate failed"); break; }
        for (int i = 0; i < 2 && rc == 0; ++i) rc = launch_gemm(g, F5_PREC_BF16, GEMM_DENSE, epi, kernel, st);
        if (rc) break;
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters && rc == 0; ++i) rc = launch_gemm(g, F5_PREC_BF16, GEMM_DENSE, epi, kernel, st);
        (void)hipEventRecord(e1, st);
        if (rc) break;
        if (hipEventSynchronize(e1) != hipSuccess) { rc = f5_fail(F5_EHIP, "event sync failed: %s", hipGetErrorString(hipGetLastError())); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_avg = ms / (float)iters;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return sync_and_release(a, st, rc);
}

// same for the attention kernel: B x H heads of N x 64, bf16
extern "C" int f5_bench_attention(int kernel, int B, int N, int H, int iters, float* ms_avg, f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (!ms_avg || B <= 0 || N <= 0 || H <= 0 || iters <= 0) return f5_fail(F5_EINVAL, "bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int inner = H * 64;
    const size_t rows = (size_t)B * N;
    DevArena a;
    void *q = nullptr, *o = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    do {
        if ((rc = a.alloc(&q, rows * 3 * inner * 2))) break;
        if ((rc = a.alloc(&o, rows * inner * 2))) break;
        hipLaunchKernelGGL(fill_random_kernel, dim3((unsigned)((rows * 3 * inner + 255) / 256)), dim3(256), 0, st, (uint16_t*)q, rows * 3 * inner, 7u, 1.5f);
        if (kernel == 1 && !attention_fast_supported(F5_PREC_BF16, N, H)) { rc = f5_fail(F5_ENOTSUP, "tuned attention cannot run this shape"); break; }
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = f5_fail(F5_EHIP, "hipEventCreate failed"); break; }
        for (int i = 0; i < 2 && rc == 0; ++i) rc = launch_attention(F5_PREC_BF16, kernel, B, N, H, q, 3 * inner, nullptr, o, inner, st);
        if (rc) break;
        (void)hipEventRecord(e0, st);
        for (int i = 0; i < iters && rc == 0; ++i) rc = launch_attention(F5_PREC_BF16, kernel, B, N, H, q, 3 * inner, nullptr, o, inner, st);
        (void)hipEventRecord(e1, st);
        if (rc) break;
        if (hipEventSynchronize(e1) != hipSuccess) { rc = f5_fail(F5_EHIP, "event sync failed"); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *ms_avg = ms / (float)iters;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return sync_and_release(a, st, rc);
}

// ----------------------------------------------------------------------------- what the matrix pipe sustains on this device
// Register-resident v_mfma_f32_16x16x32_bf16 stream: 4 waves per workgroup (one per SIMD), 64 independent accumulator tiles per
// wave, no memory traffic.  With zero operands the part holds its clock; with realistic (pseudo-random) operands the MFMA rate is
// power-limited -- that second number, not the data-sheet peak, is what a dense bf16 GEMM can approach here.
__global__ __launch_bounds__(256, 1) void mfma_rate_kernel(int iters, int random_operands, float* sink) {
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 w[8], a[8];
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u;
            w[i][e] = random_operands ? (bf16_t)(((int)(h >> 20) & 0xfff) * (1.0f / 1024.0f) - 2.0f) : (bf16_t)0.f;
            h = h * 1664525u + 1013904223u;
            a[i][e] = random_operands ? (bf16_t)(((int)(h >> 20) & 0xfff) * (1.0f / 32768.0f) - 0.0625f) : (bf16_t)0.f;
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(w[i]), "v"(a[j]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0];
    if (s == 12345.678f) sink[0] = s;  // keeps the accumulators alive
}

extern "C" int f5_bench_mfma_rate(int random_operands, float* tflops, f5_stream_t stream) {
    F5_TRY(f5_check_device());
    if (!tflops) return f5_fail(F5_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    int dev = 0, cus = 0;
    F5_HIP(hipGetDevice(&dev));
    F5_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int iters = 2000, blocks = cus * 4;  // 4 workgroups per CU back to back: ~1.5 ms per launch
    DevArena a;
    float* sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    do {
        if ((rc = a.alloc_t(&sink, 16))) break;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = f5_fail(F5_EHIP, "hipEventCreate failed"); break; }
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, st, iters, random_operands, sink);  // clocks settle
        (void)hipEventRecord(e0, st);
        const int reps = 10;
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(mfma_rate_kernel, dim3(blocks), dim3(256), 0, st, iters, random_operands, sink);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { rc = f5_fail(F5_EHIP, "event sync failed"); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)reps * blocks * 4.0 * iters * 64.0 * (2.0 * 16 * 16 * 32);
        *tflops = (float)(flops / (ms * 1e-3) / 1e12);
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return sync_and_release(a, st, rc);
}

extern int g_sync_evals, g_attn_persist, g_attn_stagger, g_resid_rmw, g_ln_fold, g_ln_fold_inkernel, g_ln_fold_fin, g_gemm_pad_rows;
extern int g_conv31_tok;
extern int g_gemm_bm128, g_gemm_tile, g_gemm_group_sites, g_gemm_reverse_sites;
extern int g_gemm_split_tail, g_gemm_w4, g_gemm_w4_bm, g_gemm_w4_ink;
extern int g_gemm_variant, g_gemm_group, g_gemm_persist_grid, g_gemm_persist, g_gemm_lean, g_ln_defer, g_ln_wide, g_ln_rows, g_ln_rows_min, g_w_prefetch, g_res_f16, g_conv31, g_attn_variant, g_vocos_fft;
int g_tuning_epoch = 0;
extern unsigned long long* g_attn_stamp_buf;
extern unsigned long long* g_gemm_clk_buf;
// diagnostic: while dev_buf (u64 [workgroups * 4]) is non-null the tuned GEMM writes (s_memtime, s_memrealtime) at workgroup start and end
extern "C" int f5_debug_gemm_clock(void* dev_buf) {
    g_gemm_clk_buf = reinterpret_cast<unsigned long long*>(dev_buf);
    return 0;
}
// diagnostic: the persistent attention kernel writes shader-clock stamps (wave 0 of every workgroup, first 8 items, 8 stamps each) into
// `dev_buf` (u64 [workgroups * 64]) while it is non-null
extern "C" int f5_debug_attn_stamps(void* dev_buf) {
    g_attn_stamp_buf = reinterpret_cast<unsigned long long*>(dev_buf);
    return 0;
}

extern "C" int f5_tuning_set(const char* key, int value) {
    if (!key) return f5_fail(F5_EINVAL, "null key");
    ++g_tuning_epoch;  // hipGraphs captured by plans under the previous knob values are dropped at their next use (model.hip)
    if (strcmp(key, "gemm_variant") == 0) {
        g_gemm_variant = value;
        return 0;
    }
    if (strcmp(key, "resid_rmw") == 0) {
        g_resid_rmw = value != 0;
        return 0;
    }
    if (strcmp(key, "attn_stagger") == 0) {
        g_attn_stagger = value;
        return 0;
    }
    if (strcmp(key, "attn_persist") == 0) {
        g_attn_persist = value != 0;
        return 0;
    }
    if (strcmp(key, "sync_evals") == 0) {
        g_sync_evals = value != 0;
        return 0;
    }
    if (strcmp(key, "w_prefetch") == 0) {
        g_w_prefetch = value;
        return 0;
    }
    if (strcmp(key, "gemm_reverse_sites") == 0) {
        g_gemm_reverse_sites = value;
        return 0;
    }
    if (strcmp(key, "gemm_group_sites") == 0) {
        g_gemm_group_sites = value;
        return 0;
    }
    if (strcmp(key, "conv31_tok") == 0) {
        g_conv31_tok = value;
        return 0;
    }
    if (strcmp(key, "gemm_tile") == 0) {
        g_gemm_tile = value;
        return 0;
    }
    if (strcmp(key, "gemm_bm128") == 0) {
        g_gemm_bm128 = value;
        return 0;
    }
    if (strcmp(key, "ln_rows") == 0) {
        g_ln_rows = value;
        return 0;
    }
    if (strcmp(key, "ln_rows_min") == 0) {
        g_ln_rows_min = value;
        return 0;
    }
    if (strcmp(key, "ln_wide") == 0) {
        g_ln_wide = value != 0;
        return 0;
    }
    if (strcmp(key, "residual_f16") == 0) {
        g_res_f16 = value != 0;
        return 0;
    }
    if (strcmp(key, "vocos_fft") == 0) {
        g_vocos_fft = value != 0;
        return 0;
    }
    if (strcmp(key, "attn_variant") == 0) {
        g_attn_variant = value;
        return 0;
    }
    if (strcmp(key, "gemm_group") == 0) {
        g_gemm_group = value;
        return 0;
    }
    if (strcmp(key, "op_conv_kernel") == 0) {
        g_op_conv_kernel = value != 0;
        return 0;
    }
    if (strcmp(key, "conv31") == 0) {
        g_conv31 = value != 0;
        return 0;
    }
    if (strcmp(key, "ln_defer") == 0) {
        g_ln_defer = value != 0;
        return 0;
    }
    if (strcmp(key, "gemm_lean") == 0) {
        g_gemm_lean = value != 0;
        return 0;
    }
    if (strcmp(key, "gemm_pad_rows") == 0) {
        g_gemm_pad_rows = value != 0;
        return 0;
    }
    if (strcmp(key, "ln_fold_inkernel") == 0) {
        g_ln_fold_inkernel = value != 0;
        return 0;
    }
    if (strcmp(key, "ln_fold_fin") == 0) {
        g_ln_fold_fin = value != 0;
        return 0;
    }
    if (strcmp(key, "ln_fold") == 0) {
        g_ln_fold = value != 0;
        return 0;
    }
    if (strcmp(key, "gemm_split_tail") == 0) {
        g_gemm_split_tail = value != 0;
        return 0;
    }
    if (strcmp(key, "gemm_w4_ink") == 0) {
        g_gemm_w4_ink = value;
        return 0;
    }
    if (strcmp(key, "gemm_w4_bm") == 0) {
        g_gemm_w4_bm = value;
        return 0;
    }
    if (strcmp(key, "gemm_w4") == 0) {
        g_gemm_w4 = value != 0;
        return 0;
    }
    if (strcmp(key, "gemm_persist") == 0) {
        g_gemm_persist = value != 0;
        return 0;
    }
    if (strcmp(key, "gemm_persist_grid") == 0) {
        if (value != 0 && (value < 8 || value > 4096)) return f5_fail(F5_EINVAL, "gemm_persist_grid must be 0 (= CU count) or in [8, 4096]");
        g_gemm_persist_grid = value;
        return 0;
    }
    if (strcmp(key, "bench_pad_a") == 0) {
        g_bench_pad_a = value;
        return 0;
    }
    if (strcmp(key, "bench_pad_w") == 0) {
        g_bench_pad_w = value;
        return 0;
    }
    return f5_fail(F5_EINVAL, "unknown tuning key '%s'", key);
}
