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
        if ((rc = launch_gemm(g, precision, GEMM_CONV31, EPI_STORE_T, 0, st))) break;
        g.A = c1; g.W = wr[1]; g.bias = b1; g.out_t = nullptr; g.out_f = acc; g.ldof = dim;
        if ((rc = launch_gemm(g, precision, GEMM_CONV31, EPI_RESID, 0, st))) break;
        rc = launch_convert_back(F5_PREC_FP32, acc, dim, rows, dim, out, dim, st);
    } while (0);
    return sync_and_release(a, st, rc);
}
