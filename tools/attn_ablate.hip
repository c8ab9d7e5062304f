// attn_ablate.hip -- timing-only ablations of the software-pipelined attention kernel (results are wrong by construction).
// Standalone: includes the kernel source and instantiates its ABL != 0 builds, which the library never does.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize tools/attn_ablate.hip -o tools/bin/attn_ablate
//   tools/bin/attn_ablate [B N H]       (defaults: the C2 shape, 64 x 1024 x 16)
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../eraxvif5tts_amd/csrc/attention_pipe.hip"

void f5_set_error(const char*, ...) {}
int f5_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    va_end(ap);
    fputc('\n', stderr);
    return code;
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

template <int WAVES, int ABL, int WPE = 2, int PADLDS = 0> static float run(int B, int N, int H, const bf16_t* qkv, bf16_t* out, int iters) {
    const float c = 0.125f * 1.4426950408889634f;
    if (PADLDS) (void)hipFuncSetAttribute((const void*)attn_pipe_kernel<false, WAVES, ABL, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, PADLDS);
    const dim3 grid(cdiv(N, 32 * WAVES), H, B);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i)
        hipLaunchKernelGGL((attn_pipe_kernel<false, WAVES, ABL, WPE>), grid, dim3(WAVES * 64), PADLDS, 0, qkv, 3 * H * 64, H * 64, (const uint8_t*)nullptr, out, H * 64, N, N, c, AttnSegs{});
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL((attn_pipe_kernel<false, WAVES, ABL, WPE>), grid, dim3(WAVES * 64), PADLDS, 0, qkv, 3 * H * 64, H * 64, (const uint8_t*)nullptr, out, H * 64, N, N, c, AttnSegs{});
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    const int B = argc > 3 ? atoi(argv[1]) : 64, N = argc > 3 ? atoi(argv[2]) : 1024, H = argc > 3 ? atoi(argv[3]) : 16;
    const size_t rows = (size_t)B * N, nq = rows * 3 * H * 64;
    std::vector<uint16_t> h(nq);
    uint32_t x = 12345u;
    for (size_t i = 0; i < nq; ++i) {  // uniform [-1.5, 1.5) in bf16
        x = x * 1664525u + 1013904223u;
        const float u = ((float)(x >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1.5f;
        uint32_t bits;
        memcpy(&bits, &u, 4);
        h[i] = (uint16_t)(bits >> 16);
    }
    bf16_t *qkv, *out;
    CK(hipMalloc(&qkv, nq * 2));
    CK(hipMalloc(&out, rows * H * 64 * 2));
    CK(hipMemcpy(qkv, h.data(), nq * 2, hipMemcpyHostToDevice));
    const double flops = 4.0 * B * H * (double)N * N * 64;
    if (argc > 4 || (argc == 2 && atoi(argv[1]) < 0)) {  // data dependence of the full kernel: zeros / tiny values instead of uniform random
        bf16_t* z;
        CK(hipMalloc(&z, nq * 2));
        CK(hipMemset(z, 0, nq * 2));
        for (int rnd = 0; rnd < 3; ++rnd) {
            const float a = run<8, 0>(B, N, H, qkv, out, 10), zr = run<8, 0>(B, N, H, z, out, 10);
            const float a4 = run<4, 0>(B, N, H, qkv, out, 10), z4 = run<4, 0>(B, N, H, z, out, 10);
            printf("full kernel: random data %.1f us, all-zero data %.1f us | 4-wave: random %.1f us, zero %.1f us\n", a, zr, a4, z4);
        }
        return 0;
    }
    struct { const char* name; float us; } r[48];
    int n = 0;
    for (int rnd = 0; rnd < 2; ++rnd) {
        n = 0;
#define RUN(W, A, label) r[n++] = {label, run<W, A>(B, N, H, qkv, out, 10)}
        RUN(8, 0, "full kernel");
        RUN(8, 1, "no exp2");
        RUN(8, 16, "no row-sum adds");
        RUN(8, 32, "no row max");
        RUN(8, 1 | 16 | 32, "no exp2, sums, max");
        RUN(8, 2, "no PV MFMAs");
        RUN(8, 4, "no QK^T MFMAs");
        RUN(8, 2 | 4, "no MFMAs at all");
        RUN(8, 8, "no K/V tile refresh / barrier");
        RUN(8, 64, "no LDS fragment reads");
        RUN(8, 8 | 64, "no refresh, no fragment reads");
        RUN(8, 128, "no K fragment reads");
        RUN(8, 256, "no V fragment (transposed) reads");
        RUN(8, 512, "fragments read, MFMAs from registers");
        RUN(8, 8 | 512, "same, no refresh");
        RUN(4, 0, "full kernel, 4-wave workgroups");
        RUN(4, 64, "4-wave, no LDS fragment reads");
        r[n++] = {"4-wave WGs, ONE per CU (1 wave/SIMD)", run<4, 0, 2, 100 * 1024>(B, N, H, qkv, out, 10)};
#define RUN1(A, label) r[n++] = {"1 wave/SIMD: " label, run<4, A, 2, 100 * 1024>(B, N, H, qkv, out, 10)}
        RUN1(64, "no LDS fragment reads");
        RUN1(8, "no refresh / barrier");
        RUN1(8 | 64, "no refresh, no fragment reads");
        RUN1(2 | 4 | 8 | 64, "softmax VALU only");
        RUN1(1 | 16 | 32 | 8 | 64, "MFMAs + cvt only");
        RUN1(2 | 4, "no MFMAs at all");
        RUN1(1, "no exp2");
        RUN1(1 | 16 | 32, "no exp2, sums, max");
        RUN1(2 | 4 | 8 | 64 | 1, "VALU only, no exp2");
        RUN1(2 | 4 | 8 | 64 | 16, "VALU only, no row sums");
        RUN1(2 | 4 | 8 | 64 | 32, "VALU only, no row max");
        r[n++] = {"4-wave WGs, 3 waves/SIMD build", run<4, 0, 3>(B, N, H, qkv, out, 10)};
        r[n++] = {"same, no LDS fragment reads", run<4, 64, 3>(B, N, H, qkv, out, 10)};
        RUN(8, 1 | 16 | 32 | 8 | 64, "MFMAs + cvt only");
        RUN(8, 2 | 4 | 8 | 64, "softmax VALU only");
    }
    printf("B=%d N=%d H=%d, 8 waves x 32 queries per workgroup (second of two rounds)\n", B, N, H);
    for (int i = 0; i < n; ++i) printf("  %-34s %8.1f us  (%.0f TFLOP/s-equivalent)\n", r[i].name, r[i].us, flops / r[i].us / 1e6);
    return 0;
}
