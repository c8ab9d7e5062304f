// valu_probe.hip -- issue-rate microbenchmarks on gfx950: what one SIMD sustains for each vector instruction class at 1-4 waves per SIMD,
// alone and beside v_mfma_f32_32x32x16_bf16 with the accumulator in arch VGPRs or in AccVGPRs.  In-kernel s_memtime stamps (shader cycles).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/valu_probe.hip -o tools/bin/valu_probe && tools/bin/valu_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

enum Op { FMA, PK_FMA, ADD, PK_ADD, MUL, PK_MUL, EXP, CVT, MAX3, NONE };
static const char* OPN[] = {"v_fma_f32", "v_pk_fma_f32", "v_add_f32", "v_pk_add_f32", "v_mul_f32", "v_pk_mul_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "v_max3_f32", "(none)"};

// MF: 0 no MFMA, 1 one MFMA per NV vector instructions with the accumulator in arch VGPRs, 2 the same with AccVGPRs,
//     3 one ds_read_b128 per NV vector instructions, 4 two ds_read_b64_tr_b16 per NV, 5 MFMA (VGPR acc) + ds_read_b128 per NV
template <int OP, int MF, int NV>
__global__ __launch_bounds__(1024) void probe(int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ char lds[];  // sized by the host so that exactly one workgroup fits a CU
    float r[16];
    f32x2 q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        r[i] = 1.0f + 1e-3f * (threadIdx.x + i);
        q[i] = f32x2{r[i], r[i] * 0.5f};
    }
    const float a = 1.0001f, b = 1e-7f;
    const f32x2 a2 = {a, a}, b2 = {b, b};
    f32x16 acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
    bf16x8 ma, mb;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        ma[i] = (__bf16)(0.01f * (threadIdx.x % 7 + i));
        mb[i] = (__bf16)(0.02f * (threadIdx.x % 5 + i));
    }
    auto valu = [&](int i) {
        const int k = i & 15;
        if constexpr (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
        if constexpr (OP == ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[k]) : "v"(b));
        if constexpr (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a));
        if constexpr (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(r[k]));
        if constexpr (OP == MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[k]) : "v"(a), "v"(b));
        if constexpr (OP == CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[k]) : "v"(a));
        if constexpr (OP == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q[k]) : "v"(a2), "v"(b2));
        if constexpr (OP == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[k]) : "v"(b2));
        if constexpr (OP == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[k]) : "v"(a2));
    };
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    u32x4 ld[4];
    u32x2 lt[4];
    const unsigned laddr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 4096;
    const unsigned taddr = (threadIdx.x & 63) * 8 + (threadIdx.x >> 6) * 4096;
    auto mfma = [&](int j) {
        if constexpr (MF == 3 || MF == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[j & 3]) : "v"(laddr), "n"(1024 * 0));
        if constexpr (MF == 4) {
            asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lt[j & 3]) : "v"(taddr));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:512" : "=v"(lt[(j + 1) & 3]) : "v"(taddr));
        }
        if constexpr (MF == 5) {
            if (j & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(ma), "v"(mb));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(ma), "v"(mb));
        }
        if constexpr (MF == 1) {
            if (j & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(ma), "v"(mb));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(ma), "v"(mb));
        }
        if constexpr (MF == 2) {
            if (j & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(ma), "v"(mb));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(ma), "v"(mb));
        }
    };
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            mfma(g);
#pragma unroll
            for (int i = 0; i < NV; ++i) valu(g * NV + i);
        }
        if constexpr (MF >= 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if constexpr (MF >= 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(ld[i]), "v"(lt[i]));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = lds[threadIdx.x & 63] ? 0.f : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + q[i][0] + q[i][1] + acc0[i] + acc1[i];
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, int MF, int NV> static void run(int wps, int cus, unsigned long long* dcyc, float* sink) {
    const int iters = 2000, waves = 4 * wps, blocks = cus;
    static bool attr = false;
    (void)attr;
    (void)hipFuncSetAttribute((const void*)probe<OP, MF, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL((probe<OP, MF, NV>), dim3(blocks), dim3(64 * waves), 100 * 1024, 0, 50, dcyc, sink);  // warm-up
    hipLaunchKernelGGL((probe<OP, MF, NV>), dim3(blocks), dim3(64 * waves), 100 * 1024, 0, iters, dcyc, sink);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("launch failed\n");
        exit(1);
    }
    std::vector<unsigned long long> h((size_t)blocks * waves);
    (void)hipMemcpy(h.data(), dcyc, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (auto v : h) {
        sum += (double)v;
        if ((double)v > mx) mx = (double)v;
    }
    // with several waves per SIMD the older wave wins arbitration and finishes first: the SIMD's rate is what the SLOWEST wave saw
    const double per_wave = mx / iters;  // cycles per loop body per wave
    (void)sum;
    const int nv = 8 * NV, nm = MF ? 8 : 0;
    static const char* MFN[] = {"no MFMA       ", "MFMA(VGPR acc)", "MFMA(AGPR acc)", "ds_read_b128  ", "2x ds_read_tr ", "MFMA+ds_b128  "};
    printf("  %-18s x%-2d per %s | %d waves/SIMD | body %7.1f cyc (slowest wave)", OPN[OP], NV, MFN[MF], wps, per_wave);
    if (nv) printf(" | %5.2f cyc per vector instr per SIMD", per_wave / wps / nv);
    if (nm) printf(" | %5.1f cyc per group per SIMD", per_wave / wps / nm);
    printf("\n");
}

int main() {
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    unsigned long long* dcyc;
    float* sink;
    (void)hipMalloc(&dcyc, (size_t)cus * 16 * 8);
    (void)hipMalloc(&sink, 64);
    printf("vector instruction issue, alone (8 x NV independent instructions per loop body)\n");
#define ALONE(OP)                                  \
    for (int w = 1; w <= 4; w *= 2) run<OP, 0, 8>(w, cus, dcyc, sink);
    ALONE(FMA) ALONE(PK_FMA) ALONE(ADD) ALONE(PK_ADD) ALONE(MUL) ALONE(PK_MUL) ALONE(EXP) ALONE(CVT) ALONE(MAX3)
    printf("MFMA alone (two independent accumulators)\n");
    for (int w = 1; w <= 2; ++w) { run<NONE, 1, 0>(w, cus, dcyc, sink); run<NONE, 2, 0>(w, cus, dcyc, sink); }
    printf("one MFMA per NV vector instructions\n");
#define BESIDE(OP, NV)                                                                      \
    for (int w = 1; w <= 2; ++w) { run<OP, 1, NV>(w, cus, dcyc, sink); run<OP, 2, NV>(w, cus, dcyc, sink); }
    BESIDE(FMA, 4) BESIDE(FMA, 8) BESIDE(PK_FMA, 4) BESIDE(EXP, 2) BESIDE(EXP, 4) BESIDE(ADD, 8) BESIDE(CVT, 4) BESIDE(FMA, 12)
    printf("LDS reads per NV vector instructions (lgkmcnt(0) once per body of 8 groups)\n");
    for (int w = 1; w <= 2; ++w) {
        run<NONE, 3, 0>(w, cus, dcyc, sink);
        run<NONE, 4, 0>(w, cus, dcyc, sink);
        run<FMA, 3, 8>(w, cus, dcyc, sink);
        run<FMA, 4, 8>(w, cus, dcyc, sink);
        run<FMA, 5, 8>(w, cus, dcyc, sink);
        run<FMA, 3, 4>(w, cus, dcyc, sink);
        run<FMA, 4, 4>(w, cus, dcyc, sink);
        run<FMA, 5, 4>(w, cus, dcyc, sink);
    }
    return 0;
}
