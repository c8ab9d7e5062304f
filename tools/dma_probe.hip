// dma_probe.hip -- what bounds the LDS-DMA operand feed of the GEMM tiles?  Standalone microbenchmark (no MFMA, no LDS reads):
// every workgroup walks the K range of one 256 x 256 tile of A[65536,1024] . W[1024,1024]^T and only moves the operand slices
// into a 5-slot LDS ring, exactly like the GEMM kernels.  hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o /tmp/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void dma16(const void* src, char* lds) { __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds, 16, 0, 0); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SHAPE 0: piece = 16 rows x 64 B (one 32-deep K-step of 16 rows); SHAPE 1: piece = 8 rows x 128 B (64-deep, whole lines)
// PAIR: SHAPE 0 only, two consecutive K-steps of the same rows issued back to back
// WAVES: 4 or 8 per workgroup; SYNC: barrier per K-step; WONLY/AONLY: move only one operand (half the bytes)
template <int WAVES, int SHAPE, int PAIR, int SYNC, int OPS, int GAP = 0>
__global__ __launch_bounds__(WAVES * 64) void probe(const char* A, const char* W, int lda_b, int ldw_b, int nk, int* sink) {
    __shared__ __attribute__((aligned(16))) char smem[5 * 32768];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = gridDim.x >> 3;
    const int swz = xcd * q + (bid >> 3);
    const int tile_m = swz >> 2, tile_n = swz & 3;
    constexpr int PIECES = 32 / WAVES;            // 1 KB pieces per wave per K-step (both operands)
    constexpr int PER_OP = PIECES / 2;
    const char* src[PIECES];
#pragma unroll
    for (int j = 0; j < PIECES; ++j) {
        const bool isw = j >= PER_OP;
        const int pj = wave * PER_OP + (isw ? j - PER_OP : j);  // piece index inside the operand tile (16 per operand)
        int row, colb;
        if (SHAPE == 0) { row = pj * 16 + (lane >> 2); colb = (lane & 3) * 16; }
        else { row = pj * 8 + (lane >> 3); colb = (lane & 7) * 16; }   // SHAPE 1: 8 rows per piece -> only 128 of 256 rows per 32-deep equivalent
        src[j] = isw ? W + (size_t)(tile_n * 256 + row) * ldw_b + colb : A + (size_t)(tile_m * 256 + row) * lda_b + colb;
    }
    auto issue = [&](int kt) {
        char* sb = smem + (kt % 5) * 32768 + wave * (PIECES * 1024);
#pragma unroll
        for (int j = 0; j < PIECES; ++j) {
            if (OPS == 1 && j >= PER_OP) continue;   // activations only
            if (OPS == 2 && j < PER_OP) continue;    // weights only
            const size_t ko = SHAPE == 0 ? (size_t)kt * 64 : (size_t)(kt >> 1) * 128 + 0;  // SHAPE 1: same bytes per step, lines revisited every other step at +8 rows
            const char* s = src[j] + ko + (SHAPE == 1 && (kt & 1) ? (size_t)128 * (j >= PER_OP ? ldw_b : lda_b) : 0);
            dma16(s, sb + j * 1024);
        }
    };
    constexpr int OPN = OPS == 0 ? PIECES : PER_OP;
    if (PAIR) {
        auto issue_pair = [&](int kt) {
            char* s0 = smem + (kt % 5) * 32768 + wave * (PIECES * 1024);
            char* s1 = smem + ((kt + 1) % 5) * 32768 + wave * (PIECES * 1024);
#pragma unroll
            for (int j = 0; j < PIECES; ++j) {
                if (OPS == 1 && j >= PER_OP) continue;
                if (OPS == 2 && j < PER_OP) continue;
                dma16(src[j] + (size_t)kt * 64, s0 + j * 1024);
                if (GAP == 1) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");          // ~64 cycles between the halves
                if (GAP == 3) asm volatile("s_sleep 4");                                                  // ~256 cycles between the halves
                dma16(src[j] + (size_t)kt * 64 + 64, s1 + j * 1024);
                if (GAP == 2) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");  // halves adjacent, ~128 cycles between pairs
            }
        };
        issue_pair(0); issue_pair(2);
        for (int kt = 0; kt < nk; kt += 2) {
            wait_vm<2 * OPN>();
            if (SYNC) __builtin_amdgcn_s_barrier();
            if (kt + 4 < nk) issue_pair(kt + 4);
            wait_vm<2 * OPN>();
            if (SYNC) __builtin_amdgcn_s_barrier();
        }
    } else {
        for (int d = 0; d < 4; ++d) issue(d);
        for (int kt = 0; kt < nk; ++kt) {
            wait_vm<2 * OPN>();
            if (SYNC) __builtin_amdgcn_s_barrier();
            if (kt + 4 < nk) issue(kt + 4);
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = *(int*)(smem + (bid & 1023));
}

template <int WAVES, int SHAPE, int PAIR, int SYNC, int OPS, int GAP = 0> void run(const char* name, const char* A, const char* W, int K, int* sink) {
    const int nk = K / 32, blocks = 1024;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((probe<WAVES, SHAPE, PAIR, SYNC, OPS, GAP>), dim3(blocks), dim3(WAVES * 64), 0, 0, A, W, K * 2, K * 2, nk, sink);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<WAVES, SHAPE, PAIR, SYNC, OPS, GAP>), dim3(blocks), dim3(WAVES * 64), 0, 0, A, W, K * 2, K * 2, nk, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, per_step = us / 4.0 / nk;
    const double bytes = (OPS == 0 ? 32768.0 : 16384.0);
    printf("%-58s %8.1f us  %6.3f us/K-step  %6.1f GB/s per CU\n", name, us, per_step, bytes / per_step * 1e-3);
    fflush(stdout);
}

int main() {
    const int M = 65536, N = 1024, K = 1024;
    char *A, *W; int* sink;
    hipMalloc(&A, (size_t)M * K * 2 + (1 << 20)); hipMalloc(&W, (size_t)N * K * 2 + (1 << 20)); hipMalloc(&sink, 4096 * 4);
    hipMemset(A, 1, (size_t)M * K * 2); hipMemset(W, 1, (size_t)N * K * 2);
    run<4, 0, 0, 1, 0>("4 waves, 16x64B pieces, barrier", A, W, K, sink);
    run<4, 0, 0, 0, 0>("4 waves, 16x64B pieces, no barrier", A, W, K, sink);
    run<8, 0, 0, 1, 0>("8 waves, 16x64B pieces, barrier", A, W, K, sink);
    run<8, 0, 0, 0, 0>("8 waves, 16x64B pieces, no barrier", A, W, K, sink);
    run<4, 0, 1, 1, 0>("4 waves, 16x64B pieces, paired K-steps, barrier", A, W, K, sink);
    run<8, 0, 1, 1, 0>("8 waves, 16x64B pieces, paired K-steps, barrier", A, W, K, sink);
    run<4, 0, 1, 1, 0, 1>("4 waves, paired, ~64 cycles between the two halves", A, W, K, sink);
    run<4, 0, 1, 1, 0, 3>("4 waves, paired, ~256 cycles between the two halves", A, W, K, sink);
    run<4, 0, 1, 1, 0, 2>("4 waves, paired, halves adjacent, ~128 cycles between pairs", A, W, K, sink);
    run<4, 1, 0, 1, 0>("4 waves, 8x128B pieces, barrier", A, W, K, sink);
    run<8, 1, 0, 1, 0>("8 waves, 8x128B pieces, barrier", A, W, K, sink);
    run<4, 0, 0, 1, 1>("4 waves, 16x64B, activations only (half bytes)", A, W, K, sink);
    run<4, 0, 0, 1, 2>("4 waves, 16x64B, weights only (half bytes)", A, W, K, sink);
    run<4, 1, 0, 1, 1>("4 waves, 8x128B, activations only (half bytes)", A, W, K, sink);
    run<4, 1, 0, 1, 2>("4 waves, 8x128B, weights only (half bytes)", A, W, K, sink);
    return 0;
}
