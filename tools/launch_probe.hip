// launch_probe.hip -- what a kernel boundary costs on MI355X, against a grid-wide barrier inside one persistent kernel.
//   (a) a chain of dependent kernel nodes in one hipGraph: empty 1-workgroup kernels, empty 256 x 512-thread kernels that declare the whole
//       160 KiB of LDS (the GEMM's launch shape), and 256-workgroup kernels that read 4 MiB written by their predecessor and write 4 MiB;
//   (b) one persistent 256-workgroup kernel that does the same read/write step between device-scope barriers (atomic counter + spin).
// The difference (a) - (b) is what fusing two kernels of the single-utterance DiT block behind a grid barrier could save per boundary.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/launch_probe.hip -o tools/bin/launch_probe && tools/bin/launch_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

__global__ void empty_small() {}

__global__ __launch_bounds__(512) void empty_big() {
    extern __shared__ char lds[];
    if (threadIdx.x == 9999) lds[0] = 1;
}

// every thread moves 32 bytes: 256 x 512 threads x 32 B = 4 MiB in, 4 MiB out; the read pattern crosses workgroups (and XCDs)
__global__ __launch_bounds__(512) void touch(const uint4* __restrict__ in, uint4* __restrict__ out, int n16) {
    const int t = blockIdx.x * 512 + threadIdx.x;
    const int src = (t * 2 + 12345) % n16;
    uint4 a = in[src], b = in[(src + 1) % n16];
    a.x += b.y;
    out[t * 2] = a;
    out[t * 2 + 1] = b;
}

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();  // release: make this workgroup's stores visible device-wide
        atomicAdd(counter, 1u);
        // bounded spin: a workgroup that is never co-resident with the others must not hang the grid
        for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin)
            __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// the same data movement, `steps` times inside one kernel, ping-ponging between two buffers across grid barriers
__global__ __launch_bounds__(512) void persistent_touch(uint4* a, uint4* b, int n16, int steps, unsigned* counter, int do_touch) {
    extern __shared__ char lds[];
    if (threadIdx.x == 9999) lds[0] = 1;
    const int t = blockIdx.x * 512 + threadIdx.x;
    for (int s = 0; s < steps; ++s) {
        if (do_touch) {
            const uint4* in = (s & 1) ? b : a;
            uint4* out = (s & 1) ? a : b;
            const int src = (t * 2 + 12345) % n16;
            // data written by other workgroups before the barrier: read past the (non-coherent across XCDs) L2 with device-scope loads
            uint4 x, y;
            x.x = __hip_atomic_load(&in[src].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            x.y = in[src].y, x.z = in[src].z, x.w = in[src].w;
            y = in[(src + 1) % n16];
            x.x += y.y;
            out[t * 2] = x;
            out[t * 2 + 1] = y;
        }
        grid_barrier(counter, (unsigned)(s + 1) * gridDim.x);
    }
}

template <class F>
static double time_graph(hipStream_t st, int nodes, int reps, F&& enqueue) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < nodes; ++i) enqueue(i);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
    return us / (double(reps) * nodes);
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int n16 = 256 * 512 * 2;  // uint4 elements = 4 MiB
    uint4 *a, *b;
    unsigned* counter;
    CK(hipMalloc(&a, size_t(n16) * 16));
    CK(hipMalloc(&b, size_t(n16) * 16));
    CK(hipMalloc(&counter, 4));
    CK(hipMemset(a, 1, size_t(n16) * 16));
    CK(hipMemset(b, 2, size_t(n16) * 16));
    CK(hipFuncSetAttribute((const void*)empty_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)persistent_touch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%d CUs\n", cus);
    const int nodes = 500, reps = 10;
    printf("hipGraph chain, per node:\n");
    printf("  empty kernel, 1 workgroup x 64                  : %6.2f us\n", time_graph(st, nodes, reps, [&](int) { empty_small<<<1, 64, 0, st>>>(); }));
    printf("  empty kernel, 256 workgroups x 512, 160 KiB LDS : %6.2f us\n",
           time_graph(st, nodes, reps, [&](int) { empty_big<<<256, 512, 160 * 1024, st>>>(); }));
    printf("  empty kernel, 256 workgroups x 512, no LDS      : %6.2f us\n", time_graph(st, nodes, reps, [&](int) { empty_big<<<256, 512, 0, st>>>(); }));
    printf("  4 MiB in -> 4 MiB out, 256 x 512, dependent     : %6.2f us\n",
           time_graph(st, nodes, reps, [&](int i) { touch<<<256, 512, 0, st>>>((i & 1) ? b : a, (i & 1) ? a : b, n16); }));
    printf("one persistent kernel (256 workgroups x 512, 160 KiB LDS), per step:\n");
    for (int do_touch = 0; do_touch < 2; ++do_touch) {
        const int steps = 500;
        double best = 1e30;
        for (int r = 0; r < 5; ++r) {
            CK(hipMemsetAsync(counter, 0, 4, st));
            CK(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            persistent_touch<<<cus < 256 ? cus : 256, 512, 160 * 1024, st>>>(a, b, n16, steps, counter, do_touch);
            CK(hipStreamSynchronize(st));
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            if (us < best) best = us;
        }
        printf("  %-47s : %6.2f us\n", do_touch ? "grid barrier + 4 MiB in -> 4 MiB out" : "grid barrier only", best / steps);
    }
    return 0;
}
