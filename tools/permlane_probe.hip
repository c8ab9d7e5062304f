// prints which source lane each lane of the two results of v_permlane16_swap_b32 holds (a = lane id, b = 100 + lane id)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
__global__ void k(unsigned* o) {
    unsigned a = threadIdx.x, b = 100 + threadIdx.x;
    u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main() {
    unsigned* d;
    unsigned h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int r = 0; r < 2; ++r) {
        printf("result %d rows:", r);
        for (int row = 0; row < 4; ++row) printf(" [%u..%u]", h[r * 64 + row * 16], h[r * 64 + row * 16 + 15]);
        printf("\n");
    }
    return 0;
}
