// gemm_fast.hip -- tuned bf16 GEMM (placeholder until the LDS-DMA kernel lands: reports "unsupported").
#include "gemm.h"
bool gemm_fast_supported(const GemmParams&, int, int, int) { return false; }
int launch_gemm_fast(const GemmParams&, int, int, hipStream_t) { return f5_fail(F5_ENOTSUP, "tuned GEMM not built"); }
