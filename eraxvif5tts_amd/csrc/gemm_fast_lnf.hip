// gemm_fast_lnf.hip -- second translation unit of gemm_fast.hip: the LayerNorm-fold instantiations of the tuned GEMM (fp16 operands,
// statistics / column constants in the epilogue), behind launch_gemm_fast_lnf().  Split off for build time only.
#define F5_LNF_TU 1
#include "gemm_fast.hip"
