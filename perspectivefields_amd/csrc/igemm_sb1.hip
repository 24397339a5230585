// Split-bf16 implicit GEMM, reduced-precision form with 1 partial product per element product
// (plain bf16 operands, fp32 accumulation).  Kernel: igemm_sb_impl.h.
#include "igemm_sb_impl.h"

namespace pf {
void launch_conv_sb1(const ConvParams& p, int sb_tile, hipStream_t s) { launch_conv_sb_nt<1>(p, sb_tile, s); }
}  // namespace pf
