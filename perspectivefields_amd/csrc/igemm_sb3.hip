// Split-bf16 implicit GEMM, reduced-precision form with 3 partial products per element product
// (h*h + h*m + m*h: ~16 significant bits per operand).  Kernel: igemm_sb_impl.h.
#include "igemm_sb_impl.h"

namespace pf {
void launch_conv_sb3(const ConvParams& p, int sb_tile, hipStream_t s) { launch_conv_sb_nt<3>(p, sb_tile, s); }
}  // namespace pf
