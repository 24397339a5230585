// Split-f16 implicit GEMM (NT_F16X3): the default parity scheme -- fp32 activations split on the fly into two fp16
// planes (a ~ ah + al 2^-11), per-channel power-of-two scaled weights as wh + wl, 3 x v_mfma_f32_32x32x16_f16 per
// element product into one fp32 accumulator.  Kernel: igemm_sb_impl.h.
#include "igemm_sb_impl.h"

namespace pf {
void launch_conv_sbf(const ConvParams& p, int sb_tile, hipStream_t s) { launch_conv_sb_nt<NT_F16X3>(p, sb_tile, s); }
}  // namespace pf
