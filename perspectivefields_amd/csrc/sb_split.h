// Split-bf16 activation format ("SBA"): an fp32 tensor stored as three bf16 planes h, m, l with x == h + m + l EXACTLY
// (truncation split: 8 + 8 + 8 significant bits; every remainder is exact in fp32), plane p at `base + p * plane_elems`,
// each plane laid out like the fp32 tensor it replaces (NHWC).  Producers (conv / LayerNorm / depthwise / attention /
// bilinear epilogues) write it once per element; the split-bf16 implicit GEMM (igemm_sb.hip) then stages its A operand
// with plain 16-byte copies instead of re-splitting every element once per (tap, n-tile) in its inner loop.
#pragma once
#include <hip/hip_runtime.h>

#include "pf_kernels.h"

namespace pf {

// One output of the bilinear x2 up-sampling (align_corners = False): the same expression, with the roundings pinned by
// explicit multiplies / fmas, in the stand-alone kernels (elem.hip) and in the halo staging of the fused conv (igemm_sbh.hip),
// so that fused and unfused paths give bit-identical results.
__device__ __forceinline__ float bilerp2x(float v00, float v01, float v10, float v11, float hx, float lx, float hy, float ly) {
  const float t = fmaf(lx, v01, __fmul_rn(hx, v00));
  const float b = fmaf(lx, v11, __fmul_rn(hx, v10));
  return fmaf(ly, b, __fmul_rn(hy, t));
}

// 4-term partial dot product of the regression prediction heads: the same expression (roundings pinned) in the stand-alone
// kernel (elem.hip pred_regression_kernel) and in the fused conv epilogue (igemm_common.h), so that both paths agree bitwise
__device__ __forceinline__ float head_dot4(const float4 a, const float4 w) {
  return fmaf(a.y, w.y, __fmul_rn(a.x, w.x)) + fmaf(a.w, w.w, __fmul_rn(a.z, w.z));
}

__device__ __forceinline__ unsigned sb_pack_hi16(unsigned lo_src, unsigned hi_src) { return (lo_src >> 16) | (hi_src & 0xffff0000u); }

// exact 3-way truncation split of 4 floats -> three 8-byte groups of 4 bf16
__device__ __forceinline__ void split4(const float4 v, uint2& h, uint2& m, uint2& l) {
  const float a[4] = {v.x, v.y, v.z, v.w};
  unsigned hb[4], mb[4], lb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned u = __float_as_uint(a[e]);
    hb[e] = u & 0xffff0000u;
    const float r = a[e] - __uint_as_float(hb[e]);
    mb[e] = __float_as_uint(r) & 0xffff0000u;
    const float r2 = r - __uint_as_float(mb[e]);
    lb[e] = __float_as_uint(r2);  // <= 8 significant bits left: exactly representable
  }
  h = make_uint2(sb_pack_hi16(hb[0], hb[1]), sb_pack_hi16(hb[2], hb[3]));
  m = make_uint2(sb_pack_hi16(mb[0], mb[1]), sb_pack_hi16(mb[2], mb[3]));
  l = make_uint2(sb_pack_hi16(lb[0], lb[1]), sb_pack_hi16(lb[2], lb[3]));
}

// ---- split-f16 scheme (NT_F16X3): x ~ hi + lo with hi = fp16_rn(x), lo = fp16_rn(x - hi), the low part UNSCALED (r03; r02 carried lo 2^11 and paid for it with a
// third weight operand wh 2^-11 made per fragment in registers: 72 of the 242 VALU instructions of a halo-kernel chunk).  The matrix cores of gfx950 keep fp16
// subnormal inputs (measured, profiles/r02_mfma_f16_subnormals.md), so lo stays usable below 2^-14.  Representation error: <= 2^-22 |x| for |x| >= 2^-3 (lo is a
// normal fp16 number there), <= 2^-25 ABSOLUTE below (lo is a subnormal: quantum 2^-24) -- an error of 3e-8 per element, i.e. fp32 rounding at unit scale; it is a
// RELATIVE loss only for a tensor whose every element is << 0.1 (PerspectiveFields.check_range / pf_debug_forward_u8 report such tensors; precision "fp32_bf16x6" has no such window).  |x| is clamped to
// the fp16 range (65504) first, so an out-of-range activation saturates instead of turning into inf - inf.
// Instruction count: v_med3 + (half a) v_cvt_pk_f16_f32 + one v_fma_mix{lo,hi}_f16 per element -- the mixed-precision fma takes the fp32 value and the fp16 hi part
// (fp16 source operand, negated) and rounds x - hi straight to fp16: 2.5 VALU instructions per element where the C expression compiles to 4.25 (cvt back to fp32,
// subtract, convert, pack).  x - hi is exact in fp32, so the fma rounds once, like the C expression it replaces (bit check: tests/test_gpu_ops.py::test_split_planes_bits).
typedef _Float16 sb_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_f16(const float a0, const float a1, unsigned& h, unsigned& l) {
  const float c0 = __builtin_amdgcn_fmed3f(a0, -65504.f, 65504.f), c1 = __builtin_amdgcn_fmed3f(a1, -65504.f, 65504.f);
  const sb_h2 hv = {(_Float16)c0, (_Float16)c1};
  h = __builtin_bit_cast(unsigned, hv);
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(c0), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(c1), "v"(h));
}
// A low part that feeds an MFMA straight from REGISTERS (attn.hip, attn_block.hip, stem7.hip, thin_linear.hip, the row fragments of cnx_mlp.hip / mit_mlp.hip and cnx_mlp's
// hidden map; every other user writes it to LDS first) passes
// through this.  hipcc pads the VALU-write -> MFMA-operand-read wait states (two) for the instructions IT emits; what an asm statement writes is invisible to its
// hazard pass beyond one boundary state, and a v_fma_mixhi_f16 one issue slot ahead of the MFMA that reads its register hands the matrix core the register's OLD
// contents (r06: thin128_kernel<false>, columns 0 - 31 wrong by ~0.7, found by its op test; profiles/r06_asm_mfma_hazard.md).  The s_nop sits INSIDE an asm that
// redefines the value, so whatever the scheduler does, two states separate the last v_fma_mix from the first MFMA.  Static check over the built library:
// tests/test_host_logic.py::test_no_valu_write_within_two_states_of_an_mfma_read.
template <typename V>
__device__ __forceinline__ void split_f16_mfma_pad(V& l) {
  asm("s_nop 1" : "+v"(l));
}
__device__ __forceinline__ void split4_f16(const float4 v, uint2& h, uint2& l) {
  split2_f16(v.x, v.y, h.x, l.x);
  split2_f16(v.z, v.w, h.y, l.y);
}

// ---- plane formats.  Every `plane` argument of the producers / consumers below is (elements between consecutive planes,
// always even) | format bit:  0 = three exact bf16 planes (x == h + m + l),  SB_FMT_F16 = two fp16 planes of the split-f16
// scheme (x ~ hi + lo: the same two values the split-f16 GEMM would compute from the fp32 tensor while staging it,
// 4 bytes per element like the fp32 tensor itself -- the consuming GEMM's A staging becomes a plain copy).
// (SB_FMT_F16 is declared in pf_kernels.h)

// store 4 consecutive elements (index idx, a multiple of 4) of a split-plane tensor
__device__ __forceinline__ void store_sb4(unsigned short* base, size_t plane_fmt, size_t idx, const float4 v) {
  const size_t plane_elems = plane_fmt & ~(size_t)1;
  if (plane_fmt & SB_FMT_F16) {
    uint2 h, l;
    split4_f16(v, h, l);
    *reinterpret_cast<uint2*>(base + idx) = h;
    *reinterpret_cast<uint2*>(base + plane_elems + idx) = l;
    return;
  }
  uint2 h, m, l;
  split4(v, h, m, l);
  *reinterpret_cast<uint2*>(base + idx) = h;
  *reinterpret_cast<uint2*>(base + plane_elems + idx) = m;
  *reinterpret_cast<uint2*>(base + 2 * plane_elems + idx) = l;
}

__device__ __forceinline__ float4 load_sb4(const unsigned short* base, size_t plane_fmt, size_t idx) {
  const size_t plane_elems = plane_fmt & ~(size_t)1;
  if (plane_fmt & SB_FMT_F16) {  // hi + lo (the value the split-f16 GEMM works with; not the original fp32 bits)
    const uint2 h = *reinterpret_cast<const uint2*>(base + idx);
    const uint2 l = *reinterpret_cast<const uint2*>(base + plane_elems + idx);
    auto f = [](unsigned w, int hi) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(hi ? (w >> 16) : (w & 0xffffu))); };
    return make_float4(f(l.x, 0) + f(h.x, 0), f(l.x, 1) + f(h.x, 1), f(l.y, 0) + f(h.y, 0), f(l.y, 1) + f(h.y, 1));
  }
  const uint2 h = *reinterpret_cast<const uint2*>(base + idx);
  const uint2 m = *reinterpret_cast<const uint2*>(base + plane_elems + idx);
  const uint2 l = *reinterpret_cast<const uint2*>(base + 2 * plane_elems + idx);
  auto f = [](unsigned w, int hi) { return __uint_as_float(hi ? (w & 0xffff0000u) : (w << 16)); };
  float4 v;
  v.x = (f(h.x, 0) + f(m.x, 0)) + f(l.x, 0);
  v.y = (f(h.x, 1) + f(m.x, 1)) + f(l.x, 1);
  v.z = (f(h.y, 0) + f(m.y, 0)) + f(l.y, 0);
  v.w = (f(h.y, 1) + f(m.y, 1)) + f(l.y, 1);
  return v;
}

}  // namespace pf
