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

// ---- split-f16 scheme (NT_F16X3): x ~ hi + lo * 2^-11 with hi = fp16_rn(x), lo = fp16_rn((x - hi) * 2^11): 22-23 significant
// bits in two fp16 values (representation error <= 2^-22 |x|).  |x| is clamped to the fp16 range (65504) first, so an
// out-of-range activation saturates instead of turning into inf - inf; the scaled remainder is at most |x| / 2.
// PF_LO_UNSCALED (build switch, python -m perspectivefields_amd.build with PF_LO_UNSCALED=1; candidate for the next round, profiles/r02_mfma_f16_subnormals.md): the low
// plane is carried UNSCALED, lo = fp16_rn(x - hi).  The matrix cores of gfx950 keep fp16 subnormal inputs (measured), so nothing is lost below 2^-14 either: same
// accuracy over the whole network in the CPU emulation (scripts/emulate_split.py f16x3u) -- and the third weight operand wh 2^-11 (4 v_pk_mul_f16 per weight
// fragment in every split kernel) and one multiply per split element disappear.  SB_LO_SCALE / SB_LO_UNSCALE are the factors on the stored low part / on its use.
#ifdef PF_LO_UNSCALED
#define SB_LO_SCALE 1.0f
#define SB_LO_UNSCALE 1.0f
#else
#define SB_LO_SCALE 2048.f
#define SB_LO_UNSCALE 0.00048828125f
#endif
typedef _Float16 sb_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4_f16(const float4 v, uint2& h, uint2& l) {
  const float a[4] = {v.x, v.y, v.z, v.w};
  _Float16 hh[4], ll[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float c = __builtin_amdgcn_fmed3f(a[e], -65504.f, 65504.f);
    hh[e] = (_Float16)c;
    ll[e] = (_Float16)((c - (float)hh[e]) * SB_LO_SCALE);
  }
  const sb_h2 h0 = {hh[0], hh[1]}, h1 = {hh[2], hh[3]}, l0 = {ll[0], ll[1]}, l1 = {ll[2], ll[3]};
  h = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
  l = make_uint2(__builtin_bit_cast(unsigned, l0), __builtin_bit_cast(unsigned, l1));
}
// 8 fp16 values (one 16-byte piece) times 2^-11 (exact unless the result is subnormal): the third weight plane wh2 = wh * 2^-11
__device__ __forceinline__ float4 scale8_f16_2m11(const float4 v) {
#ifdef PF_LO_UNSCALED
  return v;  // the low activation plane is unscaled: the product al * wh needs no 2^-11
#endif
  const sb_h2 k = {(_Float16)0.00048828125f, (_Float16)0.00048828125f};
  auto mul = [&](float f) { return __builtin_bit_cast(float, (sb_h2)(__builtin_bit_cast(sb_h2, f) * k)); };
  return make_float4(mul(v.x), mul(v.y), mul(v.z), mul(v.w));
}

// ---- plane formats.  Every `plane` argument of the producers / consumers below is (elements between consecutive planes,
// always even) | format bit:  0 = three exact bf16 planes (x == h + m + l),  SB_FMT_F16 = two fp16 planes of the split-f16
// scheme (x ~ hi + lo 2^-11: the same two values the split-f16 GEMM would compute from the fp32 tensor while staging it,
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
  if (plane_fmt & SB_FMT_F16) {  // hi + lo 2^-11 (the value the split-f16 GEMM works with; not the original fp32 bits)
    const uint2 h = *reinterpret_cast<const uint2*>(base + idx);
    const uint2 l = *reinterpret_cast<const uint2*>(base + plane_elems + idx);
    auto f = [](unsigned w, int hi) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(hi ? (w >> 16) : (w & 0xffffu))); };
    const float s = SB_LO_UNSCALE;
    return make_float4(fmaf(f(l.x, 0), s, f(h.x, 0)), fmaf(f(l.x, 1), s, f(h.x, 1)), fmaf(f(l.y, 0), s, f(h.y, 0)), fmaf(f(l.y, 1), s, f(h.y, 1)));
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
