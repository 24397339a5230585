// Split-bf16 activation format ("SBA"): an fp32 tensor stored as three bf16 planes h, m, l with x == h + m + l EXACTLY
// (truncation split: 8 + 8 + 8 significant bits; every remainder is exact in fp32), plane p at `base + p * plane_elems`,
// each plane laid out like the fp32 tensor it replaces (NHWC).  Producers (conv / LayerNorm / depthwise / attention /
// bilinear epilogues) write it once per element; the split-bf16 implicit GEMM (igemm_sb.hip) then stages its A operand
// with plain 16-byte copies instead of re-splitting every element once per (tap, n-tile) in its inner loop.
#pragma once
#include <hip/hip_runtime.h>

namespace pf {

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

// store 4 consecutive elements (index idx, a multiple of 4) of an SBA tensor
__device__ __forceinline__ void store_sb4(unsigned short* base, size_t plane_elems, size_t idx, const float4 v) {
  uint2 h, m, l;
  split4(v, h, m, l);
  *reinterpret_cast<uint2*>(base + idx) = h;
  *reinterpret_cast<uint2*>(base + plane_elems + idx) = m;
  *reinterpret_cast<uint2*>(base + 2 * plane_elems + idx) = l;
}

__device__ __forceinline__ float4 load_sb4(const unsigned short* base, size_t plane_elems, size_t idx) {
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
