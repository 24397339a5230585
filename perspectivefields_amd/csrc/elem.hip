// HBM-bound kernels of the PerspectiveFields path for gfx950: LayerNorm, depthwise
// 3x3(+GELU) / 7x7 with LDS-staged halo tiles, bilinear x2, prediction heads,
// post-process, input normalisation, ConvNeXt tail.  All NHWC fp32, 16-byte accesses.
#include <stdlib.h>

#include <algorithm>

#include "pf_kernels.h"
#include "sb_split.h"

namespace pf {

__device__ __forceinline__ float gelu_erf_e(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
// Branch-free erf-GELU for the HBM-bound kernels: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7) on v_rcp_f32 /
// v_exp_f32 -- ~20 VALU slots per value instead of libm erff's ~45 with divergent branches, which made the depthwise
// 3x3 kernel VALU-bound (232 VALU instructions per float4).  In fp32 the result is as close to the exact GELU as the
// libm-based form (max |error| 4.6e-7 vs 4.5e-7 over [-12, 12], tests/test_host_logic.py).
__device__ __forceinline__ float gelu_fast(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(x * x * -1.4426950408889634f);
  const float er = copysignf(fmaf(-p, e, 1.0f), v);
  return 0.5f * v * (1.0f + er);
}
typedef float f32x2e __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4e __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 fma4(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}

// ------------------------------------------------------------------------------ LayerNorm
// Reference: nn.LayerNorm in mix_transformers.py:89,160,172,224,327-387 (eps 1e-5 / 1e-6) and
// convnext.py:155-182 (both data formats reduce over C, which is the NHWC row here).
// LPR lanes cooperate on one row (wave-shuffle reductions); two-pass mean / centred variance.
template <int LPR, int VPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, float* __restrict__ y, unsigned short* __restrict__ y_sb,
                                                        size_t sb_plane, long rows, int C, float eps) {
  constexpr int RPB = 256 / LPR;
  const int sub = threadIdx.x % LPR;
  const long row = (long)blockIdx.x * RPB + threadIdx.x / LPR;
  const int nv = C >> 2;
  const bool rok = row < rows;
  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = sub + i * LPR;
    if (rok && c < nv) {
      v[i] = reinterpret_cast<const float4*>(x + row * C)[c];
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = f4(0.f);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, LPR);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = sub + i * LPR;
    if (c < nv) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, LPR);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = sub + i * LPR;
    if (rok && c < nv) {
      const float4 gg = reinterpret_cast<const float4*>(g)[c];
      const float4 bb = reinterpret_cast<const float4*>(b)[c];
      float4 o;
      o.x = (v[i].x - mean) * rstd * gg.x + bb.x;
      o.y = (v[i].y - mean) * rstd * gg.y + bb.y;
      o.z = (v[i].z - mean) * rstd * gg.z + bb.z;
      o.w = (v[i].w - mean) * rstd * gg.w + bb.w;
      if (y) reinterpret_cast<float4*>(y + row * C)[c] = o;
      if (y_sb) store_sb4(y_sb, sb_plane, (size_t)row * C + 4 * c, o);
    }
  }
}

void launch_layernorm(const float* x, const float* g, const float* b, float* y, long rows, int C, float eps, hipStream_t s,
                      unsigned short* y_sb, size_t sb_plane) {
  const int nv = C / 4;
  if (nv <= 16) {
    hipLaunchKernelGGL((layernorm_kernel<16, 1>), dim3((rows + 15) / 16), dim3(256), 0, s, x, g, b, y, y_sb, sb_plane, rows, C, eps);
  } else if (nv <= 32) {
    hipLaunchKernelGGL((layernorm_kernel<32, 1>), dim3((rows + 7) / 8), dim3(256), 0, s, x, g, b, y, y_sb, sb_plane, rows, C, eps);
  } else if (nv <= 64) {
    hipLaunchKernelGGL((layernorm_kernel<64, 1>), dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, y_sb, sb_plane, rows, C, eps);
  } else if (nv <= 128) {
    hipLaunchKernelGGL((layernorm_kernel<64, 2>), dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, y_sb, sb_plane, rows, C, eps);
  } else {
    hipLaunchKernelGGL((layernorm_kernel<64, 4>), dim3((rows + 3) / 4), dim3(256), 0, s, x, g, b, y, y_sb, sb_plane, rows, C, eps);
  }
}

// -------------------------------------------------------------------- depthwise 3x3 + GELU
// Reference: DWConv + nn.GELU inside Mlp (mix_transformers.py:49-56,497-508).
// Block = 8x8 output pixels x 128 channels.  The 10x10x128 input tile (with halo) is staged
// in LDS once (51.2 KB); thread (q = channel quad, y = row) then marches along x keeping a
// 3x3 window of float4 in registers: 3 ds_read_b128 + 9 fma4 + erf per output float4.
#ifdef PF_TUNING_BUILD
static constexpr int DW3_T = 8;
static constexpr int DW3_CQ = 32;  // channel quads per block (128 channels)

__global__ __launch_bounds__(256) void dwconv3x3_gelu_kernel(const float* __restrict__ x, const float* __restrict__ w9c,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             unsigned short* __restrict__ y_sb, size_t sb_plane,
                                                             int B, int H, int W, int C) {
  __shared__ __attribute__((aligned(16))) float4 tile[(DW3_T + 2) * (DW3_T + 2) * DW3_CQ];
  const int tilesX = (W + DW3_T - 1) / DW3_T, tilesY = (H + DW3_T - 1) / DW3_T;
  const int slabs = C / (DW3_CQ * 4);
  int bid = blockIdx.x;
  const int slab = bid % slabs; bid /= slabs;
  const int tx = bid % tilesX; bid /= tilesX;
  const int ty = bid % tilesY; bid /= tilesY;
  const int b = bid;
  const int x0 = tx * DW3_T, y0 = ty * DW3_T, cq0 = slab * DW3_CQ;
  const int CQ = C >> 2;
  const float4* xin = reinterpret_cast<const float4*>(x) + (long)b * H * W * CQ;
  for (int i = threadIdx.x; i < (DW3_T + 2) * (DW3_T + 2) * DW3_CQ; i += 256) {
    const int q = i % DW3_CQ, pix = i / DW3_CQ;
    const int py = pix / (DW3_T + 2), px = pix % (DW3_T + 2);
    const int iy = y0 + py - 1, ix = x0 + px - 1;
    float4 v = f4(0.f);
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xin[((long)iy * W + ix) * CQ + cq0 + q];
    tile[i] = v;
  }
  const int q = threadIdx.x % DW3_CQ, ry = threadIdx.x / DW3_CQ;  // ry in 0..7
  float4 wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = reinterpret_cast<const float4*>(w9c)[(long)k * CQ + cq0 + q];
  const float4 bv = reinterpret_cast<const float4*>(bias)[cq0 + q];
  __syncthreads();
  const int oy = y0 + ry;
  if (oy >= H) return;
  float4 win[3][3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    win[r][1] = tile[((ry + r) * (DW3_T + 2) + 0) * DW3_CQ + q];
    win[r][2] = tile[((ry + r) * (DW3_T + 2) + 1) * DW3_CQ + q];
  }
  const long yrow = ((long)b * H + oy) * W * CQ + cq0 + q;  // float4 index of (b, oy, 0, q)
#pragma unroll
  for (int ox = 0; ox < DW3_T; ++ox) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      win[r][0] = win[r][1];
      win[r][1] = win[r][2];
      win[r][2] = tile[((ry + r) * (DW3_T + 2) + ox + 2) * DW3_CQ + q];
    }
    float4 a = bv;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) a = fma4(win[r][c], wk[r * 3 + c], a);
    a.x = gelu_erf_e(a.x); a.y = gelu_erf_e(a.y); a.z = gelu_erf_e(a.z); a.w = gelu_erf_e(a.w);
    if (x0 + ox < W) {
      const long o4 = yrow + (long)(x0 + ox) * CQ;
      if (y) reinterpret_cast<float4*>(y)[o4] = a;
      if (y_sb) store_sb4(y_sb, sb_plane, (size_t)o4 * 4, a);
    }
  }
}

#endif  // PF_TUNING_BUILD

// Default: no LDS -- thread (channel quad q, column x) marches down a strip of rows keeping a 3x3 window of
// float4 in registers and fetching 3 new values per output straight from global memory; the x-1/x/x+1 overlap
// between neighbouring lanes/waves is served by L1/L2, HBM sees each input once (plus strip halos).
template <int CQB /*quads per block*/, int XB /*columns per block*/, int TH /*rows per strip*/, int DIAG = 0 /*1: no stores, 2: no loads (diagnostics)*/>
__global__ __launch_bounds__(CQB * XB) void dwconv3x3_gelu_direct_kernel(const float* __restrict__ x, const float* __restrict__ w9c,
                                                                         const float* __restrict__ bias, float* __restrict__ y,
                                                                         unsigned short* __restrict__ y_sb, size_t sb_plane,
                                                                         int B, int H, int W, int C) {
  const int CQ = C >> 2;
  const int slabs = CQ / CQB, tilesX = (W + XB - 1) / XB, strips = (H + TH - 1) / TH;
  const int nblk = B * strips * tilesX * slabs;
  int t;
  {  // XCD-aware order: consecutive work items (slabs of a tile, then x-neighbours) share one L2
    const int b = blockIdx.x, qd = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    t = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int slab = t % slabs; t /= slabs;
  const int tx = t % tilesX; t /= tilesX;
  const int st = t % strips; t /= strips;
  const int b = t;
  const int q = slab * CQB + threadIdx.x % CQB;
  const int ox = tx * XB + threadIdx.x / CQB;
  if (ox >= W) return;
  const int y0 = st * TH, y1 = min(y0 + TH, H);
  const float4* xin = reinterpret_cast<const float4*>(x) + (long)b * H * W * CQ + q;
  const long ybase = (long)b * H * W * CQ + q;  // float4 index of (b, 0, 0, q)
  float4 wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = reinterpret_cast<const float4*>(w9c)[(long)k * CQ + q];
  const float4 bv = reinterpret_cast<const float4*>(bias)[q];
  const bool lx = ox > 0, rx = ox < W - 1;
  auto load_row = [&](int iy, float4 (&row)[3]) {
    if (DIAG == 2) {
      row[0] = row[1] = row[2] = f4((float)(iy & 7) * 0.125f);
    } else if ((unsigned)iy < (unsigned)H) {
      const float4* p = xin + ((long)iy * W + ox) * CQ;
      row[0] = lx ? p[-CQ] : f4(0.f);
      row[1] = p[0];
      row[2] = rx ? p[CQ] : f4(0.f);
    } else {
      row[0] = row[1] = row[2] = f4(0.f);
    }
  };
  float4 r0[3], r1[3], r2[3];
  load_row(y0 - 1, r0);
  load_row(y0, r1);
  // the 9 taps as 18 v_pk_fma_f32 (a float4 = two aligned register pairs)
  auto lo = [](const float4& v) { return f32x2e{v.x, v.y}; };
  auto hi = [](const float4& v) { return f32x2e{v.z, v.w}; };
  auto emit = [&](int oy, const float4 (&t)[3], const float4 (&m)[3], const float4 (&b)[3]) {
    f32x2e a0 = lo(bv), a1 = hi(bv);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a0 = __builtin_elementwise_fma(lo(t[c]), lo(wk[c]), a0);     a1 = __builtin_elementwise_fma(hi(t[c]), hi(wk[c]), a1);
      a0 = __builtin_elementwise_fma(lo(m[c]), lo(wk[3 + c]), a0); a1 = __builtin_elementwise_fma(hi(m[c]), hi(wk[3 + c]), a1);
      a0 = __builtin_elementwise_fma(lo(b[c]), lo(wk[6 + c]), a0); a1 = __builtin_elementwise_fma(hi(b[c]), hi(wk[6 + c]), a1);
    }
    const float4 a = make_float4(gelu_fast(a0.x), gelu_fast(a0.y), gelu_fast(a1.x), gelu_fast(a1.y));
    const long o4 = ybase + ((long)oy * W + ox) * CQ;
    if (DIAG == 1 && a.x != 123456.789f) return;  // keeps the arithmetic alive, stores nothing
    if (y) reinterpret_cast<float4*>(y)[o4] = a;
    if (y_sb) store_sb4(y_sb, sb_plane, (size_t)o4 * 4, a);
  };
  // three rows per iteration: the window rows rotate by name, not by register moves
  int oy = y0;
  for (; oy + 3 <= y1; oy += 3) {
    load_row(oy + 1, r2); emit(oy, r0, r1, r2);
    load_row(oy + 2, r0); emit(oy + 1, r1, r2, r0);
    load_row(oy + 3, r1); emit(oy + 2, r2, r0, r1);
  }
  if (oy < y1) {
    load_row(oy + 1, r2); emit(oy, r0, r1, r2);
    if (oy + 1 < y1) { load_row(oy + 2, r0); emit(oy + 1, r1, r2, r0); }
  }
}

// Multi-column form with a row prefetch.  The kernel above keeps 3 x 16 bytes per lane in flight and then waits for them
// (the row it loads is the row it needs next): with 16 waves per CU that is ~49 KB in flight per CU, i.e. by Little's law
// ~4.8 TB/s at the ~2.5 us loaded HBM latency -- exactly what its loads-only ablation measures.  Here a thread owns NC adjacent
// columns of one channel quad (NC + 2 loads per row for NC outputs instead of 3 per output) and loads row oy + 1 + PF while it
// computes row oy from rows loaded earlier: (1 + PF) x (NC + 2) x 16 bytes per lane in flight.  Same tap order as above:
// bit-identical results.  Row buffers rotate by name (the loop is unrolled over the 3 + PF buffers).
template <int CQB /*quads per block*/, int XB /*threads along x*/, int TH /*rows per strip*/, int NC /*columns per thread*/, int PF /*rows loaded ahead*/, bool SB = false /*split-plane output (y_sb) instead of / next to fp32*/>
__global__ __launch_bounds__(CQB * XB, (NC + 2) * (3 + PF) < 16 ? 4 : 3) void dwconv3x3_gelu_mc_kernel(const float* __restrict__ x, const float* __restrict__ w9c,
                                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                                     unsigned short* __restrict__ y_sb, size_t sb_plane,
                                                                     int B, int H, int W, int C) {
  constexpr int NB = 3 + PF, NL = NC + 2;
  const int CQ = C >> 2;
  const int slabs = CQ / CQB, tilesX = (W + XB * NC - 1) / (XB * NC), strips = (H + TH - 1) / TH;
  const int nblk = B * strips * tilesX * slabs;
  int t;
  {
    const int b = blockIdx.x, qd = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    t = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int slab = t % slabs; t /= slabs;
  const int tx = t % tilesX; t /= tilesX;
  const int st = t % strips; t /= strips;
  const int b = t;
  const int q = slab * CQB + threadIdx.x % CQB;
  const int ox = (tx * XB + threadIdx.x / CQB) * NC;
  if (ox >= W) return;
  const int y0 = st * TH, y1 = min(y0 + TH, H);
  // one buffer descriptor per image (an image of the largest hidden map is 6.5 MB): 32-bit byte offsets, a column outside the map
  // or a row outside it / past the strip's halo row gets an out-of-range offset (loads return zeros, stores are dropped) -- no
  // 64-bit address arithmetic and no branch around any memory instruction
  const unsigned img_bytes = (unsigned)H * W * C * 4u;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) + (size_t)b * H * W * C, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((SB && !y) ? const_cast<float*>(x) : y + (size_t)b * H * W * C, 0, (SB && !y) ? 0 : img_bytes, 0x00020000);
  const long ybase = (long)b * H * W * CQ;  // float4 index of (b, 0, 0, 0): split-plane output only
  float4 wk[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) wk[k] = reinterpret_cast<const float4*>(w9c)[(long)k * CQ + q];
  const float4 bv = reinterpret_cast<const float4*>(bias)[q];
  unsigned coff[NL];  // byte offset of (row 0, column ox - 1 + j, quad q); 2^30 (>= any image) when the column is outside the map
#pragma unroll
  for (int j = 0; j < NL; ++j) coff[j] = (unsigned)(ox - 1 + j) < (unsigned)W ? (unsigned)((ox - 1 + j) * CQ + q) * 16u : 0x40000000u;
  const unsigned row_bytes = (unsigned)W * C * 4u;
  auto load_row = [&](int iy, float4 (&row)[NL]) {
    const unsigned rb = ((unsigned)iy < (unsigned)H && iy <= y1) ? (unsigned)iy * row_bytes : 0x80000000u;  // block-uniform
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const u32x4e v = __builtin_amdgcn_raw_buffer_load_b128(rx, rb + coff[j], 0, 0);
      row[j] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
  };
  auto lo = [](const float4& v) { return f32x2e{v.x, v.y}; };
  auto hi = [](const float4& v) { return f32x2e{v.z, v.w}; };
  auto emit = [&](int oy, const float4 (&tr)[NL], const float4 (&mr)[NL], const float4 (&br)[NL]) {
    const unsigned rb = (unsigned)oy * row_bytes;
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      f32x2e a0 = lo(bv), a1 = hi(bv);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        a0 = __builtin_elementwise_fma(lo(tr[k + c]), lo(wk[c]), a0);     a1 = __builtin_elementwise_fma(hi(tr[k + c]), hi(wk[c]), a1);
        a0 = __builtin_elementwise_fma(lo(mr[k + c]), lo(wk[3 + c]), a0); a1 = __builtin_elementwise_fma(hi(mr[k + c]), hi(wk[3 + c]), a1);
        a0 = __builtin_elementwise_fma(lo(br[k + c]), lo(wk[6 + c]), a0); a1 = __builtin_elementwise_fma(hi(br[k + c]), hi(wk[6 + c]), a1);
      }
      const float4 a = make_float4(gelu_fast(a0.x), gelu_fast(a0.y), gelu_fast(a1.x), gelu_fast(a1.y));
      if (!SB || y) {
        const u32x4e v = {__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w)};
        __builtin_amdgcn_raw_buffer_store_b128(v, ry, rb + coff[k + 1], 0, 0);  // column outside the map: dropped
      }
      if (SB && coff[k + 1] != 0x40000000u) store_sb4(y_sb, sb_plane, (size_t)(ybase + ((long)oy * W + ox + k) * CQ + q) * 4, a);
    }
  };
  float4 R[NB][NL];
#pragma unroll
  for (int k = 0; k < 2 + PF; ++k) load_row(y0 - 1 + k, R[k]);  // rows y0 - 1 .. y0 + PF
  int oy = y0;
  for (; oy + NB <= y1; oy += NB) {  // whole groups: no branch around any load
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      load_row(oy + u + 1 + PF, R[(u + 2 + PF) % NB]);
      __builtin_amdgcn_sched_barrier(0);  // keep the issue order: hipcc otherwise hoists the loads of the whole group (and spills)
      emit(oy + u, R[u % NB], R[(u + 1) % NB], R[(u + 2) % NB]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll 1
  for (; oy < y1; ++oy) {  // fewer than 3 + PF rows left: rolled, buffers rotate by register moves
    load_row(oy + 1 + PF, R[NB - 1]);
    emit(oy, R[0], R[1], R[2]);
#pragma unroll
    for (int k = 0; k + 1 < NB; ++k)
#pragma unroll
      for (int j = 0; j < NL; ++j) R[k][j] = R[k + 1][j];
  }
}

#ifdef PF_TUNING_BUILD
__global__ __launch_bounds__(256) void copy_f4_kernel(const float4* __restrict__ x, float4* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = x[i];
}
#endif

template <int CQB, int XB, int TH, int NC, int PF>
static bool launch_dw3_mc(const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s, unsigned short* y_sb, size_t sb_plane) {
  const int CQ = C / 4;
  if (CQ % CQB != 0 || W % (XB * NC) != 0) return false;
  const long blocks = (long)B * ((H + TH - 1) / TH) * (W / (XB * NC)) * (CQ / CQB);
  if (y_sb) hipLaunchKernelGGL((dwconv3x3_gelu_mc_kernel<CQB, XB, TH, NC, PF, true>), dim3((unsigned)blocks), dim3(CQB * XB), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
  else      hipLaunchKernelGGL((dwconv3x3_gelu_mc_kernel<CQB, XB, TH, NC, PF, false>), dim3((unsigned)blocks), dim3(CQB * XB), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
  return true;
}
// code = 100 shape + 10 strip + np.  shape (quads x x-threads per block): 0 = 32 x 8, 1 = 64 x 4, 2 = 64 x 2, 3 = 64 x 5; strip: 8 / 16 / 40 rows;
// np (columns per thread, rows loaded ahead): 0 (1, 1), 1 (2, 0), 2 (2, 1), 3 (2, 2), 4 (1, 2).  The product build carries the four forms the
// sweeps picked (profiles/r02_tune_dw3_mc.txt); everything else is a tuning build (PF_TUNING_BUILD=1) -- unknown codes return false (-> default).
#define PF_DW3_ARGS x, w9c, bias, y, B, H, W, C, s, y_sb, sb_plane
#ifdef PF_TUNING_BUILD
template <int CQB, int XB, int TH>
static bool launch_dw3_mc_np(int np, const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s, unsigned short* y_sb, size_t sb_plane) {
  switch (np) {
    case 0: return launch_dw3_mc<CQB, XB, TH, 1, 1>(PF_DW3_ARGS);
    case 1: return launch_dw3_mc<CQB, XB, TH, 2, 0>(PF_DW3_ARGS);
    case 2: return launch_dw3_mc<CQB, XB, TH, 2, 1>(PF_DW3_ARGS);
    case 3: return launch_dw3_mc<CQB, XB, TH, 2, 2>(PF_DW3_ARGS);
    case 4: return launch_dw3_mc<CQB, XB, TH, 1, 2>(PF_DW3_ARGS);
    default: return false;
  }
}
template <int CQB, int XB>
static bool launch_dw3_mc_th(int th, int np, const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s, unsigned short* y_sb, size_t sb_plane) {
  switch (th) {
    case 0: return launch_dw3_mc_np<CQB, XB, 8>(np, PF_DW3_ARGS);
    case 1: return launch_dw3_mc_np<CQB, XB, 16>(np, PF_DW3_ARGS);
    case 2: return launch_dw3_mc_np<CQB, XB, 40>(np, PF_DW3_ARGS);
    default: return false;
  }
}
#endif
static bool launch_dw3_mc_variant(int code, const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s, unsigned short* y_sb, size_t sb_plane) {
#ifdef PF_TUNING_BUILD
  const int shape = code / 100, th = (code / 10) % 10, np = code % 10;
  switch (shape) {
    case 0: return launch_dw3_mc_th<32, 8>(th, np, PF_DW3_ARGS);
    case 1: return launch_dw3_mc_th<64, 4>(th, np, PF_DW3_ARGS);
    case 2: return launch_dw3_mc_th<64, 2>(th, np, PF_DW3_ARGS);
    case 3: return launch_dw3_mc_th<64, 5>(th, np, PF_DW3_ARGS);
    default: return false;
  }
#else
  switch (code) {
    case 223: return launch_dw3_mc<64, 2, 40, 2, 2>(PF_DW3_ARGS);  // 80^2 maps: 4 columns x 64 quads per 128-thread block, two rows ahead
    case 100: return launch_dw3_mc<64, 4, 8, 1, 1>(PF_DW3_ARGS);   // 20^2
    case 102: return launch_dw3_mc<64, 4, 8, 2, 1>(PF_DW3_ARGS);   // 40^2 (in the pipeline 46.6 vs 51.5 us for the one-column form, profiles/r02_tune_dw3_mc.txt)
    case 314: return launch_dw3_mc<64, 5, 16, 1, 2>(PF_DW3_ARGS);  // 10^2
    default: return false;
  }
#endif
}

static int g_dw3_variant = -1;
void launch_dwconv3x3_gelu_variant(int variant, const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s,
                                   unsigned short* y_sb, size_t sb_plane) {
  const int CQ = C / 4;
#ifdef PF_TUNING_BUILD  // measured-and-rejected variants + diagnostics: tuning builds only (PF_TUNING_BUILD=1 python -m perspectivefields_amd.build)
  if (variant == 0) {
    const int tilesX = (W + DW3_T - 1) / DW3_T, tilesY = (H + DW3_T - 1) / DW3_T;
    const long blocks = (long)B * tilesY * tilesX * (C / (DW3_CQ * 4));
    hipLaunchKernelGGL(dwconv3x3_gelu_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
    return;
  } else if (variant == 1) {  // 32 quads x 8 columns, strips of 16 rows
    const long blocks = (long)B * ((H + 15) / 16) * ((W + 7) / 8) * (CQ / 32);
    hipLaunchKernelGGL((dwconv3x3_gelu_direct_kernel<32, 8, 16>), dim3((unsigned)blocks), dim3(256), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
    return;
  } else if (variant == 3) {  // 32 quads x 8 columns, strips of 40 rows
    const long blocks = (long)B * ((H + 39) / 40) * ((W + 7) / 8) * (CQ / 32);
    hipLaunchKernelGGL((dwconv3x3_gelu_direct_kernel<32, 8, 40>), dim3((unsigned)blocks), dim3(256), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
    return;
  } else if (variant == 51 || variant == 52) {  // diagnostics of variant 3: 51 = no stores, 52 = no loads
    const long blocks = (long)B * ((H + 39) / 40) * ((W + 7) / 8) * (CQ / 32);
    if (variant == 51) hipLaunchKernelGGL((dwconv3x3_gelu_direct_kernel<32, 8, 40, 1>), dim3((unsigned)blocks), dim3(256), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
    else               hipLaunchKernelGGL((dwconv3x3_gelu_direct_kernel<32, 8, 40, 2>), dim3((unsigned)blocks), dim3(256), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
    return;
  } else if (variant == 99) {  // plain copy of the same bytes (achievable streaming ceiling, diagnostic only)
    const long n = (long)B * H * W * CQ;
    hipLaunchKernelGGL(copy_f4_kernel, dim3(256 * 16), dim3(256), 0, s, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), n);
    return;
  }
#endif
  if (variant >= 1000) {  // multi-column / prefetching kernel: 1000 + 100 block shape + 10 strip height + (columns, prefetch) code
    if (launch_dw3_mc_variant(variant - 1000, x, w9c, bias, y, B, H, W, C, s, y_sb, sb_plane)) return;
    variant = H >= 16 ? 4 : 2;
  }
  if (variant == 2 && CQ % 64 == 0) {  // 64 quads x 4 columns, strips of 16 rows (10^2 maps)
    const long blocks = (long)B * ((H + 15) / 16) * ((W + 3) / 4) * (CQ / 64);
    hipLaunchKernelGGL((dwconv3x3_gelu_direct_kernel<64, 4, 16>), dim3((unsigned)blocks), dim3(256), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
  } else {  // 4 (default): 32 quads x 8 columns, strips of 8 rows
    const long blocks = (long)B * ((H + 7) / 8) * ((W + 7) / 8) * (CQ / 32);
    hipLaunchKernelGGL((dwconv3x3_gelu_direct_kernel<32, 8, 8>), dim3((unsigned)blocks), dim3(256), 0, s, x, w9c, bias, y, y_sb, sb_plane, B, H, W, C);
  }
}

void launch_dwconv3x3_gelu(const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s,
                           unsigned short* y_sb, size_t sb_plane) {
  if (g_dw3_variant == -1) {
    const char* e = getenv("PF_DW3_VARIANT");
    g_dw3_variant = e ? atoi(e) : -2;
  }
  // default: register-window kernel; strip height / block shape by map size (scripts/tune_dw.py, profiles/r01_tune_dw_v2.txt:
  // 8-row strips win or tie at 80^2 .. 20^2 now that the GELU is branch-free; 64-quad blocks at 10^2)
  // r02: the multi-column / prefetching kernel where the map fits one of its block shapes (sweep: profiles/r02_tune_dw3_mc.txt), else as before
  int v = g_dw3_variant >= 0 ? g_dw3_variant : (H >= 16 ? 4 : 2);
  if (g_dw3_variant < 0 && (C / 4) % 64 == 0) {
    if (W >= 64 && W % 4 == 0) v = 1223;
    else if (W >= 32 && W % 8 == 0) v = 1102;
    else if (W >= 16 && W % 4 == 0) v = 1100;
    else if (W % 5 == 0) v = 1314;
  }
  launch_dwconv3x3_gelu_variant(v, x, w9c, bias, y, B, H, W, C, s, y_sb, sb_plane);
}

// ------------------------------------------------------------------------- depthwise 7x7
// Reference: ConvNeXt Block.dwconv (convnext.py:30-32,48).  Block = 8x8 outputs x 96 channels
// (24 quads; all ConvNeXt-T widths are multiples of 96), 192 threads.  14x14x96 halo tile in
// LDS (75 KB) + the 49x96 weights (18.8 KB); thread (q, row) accumulates its 8 outputs while
// streaming the 7 input rows: each staged value is read once per thread and reused for up
// to 7 outputs.
#ifdef PF_TUNING_BUILD
static constexpr int DW7_T = 8;
static constexpr int DW7_CQ = 24;
static constexpr int DW7_IN = DW7_T + 6;

__global__ __launch_bounds__(192) void dwconv7x7_kernel(const float* __restrict__ x, const float* __restrict__ w49c,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        int B, int H, int W, int C) {
  __shared__ __attribute__((aligned(16))) float4 tile[DW7_IN * DW7_IN * DW7_CQ];
  __shared__ __attribute__((aligned(16))) float4 wt[49 * DW7_CQ];
  const int tilesX = (W + DW7_T - 1) / DW7_T, tilesY = (H + DW7_T - 1) / DW7_T;
  const int slabs = C / (DW7_CQ * 4);
  int bid = blockIdx.x;
  const int slab = bid % slabs; bid /= slabs;
  const int tx = bid % tilesX; bid /= tilesX;
  const int ty = bid % tilesY; bid /= tilesY;
  const int b = bid;
  const int x0 = tx * DW7_T, y0 = ty * DW7_T, cq0 = slab * DW7_CQ;
  const int CQ = C >> 2;
  const float4* xin = reinterpret_cast<const float4*>(x) + (long)b * H * W * CQ;
  for (int i = threadIdx.x; i < DW7_IN * DW7_IN * DW7_CQ; i += 192) {
    const int q = i % DW7_CQ, pix = i / DW7_CQ;
    const int py = pix / DW7_IN, px = pix % DW7_IN;
    const int iy = y0 + py - 3, ix = x0 + px - 3;
    float4 v = f4(0.f);
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xin[((long)iy * W + ix) * CQ + cq0 + q];
    tile[i] = v;
  }
  for (int i = threadIdx.x; i < 49 * DW7_CQ; i += 192) {
    const int q = i % DW7_CQ, k = i / DW7_CQ;
    wt[i] = reinterpret_cast<const float4*>(w49c)[(long)k * CQ + cq0 + q];
  }
  __syncthreads();
  const int q = threadIdx.x % DW7_CQ, ry = threadIdx.x / DW7_CQ;  // ry 0..7
  const int oy = y0 + ry;
  if (oy >= H) return;
  const float4 bv = reinterpret_cast<const float4*>(bias)[cq0 + q];
  float4 acc[DW7_T];
#pragma unroll
  for (int o = 0; o < DW7_T; ++o) acc[o] = bv;
#pragma unroll 1
  for (int ky = 0; ky < 7; ++ky) {  // rolled: keeps only one weight row (7 float4) live
    float4 wr[7];
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) wr[kx] = wt[(ky * 7 + kx) * DW7_CQ + q];
#pragma unroll
    for (int ix = 0; ix < DW7_IN; ++ix) {
      const float4 v = tile[((ry + ky) * DW7_IN + ix) * DW7_CQ + q];
#pragma unroll
      for (int o = 0; o < DW7_T; ++o) {
        const int kx = ix - o;
        if (kx >= 0 && kx < 7) acc[o] = fma4(v, wr[kx], acc[o]);
      }
    }
  }
  float4* yout = reinterpret_cast<float4*>(y) + ((long)b * H + oy) * W * CQ + cq0 + q;
#pragma unroll
  for (int o = 0; o < DW7_T; ++o)
    if (x0 + o < W) yout[(long)(x0 + o) * CQ] = acc[o];
}

// Register/ring variant: thread (channel quad q, column x) streams INPUT rows of a strip.  Its 49 per-channel
// weights stay in registers (196 VGPRs: one wave per SIMD, the whole unified register file), the next input row
// (7 float4 from global/L1: x-3..x+3) is prefetched while the current one is scattered into a ring of 7 output-row
// accumulators (tap ky feeds the row 6-ky ahead); the oldest accumulator is complete after each row and is stored.
// No LDS, no block-wide sync.  Measured 1.27 ms per B=32 forward vs 1.44 ms for the halo-tile kernel; variants with
// LDS-resident weights / two columns per thread spill under hipcc and are slower (2.3 ms) -- see DESIGN.md.
template <int CQB, int XB, int TH>
__global__ __launch_bounds__(CQB * XB) void dwconv7x7_ring_kernel(const float* __restrict__ x, const float* __restrict__ w49c,
                                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                                  int B, int H, int W, int C) {
  const int CQ = C >> 2;
  const int slabs = CQ / CQB, tilesX = (W + XB - 1) / XB, strips = (H + TH - 1) / TH;
  const int nblk = B * strips * tilesX * slabs;
  int t;
  {
    const int b = blockIdx.x, qd = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    t = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int slab = t % slabs; t /= slabs;
  const int tx = t % tilesX; t /= tilesX;
  const int st = t % strips; t /= strips;
  const int b = t;
  const int q = slab * CQB + threadIdx.x % CQB;
  const int ox = tx * XB + threadIdx.x / CQB;
  if (ox >= W) return;
  const int y0 = st * TH, y1 = min(y0 + TH, H);
  const float4* xin = reinterpret_cast<const float4*>(x) + (long)b * H * W * CQ + q;
  float4* yout = reinterpret_cast<float4*>(y) + (long)b * H * W * CQ + q;
  const float4 bv = reinterpret_cast<const float4*>(bias)[q];
  float4 wk[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) wk[k] = reinterpret_cast<const float4*>(w49c)[(long)k * CQ + q];
  auto load_row = [&](int iy, float4 (&row)[7]) {
    const bool rowok = (unsigned)iy < (unsigned)H;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) {
      const int ix = ox + kx - 3;
      row[kx] = (rowok && (unsigned)ix < (unsigned)W) ? xin[((long)iy * W + ix) * CQ] : f4(0.f);
    }
  };
  // acc[r] belongs to output row (iy - 3 + r) while input row iy is processed: tap ky feeds acc[6 - ky];
  // afterwards acc[0] is complete, the ring rotates by one slot and acc[6] restarts from the bias.
  float4 acc[7], cur[7], nxt[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[j] = bv;
  load_row(y0 - 3, cur);
#pragma unroll 1
  for (int iy = y0 - 3; iy < y1 + 3; ++iy) {
    load_row(iy + 1 < y1 + 3 ? iy + 1 : -1, nxt);  // prefetch (row -1 = zeros, unused)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      float4 a = acc[6 - ky];
#pragma unroll
      for (int kx = 0; kx < 7; ++kx) a = fma4(cur[kx], wk[ky * 7 + kx], a);
      acc[6 - ky] = a;
    }
    const int oy = iy - 3;
    if (oy >= y0 && oy < y1) yout[((long)oy * W + ox) * CQ] = acc[0];
#pragma unroll
    for (int j = 0; j < 6; ++j) acc[j] = acc[j + 1];
    acc[6] = bv;
#pragma unroll
    for (int j = 0; j < 7; ++j) cur[j] = nxt[j];
  }
}


#endif  // PF_TUNING_BUILD

// The default kernels (column-blocked / one column per lane streaming kernels) live in dw7.hip.
void launch_dwconv7x7_lane(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s);
void launch_dwconv7x7_cb(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s);

void launch_dwconv7x7_cb_cfg(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int nc, int nb, int th, hipStream_t s);
bool dwconv7x7_lds_ok(int H, int W, int C);
void launch_dwconv7x7_lds(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int th, hipStream_t s);
// packed-fp32 forms (dw7_pk.hip)
void launch_dwconv7x7_cbp_cfg(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int nc, int nb, int th, hipStream_t s);
bool launch_dwconv7x7_ldsp(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int ch, int th, hipStream_t s);
static int g_dw7_variant = -1, g_dw7_pk_ch = 0, g_dw7_pk_th = 0, g_dw7_pk_wmax = 20, g_dw7_pk_onlyw = 0;
// explicit variant / column-blocked configuration (tests, tuning); variant < 0: the default path
void launch_dwconv7x7_cfg(int variant, int nc, int nb, int th, const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s) {
  if (variant == 6 && C % 32 == 0 && launch_dwconv7x7_ldsp(x, w49c, bias, y, B, H, W, C, nc /*channels per block*/, th, s)) return;  // packed LDS-tile kernel; shapes it does not cover fall through
  if ((variant == 5 || variant == 6) && C % 32 == 0) { launch_dwconv7x7_cbp_cfg(x, w49c, bias, y, B, H, W, C, variant == 5 ? nc : 0, variant == 5 ? nb : 0, variant == 5 ? th : 0, s); return; }
  if (variant == 4 && dwconv7x7_lds_ok(H, W, C)) { launch_dwconv7x7_lds(x, w49c, bias, y, B, H, W, C, th, s); return; }
  if (variant == 3 || variant == 4) { launch_dwconv7x7_cb_cfg(x, w49c, bias, y, B, H, W, C, nc, nb, th, s); return; }
  if (variant == 2) { launch_dwconv7x7_lane(x, w49c, bias, y, B, H, W, C, s); return; }
  launch_dwconv7x7(x, w49c, bias, y, B, H, W, C, s);
}
void launch_dwconv7x7(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s) {
  if (g_dw7_variant == -1) {
    const char* e = getenv("PF_DW7_VARIANT");
    g_dw7_variant = e ? atoi(e) : 7;  // 7 (default since r04): the packed-fp32 kernels of dw7_pk.hip -- tile-in-parts LDS kernel on maps of <= 20 columns, streaming kernel otherwise; 4: their scalar forms (dw7.hip: LDS-tile / column-blocked); 3: column-blocked everywhere; 2: one column per lane; 1: ring, 0: LDS halo tile (tuning builds)
    const char* ch = getenv("PF_DW7_PK_CH"); g_dw7_pk_ch = ch ? atoi(ch) : 0;
    const char* th = getenv("PF_DW7_PK_TH"); g_dw7_pk_th = th ? atoi(th) : 0;
    const char* ow = getenv("PF_DW7_PK_ONLYW"); g_dw7_pk_onlyw = ow ? atoi(ow) : 0;  // diagnosis: packed kernels only on maps of this width
    const char* wm = getenv("PF_DW7_PK_WMAX"); g_dw7_pk_wmax = wm ? atoi(wm) : 20;  // widest map on the packed tile kernel (wider ones stream)
  }
  if (g_dw7_variant == 7 && C % 32 == 0 && (g_dw7_pk_onlyw == 0 || W == g_dw7_pk_onlyw)) {
    if (W <= g_dw7_pk_wmax && launch_dwconv7x7_ldsp(x, w49c, bias, y, B, H, W, C, g_dw7_pk_ch, g_dw7_pk_th, s)) return;
    if (!dwconv7x7_lds_ok(H, W, C)) { launch_dwconv7x7_cbp_cfg(x, w49c, bias, y, B, H, W, C, 0, 0, 0, s); return; }
    launch_dwconv7x7_lds(x, w49c, bias, y, B, H, W, C, 0, s);  // small maps the packed tile kernel does not cover (ragged strips, < 8 columns)
    return;
  }
  if ((g_dw7_variant == 4 || g_dw7_variant == 7) && dwconv7x7_lds_ok(H, W, C)) { launch_dwconv7x7_lds(x, w49c, bias, y, B, H, W, C, 0, s); return; }
  if (g_dw7_variant == 3 || g_dw7_variant == 4 || g_dw7_variant == 7) { launch_dwconv7x7_cb(x, w49c, bias, y, B, H, W, C, s); return; }
  if (g_dw7_variant == 2) { launch_dwconv7x7_lane(x, w49c, bias, y, B, H, W, C, s); return; }
#ifdef PF_TUNING_BUILD
  const int CQ = C / 4;
  if (g_dw7_variant == 0) {  // LDS halo-tile kernel (kept for A/B)
    const int tilesX = (W + DW7_T - 1) / DW7_T, tilesY = (H + DW7_T - 1) / DW7_T;
    const long blocks = (long)B * tilesY * tilesX * (C / (DW7_CQ * 4));
    hipLaunchKernelGGL(dwconv7x7_kernel, dim3((unsigned)blocks), dim3(192), 0, s, x, w49c, bias, y, B, H, W, C);
  } else if (H >= 40) {
    const long blocks = (long)B * ((H + 39) / 40) * ((W + 31) / 32) * (CQ / 8);
    hipLaunchKernelGGL((dwconv7x7_ring_kernel<8, 32, 40>), dim3((unsigned)blocks), dim3(256), 0, s, x, w49c, bias, y, B, H, W, C);
  } else {
    const long blocks = (long)B * ((H + 19) / 20) * ((W + 7) / 8) * (CQ / 24);
    hipLaunchKernelGGL((dwconv7x7_ring_kernel<24, 8, 20>), dim3((unsigned)blocks), dim3(192), 0, s, x, w49c, bias, y, B, H, W, C);
  }
#else
  launch_dwconv7x7_cb(x, w49c, bias, y, B, H, W, C, s);  // variants 0 / 1 (LDS halo tile, ring) exist in tuning builds only
#endif
}

// --------------------------------------------------------------------------- bilinear x2
// Reference: F.interpolate(scale_factor=2, mode="bilinear", align_corners=False)
// (decode_head.py:284-286; gravity_head.py:172).  src = (dst + 0.5) * 0.5 - 0.5 clamped at 0.
__global__ __launch_bounds__(256) void upsample2x_kernel(const float4* __restrict__ x, float4* __restrict__ y, unsigned short* __restrict__ y_sb,
                                                         size_t sb_plane, int B, int H, int W, int CQ) {
  const int Ho = 2 * H, Wo = 2 * W;
  const long total = (long)B * Ho * Wo * CQ;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int q = (int)(i % CQ);
    long r = i / CQ;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    float sy = ((float)oy + 0.5f) * 0.5f - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = ((float)ox + 0.5f) * 0.5f - 0.5f; sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float4* base = x + (long)b * H * W * CQ + q;
    const float4 v00 = base[((long)y0 * W + x0) * CQ], v01 = base[((long)y0 * W + x1) * CQ];
    const float4 v10 = base[((long)y1 * W + x0) * CQ], v11 = base[((long)y1 * W + x1) * CQ];
    float4 o;
    o.x = bilerp2x(v00.x, v01.x, v10.x, v11.x, hx, lx, hy, ly);
    o.y = bilerp2x(v00.y, v01.y, v10.y, v11.y, hx, lx, hy, ly);
    o.z = bilerp2x(v00.z, v01.z, v10.z, v11.z, hx, lx, hy, ly);
    o.w = bilerp2x(v00.w, v01.w, v10.w, v11.w, hx, lx, hy, ly);
    if (y) y[i] = o;
    if (y_sb) store_sb4(y_sb, sb_plane, (size_t)i * 4, o);
  }
}

// ---------------------------------------------------------- fp32 <-> split-bf16 planes (sb_split.h), 4 elements per thread
__global__ __launch_bounds__(256) void split_planes_kernel(const float4* __restrict__ x, unsigned short* __restrict__ y_sb, size_t sb_plane, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) store_sb4(y_sb, sb_plane, (size_t)i * 4, x[i]);
}
__global__ __launch_bounds__(256) void merge_planes_kernel(const unsigned short* __restrict__ x_sb, size_t sb_plane, float4* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) y[i] = load_sb4(x_sb, sb_plane, (size_t)i * 4);
}
// uniform [-scale, scale) pseudo-random fill (benchmark inputs: never time kernels on zeros, DVFS clocks them higher)
__global__ __launch_bounds__(256) void fill_random_kernel(float* __restrict__ p, long n, unsigned seed, float scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    p[i] = ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale;
  }
}
// Range record of one tensor (debug forward, DebugSink in engine.hip): out[0] = max |x| (atomicMax on the bit pattern of a non-negative float), out[1] = sum x^2,
// out[2] = elements with |x| > 65504 (saturate in the split-f16 scheme), out[3] = non-finite elements -- both COUNTS as unsigned integers in the float slots (float
// atomics would stop counting exactly at 2^24; pf_debug_ranges converts them for the caller).  `out` must be zeroed before the launch.
__global__ __launch_bounds__(256) void range_stats_kernel(const float* __restrict__ x, long n, float* __restrict__ out) {
  float mx = 0.f, ss = 0.f;
  unsigned sat = 0u, bad = 0u;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i], a = fabsf(v);
    if (!(a <= 3.4e38f)) { ++bad; continue; }
    mx = fmaxf(mx, a);
    ss = fmaf(v, v, ss);
    if (a > 65504.f) ++sat;
  }
#pragma unroll
  for (int sh = 32; sh > 0; sh >>= 1) {
    mx = fmaxf(mx, __shfl_xor(mx, sh));
    ss += __shfl_xor(ss, sh);
    sat += __shfl_xor(sat, sh);
    bad += __shfl_xor(bad, sh);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(mx));
    atomicAdd(out + 1, ss);
    if (sat) atomicAdd(reinterpret_cast<unsigned*>(out) + 2, sat);
    if (bad) atomicAdd(reinterpret_cast<unsigned*>(out) + 3, bad);
  }
}
void launch_range_stats(const float* x, long n, float* out4, hipStream_t s) {
  const unsigned blocks = (unsigned)std::min<long>((n + 255) / 256, 1024);
  hipLaunchKernelGGL(range_stats_kernel, dim3(blocks), dim3(256), 0, s, x, n, out4);
}

void launch_fill_random(float* p, long n, unsigned seed, float scale, hipStream_t s) {
  hipLaunchKernelGGL(fill_random_kernel, dim3(256 * 16), dim3(256), 0, s, p, n, seed, scale);
}

void launch_split_planes(const float* x, unsigned short* y_sb, size_t sb_plane, long n, hipStream_t s) {
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(x), y_sb, sb_plane, n / 4);
}
void launch_merge_planes(const unsigned short* x_sb, size_t sb_plane, float* y, long n, hipStream_t s) {
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(merge_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x_sb, sb_plane, reinterpret_cast<float4*>(y), n / 4);
}

// Cell form (default): thread = (input cell (i, j) .. (i+1, j+1), channel quad).  The four outputs between those four
// inputs (rows 2i+1, 2i+2 x columns 2j+1, 2j+2) use only them, so each input is loaded once per cell instead of once
// per output (4 loads -> 4 stores instead of 16 -> 4), and the index arithmetic is one division per thread.  Cells
// i = -1 / j = -1 (clamped onto row / column 0) produce the first output row / column.  Same expression and the same
// weights (0.25 / 0.75, or 0 / 1 on the clamped border) as the per-output kernel above -> bit-identical results.
__global__ __launch_bounds__(256) void upsample2x_cell_kernel(const float4* __restrict__ x, float4* __restrict__ y, unsigned short* __restrict__ y_sb,
                                                              size_t sb_plane, int H, int W, int CQ) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // (cell column + 1, channel quad)
  if (idx >= (W + 1) * CQ) return;
  const int jq = idx / CQ;
  const int q = idx - jq * CQ;
  const int jc = jq - 1, ic = (int)blockIdx.y - 1, b = blockIdx.z;
  const int r0 = max(ic, 0), r1 = min(ic + 1, H - 1), c0 = max(jc, 0), c1 = min(jc + 1, W - 1);
  const float4* base = x + (long)b * H * W * CQ + q;
  const float4 v00 = base[((long)r0 * W + c0) * CQ], v01 = base[((long)r0 * W + c1) * CQ];
  const float4 v10 = base[((long)r1 * W + c0) * CQ], v11 = base[((long)r1 * W + c1) * CQ];
  const int Ho = 2 * H, Wo = 2 * W;
#pragma unroll
  for (int dy = 0; dy < 2; ++dy) {
    const int oy = 2 * ic + 1 + dy;
    if (oy < 0 || oy >= Ho) continue;
    const float ly = ic < 0 ? 0.f : (dy ? 0.75f : 0.25f), hy = 1.f - ly;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) {
      const int ox = 2 * jc + 1 + dx;
      if (ox < 0 || ox >= Wo) continue;
      const float lx = jc < 0 ? 0.f : (dx ? 0.75f : 0.25f), hx = 1.f - lx;
      float4 o;
      o.x = bilerp2x(v00.x, v01.x, v10.x, v11.x, hx, lx, hy, ly);
      o.y = bilerp2x(v00.y, v01.y, v10.y, v11.y, hx, lx, hy, ly);
      o.z = bilerp2x(v00.z, v01.z, v10.z, v11.z, hx, lx, hy, ly);
      o.w = bilerp2x(v00.w, v01.w, v10.w, v11.w, hx, lx, hy, ly);
      const long i = (((long)b * Ho + oy) * Wo + ox) * CQ + q;
      if (y) y[i] = o;
      if (y_sb) store_sb4(y_sb, sb_plane, (size_t)i * 4, o);
    }
  }
}

static int g_up_variant = -1;
void launch_upsample2x(const float* x, float* y, int B, int H, int W, int C, hipStream_t s, unsigned short* y_sb, size_t sb_plane) {
  if (g_up_variant == -1) {
    const char* e = getenv("PF_UPSAMPLE_VARIANT");
    g_up_variant = e ? atoi(e) : 1;
  }
  if (g_up_variant == 1 && H + 1 <= 65535 && B <= 65535) {
    const dim3 grid(((W + 1) * (C / 4) + 255) / 256, H + 1, B);
    hipLaunchKernelGGL(upsample2x_cell_kernel, grid, dim3(256), 0, s, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), y_sb, sb_plane, H, W, C / 4);
    return;
  }
  const long total = (long)B * 4 * H * W * (C / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(upsample2x_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<float4*>(y), y_sb, sb_plane, B, H, W, C / 4);
}

// ---------------------------------------------------------------------- input normalise
// Reference: (x - pixel_mean) / pixel_std then stack (perspectivefields.py:234-236); BGR order.
__global__ __launch_bounds__(256) void prep_u8_kernel(const uint8_t* __restrict__ in, float4* __restrict__ out, long npix,
                                                      float m0, float m1, float m2, float s0, float s1, float s2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const uint8_t* p = in + i * 3;
    out[i] = make_float4(((float)p[0] - m0) / s0, ((float)p[1] - m1) / s1, ((float)p[2] - m2) / s2, 0.f);
  }
}
__global__ __launch_bounds__(256) void prep_f32_nchw_kernel(const float* __restrict__ in, float4* __restrict__ out, int B, int HW,
                                                            float m0, float m1, float m2, float s0, float s1, float s2) {
  const long npix = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const long b = i / HW, pix = i - b * HW;
    const float* p = in + b * 3 * HW + pix;
    out[i] = make_float4((p[0] - m0) / s0, (p[HW] - m1) / s1, (p[2L * HW] - m2) / s2, 0.f);
  }
}
static unsigned grid_for(long n) { long b = (n + 255) / 256; return (unsigned)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

void launch_prep_u8(const uint8_t* in, float* out, long npix, const float* mean3, const float* std3, hipStream_t s) {
  hipLaunchKernelGGL(prep_u8_kernel, dim3(grid_for(npix)), dim3(256), 0, s, in, reinterpret_cast<float4*>(out), npix,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
}
void launch_prep_f32_nchw(const float* in, float* out, int B, int HW, const float* mean3, const float* std3, hipStream_t s) {
  hipLaunchKernelGGL(prep_f32_nchw_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, s, in, reinterpret_cast<float4*>(out), B, HW,
                     mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
}

// -------------------------------------------------------------- regression prediction heads
// Reference: linear_pred_gravity (1x1, 32->2) + F.normalize(dim=1) (gravity_head.py:117,190-193);
// linear_pred_latitude (1x1, 32->1) + clamp(-1,1) (latitude_head.py:118,189-192).
// 8 lanes per pixel, one float4 of the 32 channels each, xor-shuffle reduction.
__global__ __launch_bounds__(256) void pred_regression_kernel(const float4* __restrict__ tg, const float4* __restrict__ tl,
                                                              const float4* __restrict__ wg, const float* __restrict__ bg,
                                                              const float4* __restrict__ wl, const float* __restrict__ bl,
                                                              float* __restrict__ pg, float* __restrict__ pl, float4* __restrict__ pn,
                                                              int B, int HW) {
  const long npix = (long)B * HW;
  const int sub = threadIdx.x & 7;
  const float4 wg0 = wg[sub], wg1 = wg[8 + sub], wl0 = wl[sub];
  const float bg0 = bg[0], bg1 = bg[1], bl0 = bl[0];
  for (long pix = ((long)blockIdx.x * 256 + threadIdx.x) >> 3; pix < npix; pix += ((long)gridDim.x * 256) >> 3) {
    const float4 a = tg[pix * 8 + sub], c = tl[pix * 8 + sub];
    float g0 = head_dot4(a, wg0);
    float g1 = head_dot4(a, wg1);
    float l0 = head_dot4(c, wl0);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) {
      g0 += __shfl_xor(g0, o, 8);
      g1 += __shfl_xor(g1, o, 8);
      l0 += __shfl_xor(l0, o, 8);
    }
    if (sub == 0) {
      g0 += bg0; g1 += bg1; l0 += bl0;
      const float nrm = fmaxf(sqrtf(fmaf(g1, g1, __fmul_rn(g0, g0))), 1e-12f);  // F.normalize eps
      g0 /= nrm; g1 /= nrm;
      l0 = fminf(fmaxf(l0, -1.f), 1.f);
      const long b = pix / HW, r = pix - b * HW;
      pg[(b * 2) * HW + r] = g0;
      pg[(b * 2 + 1) * HW + r] = g1;
      pl[pix] = l0;
      if (pn) pn[pix] = make_float4(g0, g1, l0, 0.f);
    }
  }
}

void launch_pred_regression(const float* tg, const float* tl, const float* wg, const float* bg, const float* wl, const float* bl,
                            float* pred_g_nchw, float* pred_l_nchw, float* pn_in_nhwc4, int B, int HW, hipStream_t s) {
  const long threads = (long)B * HW * 8;
  hipLaunchKernelGGL(pred_regression_kernel, dim3(grid_for(threads)), dim3(256), 0, s, reinterpret_cast<const float4*>(tg),
                     reinterpret_cast<const float4*>(tl), reinterpret_cast<const float4*>(wg), bg, reinterpret_cast<const float4*>(wl), bl,
                     pred_g_nchw, pred_l_nchw, reinterpret_cast<float4*>(pn_in_nhwc4), B, HW);
}

// --------------------------------------------------------------------- classification decode
// Reference: argmax(dim=0) then decode_bin (utils/utils.py:114-130) / decode_bin_latitude (:148-162).
__global__ __launch_bounds__(256) void decode_cls_kernel(const float* __restrict__ lg, int ng, const float* __restrict__ ll, int nl,
                                                         float* __restrict__ dg, float* __restrict__ dl, int B, int HW) {
  const long npix = (long)B * HW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix; i += (long)gridDim.x * 256) {
    const long b = i / HW, r = i - b * HW;
    const float* pgp = lg + b * ng * HW + r;
    int best = 0; float bv = pgp[0];
    for (int c = 1; c < ng; ++c) { const float v = pgp[(long)c * HW]; if (v > bv) { bv = v; best = c; } }  // first max wins, as torch.argmax
    float cx = 0.f, sy = 0.f;
    if (best != ng - 1) {
      const double ang = ((double)best * (360.0 / (double)(ng - 1)) - 180.0) / 180.0 * 3.14159265358979323846;
      cx = (float)cos(ang); sy = (float)sin(ang);
    }
    dg[(b * 2) * HW + r] = cx;
    dg[(b * 2 + 1) * HW + r] = sy;
    const float* plp = ll + b * nl * HW + r;
    best = 0; bv = plp[0];
    for (int c = 1; c < nl; ++c) { const float v = plp[(long)c * HW]; if (v > bv) { bv = v; best = c; } }
    const float size = 180.f / (float)nl;
    dl[i] = (-90.f + (float)best * size) + size * 0.5f;
  }
}
void launch_decode_cls(const float* logit_g, int ng, const float* logit_l, int nl, float* dec_g, float* dec_l, int B, int HW, hipStream_t s) {
  hipLaunchKernelGGL(decode_cls_kernel, dim3(grid_for((long)B * HW)), dim3(256), 0, s, logit_g, ng, logit_l, nl, dec_g, dec_l, B, HW);
}

// ------------------------------------------------------------------------------ post-process
// Reference: GravityDecoder.postprocess (gravity_head.py:237-261), LatitudeDecoder.postprocess
// (latitude_head.py:195-219), pf_postprocess (utils/utils.py:483-507): bilinear (align_corners
// False, scale = in/out in fp32) of the (2,h,w)*[W/w, H/h] field then L2-normalise; latitude:
// bilinear then asin -> degrees (regression) or degrees directly (classification).
__device__ __forceinline__ void postprocess_pixels(const float* __restrict__ g2, const float* __restrict__ l1, int h, int w,
                                                   float* __restrict__ up, float* __restrict__ lat, int H, int W,
                                                   float sx_scale, float sy_scale, float rh, float rw, int lat_is_sin) {
  const long total = (long)H * W;
  const long hw = (long)h * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int oy = (int)(i / W), ox = (int)(i - (long)oy * W);
    float sy = rh * ((float)oy + 0.5f) - 0.5f; sy = sy < 0.f ? 0.f : sy;
    float sx = rw * ((float)ox + 0.5f) - 0.5f; sx = sx < 0.f ? 0.f : sx;
    int y0 = (int)sy, x0 = (int)sx;
    y0 = y0 > h - 1 ? h - 1 : y0; x0 = x0 > w - 1 ? w - 1 : x0;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    float ly = sy - (float)y0, lx = sx - (float)x0;
    ly = fminf(fmaxf(ly, 0.f), 1.f); lx = fminf(fmaxf(lx, 0.f), 1.f);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const long i00 = (long)y0 * w + x0, i01 = (long)y0 * w + x1, i10 = (long)y1 * w + x0, i11 = (long)y1 * w + x1;
    float gx = hy * (hx * (g2[i00] * sx_scale) + lx * (g2[i01] * sx_scale)) + ly * (hx * (g2[i10] * sx_scale) + lx * (g2[i11] * sx_scale));
    float gy = hy * (hx * (g2[hw + i00] * sy_scale) + lx * (g2[hw + i01] * sy_scale)) +
               ly * (hx * (g2[hw + i10] * sy_scale) + lx * (g2[hw + i11] * sy_scale));
    const float nrm = fmaxf(sqrtf(gx * gx + gy * gy), 1e-12f);
    up[i] = gx / nrm;
    up[total + i] = gy / nrm;
    float lv = hy * (hx * l1[i00] + lx * l1[i01]) + ly * (hx * l1[i10] + lx * l1[i11]);
    if (lat_is_sin) lv = asinf(lv) * 57.295779513082320876798f;  // torch.rad2deg
    lat[i] = lv;
  }
}
__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ g2, const float* __restrict__ l1, int h, int w,
                                                          float* __restrict__ up, float* __restrict__ lat, int H, int W,
                                                          float sx_scale, float sy_scale, float rh, float rw, int lat_is_sin) {
  postprocess_pixels(g2, l1, h, w, up, lat, H, W, sx_scale, sy_scale, rh, rw, lat_is_sin);
}
// the reference's per-image Python loop (gravity_head.py:244-260) as ONE launch: blockIdx.y = image, per-image sizes and
// output pointers travel in the kernel arguments
__global__ __launch_bounds__(256) void postprocess_batch_kernel(const PostBatch pb, int h, int w, int lat_is_sin) {
  const int k = blockIdx.y;
  postprocess_pixels(pb.g2[k], pb.l1[k], h, w, pb.up[k], pb.lat[k], pb.H[k], pb.W[k], pb.sxs[k], pb.sys[k], pb.rh[k], pb.rw[k], lat_is_sin);
}
void launch_postprocess(const float* g2, const float* l1, int h, int w, float* up_out, float* lat_out, int H, int W, int lat_is_sin, hipStream_t s) {
  const float rh = (float)h / (float)H, rw = (float)w / (float)W;  // area_pixel_compute_scale, fp32
  const float sxs = (float)((double)W / (double)w), sys = (float)((double)H / (double)h);  // python float -> float32 tensor
  hipLaunchKernelGGL(postprocess_kernel, dim3(grid_for((long)H * W)), dim3(256), 0, s, g2, l1, h, w, up_out, lat_out, H, W, sxs, sys, rh, rw, lat_is_sin);
}
void launch_postprocess_batch(PostBatch& pb, int h, int w, int lat_is_sin, hipStream_t s) {
  long mx = 1;
  for (int k = 0; k < pb.n; ++k) {
    const int H = pb.H[k], W = pb.W[k];
    pb.rh[k] = (float)h / (float)H; pb.rw[k] = (float)w / (float)W;
    pb.sxs[k] = (float)((double)W / (double)w); pb.sys[k] = (float)((double)H / (double)h);
    mx = std::max(mx, (long)H * W);
  }
  unsigned gx = (unsigned)std::min<long>((mx + 255) / 256, 2048);
  hipLaunchKernelGGL(postprocess_batch_kernel, dim3(gx, pb.n), dim3(256), 0, s, pb, h, w, lat_is_sin);
}

// --------------------------------------------------------------------------- nearest resize
// Reference: F.interpolate(images, (S,S)) default 'nearest' (param_network.py:197): src = floor(dst * in/out).
__global__ __launch_bounds__(256) void nearest_nhwc4_kernel(const float4* __restrict__ x, float4* __restrict__ y, int B, int H, int W, int Ho, int Wo,
                                                            float sh, float sw) {
  const long total = (long)B * Ho * Wo;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % Wo);
    long r = i / Wo;
    const int oy = (int)(r % Ho);
    const int b = (int)(r / Ho);
    int iy = (int)floorf((float)oy * sh), ix = (int)floorf((float)ox * sw);
    iy = iy > H - 1 ? H - 1 : iy; ix = ix > W - 1 ? W - 1 : ix;
    y[i] = x[((long)b * H + iy) * W + ix];
  }
}
void launch_nearest_nhwc4(const float* x, float* y, int B, int H, int W, int Ho, int Wo, hipStream_t s) {
  hipLaunchKernelGGL(nearest_nhwc4_kernel, dim3(grid_for((long)B * Ho * Wo)), dim3(256), 0, s, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<float4*>(y), B, H, W, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo);
}

// ------------------------------------------------------------------------------ ConvNeXt tail
// Reference: x.mean([-2,-1]) -> nn.LayerNorm(768, eps 1e-6) -> head Linear (convnext.py:144-151).
// One block per image.
__global__ __launch_bounds__(256) void gap_ln_head_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                                          const float* __restrict__ w, const float* __restrict__ hb, float* __restrict__ out,
                                                          int HW, int C, int nout, float eps) {
  __shared__ float pooled[1024];
  __shared__ float red[256];
  const int img = blockIdx.x, tid = threadIdx.x;
  const float* xi = x + (long)img * HW * C;
  for (int c = tid; c < C; c += 256) {
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += xi[(long)p * C + c];
    pooled[c] = s / (float)HW;
  }
  __syncthreads();
  auto block_sum = [&](float v) {
    red[tid] = v;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) red[tid] += red[tid + o];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  float s = 0.f;
  for (int c = tid; c < C; c += 256) s += pooled[c];
  const float mean = block_sum(s) / (float)C;
  float q = 0.f;
  for (int c = tid; c < C; c += 256) { const float d = pooled[c] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(block_sum(q) / (float)C + eps);
  for (int c = tid; c < C; c += 256) pooled[c] = (pooled[c] - mean) * rstd * g[c] + b[c];
  __syncthreads();
  for (int o = 0; o < nout; ++o) {
    float d = 0.f;
    for (int c = tid; c < C; c += 256) d += pooled[c] * w[(long)o * C + c];
    const float r = block_sum(d);
    if (tid == 0) out[(long)img * nout + o] = r + hb[o];
  }
}
void launch_gap_ln_head(const float* x, const float* g, const float* b, const float* w, const float* hb, float* out, int B, int HW, int C, int nout, float eps, hipStream_t s) {
  hipLaunchKernelGGL(gap_ln_head_kernel, dim3(B), dim3(256), 0, s, x, g, b, w, hb, out, HW, C, nout, eps);
}

// ------------------------------------------------------ camera parameters -> perspective fields
// Reference: PanoCam.get_up_general / get_lat_general (utils/panocam.py:451-556), the step the demos run on the ParamNet
// output (utils/utils.py:325-381).  cam = {roll, elevation (radians), focal_rel, cx_rel, cy_rel} on the device.
// Quirks kept: up vectors at pixel centres (j + 0.5); latitude on linspace(-c, size - c, size) (end points included);
// elevation == 0 -> constant up field.  Output layout matches pred_*_original: up [2][H][W], latitude [H][W] degrees.
__global__ __launch_bounds__(256) PF_NO_PK_F32 void fields_from_params_kernel(const float* __restrict__ cam, int H, int W, float* __restrict__ up,
                                                                 float* __restrict__ lat) {
  const float roll = cam[0], el = cam[1], f = cam[2] * (float)H;
  const float cx = (cam[3] + 0.5f) * (float)W, cy = (cam[4] + 0.5f) * (float)H;
  float sr, cr, se, ce;
  sincosf(roll, &sr, &cr);
  sincosf(el, &se, &ce);
  const float sx = W > 1 ? (float)W / (float)(W - 1) : 0.f, sy = H > 1 ? (float)H / (float)(H - 1) : 0.f;
  const float vx = el != 0.f ? sr * ce * f / -se + cx : 0.f, vy = el != 0.f ? cr * ce * f / -se + cy : 0.f;
  const float sgn = el > 0.f ? 1.f : -1.f;
  const long n = (long)H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int row = (int)(i / W), col = (int)(i - (long)row * W);
    float ux, uy;
    if (el == 0.f) { ux = -sr; uy = -cr; }
    else { ux = (vx - ((float)col + 0.5f)) * sgn; uy = (vy - ((float)row + 0.5f)) * sgn; }
    const float inv = 1.0f / sqrtf(ux * ux + uy * uy);
    up[i] = ux * inv;
    up[n + i] = uy * inv;
    const float x = (-cx + (float)col * sx) / f, y = (-cy + (float)row * sy) / f;
    const float xw = x * cr - y * sr;
    const float yw = x * ce * sr + y * ce * cr - se;
    const float zw = x * se * sr + y * se * cr + ce;
    lat[i] = -atan2f(yw, sqrtf(xw * xw + zw * zw)) * 57.29577951308232f;
  }
}
void launch_fields_from_params(const float* cam5, int H, int W, float* up, float* lat, hipStream_t s) {
  long blocks = ((long)H * W + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fields_from_params_kernel, dim3((unsigned)blocks), dim3(256), 0, s, cam5, H, W, up, lat);
}

// ------------------------------------------------------------------------- bit-exact PIL resize
// Reference: ResizeTransform.apply_image -> PIL Image.resize(BILINEAR) on uint8 (perspectivefields.py:34-46,201).
// Pillow's resample is a two-pass (horizontal, then vertical) antialiased triangle filter in 22-bit fixed point;
// the coefficient tables are computed on the host in double exactly as Pillow does (engine.hip resize_coeffs),
// the passes below are pure integer arithmetic, so the result is bit-identical to PIL (tests/test_gpu_resize.py).
static constexpr int RS_PREC = 32 - 8 - 2;
__device__ __forceinline__ uint8_t rs_clip8(int v) { v >>= RS_PREC; return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int OW,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long total = (long)H * OW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xx = (int)(i % OW);
    const long y = i / OW;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ksize;
    const uint8_t* p = in + (y * W + xmin) * 3;
    int s0 = 1 << (RS_PREC - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
      const int w = k[x];
      s0 += p[3 * x] * w; s1 += p[3 * x + 1] * w; s2 += p[3 * x + 2] * w;
    }
    uint8_t* o = out + i * 3;
    o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
  }
}
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int OW, int OH,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long total = (long)OH * OW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xx = (int)(i % OW);
    const int yy = (int)(i / OW);
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = kk + (long)yy * ksize;
    const uint8_t* p = in + ((long)ymin * OW + xx) * 3;
    int s0 = 1 << (RS_PREC - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
      const int w = k[y];
      s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
      p += (long)OW * 3;
    }
    uint8_t* o = out + i * 3;
    o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
  }
}
// The same two passes for up to ResizeBatch::MAX images per launch pair (blockIdx.y = image; per-image pointers, sizes and
// coefficient tables travel in the kernel arguments): inference_batch's per-image resize loop as two launches.
__device__ __forceinline__ void resize_h_pixels(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H, int W, int OW,
                                                const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long total = (long)H * OW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xx = (int)(i % OW);
    const long y = i / OW;
    const int xmin = bounds[2 * xx], n = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ksize;
    const uint8_t* p = in + (y * W + xmin) * 3;
    int s0 = 1 << (RS_PREC - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < n; ++x) {
      const int w = k[x];
      s0 += p[3 * x] * w; s1 += p[3 * x + 1] * w; s2 += p[3 * x + 2] * w;
    }
    uint8_t* o = out + i * 3;
    o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
  }
}
__device__ __forceinline__ void resize_v_pixels(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int OW, int OH,
                                                const int* __restrict__ bounds, const int* __restrict__ kk, int ksize) {
  const long total = (long)OH * OW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int xx = (int)(i % OW);
    const int yy = (int)(i / OW);
    const int ymin = bounds[2 * yy], n = bounds[2 * yy + 1];
    const int* k = kk + (long)yy * ksize;
    const uint8_t* p = in + ((long)ymin * OW + xx) * 3;
    int s0 = 1 << (RS_PREC - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < n; ++y) {
      const int w = k[y];
      s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
      p += (long)OW * 3;
    }
    uint8_t* o = out + i * 3;
    o[0] = rs_clip8(s0); o[1] = rs_clip8(s1); o[2] = rs_clip8(s2);
  }
}
__global__ __launch_bounds__(256) void resize_h_batch_kernel(const ResizeBatch rb, int OW) {
  const int k = blockIdx.y;
  resize_h_pixels(rb.in[k], rb.tmp[k], rb.H[k], rb.W[k], OW, rb.bh[k], rb.kh[k], rb.ksh[k]);
}
__global__ __launch_bounds__(256) void resize_v_batch_kernel(const ResizeBatch rb, int OW, int OH) {
  const int k = blockIdx.y;
  resize_v_pixels(rb.tmp[k], rb.out[k], OW, OH, rb.bv[k], rb.kv[k], rb.ksv[k]);
}
void launch_resize_batch_u8(const ResizeBatch& rb, int OH, int OW, hipStream_t s) {
  long mx = 1;
  for (int k = 0; k < rb.n; ++k) mx = std::max(mx, (long)rb.H[k] * OW);
  hipLaunchKernelGGL(resize_h_batch_kernel, dim3(grid_for(mx), rb.n), dim3(256), 0, s, rb, OW);
  hipLaunchKernelGGL(resize_v_batch_kernel, dim3(grid_for((long)OH * OW), rb.n), dim3(256), 0, s, rb, OW, OH);
}

void launch_resize_u8(const uint8_t* in, int H, int W, uint8_t* tmp, uint8_t* out, int OH, int OW, const int* bh, const int* kh, int ksh,
                      const int* bv, const int* kv, int ksv, hipStream_t s) {
  hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((long)H * OW)), dim3(256), 0, s, in, tmp, H, W, OW, bh, kh, ksh);
  hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((long)OH * OW)), dim3(256), 0, s, tmp, out, H, OW, OH, bv, kv, ksv);
}

// Reference: ParamNet.forward eval branch (param_network.py:62-67).  out is [B][8]:
// mode 0 (centered): roll, pitch, vfov (deg), rel_focal = 1/(2 tan x2), raw x0..x3
// mode 1 (uncentered, ParamNetConvNextRegress, param_network.py:204-220): raw x0..x(n-1), zero padded; with the zoo's output order (roll, pitch, general_vfov, rel_cx,
//   rel_cy: n = 5) out[5] = rel_focal from the general vertical FoV (utils/utils.py:47-91 with h = 1, degree = True).  The reference runs scipy.fsolve from 1.5 on
//   cos(gvfov) = (p^2 + q^2 - 1) / (2 p q),  p^2 = f^2 + cx^2 + (cy + 1/2)^2,  q^2 = f^2 + cx^2 + (cy - 1/2)^2  and returns |f|.  With u = f^2 + cx^2 + cy^2 + 1/4:
//   p^2 q^2 = u^2 - cy^2 and p^2 + q^2 - 1 = 2 u - 1, so cos^2 (u^2 - cy^2) = (u - 1/2)^2 is a quadratic in u: solved in closed form in fp64 (the same expression
//   as perspectivefields.py general_vfov_to_focal, which stays the host path of non-zoo output orders) -- no host round trip on the uncentered models' hot path.
__global__ void paramnet_scalars_kernel(const float* __restrict__ raw, int nraw, float* __restrict__ out8, int B, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B) return;
  const float* r = raw + (long)i * nraw;
  float* o = out8 + (long)i * 8;
  if (mode == 0) {
    o[0] = r[0] * 90.0f;
    o[1] = r[1] * 90.0f;
    o[2] = r[2] * 90.0f;
    o[3] = 1.0f / 2.0f / tanf(r[2]);
    o[4] = r[0]; o[5] = r[1]; o[6] = r[2]; o[7] = r[3];
  } else {
    for (int k = 0; k < 8; ++k) o[k] = k < nraw ? r[k] : 0.f;
    if (nraw == 5) {
      const double gv = (double)(r[2] * 90.0f), cx = (double)r[3], cy = (double)r[4];  // the factors in fp32, as the reference's x[:, idx] * factor
      const double c = cos(gv * 0.017453292519943295), s2 = 1.0 - c * c;
      const double disc = sqrt(fmax(1.0 - 4.0 * s2 * (c * c * cy * cy + 0.25), 0.0));
      const double u = (c >= 0.0 ? 1.0 + disc : 1.0 - disc) / (2.0 * s2);  // cos > 0 needs u > 1/2: the '+' root; gvfov > 90 deg: the '-' root
      o[5] = (float)sqrt(fabs(u - cy * cy - 0.25 - cx * cx));
    }
  }
}
void launch_paramnet_scalars(const float* raw, int nraw, float* out8, int B, int mode, hipStream_t s) {
  hipLaunchKernelGGL(paramnet_scalars_kernel, dim3((B + 63) / 64), dim3(64), 0, s, raw, nraw, out8, B, mode);
}


}  // namespace pf
