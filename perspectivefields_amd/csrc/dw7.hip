// Depthwise 7x7 (ConvNeXt Block.dwconv, convnext.py:30-32,48) for gfx950: channels-per-lane streaming kernel.
// Compiled with -fno-slp-vectorize (build.py): the one-channel-per-lane form must stay scalar v_fma_f32 -- the SLP
// vectoriser otherwise pairs accumulators into v_pk_fma_f32 with register shuffles and doubles the VGPR count.
//
// Lane = CPL neighbouring channels (1 or 2); 32 / CPL lanes = 32 channels of one column; a wave = 2 CPL neighbouring
// columns.  Thread (channels, column x) streams the input rows of a strip:
//   * 7 buffer loads per row (x-3..x+3; the overlap between neighbouring columns is served by L1): the per-lane
//     column/channel byte offset is fixed (out-of-image columns use the out-of-range offset -> hardware zeros), the
//     row offset is a scalar register, so a row costs NO address VALU work;
//   * 49 FMAs (CPL = 2: v_pk_fma_f32, both channels per instruction) into 7 output-row accumulators whose ring
//     position is a compile-time constant (loop unrolled 7 NB times: 7 ring slots x NB row buffers, no register moves);
//   * one store per row.
// The 49 per-channel weights live in VGPRs: 98 registers at CPL = 2 (2-3 waves per SIMD), 49 at CPL = 1 (5 waves).
// The op is latency-bound, not VALU- or HBM-bound, until enough rows are in flight per SIMD (PMC, profiles/): at 2 waves
// x 2 rows ahead a row costs ~1800 cycles of which ~210 are VALU.  FLOP-heavy for a "memory-bound" op all the same: 98
// flops per 8 bytes -- 0.09 ms (packed) / 0.18 ms (scalar) of VALU time per forward against 0.14 ms of HBM time at 8 TB/s.
#include <stdlib.h>

#include <algorithm>

#include "pf_kernels.h"

namespace pf {

template <int N> struct LaneVec;
template <> struct LaneVec<1> {
  typedef float T;
  static __device__ __forceinline__ T zero() { return 0.f; }
  static __device__ __forceinline__ T fma(T a, T b, T c) { return fmaf(a, b, c); }
  static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0)); }
  static __device__ __forceinline__ void store(T v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0); }
};
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
template <> struct LaneVec<2> {
  typedef f32x2 T;
  static __device__ __forceinline__ T zero() { return f32x2{0.f, 0.f}; }
  static __device__ __forceinline__ T fma(T a, T b, T c) { return __builtin_elementwise_fma(a, b, c); }
  static __device__ __forceinline__ T load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const u32x2v q = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
    return f32x2{__uint_as_float(q.x), __uint_as_float(q.y)};
  }
  static __device__ __forceinline__ void store(T v, __amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_buffer_store_b64(u32x2v{__float_as_uint(v.x), __float_as_uint(v.y)}, r, voff, soff, 0);
  }
};

template <int CPL /*channels per lane*/, int NB /*row buffers: loads run NB - 1 rows ahead*/, int MAXT /*max threads per block*/,
          int DIAG = 0 /*diagnostics: 1 = no stores, 2 = no loads*/>
__global__ __launch_bounds__(MAXT) void dwconv7x7_lane_kernel(const float* __restrict__ x, const float* __restrict__ w49c,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int B, int H, int W, int C, int TH, int XB /*columns per block*/, int CPB /*channels per block*/) {
  typedef LaneVec<CPL> V;
  typedef typename V::T T;
  const int LPC = CPB / CPL;  // lanes per column
  const int slabs = C / CPB, tilesX = (W + XB - 1) / XB, strips = (H + TH - 1) / TH;
  const int nblk = B * strips * tilesX * slabs;
  int t;
  {  // XCD-aware order: the slabs / x-neighbours of one image region share an L2
    const int b = blockIdx.x, qd = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    t = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int slab = t % slabs; t /= slabs;
  const int tx = t % tilesX; t /= tilesX;
  const int st = t % strips; t /= strips;
  const int b = t;
  const int col = threadIdx.x / LPC;
  const int c = slab * CPB + CPL * (threadIdx.x - col * LPC);
  const int ox = tx * XB + col;
  const bool col_ok = ox < W;
  const int y0 = st * TH, y1 = min(y0 + TH, H);
  T wk[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) wk[k] = *reinterpret_cast<const T*>(w49c + (long)k * C + c);
  const T bv = *reinterpret_cast<const T*>(bias + c);
  float* const xb = const_cast<float*>(x + (long)b * H * W * C);
  float* const yb = y + (long)b * H * W * C;
  const unsigned img_bytes = (unsigned)((long)H * W * C * 4);
  unsigned voff[7];
#pragma unroll
  for (int kx = 0; kx < 7; ++kx) {
    const int ix = ox + kx - 3;
    voff[kx] = (col_ok && (unsigned)ix < (unsigned)W) ? (unsigned)(ix * C + c) * 4u : 0x80000000u;
  }
  const unsigned voff_out = col_ok ? (unsigned)(ox * C + c) * 4u : 0x80000000u;
  const unsigned row_bytes = (unsigned)(W * C) * 4u;
  T acc[7], in[NB][7];
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[j] = bv;
  // Everything touching memory in the row loop is branch-free (a descriptor with zero records turns the loads of a row
  // outside the image into hardware zeros and drops the stores of rows above the strip): with branches around loads the
  // compiler's s_waitcnt analysis merges different load orders and drains every load each row -- no prefetch overlap.
  auto load_row = [&](int iy, T (&v)[7]) {
    const bool ok = (unsigned)iy < (unsigned)H;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(xb, 0, ok ? img_bytes : 0u, 0x00020000);
    const unsigned soff = ok ? (unsigned)iy * row_bytes : 0u;
#pragma unroll
    for (int kx = 0; kx < 7; ++kx) v[kx] = DIAG == 2 ? bv : V::load(r, voff[kx], soff);
  };
  // relative row index tt: input row iy = y0 - 3 + tt, tt in [0, nrows); it feeds output rows oy = iy - ky + 3, whose
  // accumulator slot is (tt - ky + 3) mod 7; after row tt the output row iy - 3 (slot (tt + 4) mod 7) is complete.
  // r = tt mod (7 NB) is a compile-time constant in the unrolled bodies: ring slot and row buffer are fixed registers.
  const int nrows = (y1 - y0) + 6;
  auto do_row = [&](int tt, int r) {
    const int iy = y0 - 3 + tt;
    load_row(iy + NB - 1, in[(r + NB - 1) % NB]);  // prefetch; rows past the strip are loaded but never used
    if ((unsigned)iy < (unsigned)H) {              // block-uniform, no memory operation inside
#pragma unroll
      for (int ky = 0; ky < 7; ++ky) {
        T a = acc[(r - ky + 3 + 7 * NB) % 7];
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) a = V::fma(in[r % NB][kx], wk[ky * 7 + kx], a);
        acc[(r - ky + 3 + 7 * NB) % 7] = a;
      }
    }
    const int oy = iy - 3;  // < y1 always (tt < nrows)
    const bool st_ok = oy >= y0;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, st_ok ? img_bytes : 0u, 0x00020000);
    if (DIAG != 1) V::store(acc[(r + 4) % 7], ry, voff_out, st_ok ? (unsigned)oy * row_bytes : 0u);
    else if (tt == nrows - 1) V::store(acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5] + acc[6], ry, voff_out, 0u);  // keep the arithmetic alive
    acc[(r + 4) % 7] = bv;
  };
#pragma unroll
  for (int d = 0; d < NB - 1; ++d) load_row(y0 - 3 + d, in[d]);
  int t0 = 0;
  for (; t0 + 7 * NB <= nrows; t0 += 7 * NB) {  // whole groups: no exit inside
#pragma unroll
    for (int r = 0; r < 7 * NB; ++r) do_row(t0 + r, r);
  }
#pragma unroll
  for (int r = 0; r < 7 * NB; ++r) {  // straight-line tail
    if (t0 + r >= nrows) break;
    do_row(t0 + r, r);
  }
}

// ---- column-blocked form (default for the larger maps).  The lane kernel above issues 7 dword loads per output and lane:
// with 4 waves per CU that is 112 texture-address cycles per 98 VALU cycles and row -- the kernel sits on the L1 / TA
// request rate, which is why it did not react to occupancy, prefetch depth or packed FMAs.  Here a thread
// owns NC ADJACENT output columns of one channel: a row costs NC + 6 loads for 49 NC FMAs (2.5 loads per output at NC = 4),
// the x-overlap lives in registers instead of L1.  Same streaming structure otherwise: lane = channel (32 consecutive
// channels = one 128-byte piece of a pixel), buffer loads with per-lane fixed offsets + scalar row offset (hardware range
// check = zero padding), compile-time ring of 7 output-row accumulators x NC columns, loads NB - 1 rows ahead, branch-free
// memory operations.  Taps whose output row lies outside the strip are skipped with block-uniform branches, so a strip's
// six halo rows cost loads only, no FMAs.
template <int NC /*adjacent output columns per thread*/, int NB /*row buffers*/, int DIAG = 0>
__global__ __launch_bounds__(256) void dwconv7x7_cb_kernel(const float* __restrict__ x, const float* __restrict__ w49c,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           int B, int H, int W, int C, int TH, int GB /*column groups per block*/) {
  constexpr int NI = NC + 6;  // input columns per thread
  const int groups = (W + NC - 1) / NC;
  const int slabs = C / 32, tilesX = (groups + GB - 1) / GB, strips = (H + TH - 1) / TH;
  const int nblk = B * strips * tilesX * slabs;
  int t;
  {  // XCD-aware order: the slabs / x-neighbours of one image region share an L2
    const int b = blockIdx.x, qd = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    t = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int slab = t % slabs; t /= slabs;
  const int tx = t % tilesX; t /= tilesX;
  const int st = t % strips; t /= strips;
  const int b = t;
  const int g = tx * GB + (int)threadIdx.x / 32;
  const int c = slab * 32 + ((int)threadIdx.x & 31);
  const int x0 = g * NC;
  const bool g_ok = g < groups;
  const int y0 = st * TH, y1 = min(y0 + TH, H);
  float wk[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) wk[k] = w49c[(long)k * C + c];
  const float bv = bias[c];
  float* const xb = const_cast<float*>(x + (long)b * H * W * C);
  float* const yb = y + (long)b * H * W * C;
  const unsigned img_bytes = (unsigned)((long)H * W * C * 4);
  unsigned voff[NI], voff_out[NC];
#pragma unroll
  for (int j = 0; j < NI; ++j) {
    const int ix = x0 - 3 + j;
    voff[j] = (g_ok && (unsigned)ix < (unsigned)W) ? (unsigned)(ix * C + c) * 4u : 0x80000000u;
  }
#pragma unroll
  for (int j = 0; j < NC; ++j) voff_out[j] = (g_ok && x0 + j < W) ? (unsigned)((x0 + j) * C + c) * 4u : 0x80000000u;
  const unsigned row_bytes = (unsigned)(W * C) * 4u;
  float acc[7][NC], in[NB][NI];
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[s][j] = bv;
  auto load_row = [&](int iy, float (&v)[NI]) {
    const bool ok = (unsigned)iy < (unsigned)H;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(xb, 0, ok ? img_bytes : 0u, 0x00020000);
    const unsigned soff = ok ? (unsigned)iy * row_bytes : 0u;
#pragma unroll
    for (int j = 0; j < NI; ++j) v[j] = DIAG == 2 ? bv : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff[j], soff, 0));
  };
  // relative row tt: input row iy = y0 - 3 + tt feeds output rows oy = iy - ky + 3 (accumulator slot (tt - ky + 3) mod 7);
  // after row tt the output row iy - 3 (slot (tt + 4) mod 7) is complete.  r = tt mod (7 NB) is a compile-time constant.
  const int nrows = (y1 - y0) + 6;
  auto do_row = [&](int tt, int r) {
    const int iy = y0 - 3 + tt;
    load_row(iy + NB - 1, in[(r + NB - 1) % NB]);  // prefetch; rows past the strip are loaded but never used
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int oy = iy - ky + 3;
      if (oy >= y0 && oy < y1) {  // block-uniform; rows outside the image were loaded as zeros
        constexpr int dummy = 0; (void)dummy;
        const int s = (r - ky + 3 + 7 * NB) % 7;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          float a = acc[s][j];
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) a = fmaf(in[r % NB][j + kx], wk[ky * 7 + kx], a);
          acc[s][j] = a;
        }
      }
    }
    const int oy = iy - 3;  // < y1 always (tt < nrows)
    const bool st_ok = oy >= y0;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, st_ok ? img_bytes : 0u, 0x00020000);
    const unsigned soff = st_ok ? (unsigned)oy * row_bytes : 0u;
    const int so = (r + 4) % 7;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      if (DIAG != 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[so][j]), ry, voff_out[j], soff, 0);
      else if (tt == nrows - 1) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[0][j] + acc[1][j] + acc[2][j] + acc[3][j] + acc[4][j] + acc[5][j] + acc[6][j]), ry, voff_out[j], 0u, 0);
      acc[so][j] = bv;
    }
  };
#pragma unroll
  for (int d = 0; d < NB - 1; ++d) load_row(y0 - 3 + d, in[d]);
  int t0 = 0;
  for (; t0 + 7 * NB <= nrows; t0 += 7 * NB) {  // whole groups: no exit inside
#pragma unroll
    for (int r = 0; r < 7 * NB; ++r) do_row(t0 + r, r);
  }
#pragma unroll
  for (int r = 0; r < 7 * NB; ++r) {  // straight-line tail
    if (t0 + r >= nrows) break;
    do_row(t0 + r, r);
  }
}

// ---- LDS-tile form for the SMALL maps (20^2 x 384 and 10^2 x 768: 12 of the 18 launches of a ConvNeXt-T forward, convnext.py:30-32,48).  On a map that is only 10-20
// rows tall the streaming kernel above is a chain of 10-20 dependent row steps, each waiting for its loads (r02: 21.4 us for a 39 MB launch whose bandwidth time is
// 8 us; no sweep setting moved it below 16.9 us).  Here a block = (image, strip of TH rows, 32-channel slab) first pulls its WHOLE input tile -- (TH + 6) x (W + 6)
// pixels x 128 B, out-of-image pixels as hardware zeros -- into LDS with every load of the block in flight at once, and then runs the same compile-time accumulator
// ring from LDS: a row step now costs NC + 6 ds_read_b32 (lane = channel: 32 consecutive banks, conflict-free) instead of a round trip to L2 / HBM.  Same
// accumulation order as the streaming kernel: bit-identical results.
template <int NC /*adjacent output columns per thread*/, int NF4 /*float4 pieces of the tile per thread*/>
__global__ __launch_bounds__(NC == 2 ? 320 : 256) void dwconv7x7_lds_kernel(const float* __restrict__ x, const float* __restrict__ w49c, const float* __restrict__ bias,
                                                                             float* __restrict__ y, int B, int H, int W, int C, int TH) {
  constexpr int NI = NC + 6;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [TH + 6][W + 6][32]
  const int groups = (W + NC - 1) / NC;
  const int slabs = C / 32, strips = (H + TH - 1) / TH;
  const int nblk = B * strips * slabs;
  int t;
  {  // XCD-aware order: the slabs of one image strip share an L2 (each 128-byte line of a pixel is one slab's)
    const int b = blockIdx.x, qd = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    t = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int slab = t % slabs; t /= slabs;
  const int st = t % strips; t /= strips;
  const int b = t;
  const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
  const int y0 = st * TH, y1 = min(y0 + TH, H);
  const int TW = W + 6, nrows = (y1 - y0) + 6;
  float* const xb = const_cast<float*>(x + (long)b * H * W * C);
  float* const yb = y + (long)b * H * W * C;
  const unsigned img_bytes = (unsigned)((long)H * W * C * 4);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xb, 0, img_bytes, 0x00020000);
  // ---- stage: piece e = tid + nthr i -> (tile pixel e / 8, float4 e % 8 of the slab); all NF4 loads of a thread are in flight before its first LDS write
  {
    const int npieces = nrows * TW * 8;
    float4 v[NF4];
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int e = tid + nthr * i;
      const int pix = e >> 3, c4 = e & 7;
      const int ty = pix / TW, tx = pix - ty * TW;
      const int iy = y0 - 3 + ty, ix = tx - 3;
      const bool ok = e < npieces && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const unsigned off = ok ? (unsigned)(((iy * W + ix) * C + slab * 32 + c4 * 4) * 4) : 0x80000000u;
      const u32x4v q = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
      v[i] = make_float4(__uint_as_float(q.x), __uint_as_float(q.y), __uint_as_float(q.z), __uint_as_float(q.w));
    }
#pragma unroll
    for (int i = 0; i < NF4; ++i) {
      const int e = tid + nthr * i;
      if (e < npieces) *reinterpret_cast<float4*>(tile + (size_t)e * 4) = v[i];
    }
    // the NC - 1 spare pixels behind the tile (read by the last column group of an odd-width map; their products feed accumulators that are never stored) hold zeros,
    // not whatever the previous block left in LDS
    if (tid < (NC - 1) * 8) *reinterpret_cast<float4*>(tile + (size_t)npieces * 4 + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int g = tid >> 5, cl = tid & 31;
  const int c = slab * 32 + cl;
  const int x0 = g * NC;
  const bool g_ok = g < groups;
  float wk[49];
#pragma unroll
  for (int k = 0; k < 49; ++k) wk[k] = w49c[(long)k * C + c];
  const float bv = bias[c];
  unsigned voff_out[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) voff_out[j] = (g_ok && x0 + j < W) ? (unsigned)((x0 + j) * C + c) * 4u : 0x80000000u;
  const unsigned row_bytes = (unsigned)(W * C) * 4u;
  float acc[7][NC];
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int j = 0; j < NC; ++j) acc[s][j] = bv;
  __syncthreads();
  // a thread's window stays inside its tile row whenever x0 + NC <= W; the ragged last group of an odd-width map reads up to NC - 1 pixels past the row end for the
  // columns it never stores (the launcher allocates NC - 1 spare pixels behind the tile)
  const float* trow = tile + x0 * 32 + cl;
  // relative row tt: input row iy = y0 - 3 + tt feeds output rows oy = iy - ky + 3 (accumulator slot (tt - ky + 3) mod 7); after row tt the output row iy - 3
  // (slot (tt + 4) mod 7) is complete.  r = tt mod 7 is a compile-time constant.
  auto do_row = [&](int tt, int r) {
    const int iy = y0 - 3 + tt;
    float in[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) in[j] = trow[(tt * TW + j) * 32];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int oy = iy - ky + 3;
      if (oy >= y0 && oy < y1) {  // block-uniform
        const int s = (r - ky + 3 + 7) % 7;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          float a = acc[s][j];
#pragma unroll
          for (int kx = 0; kx < 7; ++kx) a = fmaf(in[j + kx], wk[ky * 7 + kx], a);
          acc[s][j] = a;
        }
      }
    }
    const int oy = iy - 3;
    const bool st_ok = oy >= y0;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, st_ok ? img_bytes : 0u, 0x00020000);
    const unsigned soff = st_ok ? (unsigned)oy * row_bytes : 0u;
    const int so = (r + 4) % 7;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[so][j]), ry, voff_out[j], soff, 0);
      acc[so][j] = bv;
    }
  };
  int t0 = 0;
  for (; t0 + 7 <= nrows; t0 += 7) {
#pragma unroll
    for (int r = 0; r < 7; ++r) do_row(t0 + r, r);
  }
#pragma unroll
  for (int r = 0; r < 7; ++r) {
    if (t0 + r >= nrows) break;
    do_row(t0 + r, r);
  }
}

// maps of at most 20 columns (threads = 32 x column groups <= 320); th <= 0: automatic strip height
bool dwconv7x7_lds_ok(int H, int W, int C) {
  if (W > 20 || W < 2 || (C % 32) != 0 || H < 1) return false;
  return (long)7 * (W + 6) * 8 <= (long)13 * 32 * ((W + 1) / 2);  // a one-row strip must fit the 13 float4 pieces per thread (fails for W = 2: the streaming kernel takes those)
}
void launch_dwconv7x7_lds(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int th, hipStream_t s) {
  constexpr int NC = 2;
  const int groups = (W + NC - 1) / NC;
  const int threads = 32 * groups;
  // strip height: the tallest tile whose pieces fit 13 float4 per thread (20^2: TH = 10 -> 16 x 26 x 8 = 3328 pieces / 320 threads = 11; 10^2: the whole map, 2048 / 160 = 13)
  int TH = th > 0 ? std::min(th, H) : H;
  auto pieces = [&](int t) { return (long)(std::min(t, H) + 6) * (W + 6) * 8; };
  auto lds_bytes = [&](int t) { return ((size_t)(std::min(t, H) + 6) * (W + 6) + (NC - 1)) * 32 * 4; };
  while (TH > 1 && ((pieces(TH) + threads - 1) / threads > 13 || lds_bytes(TH) > 64 * 1024)) --TH;  // <= 64 KB of dynamic LDS: no hipFuncSetAttribute needed (20^2: 53 KB, 10^2: 32 KB)
  if (th <= 0) {  // prefer equal strips
    const int strips = (H + TH - 1) / TH;
    TH = (H + strips - 1) / strips;
  }
  const long blocks = (long)B * ((H + TH - 1) / TH) * (C / 32);
  const size_t lds = ((size_t)(TH + 6) * (W + 6) + (NC - 1)) * 32 * 4;
  hipLaunchKernelGGL((dwconv7x7_lds_kernel<NC, 13>), dim3((unsigned)blocks), dim3(threads), lds, s, x, w49c, bias, y, B, H, W, C, TH);
}

template <int NC, int NB>
static void launch_cb(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int TH, hipStream_t s) {
  const int groups = (W + NC - 1) / NC;
  int GB = 8;  // column groups per block (x 32 channels = threads); prefer an even divisor of `groups` (whole waves, no idle groups)
  for (int cand : {8, 6, 4, 2}) if (groups % cand == 0) { GB = cand; break; }
  if (groups % 2 != 0) GB = groups <= 8 ? groups : 8;
  const long blocks = (long)B * ((H + TH - 1) / TH) * ((groups + GB - 1) / GB) * (C / 32);
  static int diag = -1;
  if (diag < 0) { const char* d = getenv("PF_DW7_DIAG"); diag = d ? atoi(d) : 0; }
#ifdef PF_TUNING_BUILD
  if (diag == 1)      { hipLaunchKernelGGL((dwconv7x7_cb_kernel<NC, NB, 1>), dim3((unsigned)blocks), dim3(GB * 32), 0, s, x, w49c, bias, y, B, H, W, C, TH, GB); return; }
  else if (diag == 2) { hipLaunchKernelGGL((dwconv7x7_cb_kernel<NC, NB, 2>), dim3((unsigned)blocks), dim3(GB * 32), 0, s, x, w49c, bias, y, B, H, W, C, TH, GB); return; }
#endif
  hipLaunchKernelGGL((dwconv7x7_cb_kernel<NC, NB, 0>), dim3((unsigned)blocks), dim3(GB * 32), 0, s, x, w49c, bias, y, B, H, W, C, TH, GB);
}

// column-blocked kernel: NC / NB / strip height TH by map size (0 = automatic; scripts/tune_dw7.py measured the table);
// PF_DW7_NC / PF_DW7_NB / PF_DW7_TH override the automatic choice
void launch_dwconv7x7_cb_cfg(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int nc, int nb, int th, hipStream_t s) {
  static int e_nc = -1, e_nb = 0, e_th = 0;
  if (e_nc == -1) {
    const char* a = getenv("PF_DW7_NC"); e_nc = a ? atoi(a) : 0;
    const char* n = getenv("PF_DW7_NB"); e_nb = n ? atoi(n) : 0;
    const char* t = getenv("PF_DW7_TH"); e_th = t ? atoi(t) : 0;
  }
  if (nc <= 0) nc = e_nc;
  if (nb <= 0) nb = e_nb;
  if (th <= 0) th = e_th;
  // measured on MI355X at B = 32 (profiles/r02_tune_dw7.txt): 80^2: nc4 nb2 th20, 40^2: nc4 nb2 th10, 20^2: nc4 nb2 th10, 10^2: nc2 nb3 th10
  const int NC = nc > 0 ? nc : (W >= 20 ? 4 : 2);
  const int NB = nb > 0 ? nb : (W >= 20 ? 2 : 3);
  int TH = th > 0 ? std::min(th, H) : std::min(H, H >= 80 ? 20 : 10);
  if (th <= 0) {  // small batches: more strips until the chip is full (256 CUs x 12-16 waves); waves = B x C x ceil(W / NC) / 64 x strips
    const long per_strip = (long)B * C * ((W + NC - 1) / NC) / 64;
    while (TH > 5 && per_strip * ((H + TH - 1) / TH) < 2048) TH = (TH + 1) / 2;
  }
  if (NC >= 4) { if (NB == 2) launch_cb<4, 2>(x, w49c, bias, y, B, H, W, C, TH, s); else launch_cb<4, 3>(x, w49c, bias, y, B, H, W, C, TH, s); }
  else         { if (NB == 2) launch_cb<2, 2>(x, w49c, bias, y, B, H, W, C, TH, s); else launch_cb<2, 3>(x, w49c, bias, y, B, H, W, C, TH, s); }
}
void launch_dwconv7x7_cb(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s) {
  launch_dwconv7x7_cb_cfg(x, w49c, bias, y, B, H, W, C, 0, 0, 0, s);
}

template <int CPL, int NB>
static void launch_lane(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int th_env, int cpb_env, hipStream_t s) {
  // block = XB columns x CPB channels.  CPB = 32: 128-byte pieces of a pixel per block; CPB = C: whole pixels (contiguous
  // C * 4 bytes per column and row), columns per block limited by the 1024-thread block
  const int CPB = cpb_env == 0 ? 32 : C;
  const int LPC = CPB / CPL;
  int XB = 1;
  for (int cand : {16, 10, 8, 5, 4, 2}) if (W % cand == 0 && cand * LPC <= (CPB == 32 ? 256 : 1024)) { XB = cand; break; }
  if (CPB == 32 && XB < 4) XB = 4;
  const int threads = (XB * LPC + 63) / 64 * 64;
  const long per_strip = (long)B * ((W + XB - 1) / XB) * (C / CPB) * threads / 256;  // 256-thread block equivalents
  int TH = th_env > 0 ? th_env : H;
  if (th_env <= 0) while (TH > 20 && per_strip * ((H + TH - 1) / TH) < 768) TH = (TH + 1) / 2;
  const long blocks = (long)B * ((H + TH - 1) / TH) * ((W + XB - 1) / XB) * (C / CPB);
  static int diag = -1;
  if (diag < 0) { const char* d = getenv("PF_DW7_DIAG"); diag = d ? atoi(d) : 0; }
#ifdef PF_TUNING_BUILD
  if (diag == 1 && threads <= 256)      { hipLaunchKernelGGL((dwconv7x7_lane_kernel<CPL, NB, 256, 1>), dim3((unsigned)blocks), dim3(threads), 0, s, x, w49c, bias, y, B, H, W, C, TH, XB, CPB); return; }
  else if (diag == 2 && threads <= 256) { hipLaunchKernelGGL((dwconv7x7_lane_kernel<CPL, NB, 256, 2>), dim3((unsigned)blocks), dim3(threads), 0, s, x, w49c, bias, y, B, H, W, C, TH, XB, CPB); return; }
#endif
  if (threads <= 256) hipLaunchKernelGGL((dwconv7x7_lane_kernel<CPL, NB, 256>), dim3((unsigned)blocks), dim3(threads), 0, s, x, w49c, bias, y, B, H, W, C, TH, XB, CPB);
  else                hipLaunchKernelGGL((dwconv7x7_lane_kernel<CPL, NB, 1024>), dim3((unsigned)blocks), dim3(threads), 0, s, x, w49c, bias, y, B, H, W, C, TH, XB, CPB);
}

static int g_cpl = -1, g_nb = 3, g_th = 0, g_cpb = 0;
void launch_dwconv7x7_lane(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s) {
  if (g_cpl == -1) {  // tuning knobs: PF_DW7_CPL (channels per lane), PF_DW7_NB (row buffers), PF_DW7_TH (strip height), PF_DW7_CPB (0: 32 channels per block, 1: all)
    const char* e = getenv("PF_DW7_CPL"); g_cpl = e ? atoi(e) : 1;
    const char* n = getenv("PF_DW7_NB"); g_nb = n ? atoi(n) : 3;
    const char* t = getenv("PF_DW7_TH"); g_th = t ? atoi(t) : 0;
    const char* c = getenv("PF_DW7_CPB"); g_cpb = c ? atoi(c) : 0;
  }
#ifdef PF_TUNING_BUILD
  if (g_cpl == 2) { launch_lane<2, 3>(x, w49c, bias, y, B, H, W, C, g_th, 0, s); return; }
  if (g_nb == 2) { launch_lane<1, 2>(x, w49c, bias, y, B, H, W, C, g_th, g_cpb, s); return; }
#endif
  launch_lane<1, 3>(x, w49c, bias, y, B, H, W, C, g_th, g_cpb, s);
}

}  // namespace pf
