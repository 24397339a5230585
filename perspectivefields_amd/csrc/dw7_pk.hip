// Depthwise 7x7 (ConvNeXt Block.dwconv, convnext.py:30-32,48) for gfx950, packed-fp32 forms [r04].
//
// The kernels of dw7.hip issue one v_fma_f32 per tap and output: 98 flops per 8 bytes of traffic put the scalar-VALU floor of the op (78.6 TFLOP/s) at 80 % of the
// HBM time, and every launch of the class sat at 1.6-3x that floor.  Here a thread's TWO ADJACENT OUTPUT COLUMNS of one channel form a register pair and every tap is
// one v_pk_fma_f32 (both columns per instruction; the weight is the same for both halves and is broadcast by op_sel, no extra register): half the VALU instructions for
// the same fused multiply-adds in the same order -- the results are bit-identical to dw7.hip's (tests/test_gpu_ops.py::test_dwconv7x7: torch.equal).
//
//   * dwconv7x7_cbp_kernel: the column-blocked streaming kernel (80^2 / 40^2 maps) with packed accumulators.
//   * dwconv7x7_ldsp_kernel: the LDS-tile kernel (maps of <= 20 columns) with packed accumulators, a tile that arrives in PARTS of 7 rows (all loads issued up
//     front, part p written to LDS and consumed while parts p + 1 ... are still in flight: the compute of a block starts after 7 rows, not after the whole tile), the
//     49 x CH weights through LDS (2-3 dwordx4 loads per thread instead of 49 dword loads) and a straight-line body (strip height is a template parameter: no branch in
//     the row loop, so the compiler hoists the next row's LDS reads over this row's FMAs).  CH = 16 lets a block own a whole 20^2 map of 16 channels (no halo re-reads).
// Compiled with -fno-slp-vectorize like dw7.hip (build.py): the packing is explicit.
#include <stdlib.h>

#include <algorithm>

#include "pf_kernels.h"

namespace pf {

typedef float pk2 __attribute__((ext_vector_type(2)));
typedef unsigned int pk_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f32;  // the tile reads keep their LDS address space through the pointer arithmetic below (a generic pointer would become flat_load)

// c += a * w2[HI] (both halves of a by the same weight).  The weights live TWO PER REGISTER PAIR and the instruction's op_sel bits pick the half: written as
// __builtin_elementwise_fma(a, pk2{w, w}, c) the compiler folds the splat into op_sel only where it has a single use and otherwise materialises all 49 weights as
// (w, w) pairs -- 98 registers and a v_mov per weight, which is what this form is here to avoid.
// THE WEIGHT PAIR MUST BE src0.  With the pair as src1 and the HIGH half broadcast (op_sel:[0,1,0] op_sel_hi:[1,1,1]) the instruction returned wrong low-lane results in
// lanes 32-63 -- only while this library's MFMA kernels ran on the same CUs from another stream (8-20 of 30 launches differing, ~10 % relative error in the affected
// outputs), never alone, never beside rocBLAS GEMMs or a streaming kernel; the same operands as src0 (op_sel:[1,0,0]), the low-half broadcast on either source, and the
// compiler's own packed FMAs are clean (profiles/r04_dw7_packed.md; found by tests/test_gpu_e2e.py::test_deferred_paramnet_branch_equals_joined_forward, pinned by
// ::test_packed_dwconv7x7_beside_forward).
template <int HI>
static __device__ __forceinline__ void pk_fma_w(pk2& c, pk2 a, pk2 w2) {
  if (HI) asm("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(a), "v"(w2));
  else    asm("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1]" : "+v"(c) : "v"(a), "v"(w2));
}
template <int K>
static __device__ __forceinline__ void pk_tap(pk2& c, pk2 a, const pk2 (&wk2)[25]) { pk_fma_w<K & 1>(c, a, wk2[K >> 1]); }
template <int KY>
static __device__ __forceinline__ void taps7_row(pk2& c, const pk2* pr, const pk2 (&wk2)[25]) {  // the 7 horizontal taps of kernel row KY; pr[kx] = (input column kx, kx + 1) relative to the output pair
  pk_tap<KY * 7 + 0>(c, pr[0], wk2); pk_tap<KY * 7 + 1>(c, pr[1], wk2); pk_tap<KY * 7 + 2>(c, pr[2], wk2); pk_tap<KY * 7 + 3>(c, pr[3], wk2);
  pk_tap<KY * 7 + 4>(c, pr[4], wk2); pk_tap<KY * 7 + 5>(c, pr[5], wk2); pk_tap<KY * 7 + 6>(c, pr[6], wk2);
}
static __device__ __forceinline__ void taps7(pk2& c, const pk2* pr, const pk2 (&wk2)[25], int ky /*compile-time after unrolling*/) {
  switch (ky) {
    case 0: taps7_row<0>(c, pr, wk2); break;
    case 1: taps7_row<1>(c, pr, wk2); break;
    case 2: taps7_row<2>(c, pr, wk2); break;
    case 3: taps7_row<3>(c, pr, wk2); break;
    case 4: taps7_row<4>(c, pr, wk2); break;
    case 5: taps7_row<5>(c, pr, wk2); break;
    default: taps7_row<6>(c, pr, wk2); break;
  }
}

// ---- column-blocked streaming kernel, packed.  Same structure as dwconv7x7_cb_kernel (dw7.hip): lane = channel, thread = NC adjacent output columns, buffer loads with
// per-lane fixed offsets + scalar row offset, compile-time ring of 7 output rows, loads NB - 1 rows ahead, branch-free memory operations.
template <int NC /*adjacent output columns per thread (even)*/, int NB /*row buffers*/>
__global__ __launch_bounds__(256) void dwconv7x7_cbp_kernel(const float* __restrict__ x, const float* __restrict__ w49c, const float* __restrict__ bias,
                                                            float* __restrict__ y, int B, int H, int W, int C, int TH, int GB /*column groups per block*/) {
  static_assert(NC % 2 == 0, "pairs of columns");
  constexpr int NI = NC + 6, NP = NC / 2;
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
  pk2 wk2[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) wk2[k] = pk2{w49c[(long)(2 * k) * C + c], k < 24 ? w49c[(long)(2 * k + 1) * C + c] : 0.f};
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
  pk2 acc[7][NP];
  pk2 in[NB][NI / 2];  // the input row as aligned pairs (columns 2k, 2k + 1)
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[s][j] = pk2{bv, bv};
  auto load_row = [&](int iy, pk2 (&v)[NI / 2]) {
    const bool ok = (unsigned)iy < (unsigned)H;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(xb, 0, ok ? img_bytes : 0u, 0x00020000);
    const unsigned soff = ok ? (unsigned)iy * row_bytes : 0u;
#pragma unroll
    for (int j = 0; j < NI / 2; ++j) {
      v[j].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff[2 * j], soff, 0));
      v[j].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff[2 * j + 1], soff, 0));
    }
  };
  const int nrows = (y1 - y0) + 6;
  auto do_row = [&](int tt, int r) {
    const int iy = y0 - 3 + tt;
    load_row(iy + NB - 1, in[(r + NB - 1) % NB]);  // prefetch; rows past the strip are loaded but never used
    pk2 pr[NI - 1];                                // (column k, column k + 1) of this input row: the even ones are the row's own pairs, an odd one is one v_pk_mov_b32 (hi of pair k / 2, lo of the next)
#pragma unroll
    for (int k = 0; k < NI - 1; ++k) pr[k] = (k & 1) ? __builtin_shufflevector(in[r % NB][k >> 1], in[r % NB][(k >> 1) + 1], 1, 2) : in[r % NB][k >> 1];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int oy = iy - ky + 3;
      if (oy >= y0 && oy < y1) {  // block-uniform; rows outside the image were loaded as zeros
        const int s = (r - ky + 3 + 7 * NB) % 7;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          pk2 a = acc[s][j];
          taps7(a, &pr[2 * j], wk2, ky);
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
    for (int j = 0; j < NP; ++j) {
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[so][j].x), ry, voff_out[2 * j], soff, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[so][j].y), ry, voff_out[2 * j + 1], soff, 0);
      acc[so][j] = pk2{bv, bv};
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

// ---- the same kernel with the strip height as a template parameter (H % TH == 0): every row of the strip is unrolled with its ring slot, row buffer and the set of
// vertical taps that land inside the strip known at compile time -- no branch between the first load and the last store, so the FMAs of different kernel rows (different
// accumulators) interleave and the scheduler is free to place the next row's loads.
template <int NC, int NB, int TH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NC == 2 ? 4 : 3))) void dwconv7x7_cbps_kernel(
    const float* __restrict__ x, const float* __restrict__ w49c, const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W, int C, int GB) {
  static_assert(NC % 2 == 0, "pairs of columns");
  constexpr int NI = NC + 6, NP = NC / 2, R = TH + 6;
  const int groups = (W + NC - 1) / NC;
  const int slabs = C / 32, tilesX = (groups + GB - 1) / GB, strips = H / TH;
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
  const int y0 = st * TH;
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
  pk2 in[NB][NI / 2];  // the input row as aligned pairs (columns 2k, 2k + 1)
  auto load_row = [&](int iy, pk2 (&v)[NI / 2]) {
    const bool ok = (unsigned)iy < (unsigned)H;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(xb, 0, ok ? img_bytes : 0u, 0x00020000);
    const unsigned soff = ok ? (unsigned)iy * row_bytes : 0u;
#pragma unroll
    for (int j = 0; j < NI / 2; ++j) {
      v[j].x = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff[2 * j], soff, 0));
      v[j].y = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff[2 * j + 1], soff, 0));
    }
  };
#pragma unroll
  for (int d = 0; d < NB - 1; ++d) load_row(y0 - 3 + d, in[d]);  // the first rows' loads go out before the weights'
  pk2 wk2[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) wk2[k] = pk2{w49c[(long)(2 * k) * C + c], k < 24 ? w49c[(long)(2 * k + 1) * C + c] : 0.f};
  const float bv = bias[c];
  pk2 acc[7][NP];
#pragma unroll
  for (int s = 0; s < 7; ++s)
#pragma unroll
    for (int j = 0; j < NP; ++j) acc[s][j] = pk2{bv, bv};
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, img_bytes, 0x00020000);
#pragma unroll
  for (int tt = 0; tt < R; ++tt) {
    if (tt + NB - 1 < R) load_row(y0 - 3 + tt + NB - 1, in[(tt + NB - 1) % NB]);
    pk2 pr[NI - 1];  // (column k, column k + 1) of this input row: the even ones are the row's own pairs, an odd one is one v_pk_mov_b32 (hi of pair k / 2, lo of the next)
#pragma unroll
    for (int k = 0; k < NI - 1; ++k) pr[k] = (k & 1) ? __builtin_shufflevector(in[tt % NB][k >> 1], in[tt % NB][(k >> 1) + 1], 1, 2) : in[tt % NB][k >> 1];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int o = tt - ky;
      if (o >= 0 && o < TH) {  // compile-time
        const int s = (tt - ky + 3) % 7;
#pragma unroll
        for (int j = 0; j < NP; ++j) taps7(acc[s][j], &pr[2 * j], wk2, ky);
      }
    }
    if (tt >= 6) {
      const int so = (tt + 4) % 7;
      const unsigned soff = (unsigned)(y0 + tt - 6) * row_bytes;
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[so][j].x), ry, voff_out[2 * j], soff, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[so][j].y), ry, voff_out[2 * j + 1], soff, 0);
        acc[so][j] = pk2{bv, bv};
      }
    }
  }
}

template <int NC, int NB>
static void launch_cbp(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int TH, hipStream_t s) {
  const int groups = (W + NC - 1) / NC;
  int GB = 8;  // column groups per block (x 32 channels = threads); prefer an even divisor of `groups` (whole waves, no idle groups)
  for (int cand : {8, 6, 4, 2}) if (groups % cand == 0) { GB = cand; break; }
  if (groups % 2 != 0) GB = groups <= 8 ? groups : 8;
  const long blocks = (long)B * ((H + TH - 1) / TH) * ((groups + GB - 1) / GB) * (C / 32);
  if (H % TH == 0 && TH == 20) { hipLaunchKernelGGL((dwconv7x7_cbps_kernel<NC, NB, 20>), dim3((unsigned)blocks), dim3(GB * 32), 0, s, x, w49c, bias, y, B, H, W, C, GB); return; }
  if (H % TH == 0 && TH == 10) { hipLaunchKernelGGL((dwconv7x7_cbps_kernel<NC, NB, 10>), dim3((unsigned)blocks), dim3(GB * 32), 0, s, x, w49c, bias, y, B, H, W, C, GB); return; }
  hipLaunchKernelGGL((dwconv7x7_cbp_kernel<NC, NB>), dim3((unsigned)blocks), dim3(GB * 32), 0, s, x, w49c, bias, y, B, H, W, C, TH, GB);
}

// nc / nb / th <= 0: the automatic choice (same table as the scalar kernel until scripts/tune_dw7.py says otherwise)
void launch_dwconv7x7_cbp_cfg(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int nc, int nb, int th, hipStream_t s) {
  // measured on MI355X at B = 32 (profiles/r04_dw7_packed.md): 80^2: nc4 nb3 th20, 40^2: nc4 nb2 th20, 20^2: nc2 nb3 th20, 10^2: nc2 nb3 th10
  const int NC = nc > 0 ? nc : (W >= 40 ? 4 : 2);
  const int NB = nb > 0 ? nb : (W >= 80 || W < 40 ? 3 : 2);
  int TH = th > 0 ? std::min(th, H) : (H % 20 == 0 ? 20 : (H % 10 == 0 ? 10 : std::min(H, 20)));
  if (th <= 0) {  // small batches: more strips until the chip is full
    const long per_strip = (long)B * C * ((W + NC - 1) / NC) / 64;
    while (TH > 5 && per_strip * ((H + TH - 1) / TH) < 2048) TH = (TH + 1) / 2;
  }
  if (NC >= 4) { if (NB == 2) launch_cbp<4, 2>(x, w49c, bias, y, B, H, W, C, TH, s); else launch_cbp<4, 3>(x, w49c, bias, y, B, H, W, C, TH, s); }
  else         { if (NB == 2) launch_cbp<2, 2>(x, w49c, bias, y, B, H, W, C, TH, s); else launch_cbp<2, 3>(x, w49c, bias, y, B, H, W, C, TH, s); }
}

// ---- LDS-tile kernel, packed, tile in parts.  Block = (image, strip of TH rows, slab of CH channels); thread = (column group of 2 output columns, channel).
// LDS: [49][CH] weights, then the input tile [TH + 6][W + 6][CH] (+ one spare pixel read by the ragged last group of an odd-width map).
template <int CH /*channels per block: 32 or 16*/, int TH /*output rows per block; H % TH == 0*/, int NF /*dwordx4 pieces per thread and part*/>
__global__ __launch_bounds__(320) __attribute__((amdgpu_waves_per_eu(CH == 32 ? 4 : 3))) void dwconv7x7_ldsp_kernel(const float* __restrict__ x, const float* __restrict__ w49c, const float* __restrict__ bias,
                                                             float* __restrict__ y, int B, int H, int W, int C) {
  constexpr int Q = CH / 4, R = TH + 6, PARTS = (R + 6) / 7, NWP = 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const wl = smem;
  float* const tile = smem + 49 * CH;
  const int TW = W + 6;
  const int groups = (W + 1) / 2;
  const int slabs = C / CH, strips = H / TH;
  const int nblk = B * strips * slabs;
  int t;
  {  // XCD-aware order: the slabs of one image strip share an L2
    const int b = blockIdx.x, qd = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    t = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + idx;
  }
  const int slab = t % slabs; t /= slabs;
  const int st = t % strips; t /= strips;
  const int b = t;
  const int tid = (int)threadIdx.x, nthr = (int)blockDim.x;
  const int y0 = st * TH;
  float* const xb = const_cast<float*>(x + (long)b * H * W * C);
  float* const yb = y + (long)b * H * W * C;
  const unsigned img_bytes = (unsigned)((long)H * W * C * 4);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(xb, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w49c), 0, (unsigned)(49 * C * 4), 0x00020000);
  const int g = min(tid / CH, groups - 1), cl = tid % CH;  // the idle lanes of the last wave shadow the last group (they stage, compute, and store nothing)
  const bool g_ok = tid / CH < groups;
  const int c = slab * CH + cl;
  const int x0 = g * 2;
  const float bv = bias[c];
  // ---- every load of the block is issued here, in the order it is needed: weights, then the tile parts (7 tile rows each)
  pk_u32x4 wv[NWP];
#pragma unroll
  for (int i = 0; i < NWP; ++i) {
    const int e = tid + nthr * i;
    const unsigned off = e < 49 * Q ? (unsigned)(((e / Q) * C + slab * CH + (e % Q) * 4) * 4) : 0x80000000u;
    wv[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, off, 0, 0);
  }
  pk_u32x4 v[PARTS][NF];
  const float inv_tw = 1.0f / (float)TW;
#pragma unroll
  for (int p = 0; p < PARTS; ++p) {
    const int prow = (R - 7 * p) < 7 ? (R - 7 * p) : 7;  // tile rows of this part
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (i * 7 >= NF * prow) break;  // a short last part needs proportionally fewer pieces per thread (compile-time)
      const int e = tid + nthr * i;
      const int pix = e / Q, c4 = e % Q;
      const int ty = (int)(((float)pix + 0.5f) * inv_tw), txp = pix - ty * TW;
      const int iy = y0 - 3 + 7 * p + ty, ix = txp - 3;
      const bool ok = ty < prow && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      const unsigned off = ok ? (unsigned)(((iy * W + ix) * C + slab * CH + c4 * 4) * 4) : 0x80000000u;
      v[p][i] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
    }
  }
  unsigned voff_out[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) voff_out[j] = (g_ok && x0 + j < W) ? (unsigned)((x0 + j) * C + c) * 4u : 0x80000000u;
  const unsigned row_bytes = (unsigned)(W * C) * 4u;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, img_bytes, 0x00020000);
  pk2 acc[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) acc[s] = pk2{bv, bv};
  pk2 wk2[25];
  const lds_f32* const wlr = (const lds_f32*)wl + cl;
  const lds_f32* trow = (const lds_f32*)tile + x0 * CH + cl;
#pragma unroll
  for (int tt = 0; tt < R; ++tt) {
    if (tt % 7 == 0) {  // ---- part tt / 7 lands
      const int p = tt / 7;
      const int prow = (R - 7 * p) < 7 ? (R - 7 * p) : 7;
      if (p == 0) {
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
          const int e = tid + nthr * i;
          if (e < 49 * Q) *reinterpret_cast<pk_u32x4*>(wl + (size_t)e * 4) = wv[i];
        }
        if (tid < Q) *reinterpret_cast<float4*>(tile + (size_t)R * TW * CH + tid * 4) = make_float4(0.f, 0.f, 0.f, 0.f);  // the spare pixel
      }
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        if (i * 7 >= NF * prow) break;
        const int e = tid + nthr * i;
        if (e < prow * TW * Q) *reinterpret_cast<pk_u32x4*>(tile + ((size_t)7 * p * TW * Q + e) * 4) = v[p][i];
      }
      __syncthreads();
      if (p == 0) {
#pragma unroll
        for (int k = 0; k < 25; ++k) wk2[k] = pk2{wlr[(2 * k) * CH], k < 24 ? wlr[(2 * k + 1) * CH] : 0.f};
      }
    }
    // ---- input row tt of the tile feeds output rows o = tt - ky (slot (r - ky + 3) mod 7, r = tt mod 7); after it output row tt - 6 is complete
    const int r = tt % 7;
    const lds_f32* te = trow + tt * TW * CH;
    pk2 pe[4], po[3];  // tile columns (2i, 2i + 1): one ds_read2_b32 each, straight into an aligned pair; (2i + 1, 2i + 2): one v_pk_mov_b32 (reading them from LDS as well measured the same)
#pragma unroll
    for (int i = 0; i < 4; ++i) pe[i] = pk2{te[(2 * i) * CH], te[(2 * i + 1) * CH]};
#pragma unroll
    for (int i = 0; i < 3; ++i) po[i] = __builtin_shufflevector(pe[i], pe[i + 1], 1, 2);
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      const int o = tt - ky;
      if (o >= 0 && o < TH) {  // compile-time
        const int s = (r - ky + 3 + 7) % 7;
        const pk2 pr[7] = {pe[0], po[0], pe[1], po[1], pe[2], po[2], pe[3]};
        pk2 a = acc[s];
        taps7(a, pr, wk2, ky);
        acc[s] = a;
      }
    }
    if (tt >= 6) {
      const int so = (r + 4) % 7;
      const unsigned soff = (unsigned)(y0 + tt - 6) * row_bytes;
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[so].x), ry, voff_out[0], soff, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[so].y), ry, voff_out[1], soff, 0);
      acc[so] = pk2{bv, bv};
    }
  }
}

template <int CH, int TH, int NF>
static bool try_ldsp(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s) {
  constexpr int Q = CH / 4, R = TH + 6;
  if (H % TH != 0 || C % CH != 0) return false;
  const int TW = W + 6, groups = (W + 1) / 2;
  const int threads = (CH * groups + 63) / 64 * 64;
  if (threads > 320 || (long)7 * TW * Q > (long)NF * threads || 49 * Q > 3 * threads) return false;
  const size_t lds = ((size_t)49 * CH + ((size_t)R * TW + 1) * CH) * 4;
  if (lds > 64 * 1024) return false;
  const long blocks = (long)B * (H / TH) * (C / CH);
  hipLaunchKernelGGL((dwconv7x7_ldsp_kernel<CH, TH, NF>), dim3((unsigned)blocks), dim3(threads), lds, s, x, w49c, bias, y, B, H, W, C);
  return true;
}

// maps of 8 .. 20 columns with 32 channels per block, up to 40 columns with 16; ch in {0 (automatic), 16, 32, < 0: never}; th in {0 (automatic), 5, 10, 20}.
// false: shape / configuration not covered (the caller falls back)
bool launch_dwconv7x7_ldsp(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, int ch, int th, hipStream_t s) {
  if (W > 40 || W < 8 || H < 5 || ch < 0) return false;  // ch < 0: the caller wants the streaming kernel
  if (ch == 0) ch = W > 20 ? 16 : 32;
  if (th > H) th = 0;  // a preference that does not apply to this map
  if (th <= 0) th = H % 10 == 0 ? 10 : (H % 5 == 0 ? 5 : 0);
  if (ch == 32) {
    if (th == 10) return try_ldsp<32, 10, 5>(x, w49c, bias, y, B, H, W, C, s);
    if (th == 5) return try_ldsp<32, 5, 5>(x, w49c, bias, y, B, H, W, C, s);
  } else if (ch == 16) {
    if (th == 20) return try_ldsp<16, 20, 5>(x, w49c, bias, y, B, H, W, C, s);
    if (th == 10) return try_ldsp<16, 10, 5>(x, w49c, bias, y, B, H, W, C, s);
    if (th == 5) return try_ldsp<16, 5, 5>(x, w49c, bias, y, B, H, W, C, s);
  }
  return false;
}

}  // namespace pf
