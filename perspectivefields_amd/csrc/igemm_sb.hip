// Split-bf16 implicit-GEMM convolution / GEMM for gfx950: fp32-accurate results at 6/16 of the
// fp32-MFMA cost, on v_mfma_f32_32x32x16_bf16.
//
// Every fp32 value is split EXACTLY into three bf16 parts by truncation, a = a_h + a_m + a_l (8 + 8 + 8
// significant bits; r = a - trunc(a) is exact in fp32).  A product is then
//   a*b = a_h b_h + (a_h b_m + a_m b_h) + (a_h b_l + a_l b_h + a_m b_m) + O(2^-24 |a b|)
// i.e. six bf16 MFMAs with fp32 accumulation reproduce the fp32 dot product to fp32 rounding level (the three
// dropped terms are < 3 * 2^-24 relative; bf16 x bf16 products are exact in the fp32 accumulator).  A CPU
// emulation of this scheme is in tests/test_host_logic.py; on the device it is held to the same oracle parity
// as the exact-fp32 kernel (tests/test_gpu_ops.py runs every tile id).  bf16 keeps the fp32 exponent range, so
// there is no overflow / underflow hazard (unlike an fp16 split).
//
// Structure = igemm.hip (same A gather with tap masks and buffer-load range checks, same grouped launch, same
// LDS-staged epilogue): weights are pre-split on the host into three bf16 planes; activations are split by the
// staging threads (5 VALU ops + packing per element, once per block) and written to three bf16 LDS planes with
// 64-byte rows and an XOR piece swizzle (conflict-free ds_read_b128 and staging writes).
// One LDS buffer + register prefetch: global loads of step t+1 fly during the MFMAs of step t.
// [r01] 3x3 256->256 @80^2 (both heads): 177 TFLOP/s fp32-equivalent vs 132 for the exact-fp32 MFMA kernel.
#include <stdlib.h>

#include "igemm_common.h"

namespace pf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
static constexpr int SB_ROW = BK;  // ushorts per LDS row: 64 bytes = four 16-byte pieces, no padding
// Bank-conflict freedom comes from an XOR swizzle of the piece index with bits 2-3 of the row: a ds_read_b128
// lane group covers 16 rows at one logical piece -> rows r, r+4, r+8, r+12 (same 16-byte slot mod 256 B) land on
// four different pieces; the b64 / b128 staging writes always cover whole 64-byte rows (PMC: the earlier 80-byte
// padded layout cost 8.8e7 SQ_LDS_BANK_CONFLICT cycles per launch on the write side and 25 % more LDS).
__device__ __forceinline__ int sb_piece(int row, int piece) { return piece ^ ((row >> 2) & 3); }

__device__ __forceinline__ unsigned pack_hi16(unsigned lo_src, unsigned hi_src) { return (lo_src >> 16) | (hi_src & 0xffff0000u); }

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
  h = make_uint2(pack_hi16(hb[0], hb[1]), pack_hi16(hb[2], hb[3]));
  m = make_uint2(pack_hi16(mb[0], mb[1]), pack_hi16(mb[2], mb[3]));
  l = make_uint2(pack_hi16(lb[0], lb[1]), pack_hi16(lb[2], lb[3]));
}

// PF2: global loads run two K steps ahead (two raw register sets) instead of one
template <int BM, int BN, int WM, int WN, int MODE, bool PF2>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && BM * BN == 128 * 128) ? 3 : 1) void igemm_sb_kernel(const ConvParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int RPP = NT / 8;    // A rows staged per pass (8 threads x float4 = 32 floats)
  constexpr int RPB = NT / 4;    // B rows staged per pass (4 threads x 16 B = 32 bf16)
  constexpr int SM = BM / (WM * 32);
  constexpr int SN = BN / (WN * 32);
  constexpr int A_ROWS = BM / RPP;
  constexpr int B_ROWS = (BN + RPB - 1) / RPB;  // BN < RPB (N = 32 tiles): the upper threads stage no B rows
  static_assert(BM % RPP == 0 && (BN % RPB == 0 || BN < RPB), "tile rows must be a multiple of the staging pass");
  constexpr int PLANE_A = BM * SB_ROW, PLANE_B = BN * SB_ROW;  // ushorts
  constexpr int SMEM_USHORTS = 3 * (PLANE_A + PLANE_B);
  __shared__ __attribute__((aligned(16))) unsigned short smem_u[SMEM_USHORTS];
  unsigned short* As = smem_u;                // [3][BM][SB_ROW]
  unsigned short* Bs = smem_u + 3 * PLANE_A;  // [3][BN][SB_ROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int nblk1 = tilesM * tilesN;
  int t = xcd_tile_index(nblk1 * p.groups);
  const bool g1 = t >= nblk1;
  if (g1) t -= nblk1;
  const ConvPtrs& P = g1 ? p.g[1] : p.g[0];
  const int m0 = (t / tilesN) * BM;
  const int n0 = (t % tilesN) * BN;

  // ---- A staging geometry (identical to igemm.hip)
  const int c4 = tid & 7;
  const int r0 = tid >> 3;
  const int HoWo = p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x2 ? P.x2 : P.x), 0, P.x2 ? p.x2_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.w_sb), 0, 3u * p.w_sb_plane_bytes, 0x00020000);
  int a_off1[A_ROWS], a_off2[A_ROWS];
  unsigned long long a_mask[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + r0 + RPP * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / HoWo;
    const int rem = mm - b * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    const int pix = (b * p.H + iy0) * p.W + ix0;
    a_off1[i] = pix * p.C1 * 4 + c4 * 16;
    a_off2[i] = pix * p.C2 * 4 + c4 * 16;
    unsigned long long mk = 0;
    if (ok)
      for (int ky = 0; ky < p.KH; ++ky)
        for (int kx = 0; kx < p.KW; ++kx)
          if ((unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)p.W) mk |= 1ull << (ky * p.KW + kx);
    a_mask[i] = mk;
  }
  // ---- B staging: thread -> (row rb0 + RPB i, 16-byte piece pc of the 64-byte K chunk), three planes
  const int pc = tid & 3;
  const int rb0 = tid >> 2;
  unsigned b_off[B_ROWS];
#pragma unroll
  for (int i = 0; i < B_ROWS; ++i) {
    const int n = n0 + rb0 + RPB * i;
    b_off[i] = (n < p.Cout && rb0 + RPB * i < BN) ? (unsigned)(n * p.KH * p.KWCp + pc * 8) * 2u : OOB;
  }

  // two raw register sets: loads run TWO K steps ahead of their split/store (one 48-MFMA step is too short to
  // cover an L2/HBM round trip)
  float4 a_raw0[A_ROWS], a_raw1[A_ROWS];
  float4 b_raw0[B_ROWS][3], b_raw1[B_ROWS][3];
  const int nJ = p.KWCp / BK;
  const int nK = p.KH * nJ;

  auto load_tiles = [&](int it, float4 (&a_reg)[A_ROWS], float4 (&b_reg)[B_ROWS][3]) {
    const int ky = it / nJ;
    const int j0 = (it - ky * nJ) * BK;
    const int kx = j0 / p.Cin;
    const int ci0 = j0 - kx * p.Cin;
    const int bit = ky * p.KW + kx;
    if (MODE == 2) {
      const bool first = ci0 < p.C1;
      const int toff = ((ky * p.W + kx) * (first ? p.C1 : p.C2) + (first ? ci0 : ci0 - p.C1)) * 4;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const unsigned off = ((a_mask[i] >> bit) & 1ull) ? (unsigned)((first ? a_off1[i] : a_off2[i]) + toff) : OOB;
        const float4 v1 = buf_load16(rx, first ? off : OOB);
        const float4 v2 = buf_load16(rx2, first ? OOB : off);
        a_reg[i] = make_float4(v1.x + v2.x, v1.y + v2.y, v1.z + v2.z, v1.w + v2.w);
      }
    } else {
      const int toff = ((ky * p.W + kx) * p.C1 + ci0) * 4;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) a_reg[i] = buf_load16(rx, ((a_mask[i] >> bit) & 1ull) ? (unsigned)(a_off1[i] + toff) : OOB);
    }
    const unsigned woff = (unsigned)(ky * p.KWCp + j0) * 2u;
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
        b_reg[i][pl] = buf_load16(rw, b_off[i] == OOB ? OOB : b_off[i] + woff + (unsigned)pl * p.w_sb_plane_bytes);
  };
  auto store_tiles = [&](const float4 (&a_reg)[A_ROWS], const float4 (&b_reg)[B_ROWS][3]) {
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      uint2 h, m, l;
      split4(a_reg[i], h, m, l);
      const int row = r0 + RPP * i;
      unsigned short* d = As + row * SB_ROW + sb_piece(row, c4 >> 1) * 8 + (c4 & 1) * 4;
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + PLANE_A) = m;
      *reinterpret_cast<uint2*>(d + 2 * PLANE_A) = l;
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i)
      if (BN % RPB == 0 || rb0 + RPB * i < BN) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          *reinterpret_cast<float4*>(Bs + pl * PLANE_B + (rb0 + RPB * i) * SB_ROW + sb_piece(rb0 + RPB * i, pc) * 8) = b_reg[i][pl];
      }
  };

  f32x16 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int wm0 = (wave / WN) * (SM * 32);
  const int wn0 = (wave % WN) * (SN * 32);
  const unsigned short* Ab = As + (wm0 + l31) * SB_ROW;
  const unsigned short* Bb = Bs + (wn0 + l31) * SB_ROW;
  const int swz = (l31 >> 2) & 3;  // wm0, wn0 and i*32 are multiples of 32: the swizzle depends on the lane only

  auto compute = [&]() {
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two 16-deep chunks per K step; this lane's 8 k-values = piece 2c + hi
      const int po = ((2 * c + hi) ^ swz) * 8;
      bf16x8 af[SM][3], bf[SN][3];
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(Ab + pl * PLANE_A + i * 32 * SB_ROW + po);
#pragma unroll
      for (int j = 0; j < SN; ++j)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(Bb + pl * PLANE_B + j * 32 * SB_ROW + po);
      // six partial products, smallest first; the (i, j) loop is innermost so that consecutive MFMAs never
      // depend on each other's accumulator
      constexpr int TA[6] = {2, 0, 1, 1, 0, 0};  // plane of A: l h m m h h
      constexpr int TB[6] = {0, 2, 1, 0, 1, 0};  // plane of B: h l m h m h
#pragma unroll
      for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
          for (int j = 0; j < SN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][TA[t6]], bf[j][TB[t6]], acc[i][j], 0, 0, 0);
    }
  };

  if (PF2) {
    // prologue: tile 0 -> LDS, tile 1 -> raw set 1
    load_tiles(0, a_raw0, b_raw0);
    if (nK > 1) load_tiles(1, a_raw1, b_raw1);
    store_tiles(a_raw0, b_raw0);
    __syncthreads();
    for (int it = 0; it < nK; it += 2) {
      // LDS holds tile it; raw1 holds tile it+1 (if any)
      if (it + 2 < nK) load_tiles(it + 2, a_raw0, b_raw0);
      compute();
      if (it + 1 >= nK) break;
      __syncthreads();  // every wave has read tile it
      store_tiles(a_raw1, b_raw1);
      __syncthreads();
      // LDS holds tile it+1; raw0 holds tile it+2 (if any)
      if (it + 3 < nK) load_tiles(it + 3, a_raw1, b_raw1);
      compute();
      if (it + 2 >= nK) break;
      __syncthreads();
      store_tiles(a_raw0, b_raw0);
      __syncthreads();
    }
  } else {
    load_tiles(0, a_raw0, b_raw0);
    store_tiles(a_raw0, b_raw0);
    __syncthreads();
    for (int it = 0; it < nK; ++it) {
      if (it + 1 < nK) load_tiles(it + 1, a_raw0, b_raw0);
      compute();
      __syncthreads();  // every wave has read this step's planes
      if (it + 1 < nK) store_tiles(a_raw0, b_raw0);
      __syncthreads();
    }
  }

  epilogue_nhwc<BM, BN, WM, WN, SM, SN, NT, SMEM_USHORTS / 2>(p, P, acc, reinterpret_cast<float*>(smem_u), m0, n0);
}

// ---------------------------------------------------------------------------------------
struct SbCfg { int bm, bn; const char* name; };
// Two-step-ahead prefetch variants (PF2 = true) were measured and rejected: the second raw register set drops
// occupancy (128x128: 350 registers -> 1 wave/SIMD) and loses 8-25 % on every shape (profiles/r01_tune_conv_v6_sb_pf2.txt).
static const SbCfg kSb[] = {{128, 128, "sb128x128"}, {64, 64, "sb64x64"}, {128, 64, "sb128x64"}, {256, 128, "sb256x128w8"},
                             {128, 256, "sb128x256w8"}, {128, 32, "sb128x32"}, {256, 256, "sb256x256w8"}};
int conv_sb_num_tiles() { return (int)(sizeof(kSb) / sizeof(kSb[0])); }
const char* conv_sb_tile_name(int id) { return kSb[id].name; }
int conv_sb_tile_bm(int id) { return kSb[id].bm; }
int conv_sb_tile_bn(int id) { return kSb[id].bn; }

bool conv_sb_eligible(const ConvParams& p) {
  if (p.nchw_out || (p.Cin % BK) != 0) return false;
  for (int g = 0; g < p.groups; ++g)
    if (!p.g[g].w_sb) return false;
  return true;
}

template <int BM, int BN, int WM, int WN, bool PF2>
static void launch_sb_cfg(const ConvParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  const dim3 grid(tilesM * tilesN * p.groups), block(WM * WN * 64);
  if (p.C2 > 0) hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 2, PF2>), grid, block, 0, s, p);
  else          hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 0, PF2>), grid, block, 0, s, p);
}

void launch_conv_sb(const ConvParams& p, int sb_tile, hipStream_t s) {
  switch (sb_tile) {
    case 0: launch_sb_cfg<128, 128, 2, 2, false>(p, s); break;
    case 1: launch_sb_cfg<64, 64, 2, 2, false>(p, s); break;
    case 2: launch_sb_cfg<128, 64, 2, 2, false>(p, s); break;
    case 3: launch_sb_cfg<256, 128, 4, 2, false>(p, s); break;
    case 4: launch_sb_cfg<128, 256, 2, 4, false>(p, s); break;  // whole N = 256 per block: A staged / split once per m-tile
    case 5: launch_sb_cfg<128, 32, 4, 1, false>(p, s); break;  // N = 32 layers (conv_fuse_conv1)
    default: launch_sb_cfg<256, 256, 2, 4, false>(p, s); break;  // half the global / LDS / split work per MFMA, one block per CU
  }
}

}  // namespace pf
