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
// LDS-staged epilogue).  Weights are pre-split on the host into three bf16 planes.  Activations come in one of two
// forms (template ASB):
//   ASB = false: fp32 NHWC; the staging threads split every element (5 VALU ops + packing) once per block and K step;
//   ASB = true : already split by the producing kernel's epilogue (sb_split.h) -- staging is a plain 16-byte copy per
//                plane, the inner loop carries no VALU work besides address selects.  The split is lossless, so both
//                forms give bit-identical results.
// LDS: three bf16 planes per operand, 64-byte rows, XOR piece swizzle (conflict-free ds_read_b128 and staging writes).
// One LDS buffer + a ring of PFD register sets: global loads run PFD K steps ahead of their LDS store (branch-free:
// loads past the last K step go to the out-of-range offset), which is what the small-M GEMMs need -- their K loop is
// a chain of L2 round trips, not of MFMAs.
// This header holds the kernel template and its per-precision launcher; igemm_sb.hip / igemm_sb3.hip / igemm_sb1.hip
// instantiate it for NT = 6 / 3 / 1 partial products (separate translation units: parallel compilation).
#pragma once
#include <stdlib.h>

#include "igemm_common.h"
#include "sb_split.h"

namespace pf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// one 32x32x16 MFMA on 16-byte operand fragments: fp16 operands for the split-f16 scheme, bf16 otherwise
template <bool F16>
__device__ __forceinline__ f32x16 mfma16(const u32x4 a, const u32x4 b, const f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
static constexpr int SB_ROW = BK;  // ushorts per LDS row: 64 bytes = four 16-byte pieces, no padding
// Bank-conflict freedom comes from an XOR swizzle of the piece index with bits 2-3 of the row: a ds_read_b128
// lane group covers 16 rows at one logical piece -> rows r, r+4, r+8, r+12 (same 16-byte slot mod 256 B) land on
// four different pieces; the b64 / b128 staging writes always cover whole 64-byte rows (PMC: the earlier 80-byte
// padded layout cost 8.8e7 SQ_LDS_BANK_CONFLICT cycles per launch on the write side and 25 % more LDS).
__device__ __forceinline__ int sb_piece(int row, int piece) { return piece ^ ((row >> 2) & 3); }

// round-to-nearest-even fp32 -> bf16 bits in the upper half (finite inputs)
__device__ __forceinline__ unsigned rne_hi16(unsigned u) { return (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u; }
// reduced-precision operand forms: one bf16 (round to nearest), or h (truncated) + m (remainder rounded to nearest)
__device__ __forceinline__ uint2 round4_bf16(const float4 v) {
  return make_uint2(sb_pack_hi16(rne_hi16(__float_as_uint(v.x)), rne_hi16(__float_as_uint(v.y))), sb_pack_hi16(rne_hi16(__float_as_uint(v.z)), rne_hi16(__float_as_uint(v.w))));
}
__device__ __forceinline__ void split4_hm(const float4 v, uint2& h, uint2& m) {
  const float a[4] = {v.x, v.y, v.z, v.w};
  unsigned hb[4], mb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hb[e] = __float_as_uint(a[e]) & 0xffff0000u;
    mb[e] = rne_hi16(__float_as_uint(a[e] - __uint_as_float(hb[e])));
  }
  h = make_uint2(sb_pack_hi16(hb[0], hb[1]), sb_pack_hi16(hb[2], hb[3]));
  m = make_uint2(sb_pack_hi16(mb[0], mb[1]), sb_pack_hi16(mb[2], mb[3]));
}

#ifndef SB_W8_WAVES
#define SB_W8_WAVES 4  // min waves per SIMD of the 256x128 / 128x256 8-wave tiles: 4 = two blocks per CU (<= 128 VGPRs)
#endif
// LNF: LayerNorm of the input rows fused into a 1x1 layer (ConvParams::ln).  The staging threads subtract a per-row pivot (the mean of the
// row's first 32 channels: LayerNorm is shift invariant, and the shifted row has no large common offset left to cancel), accumulate
// sum / sum of squares of what they stage, and the epilogue applies  y = rstd (acc - mean colsum) + bias  -- the normalised tensor
// is never written or read (mix_transformers.py:200, :123-126; convnext.py:50-51).
// Ablation / scheduling forms of the linear tiles (tuning builds only; scripts/tune_sb_ablate.py, tiles "sbA<mask>_*" -- WRONG results by construction, timing only:
// 1 = no split arithmetic while staging A, 4 = no barriers in the K loop, 8 = half the fragment reads, 16 / 32 = no global loads of A / B.
// Measured and deleted in r03 (profiles/r03_sb_ablate.txt): the K-step position carried instead of two integer divisions per step ("sbI_*": 64x64 -3 %, 128x128 / 256x128 +3...+20 %)
// and that plus scheduling barriers pinning the loads in front of the MFMAs ("sbPI_*": no better).
// Also measured and deleted: two LDS operand buffers with ONE barrier per K step, the next tile's split + stores behind this tile's MFMAs ("sbD_*", bit-identical):
// nothing at B = 32 (64x64 -1...+5 %, larger tiles +7...+42 %: twice the LDS), -3...-4 % per launch at B = 1 (profiles/r03_sbd_double_buffer.txt, patch in profiles/r03_rejected/).
#ifdef PF_TUNING_BUILD
#define SB_ABL_PARAM , int SABL = 0
#else
#define SB_ABL_PARAM
static constexpr int SABL = 0;
#endif
template <int BM, int BN, int WM, int WN, int MODE, bool ASB, int PFD, int NTERM, bool LNF = false SB_ABL_PARAM>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN == 4 && BM * BN == 128 * 128) ? (PFD == 1 ? 3 : 2) : ((WM * WN == 8 && BM * BN == 256 * 128 && (NTERM == 6 || NTERM == NT_F16X3)) ? SB_W8_WAVES : 1)) void igemm_sb_kernel(const ConvParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int RPP = NT / 8;    // fp32 A rows staged per pass (8 threads x float4 = 32 floats)
  constexpr int RPB = NT / 4;    // bf16 rows staged per pass (4 threads x 16 B = 32 bf16): B, and A when ASB
  constexpr int SM = BM / (WM * 32);
  constexpr int SN = BN / (WN * 32);
  constexpr int A_ROWS = ASB ? (BM + RPB - 1) / RPB : BM / RPP;
  // NTERM partial products per element product: 6 = fp32-accurate (3 planes per operand), 3 = h*h + h*m + m*h (2 planes,
  // ~16 significant bits), 1 = plain bf16 (1 plane)
  // NT_F16X3 (split-f16, the default parity scheme): a ~ ah + al with al = fp16(a - ah) UNSCALED (two fp16 planes, sb_split.h), weights pre-scaled per output
  // channel and split as wh + wl (two fp16 planes in global memory):  al wh + ah wl + ah wh  accumulates in ONE accumulator, 3 MFMAs per product
  // (r02 carried al 2^11 and a third weight operand wh 2^-11 made in registers; gone since r03).
  constexpr bool F16 = NTERM == NT_F16X3;
  // DIRECT: MFMA operands swapped (transposed accumulators) + register epilogue (igemm_common.h epilogue_direct) on the 64 x 64 tiles, i.e. the
  // small-M / latency-bound layers (+2...7 % there, profiles/r02_tune_conv_v3_direct_epilogue.txt).  The larger tiles keep the LDS-staged
  // epilogue: their layers are write-heavy (M >= 51 200, short K) and the 32-byte store granules of the direct form cost them 5-20 %
  constexpr bool DIRECT = BM == 64 && BN == 64;
  constexpr int NPL = F16 ? 2 : (NTERM == 6 ? 3 : (NTERM == 3 ? 2 : 1));  // A planes in LDS
  constexpr int NPB = NPL;                                                // B planes in LDS (split-f16: wh, wl; r02's third operand wh 2^-11 is gone: the low plane is unscaled)
  constexpr int NPG = F16 ? 2 : NPL;                                      // B planes loaded from global memory
  constexpr int NMF = F16 ? 3 : NTERM;                                    // MFMAs per element product
  constexpr int A_REGS = ASB ? A_ROWS * NPL : A_ROWS;  // float4 registers per staged A tile
  constexpr int B_ROWS = (BN + RPB - 1) / RPB;        // BN < RPB (N = 32 tiles): the upper threads stage no B rows
  static_assert(ASB ? (BM % RPB == 0 || BM < RPB) : BM % RPP == 0, "A tile rows must be a multiple of the staging pass");
  static_assert(BN % RPB == 0 || BN < RPB, "B tile rows must be a multiple of the staging pass");
  constexpr int PLANE_A = BM * SB_ROW, PLANE_B = BN * SB_ROW;  // ushorts
  constexpr int EPI_USHORTS = 2 * (WM * 32) * (BN + 4);        // epilogue staging chunk (igemm_common.h)
  constexpr int OPER_USHORTS = NPL * PLANE_A + NPB * PLANE_B;
  constexpr int SMEM_USHORTS = OPER_USHORTS > EPI_USHORTS ? OPER_USHORTS : EPI_USHORTS;
  static_assert(!LNF || (!ASB && MODE == 0), "fused LayerNorm: fp32 rows, no concat");
  __shared__ __attribute__((aligned(16))) unsigned short smem_u[SMEM_USHORTS + (LNF ? 4 * BM : 0)];  // + [BM] (mean, rstd) of the rows
  unsigned short* As = smem_u;                  // [NPL][BM][SB_ROW]
  unsigned short* Bs = smem_u + NPL * PLANE_A;  // [NPB][BN][SB_ROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int nblk1 = tilesM * tilesN;
  // split-K (ConvParams::splitk = S > 1, deep-K launches with too few tiles to fill the chip): the grid holds S copies of the
  // tile set, copy `sidx` contracts K steps [sidx nK / S, (sidx + 1) nK / S) and stores raw sums to P.partial[sidx]
  const int S = p.splitk > 1 ? p.splitk : 1;
  int t = xcd_tile_index(nblk1 * p.groups * S);
  const int sidx = t / (nblk1 * p.groups);
  t -= sidx * nblk1 * p.groups;
  const bool g1 = t >= nblk1;
  if (g1) t -= nblk1;
  const ConvPtrs& P = g1 ? p.g[1] : p.g[0];
  const int m0 = (t / tilesN) * BM;
  const int n0 = (t % tilesN) * BN;

  // ---- staging geometry.  fp32 A: thread -> (row r0 + RPP i, float4 c4 of the 32-float K chunk).
  //      bf16 (B, and A when ASB): thread -> (row rb0 + RPB i, 16-byte piece pc of the 64-byte K chunk), three planes.
  const int c4 = tid & 7;
  const int r0 = tid >> 3;
  const int pc = tid & 3;
  const int rb0 = tid >> 2;
  const int HoWo = p.Ho * p.Wo;
  constexpr int ESZ = ASB ? 2 : 4;  // bytes per A element in global memory
  // one buffer descriptor per A plane (a plane of the largest activation, 320^2 x 64 channels x 2 heads x batch, is
  // 0.8 GB: three of them under one descriptor would run into the out-of-range marker)
  __amdgpu_buffer_rsrc_t rx[ASB ? NPL : 1], rx2[ASB ? NPL : 1];
  if (ASB) {
#pragma unroll
    for (int pl = 0; pl < (ASB ? NPL : 1); ++pl) {
      rx[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.x_sb + pl * (p.x_sb_plane & ~(size_t)1)), 0, p.x_bytes / 2, 0x00020000);
      rx2[pl] = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.x2_sb ? P.x2_sb + pl * (p.x2_sb_plane & ~(size_t)1) : P.x_sb), 0,
                                                  P.x2_sb ? p.x2_bytes / 2 : 0, 0x00020000);
    }
  } else {
    rx[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, p.x_bytes, 0x00020000);
    rx2[0] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x2 ? P.x2 : P.x), 0, P.x2 ? p.x2_bytes : 0, 0x00020000);
  }
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(F16 ? P.w_h16 : P.w_sb), 0, (F16 ? 2u : 5u) * p.w_sb_plane_bytes, 0x00020000);
  // 1x1 / stride 1 / unpadded layers (every nn.Linear: the small-M launches whose blocks are alone on their CU and pay the set-up below in full) skip the pixel
  // arithmetic of the general convolution -- two integer divisions per staged row here, two per K step in load_tiles -- block-uniform branches around ALU code only
  // (the 8-wave tiles with the 128-register cap serve large-M launches, where the set-up is hidden behind the other resident block: not offered there -- the extra
  // live values cost them spills; the same holds for the 128 x 128 tiles of the bf16 schemes with their three operand planes.  Plain fp32 single-input form only:
  // on the split-plane and concat forms the extra values cost a resident block)
  const bool lin = MODE == 0 && !ASB && WM * WN == 4 && (F16 || BM * BN < 128 * 128) && p.KH == 1 && p.KW == 1 && p.stride == 1 && p.pad == 0;
  int a_off1[A_ROWS], a_off2[A_ROWS];
  unsigned long long a_mask[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int rl = ASB ? rb0 + RPB * i : r0 + RPP * i;
    const int m = m0 + rl;
    const bool ok = m < p.M && rl < BM;
    const int mm = ok ? m : 0;
    int iy0 = 0, ix0 = 0, pix = mm;  // a linear layer: output row m reads input pixel m
    if (!lin) {
      const int b = mm / HoWo;
      const int rem = mm - b * HoWo;
      const int oy = rem / p.Wo;
      const int ox = rem - oy * p.Wo;
      iy0 = oy * p.stride - p.pad; ix0 = ox * p.stride - p.pad;
      pix = (b * p.H + iy0) * p.W + ix0;
    }
    // MODE 1 (Cin == 4, the 3-channel stems padded to 4): a float4 is one PIXEL, the 32-float K chunk = 8 taps of one kernel row,
    // this thread's tap (kx = chunk offset / 4 + c4) is added per K step
    a_off1[i] = pix * p.C1 * ESZ + (MODE == 1 ? 0 : (ASB ? pc : c4) * 16);
    a_off2[i] = pix * p.C2 * ESZ + (ASB ? pc : c4) * 16;
    unsigned long long mk = (lin && ok) ? 1ull : 0ull;
    if (ok && !lin)
      for (int ky = 0; ky < p.KH; ++ky)
        for (int kx = 0; kx < p.KW; ++kx)
          if ((unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)p.W) mk |= 1ull << (ky * p.KW + kx);
    a_mask[i] = mk;
  }
  unsigned b_off[B_ROWS];
#pragma unroll
  for (int i = 0; i < B_ROWS; ++i) {
    const int n = n0 + rb0 + RPB * i;
    b_off[i] = (n < p.Cout && rb0 + RPB * i < BN) ? (unsigned)(n * p.KH * p.KWCp + pc * 8) * 2u : OOB;
  }

  // weight planes in global memory: 0 h, 1 m, 2 l (truncation split), 3 round-to-nearest bf16, 4 round-to-nearest m
  // (split-f16: its own two planes, 0 wh, 1 wl)
  constexpr int BPL[3] = {NTERM == 1 ? 3 : 0, NTERM == 3 ? 4 : 1, 2};
  struct Raw { float4 a[A_REGS]; float4 b[B_ROWS][NPG]; };
  Raw raw[PFD];
  const int nJ = p.KWCp / BK;
  const int nK_all = p.KH * nJ;
  const int it0 = (int)((long)sidx * nK_all / S), nK = (int)((long)(sidx + 1) * nK_all / S);  // this block's K steps [it0, nK)

  // branch-free: a step at or past nK loads from the out-of-range offset (zeros, never stored)
  auto load_tiles = [&](int it, Raw& R) {
    const bool live = it < nK;
    int ky = 0, j0 = it * BK, kx = 0, ci0 = j0;  // lin: one tap, K step = channel chunk
    if (!lin) {
      ky = it / nJ;
      j0 = (it - ky * nJ) * BK;
      kx = j0 / p.Cin;
      ci0 = j0 - kx * p.Cin;
    }
    const int bit = (ky * p.KW + kx) & 63;
    const bool first = MODE != 2 || ci0 < p.C1;
    const int toff = ((ky * p.W + kx) * (first ? p.C1 : p.C2) + (first ? ci0 : ci0 - p.C1)) * ESZ;
    const int j1 = j0 + c4 * 4, kx1 = j1 >> 2;  // MODE 1: this thread's own tap of the chunk
    const int bit1 = (ky * p.KW + kx1) & 63, toff1 = (ky * p.W + kx1) * 16;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const bool valid = MODE == 1 ? (live && j1 < p.KWC && ((a_mask[i] >> bit1) & 1ull)) : (live && ((a_mask[i] >> bit) & 1ull));
      const unsigned off = valid ? (unsigned)((first ? a_off1[i] : a_off2[i]) + (MODE == 1 ? toff1 : toff)) : OOB;
      if (ASB) {
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          if (MODE == 2) {
            const float4 v1 = buf_load16(rx[pl], first ? off : OOB);
            const float4 v2 = buf_load16(rx2[pl], first ? OOB : off);
            // one of the two is all-zero bits: OR keeps the other's bf16 bit patterns
            R.a[i * NPL + pl] = make_float4(__uint_as_float(__float_as_uint(v1.x) | __float_as_uint(v2.x)), __uint_as_float(__float_as_uint(v1.y) | __float_as_uint(v2.y)),
                                          __uint_as_float(__float_as_uint(v1.z) | __float_as_uint(v2.z)), __uint_as_float(__float_as_uint(v1.w) | __float_as_uint(v2.w)));
          } else {
            R.a[i * NPL + pl] = buf_load16(rx[pl], off);
          }
        }
      } else if (MODE == 2) {
        const float4 v1 = buf_load16(rx[0], first ? off : OOB);
        const float4 v2 = buf_load16(rx2[0], first ? OOB : off);
        R.a[i] = make_float4(v1.x + v2.x, v1.y + v2.y, v1.z + v2.z, v1.w + v2.w);
      } else if constexpr ((SABL & 16) != 0) {
        float o = __builtin_bit_cast(float, off | 0x3f000000u);
        asm volatile("" : "+v"(o));  // opaque: the split in store_tiles stays
        R.a[i] = make_float4(o, o, o, o);
      } else {
        R.a[i] = buf_load16(rx[0], off);
      }
    }
    const unsigned woff = (unsigned)(ky * p.KWCp + j0) * 2u;
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i)
#pragma unroll
      for (int pl = 0; pl < NPG; ++pl)
        if constexpr ((SABL & 32) != 0) {
          float o = __builtin_bit_cast(float, (b_off[i] + woff) | 0x3c003c00u);
          asm volatile("" : "+v"(o));
          R.b[i][pl] = make_float4(o, o, o, o);
        } else {
          R.b[i][pl] = buf_load16(rw, (live && b_off[i] != OOB) ? b_off[i] + woff + (unsigned)BPL[pl] * p.w_sb_plane_bytes : OOB);
        }
  };
  // statistics in packed fp32 (v_pk_add_f32 / v_pk_fma_f32: two lanes of the sum per instruction -- VALU instructions are paid in MFMA issue time,
  // profiles/r02_cnx_mlp.md): 6 instead of 12 VALU instructions per staged float4
  typedef float lnf2 __attribute__((ext_vector_type(2)));
  float ln_piv[LNF ? A_ROWS : 1];
  lnf2 ln_s1[LNF ? A_ROWS : 1], ln_s2[LNF ? A_ROWS : 1];
#pragma unroll
  for (int i = 0; i < (LNF ? A_ROWS : 1); ++i) { ln_piv[i] = 0.f; ln_s1[i] = lnf2{0.f, 0.f}; ln_s2[i] = lnf2{0.f, 0.f}; }
  auto store_tiles = [&](const Raw& R) {
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      if (ASB) {
        const int row = rb0 + RPB * i;
        if (BM % RPB == 0 || row < BM) {
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            *reinterpret_cast<float4*>(As + pl * PLANE_A + row * SB_ROW + sb_piece(row, pc) * 8) = R.a[i * NPL + pl];
        }
      } else {
        uint2 h, m, l;
        float4 av = R.a[i];
        if (LNF) {
          const float pv = ln_piv[LNF ? i : 0];
          const lnf2 a0 = lnf2{av.x, av.y} - lnf2{pv, pv}, a1 = lnf2{av.z, av.w} - lnf2{pv, pv};
          av = make_float4(a0.x, a0.y, a1.x, a1.y);
          ln_s1[LNF ? i : 0] = (ln_s1[LNF ? i : 0] + a0) + a1;
          ln_s2[LNF ? i : 0] = __builtin_elementwise_fma(a1, a1, __builtin_elementwise_fma(a0, a0, ln_s2[LNF ? i : 0]));
        }
        if constexpr (F16 && (SABL & 1) != 0) {
          h = make_uint2(__builtin_bit_cast(unsigned, av.x), __builtin_bit_cast(unsigned, av.y));
          m = make_uint2(__builtin_bit_cast(unsigned, av.z), __builtin_bit_cast(unsigned, av.w));
        } else if (F16) split4_f16(av, h, m);
        else if (NTERM == 6) split4(av, h, m, l);
        else if (NTERM == 3) split4_hm(av, h, m);
        else h = round4_bf16(av);
        const int row = r0 + RPP * i;
        unsigned short* d = As + row * SB_ROW + sb_piece(row, c4 >> 1) * 8 + (c4 & 1) * 4;
        *reinterpret_cast<uint2*>(d) = h;
        if (NPL > 1) *reinterpret_cast<uint2*>(d + PLANE_A) = m;
        if (NPL > 2) *reinterpret_cast<uint2*>(d + 2 * PLANE_A) = l;
      }
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i)
      if (BN % RPB == 0 || rb0 + RPB * i < BN) {
#pragma unroll
        for (int pl = 0; pl < NPG; ++pl)
          *reinterpret_cast<float4*>(Bs + pl * PLANE_B + (rb0 + RPB * i) * SB_ROW + sb_piece(rb0 + RPB * i, pc) * 8) = R.b[i][pl];
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
    u32x4 af[SM][NPL], bf[SN][F16 ? 3 : NPB];
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two 16-deep chunks per K step; this lane's 8 k-values = piece 2c + hi
      const int po = ((2 * c + hi) ^ swz) * 8;
      if ((SABL & 8) == 0 || c == 0) {
#pragma unroll
      for (int i = 0; i < SM; ++i)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) af[i][pl] = *reinterpret_cast<const u32x4*>(Ab + pl * PLANE_A + i * 32 * SB_ROW + po);
#pragma unroll
      for (int j = 0; j < SN; ++j)
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl) bf[j][pl] = *reinterpret_cast<const u32x4*>(Bb + pl * PLANE_B + j * 32 * SB_ROW + po);
      }
      // six partial products, smallest first; the (i, j) loop is innermost so that consecutive MFMAs never
      // depend on each other's accumulator
      //                                                                                                  split-f16: al ah ah
      constexpr int TA[6] = {F16 ? 1 : (NTERM == 6 ? 2 : (NTERM == 3 ? 1 : 0)), 0, NTERM == 6 ? 1 : 0, 1, 0, 0};  // plane of A: l h m m h h | m h h | h
      constexpr int TB[6] = {0, F16 ? 1 : (NTERM == 6 ? 2 : (NTERM == 3 ? 1 : 0)), NTERM == 6 ? 1 : 0, 0, 1, 0};  // plane of B: h l m h m h | h m h | h | wh2 wl wh
#pragma unroll
      for (int t6 = 0; t6 < NMF; ++t6)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
          for (int j = 0; j < SN; ++j)
            acc[i][j] = DIRECT ? mfma16<F16>(bf[j][TB[t6]], af[i][TA[t6]], acc[i][j]) : mfma16<F16>(af[i][TA[t6]], bf[j][TB[t6]], acc[i][j]);
    }
  };

  // prologue: tiles 0 .. PFD-1 in flight, tile 0 -> LDS (tile k lives in register set k % PFD)
#pragma unroll
  for (int d = 0; d < PFD; ++d) load_tiles(it0 + d, raw[d]);
  if constexpr (LNF) {
    // pivot = MEAN of the row's first 32 channels (the chunk its 8 staging lanes hold now; xor butterfly: the same bits in all 8 lanes).  A single channel as pivot
    // (r02: channel 0) turns an outlier channel -- trained transformers have channels tens of sigma off -- into a large common offset of the shifted row; the chunk
    // mean moves by 1/32 of an outlier at most.
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      float sp = (raw[0].a[i].x + raw[0].a[i].y) + (raw[0].a[i].z + raw[0].a[i].w);
      sp += __shfl_xor(sp, 1); sp += __shfl_xor(sp, 2); sp += __shfl_xor(sp, 4);
      ln_piv[i] = sp * (1.0f / 32.0f);
    }
  }
  store_tiles(raw[0]);
  __syncthreads();
  // Main loop: whole groups of PFD steps with no exit inside, so that the loop header sees ONE load order and the
  // compiler's s_waitcnt analysis keeps the partial vmcnt(N) waits (an exit inside the group rejoins the back edge
  // with a different order and every PFD-th store then drains all loads).  The last 1..PFD tiles are already in
  // flight when the loop ends and are consumed by the straight-line tail.
  int it = it0;
  for (; it + PFD < nK; it += PFD) {
#pragma unroll
    for (int d = 0; d < PFD; ++d) {
      load_tiles(it + d + PFD, raw[d]);  // refill the set whose tile (it + d) is in LDS now
      compute();                         // tile it + d
      if constexpr ((SABL & 4) == 0) __syncthreads();  // every wave has read this step's planes
      store_tiles(raw[(d + 1) % PFD]);   // tile it + d + 1 (the oldest loads in flight)
      if constexpr ((SABL & 4) == 0) __syncthreads();
    }
  }
#pragma unroll
  for (int d = 0; d < PFD; ++d) {
    compute();
    if (it + d + 1 >= nK) break;
    __syncthreads();
    store_tiles(raw[(d + 1) % PFD]);
    __syncthreads();
  }

  float* ln_stat = reinterpret_cast<float*>(smem_u + SMEM_USHORTS);  // [BM][2]: mean of the shifted row, rstd
  if constexpr (LNF) {
    const float invK = 1.0f / (float)p.Cin;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      float a = ln_s1[i].x + ln_s1[i].y, b = ln_s2[i].x + ln_s2[i].y;
#pragma unroll
      for (int sh = 1; sh < 8; sh <<= 1) { a += __shfl_xor(a, sh); b += __shfl_xor(b, sh); }
      const float mu = a * invK;
      const float var = fmaxf(fmaf(-mu, mu, b * invK), 0.f);
      if (c4 == 0) { ln_stat[2 * (r0 + RPP * i)] = mu; ln_stat[2 * (r0 + RPP * i) + 1] = 1.0f / sqrtf(var + p.ln_eps); }
    }
  }
  if (S > 1) {  // raw partial sums; scale, bias, activation and residual are applied by splitk_reduce_kernel
    ConvParams pp = p;
    pp.act = ACT_NONE; pp.post_relu = 0;
    pp.sat = nullptr;  // raw partial accumulators (weights still scaled): nothing to watch here
    ConvPtrs Q;
    Q.y = P.partial + (size_t)sidx * p.M * p.ldy;
    if constexpr (DIRECT) epilogue_direct<SM, SN>(pp, Q, acc, m0, wm0, n0 + wn0, nullptr, nullptr);
    else epilogue_nhwc<BM, BN, WM, WN, SM, SN, NT, SMEM_USHORTS / 2>(pp, Q, acc, reinterpret_cast<float*>(smem_u), m0, n0);
    return;
  }
  if constexpr (DIRECT) {
    if constexpr (LNF) __syncthreads();  // the row statistics written above
    epilogue_direct<SM, SN>(p, P, acc, m0, wm0, n0 + wn0, F16 ? P.w_h16_inv_scale : nullptr, LNF ? ln_stat : nullptr);
  } else {
    epilogue_nhwc<BM, BN, WM, WN, SM, SN, NT, SMEM_USHORTS / 2>(p, P, acc, reinterpret_cast<float*>(smem_u), m0, n0, nullptr, F16 ? P.w_h16_inv_scale : nullptr,
                                                                LNF ? ln_stat : nullptr);
  }
}

template <int BM, int BN, int WM, int WN, int PFD, int NT>
static void launch_sb_cfg(const ConvParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  const dim3 grid(tilesM * tilesN * p.groups * (p.splitk > 1 ? p.splitk : 1)), block(WM * WN * 64);
  const bool asb = p.g[0].x_sb != nullptr;
  if (p.Cin == 4) {  // stems (conv_sb_eligible: fp32 input, no concat)
    if constexpr (NT == NT_F16X3 && (BN <= 128 && BM <= 128)) hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 1, false, PFD, NT>), grid, block, 0, s, p);
  } else if (p.C2 > 0) {
    if (asb) hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 2, true, PFD, NT>), grid, block, 0, s, p);
    else     hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 2, false, PFD, NT>), grid, block, 0, s, p);
  } else if (p.ln) {  // conv_sb_tile_ok: fp32 rows, 1x1, no concat
    hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 0, false, PFD, NT, true>), grid, block, 0, s, p);
  } else {
    if (asb) hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 0, true, PFD, NT>), grid, block, 0, s, p);
    else     hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 0, false, PFD, NT>), grid, block, 0, s, p);
  }
}

#ifdef PF_TUNING_BUILD
template <int BM, int BN, int WM, int WN, int PFD, int MASK>
static void launch_sb_abl(const ConvParams& p, hipStream_t s) {  // split-f16 scheme, one fp32 input, no LayerNorm fusion, no split-K
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  const dim3 grid(tilesM * tilesN * p.groups), block(WM * WN * 64);
  if (p.nterms != NT_F16X3 || p.g[0].x_sb || p.Cin == 4 || p.C2 > 0 || p.ln || p.splitk > 1) return;
  hipLaunchKernelGGL((igemm_sb_kernel<BM, BN, WM, WN, 0, false, PFD, NT_F16X3, false, MASK>), grid, block, 0, s, p);
}
template <int BM, int BN, int WM, int WN, int PFD>
static void launch_sb_abl_set(const ConvParams& p, int k, hipStream_t s) {  // k: 0..3 -> masks 16, 32, 48, 1
  switch (k) {
    case 0: launch_sb_abl<BM, BN, WM, WN, PFD, 16>(p, s); break;
    case 1: launch_sb_abl<BM, BN, WM, WN, PFD, 32>(p, s); break;
    case 2: launch_sb_abl<BM, BN, WM, WN, PFD, 48>(p, s); break;
    default: launch_sb_abl<BM, BN, WM, WN, PFD, 1>(p, s); break;
  }
}
#endif

// tile ids: kSb[] in igemm_sb.hip
template <int NT>
static void launch_conv_sb_nt(const ConvParams& p, int sb_tile, hipStream_t s) {
#ifdef PF_TUNING_BUILD
  if constexpr (NT == NT_F16X3) {
    if (sb_tile >= 12 && sb_tile < 12 + 16) {  // kSb[]: four base tiles x four forms
      const int base = (sb_tile - 12) / 4, k = (sb_tile - 12) % 4;
      if (base == 0) launch_sb_abl_set<64, 64, 2, 2, 1>(p, k, s);
      else if (base == 1) launch_sb_abl_set<64, 64, 2, 2, 3>(p, k, s);
      else if (base == 2) launch_sb_abl_set<128, 128, 2, 2, 1>(p, k, s);
      else launch_sb_abl_set<256, 128, 4, 2, 1>(p, k, s);
      return;
    }
  }
#endif
  switch (sb_tile) {
    case 0: launch_sb_cfg<128, 128, 2, 2, 1, NT>(p, s); break;
    case 1: launch_sb_cfg<64, 64, 2, 2, 1, NT>(p, s); break;
    case 2: launch_sb_cfg<128, 64, 2, 2, 1, NT>(p, s); break;
    case 3: launch_sb_cfg<256, 128, 4, 2, 1, NT>(p, s); break;
    case 4: launch_sb_cfg<128, 256, 2, 4, 1, NT>(p, s); break;  // whole N = 256 per block: A staged once per m-tile
    case 5: launch_sb_cfg<128, 32, 4, 1, 1, NT>(p, s); break;   // N = 32 layers (conv_fuse_conv1)
    case 6: launch_sb_cfg<256, 256, 2, 4, 1, NT>(p, s); break;  // half the global / LDS work per MFMA, one block per CU
    case 7: launch_sb_cfg<64, 64, 2, 2, 2, NT>(p, s); break;
    case 8: launch_sb_cfg<64, 64, 2, 2, 3, NT>(p, s); break;
    case 9: launch_sb_cfg<128, 64, 2, 2, 2, NT>(p, s); break;
    case 10: launch_sb_cfg<128, 32, 4, 1, 2, NT>(p, s); break;
    default: launch_sb_cfg<128, 128, 2, 2, 2, NT>(p, s); break;
  }
}

}  // namespace pf
