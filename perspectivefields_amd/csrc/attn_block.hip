// The attention half of a MiT block with ONE head of 64 channels (stage 1 of MiT-B3: 80 x 80 = 6400 tokens, C = 64, spatial reduction 8 -> 100 key / value rows) as
// one kernel:
//     y = x + proj( softmax( (LayerNorm_1(x) Wq^T + bq) K^T / 8 ) V )            (Block.forward, mix_transformers.py:199; Attention.forward :108-141)
// It replaces three launches of the forward -- the q projection (204 800 x 64 x 64: a 105 MB round trip of the token map at 3 TB/s, not a GEMM), the attention core and
// the output projection with its residual (157 MB) -- and two HBM round trips of the 52 MB map: the kernel reads the block's rows of x and K / V, and writes the rows back.
// (At C = 320 the same fusion does not pay -- profiles/r06_attention.md -- because q / proj are 0.4 MB weight streams per 64-row block there; here both matrices are
// 16 KB and live in LDS next to K / V.)
//
// Everything between the row load and the row store stays in registers, because every product is TRANSPOSED (weights / keys / values as the MFMA's A operand, the
// wave's 32 token rows as its B operand) and the contraction index of an MFMA may be permuted freely as long as both operands agree:
//   * in the 32 x 32 C/D layout lane (row l = lane & 31, half hi = lane >> 5) owns, per accumulator, the channels n = 32 tile + 8 g + 4 hi + e (g, e < 4) of ITS row;
//   * the next product needs, as B operand of chunk t (16 contraction values), 8 values per lane: the accumulator registers of g = 2 (t & 1), 2 (t & 1) + 1 are exactly
//     8 values of chunk t -- in the order  k = 16 t + 8 (j >> 2) + 4 hi + (j & 3)  instead of the standard  16 t + 8 hi + j;
//   * so the A operands are PACKED in that order: Wq / Wproj fragments on the host (attn64_pack), K by the order of its 8-byte pieces in LDS (staging below).  V^T's
//     contraction index is the kv index, permuted as in attn.hip.  The row of x itself is loaded in the accumulator layout, so it is the LayerNorm input, the B operand
//     of the q product (Wq packed in the same order) and, unchanged, the residual of the epilogue.
// Block = 8 waves (two per SIMD), one block per CU: K / V of the image (hi / lo planes, 63.6 KB at 100 rows), both weight matrices (32 KB) and the per-channel tables
// are staged once and serve QT tiles of 32 rows per wave.  Arithmetic of the attention core (scales, splits, softmax) is that of sr_attention_f16_kernel.
#include <stdlib.h>

#include <vector>

#include "host_pack.h"
#include "sb_split.h"

namespace pf {

namespace {

typedef float ab_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ab_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 ab_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int ab_u32x4 __attribute__((ext_vector_type(4)));

constexpr int AB_C = 64;          // channels = head dim
constexpr int AB_KS = 72;         // halfs per K row in LDS (attn.hip AT_KS)
constexpr int AB_VS = 136;        // halfs per V^T row (attn.hip AT_VS)
constexpr float AB_KV_SCALE = 16.f, AB_Q_SCALE = 64.f, AB_QS = 0.125f * AB_Q_SCALE, AB_P_EXP = 11.f;   // attn.hip: AT_KV_SCALE, AT_Q_SCALE, AT_QS, AT_P_EXP
constexpr int AB_WBYTES = 2 * 4 * 2 * 2 * 1024;   // [matrix q / proj][chunk 4][n tile 2][plane 2] fragments of 1 KB
constexpr int AB_TAB = 6 * AB_C;                  // ln gamma, ln beta, q inverse scale, q bias, proj inverse scale, proj bias

__device__ __forceinline__ unsigned ab_pack(float a, float b) {
  const ab_h2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void ab_split8(const float (&a)[8], ab_u32x4& h, ab_u32x4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split2_f16(a[2 * e], a[2 * e + 1], hh[e], ll[e]);
  h = ab_u32x4{hh[0], hh[1], hh[2], hh[3]};
  l = ab_u32x4{ll[0], ll[1], ll[2], ll[3]};
  split_f16_mfma_pad(l);  // register-direct MFMA operand: sb_split.h
}
__device__ __forceinline__ ab_f32x16 ab_mfma(const ab_u32x4 a, const ab_u32x4 b, const ab_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(ab_f16x8, a), __builtin_bit_cast(ab_f16x8, b), c, 0, 0, 0);
}
// the three partial products of the split-f16 scheme into one accumulator, smallest first
__device__ __forceinline__ ab_f32x16 ab_mma3(const ab_u32x4 ah, const ab_u32x4 al, const ab_u32x4 bh, const ab_u32x4 bl, ab_f32x16 c) {
  c = ab_mfma(ah, bl, c);
  c = ab_mfma(al, bh, c);
  return ab_mfma(ah, bh, c);
}

}  // namespace

__global__ __launch_bounds__(512, 1) void mit_attn64_kernel(const MitAttn64Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_ab[];
  const int M = p.M, N = p.N;
  unsigned short* Kh = reinterpret_cast<unsigned short*>(smem_ab);   // [M][AB_KS]  hi of 16 K, the 8-byte pieces of every 16-chunk in the order 0, 2, 1, 3
  unsigned short* Kl = Kh + M * AB_KS;
  unsigned short* VTh = Kl + M * AB_KS;                              // [64][AB_VS] hi of 16 V, transposed, kv in MFMA k order
  unsigned short* VTl = VTh + AB_C * AB_VS;
  const unsigned char* Wf = reinterpret_cast<const unsigned char*>(VTl + AB_C * AB_VS);
  const float* tabs = reinterpret_cast<const float*>(Wf + AB_WBYTES);
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nchunk = (M + 15) >> 4;

  // ---- stage K (row-major, permuted pieces), V (transposed, permuted kv order, zero beyond M), the weight fragments and the tables
  const float* kvb = p.kv + (long)b * M * 2 * AB_C;
  for (int i = tid; i < M * (AB_C / 4); i += 512) {
    const int row = i >> 4, c4 = i & 15;
    const float4 v = *reinterpret_cast<const float4*>(kvb + (long)row * 2 * AB_C + c4 * 4);
    const float a[4] = {v.x * AB_KV_SCALE, v.y * AB_KV_SCALE, v.z * AB_KV_SCALE, v.w * AB_KV_SCALE};
    float hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float c = __builtin_amdgcn_fmed3f(a[e], -65504.f, 65504.f);
      hh[e] = (float)(_Float16)c;
      ll[e] = c - hh[e];
    }
    const int pos = (c4 & ~3) + (((c4 & 1) << 1) | ((c4 >> 1) & 1));   // piece c4 of its 16-chunk -> slot: 0, 2, 1, 3
    *reinterpret_cast<uint2*>(Kh + row * AB_KS + pos * 4) = make_uint2(ab_pack(hh[0], hh[1]), ab_pack(hh[2], hh[3]));
    *reinterpret_cast<uint2*>(Kl + row * AB_KS + pos * 4) = make_uint2(ab_pack(ll[0], ll[1]), ab_pack(ll[2], ll[3]));
  }
  for (int i = tid; i < nchunk * 16 * (AB_C / 4); i += 512) {
    const int row = i >> 4, c4 = i & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < M) v = *reinterpret_cast<const float4*>(kvb + (long)row * 2 * AB_C + AB_C + c4 * 4);
    const float a[4] = {v.x * AB_KV_SCALE, v.y * AB_KV_SCALE, v.z * AB_KV_SCALE, v.w * AB_KV_SCALE};
    const int o = row & 15;
    const int pos = (row & ~15) + 8 * ((o >> 2) & 1) + (o & 3) + 4 * (o >> 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float c = __builtin_amdgcn_fmed3f(a[e], -65504.f, 65504.f);
      const _Float16 hv = (_Float16)c;
      const _Float16 lv = (_Float16)(c - (float)hv);
      VTh[(c4 * 4 + e) * AB_VS + pos] = __builtin_bit_cast(unsigned short, hv);
      VTl[(c4 * 4 + e) * AB_VS + pos] = __builtin_bit_cast(unsigned short, lv);
    }
  }
  for (int i = tid; i < AB_WBYTES / 16; i += 512)
    reinterpret_cast<ab_u32x4*>(const_cast<unsigned char*>(Wf))[i] = reinterpret_cast<const ab_u32x4*>(p.wfr)[i];
  for (int i = tid; i < AB_TAB / 4; i += 512) reinterpret_cast<float4*>(const_cast<float*>(tabs))[i] = reinterpret_cast<const float4*>(p.tab)[i];
  __syncthreads();

  const float* t_g = tabs, *t_b = tabs + AB_C, *t_qi = tabs + 2 * AB_C, *t_qb = tabs + 3 * AB_C, *t_pi = tabs + 4 * AB_C, *t_pb = tabs + 5 * AB_C;
  auto wfrag = [&](int mat, int c, int nt, int plane) { return *reinterpret_cast<const ab_u32x4*>(Wf + ((((mat * 4 + c) * 2 + nt) * 2 + plane) * 1024) + lane * 16); };

  for (int qt = 0; qt < p.QT; ++qt) {
    const int q0 = ((blockIdx.x * p.QT + qt) * 8 + wave) * 32;
    if (q0 >= N) break;  // wave-uniform; no barrier below
    const int qrow = q0 + l31;
    const int qr = qrow < N ? qrow : N - 1;   // rows past the end: a valid row, computed and not stored
    // ---- the row in the accumulator layout: xr[nt][g] = channels 32 nt + 8 g + 4 hi .. + 3
    const float* xp = p.x + ((long)b * N + qr) * AB_C + 4 * hi;
    float4 xr[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) xr[nt][g] = *reinterpret_cast<const float4*>(xp + 32 * nt + 8 * g);
    // ---- LayerNorm_1, two passes over the registers like F.layer_norm (the other half of the row sits in lane ^ 32)
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) s += (xr[nt][g].x + xr[nt][g].y) + (xr[nt][g].z + xr[nt][g].w);
    s += __shfl_xor(s, 32, 64);
    const float mu = s * (1.0f / AB_C);
    float ss = 0.f;
    float4 xn[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = xr[nt][g];
        xn[nt][g] = make_float4(v.x - mu, v.y - mu, v.z - mu, v.w - mu);
        ss = fmaf(xn[nt][g].x, xn[nt][g].x, fmaf(xn[nt][g].y, xn[nt][g].y, fmaf(xn[nt][g].z, xn[nt][g].z, fmaf(xn[nt][g].w, xn[nt][g].w, ss))));
      }
    ss += __shfl_xor(ss, 32, 64);
    const float rs = 1.0f / sqrtf(ss * (1.0f / AB_C) + p.ln_eps);
    // ---- q^T = Wq LN(x)^T: B fragments of chunk c = 2 nt + (g >> 1) straight from the normalised registers
    ab_u32x4 bh[4], bl[4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int g = 2 * gp + u, n = 32 * nt + 8 * g + 4 * hi;
          const float4 gm = *reinterpret_cast<const float4*>(t_g + n), be = *reinterpret_cast<const float4*>(t_b + n);
          const float4 v = xn[nt][g];
          a[4 * u] = fmaf(v.x * rs, gm.x, be.x); a[4 * u + 1] = fmaf(v.y * rs, gm.y, be.y); a[4 * u + 2] = fmaf(v.z * rs, gm.z, be.z); a[4 * u + 3] = fmaf(v.w * rs, gm.w, be.w);
        }
        ab_split8(a, bh[2 * nt + gp], bl[2 * nt + gp]);
      }
    ab_f32x16 qacc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) qacc[nt][e] = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) qacc[nt] = ab_mma3(wfrag(0, c, nt, 0), wfrag(0, c, nt, 1), bh[c], bl[c], qacc[nt]);
    }
    // ---- q = acc / S + bias (watched against the attention window), x d^-0.5 x 64, split: B fragments of S^T = K Q^T, chunk t = 2 nt + (g >> 1)
    ab_u32x4 qh[4], ql[4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int g = 2 * gp + u, n = 32 * nt + 8 * g + 4 * hi;
          const float4 iv = *reinterpret_cast<const float4*>(t_qi + n), bb = *reinterpret_cast<const float4*>(t_qb + n);
          const float q0v = fmaf(qacc[nt][4 * g], iv.x, bb.x), q1v = fmaf(qacc[nt][4 * g + 1], iv.y, bb.y), q2v = fmaf(qacc[nt][4 * g + 2], iv.z, bb.z), q3v = fmaf(qacc[nt][4 * g + 3], iv.w, bb.w);
          if (p.sat) sat_watch4(p.sat, 8188.f, q0v, q1v, q2v, q3v);
          a[4 * u] = q0v * AB_QS; a[4 * u + 1] = q1v * AB_QS; a[4 * u + 2] = q2v * AB_QS; a[4 * u + 3] = q3v * AB_QS;
        }
        ab_split8(a, qh[2 * nt + gp], ql[2 * nt + gp]);
      }

    // ---- S^T[kv][q] * 16 * 64: kv blocks of 32 rows (rows past M read a clamped row and are masked below)
    ab_f32x16 sacc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[c][e] = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (32 * c < M) {  // block-uniform
        const int krow = min(c * 32 + l31, M - 1);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const ab_u32x4 kh = *reinterpret_cast<const ab_u32x4*>(Kh + krow * AB_KS + 16 * t + 8 * hi);
          const ab_u32x4 kl = *reinterpret_cast<const ab_u32x4*>(Kl + krow * AB_KS + 16 * t + 8 * hi);
          sacc[c] = ab_mma3(kh, kl, qh[t], ql[t], sacc[c]);
        }
      }
    }
    // ---- softmax over kv for query column l31; rows held by this lane: kv = 32 c + (r & 3) + 8 (r >> 2) + 4 hi (attn.hip)
    constexpr float L2E = 1.4426950408889634f / (AB_KV_SCALE * AB_Q_SCALE);
    float mx = -3.0e38f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (32 * c + 32 <= M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[c][r]);
      } else if (32 * c < M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kvi = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (kvi < M) mx = fmaxf(mx, sacc[c][r]);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mx2 = mx * L2E - AB_P_EXP;
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (32 * c + 32 <= M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pexp = __builtin_amdgcn_exp2f(fmaf(sacc[c][r], L2E, -mx2));
          sacc[c][r] = pexp;
          sum += pexp;
        }
      } else if (32 * c < M) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kvi = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float pexp = kvi < M ? __builtin_amdgcn_exp2f(fmaf(sacc[c][r], L2E, -mx2)) : 0.f;
          sacc[c][r] = pexp;
          sum += pexp;
        }
      }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / (sum * AB_KV_SCALE);  // also undoes the scale of V

    // ---- O^T[d][q] * 16 = sum_kv V^T[d][kv] P^T[kv][q]
    ab_f32x16 oacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) oacc[j][e] = 0.f;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      if (cc < nchunk) {  // block-uniform
        float pe[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) pe[e] = sacc[cc >> 1][8 * (cc & 1) + e];
        ab_u32x4 ph, pl;
        ab_split8(pe, ph, pl);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const ab_u32x4 vh = *reinterpret_cast<const ab_u32x4*>(VTh + (32 * j + l31) * AB_VS + 16 * cc + 8 * hi);
          const ab_u32x4 vl = *reinterpret_cast<const ab_u32x4*>(VTl + (32 * j + l31) * AB_VS + 16 * cc + 8 * hi);
          oacc[j] = ab_mma3(vh, vl, ph, pl, oacc[j]);
        }
      }
    }
    // ---- y^T = Wproj O^T: the attention output (x 1 / sum) is the B operand, chunk t = 2 j + (g >> 1)
    ab_u32x4 oh[4], ol[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = oacc[j][8 * gp + u] * inv;
        ab_split8(a, oh[2 * j + gp], ol[2 * j + gp]);
      }
    ab_f32x16 yacc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) yacc[nt][e] = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t) yacc[nt] = ab_mma3(wfrag(1, t, nt, 0), wfrag(1, t, nt, 1), oh[t], ol[t], yacc[nt]);
    }
    // ---- y = acc / S + bias + x: the residual is the row as it was loaded (same channels)
    if (qrow < N) {
      float* yp = p.y + ((long)b * N + qrow) * AB_C + 4 * hi;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 32 * nt + 8 * g + 4 * hi;
          const float4 iv = *reinterpret_cast<const float4*>(t_pi + n), bb = *reinterpret_cast<const float4*>(t_pb + n);
          const float4 r = xr[nt][g];
          const float4 w = make_float4(fmaf(yacc[nt][4 * g], iv.x, bb.x) + r.x, fmaf(yacc[nt][4 * g + 1], iv.y, bb.y) + r.y, fmaf(yacc[nt][4 * g + 2], iv.z, bb.z) + r.z,
                                       fmaf(yacc[nt][4 * g + 3], iv.w, bb.w) + r.w);
          if (p.sat) sat_watch4(p.sat, p.sat_limit, w.x, w.y, w.z, w.w);
          *reinterpret_cast<float4*>(yp + 32 * nt + 8 * g) = w;
        }
    }
  }
}

bool mit_attn64_supported(int C, int heads, int kv_rows) { return C == AB_C && heads == 1 && kv_rows >= 1 && kv_rows <= 128; }

void launch_mit_attn64(const MitAttn64Args& a, int num_cus, hipStream_t s) {
  MitAttn64Args p = a;
  // tiles of 32 rows, 8 waves per block, QT tiles per wave: ONE round of blocks (one block per CU) wherever the work allows it
  const int tiles = (p.N + 31) / 32;
  int QT = (int)(((long)tiles * p.B + 8L * num_cus - 1) / (8L * num_cus));
  QT = QT < 1 ? 1 : (QT > 8 ? 8 : QT);
  p.QT = QT;
  const size_t lds = ((size_t)2 * p.M * AB_KS + (size_t)2 * AB_C * AB_VS) * sizeof(unsigned short) + AB_WBYTES + AB_TAB * sizeof(float);
  {  // > 64 KB of dynamic LDS needs the attribute, once per device of the process (a handle is bound to one device; a process may hold several)
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !done[dev])
      done[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(mit_attn64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024) == hipSuccess;
  }
  const dim3 grid((tiles + 8 * QT - 1) / (8 * QT), p.B);
  hipLaunchKernelGGL(mit_attn64_kernel, grid, dim3(512), lds, s, p);
}

// Host side: LayerNorm-1 table, and both 64 x 64 matrices as split-f16 planes (per-output-channel power-of-two scale, host_pack.h split_f16x2) in MFMA fragment order
// with the PERMUTED contraction index: fragment (matrix, chunk c, n tile, plane), lane (l31, hi), element j  =  Ws[32 nt + l31][16 c + 8 (j >> 2) + 4 hi + (j & 3)]
void attn64_pack(const float* ln_g, const float* ln_b, const float* q_w, const float* q_b, const float* p_w, const float* p_b, std::vector<unsigned short>* wfr, std::vector<float>* tab) {
  const pf_host::F16Planes qs = pf_host::split_f16x2(std::vector<float>(q_w, q_w + (size_t)AB_C * AB_C), AB_C), ps = pf_host::split_f16x2(std::vector<float>(p_w, p_w + (size_t)AB_C * AB_C), AB_C);
  const std::vector<unsigned short>&q_planes = qs.planes, &p_planes = ps.planes;
  const std::vector<float>&q_inv = qs.inv_scale, &p_inv = ps.inv_scale;
  wfr->assign(AB_WBYTES / 2, 0);
  const size_t n_all = (size_t)AB_C * AB_C;
  for (int mat = 0; mat < 2; ++mat) {
    const std::vector<unsigned short>& pl = mat ? p_planes : q_planes;
    for (int c = 0; c < 4; ++c)
      for (int nt = 0; nt < 2; ++nt)
        for (int plane = 0; plane < 2; ++plane)
          for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
              const int l31 = lane & 31, hi = lane >> 5;
              const int n = 32 * nt + l31, k = 16 * c + 8 * (j >> 2) + 4 * hi + (j & 3);
              (*wfr)[((((size_t)(mat * 4 + c) * 2 + nt) * 2 + plane) * 64 + lane) * 8 + j] = pl[plane * n_all + (size_t)n * AB_C + k];
            }
  }
  tab->resize(AB_TAB);
  for (int n = 0; n < AB_C; ++n) {
    (*tab)[n] = ln_g[n]; (*tab)[AB_C + n] = ln_b[n];
    (*tab)[2 * AB_C + n] = q_inv[n]; (*tab)[3 * AB_C + n] = q_b[n];
    (*tab)[4 * AB_C + n] = p_inv[n]; (*tab)[5 * AB_C + n] = p_b[n];
  }
}

}  // namespace pf
