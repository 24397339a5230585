// Spatial-reduction attention core of MiT-B3 for gfx950, fp32 MFMA, softmax in registers.
//
// Reference: Attention.forward, mix_transformers.py:108-141 -- attn = softmax(q k^T * d^-0.5) v
// with head_dim d = 64 in every stage and kv_len = 100 always (sr 8/4/2/1 on 80^2..10^2 maps).
//
// One workgroup = 4 waves = 128 query rows of one (batch, head); K and V of that head
// (100 x 64 fp32 each) live in LDS for the whole block.  Each wave owns 32 query rows and
// computes the TRANSPOSED score tile S^T = K Q^T with v_mfma_f32_32x32x2_f32, so a query
// row's scores sit in ONE lane pair (l, l^32): row max / sum are register reductions plus a
// single cross-half shuffle, and P^T in the accumulator layout is *already* the B operand
// of the second product O^T = V^T P^T (k-pair = the two kv rows held by the half-waves).
#include <stdlib.h>

#include "pf_kernels.h"
#include "sb_split.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int HD = 64;        // head dim
static constexpr int KV_PAD = 128;   // kv rows padded to 4 MFMA row blocks
static constexpr int K_ROW = HD + 4; // 68 floats: 68*i mod 64 = 4i -> conflict-free ds_read_b128 over 16 rows

__global__ __launch_bounds__(256) void sr_attention_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                           float* __restrict__ out, unsigned short* __restrict__ out_sb, size_t sb_plane, int N, int M, int heads) {
  // K and V rows of this (batch, head): exactly M rows each (52.8 KB at M = 100: three blocks per CU); the MFMA row
  // blocks that reach past M read a clamped row, whose scores are masked / whose probabilities are zero
  extern __shared__ __attribute__((aligned(16))) float smem_attn[];
  float* Ks = smem_attn;              // [M][K_ROW]
  float* Vs = smem_attn + M * K_ROW;  // [M][HD]
  const int C = heads * HD;
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;

  // stage K, V
  const float* kvb = kv + (long)b * M * 2 * C + h * HD;
  for (int i = tid; i < M * (HD / 4); i += 256) {
    const int row = i / (HD / 4), c4 = i % (HD / 4);
    *reinterpret_cast<float4*>(Ks + row * K_ROW + c4 * 4) = *reinterpret_cast<const float4*>(kvb + (long)row * 2 * C + c4 * 4);
    *reinterpret_cast<float4*>(Vs + row * HD + c4 * 4) = *reinterpret_cast<const float4*>(kvb + (long)row * 2 * C + C + c4 * 4);
  }

  // this lane's query row (clamped; out-of-range rows are computed but not stored)
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int qrow = q0 + l31;
  const int qr = qrow < N ? qrow : N - 1;
  const float* qp = q + ((long)b * N + qr) * C + h * HD + 4 * hi;
  float4 qf[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 v = *reinterpret_cast<const float4*>(qp + 8 * t);
    qf[t] = make_float4(v.x * 0.125f, v.y * 0.125f, v.z * 0.125f, v.w * 0.125f);  // d^-0.5, exact
  }
  __syncthreads();

  // ---- S^T[kv][q] = sum_d K[kv][d] * Q[q][d]; 4 kv blocks of 32 rows
  f32x16 sacc[4];
  int krow[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    krow[c] = min(c * 32 + l31, M - 1);
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[c][e] = 0.f;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    float4 kf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) kf[c] = *reinterpret_cast<const float4*>(Ks + krow[c] * K_ROW + 8 * t + 4 * hi);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].x, qf[t].x, sacc[c], 0, 0, 0);
      sacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].y, qf[t].y, sacc[c], 0, 0, 0);
      sacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].z, qf[t].z, sacc[c], 0, 0, 0);
      sacc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[c].w, qf[t].w, sacc[c], 0, 0, 0);
    }
  }

  // ---- softmax over kv for query column (lane & 31); rows held by this lane:
  //      kv = 32 c + (r & 3) + 8 (r >> 2) + 4 hi
  // A 32-row kv block is either entirely valid (block-uniform test, no per-element masking) or the ragged last one.
  // exp(x) for x <= 0 as v_exp_f32(x * log2 e): relative error <= ~1e-6 only where the weight itself is < 1e-8.
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
  const float mx2 = mx * 1.4426950408889634f;
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (32 * c + 32 <= M) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pexp = __builtin_amdgcn_exp2f(fmaf(sacc[c][r], 1.4426950408889634f, -mx2));
        sacc[c][r] = pexp;
        sum += pexp;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kvi = 32 * c + (r & 3) + 8 * (r >> 2) + 4 * hi;
        const float pexp = kvi < M ? __builtin_amdgcn_exp2f(fmaf(sacc[c][r], 1.4426950408889634f, -mx2)) : 0.f;
        sacc[c][r] = pexp;
        sum += pexp;
      }
    }
  }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;

  // ---- O^T[d][q] = sum_kv V[kv][d] * P^T[kv][q]; A operand = V row (kv chosen per half-wave), B = P register
  f32x16 oacc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[j][e] = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (32 * c + (r & 3) + 8 * (r >> 2) < M) {  // block-uniform skip of fully masked k-pairs (kv >= M in both halves => p = 0)
        const int kvi = min(32 * c + (r & 3) + 8 * (r >> 2) + 4 * hi, M - 1);  // rows >= M carry p = 0
        const float v0 = Vs[kvi * HD + l31];
        const float v1 = Vs[kvi * HD + 32 + l31];
        oacc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sacc[c][r], oacc[0], 0, 0, 0);
        oacc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sacc[c][r], oacc[1], 0, 0, 0);
      }
    }

  // ---- store: lane holds O[q = lane&31][d = 32 j + (r & 3) + 8 (r >> 2) + 4 hi]; 4 consecutive d per float4
  if (qrow < N) {
    const size_t o0 = ((size_t)b * N + qrow) * C + h * HD;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(oacc[j][4 * g] * inv, oacc[j][4 * g + 1] * inv, oacc[j][4 * g + 2] * inv, oacc[j][4 * g + 3] * inv);
        const size_t o = o0 + 32 * j + 8 * g + 4 * hi;
        if (out) *reinterpret_cast<float4*>(out + o) = v;
        if (out_sb) store_sb4(out_sb, sb_plane, o, v);
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Split-f16 form (default): both products on v_mfma_f32_32x32x16_f16 with the 2-way fp16 split of the conv kernels
// (igemm_sb_impl.h): 3 MFMAs per product at 16x the fp32-MFMA rate = 5.3x less matrix-core time, fp32-class accuracy
// (scripts/emulate_split.py: no measurable change end to end).  K and V play the "weight" role: scaled by 16 (exact) and
// split as hi + lo with lo = fp16(16 x - hi) UNSCALED, so that  ql kh + qh kl + qh kh  accumulates in one
// accumulator (|k|, |v| < 4094, saturating beyond; entries below 2^-6 keep an absolute accuracy of 2^-29); q d^-0.5 and the probabilities
// are split as hi + lo (unscaled), after an exact power-of-two factor that keeps lo a normal fp16 number (AT_Q_SCALE, AT_P_EXP).  Same transposed formulation as above: S^T = K Q^T leaves a query's scores in one lane
// pair, and P^T in accumulator layout feeds O^T = V^T P^T directly -- the MFMA k index of that product is then a fixed
// PERMUTATION of the kv index (lane half hi, element e of 16-chunk cc  <->  kv = 16 cc + (e & 3) + 8 (e >> 2) + 4 hi), so
// V is staged TRANSPOSED and in that permuted kv order: every A fragment is one 16-byte LDS read.
typedef _Float16 at_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 at_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int at_u32x4 __attribute__((ext_vector_type(4)));
static constexpr int AT_KS = 72;    // halfs per K row: 144 B -> 16 consecutive rows hit 16 distinct 16-byte slots (36 i mod 64)
static constexpr int AT_VS = 136;   // halfs per V^T row: 128 kv positions + 8 (272 B = 68 words: 68 i mod 64 = 4 i)
static constexpr float AT_KV_SCALE = 16.f;
static constexpr float AT_Q_SCALE = 64.f;          // extra power of two on q d^-0.5 (keeps the low plane of the split out of the fp16 subnormals)
static constexpr float AT_QS = 0.125f * AT_Q_SCALE;
static constexpr float AT_P_EXP = 11.f;            // the probabilities are carried as 2^11 p in (0, 2048]: p = exp(s - max) is mostly far below 2^-3, where the unscaled low
                                                   // plane would be a subnormal (absolute error 2^-25 per element, ~n_kv 2^-25 |v| on the P V product); 2^-11 cancels in 1 / sum

__device__ __forceinline__ unsigned at_pack(float a, float b) {
  const at_h2 v = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, v);
}
// 8 floats -> hi / lo fragments of the split-f16 scheme (sb_split.h: lo = fp16(x - hi), unscaled, on both the activation and the "weight" side)
__device__ __forceinline__ void at_split8(const float (&a)[8], at_u32x4& h, at_u32x4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split2_f16(a[2 * e], a[2 * e + 1], hh[e], ll[e]);
  h = at_u32x4{hh[0], hh[1], hh[2], hh[3]};
  l = at_u32x4{ll[0], ll[1], ll[2], ll[3]};
  split_f16_mfma_pad(l);  // register-direct MFMA operand: sb_split.h
}
__device__ __forceinline__ f32x16 at_mfma(const at_u32x4 a, const at_u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(at_f16x8, a), __builtin_bit_cast(at_f16x8, b), c, 0, 0, 0);
}

// QT: query tiles of 128 rows per block -- K / V of a (batch, head) are staged (fetched, split, V transposed) once per block, so a block that walks
// QT tiles amortises that staging QT-fold (it costs about as much as one tile's two products); the launcher picks QT so that the grid still fills the chip
__global__ __launch_bounds__(256, 2) void sr_attention_f16_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                                  float* __restrict__ out, unsigned short* __restrict__ out_sb, size_t sb_plane, int N, int M, int heads, int QT) {
  extern __shared__ __attribute__((aligned(16))) unsigned short smem_at[];
  unsigned short* Kh = smem_at;              // [M][AT_KS]  hi of 16 K
  unsigned short* Kl = Kh + M * AT_KS;       //             lo (unscaled remainder)
  unsigned short* VTh = Kl + M * AT_KS;      // [64][AT_VS] hi of 16 V, transposed, kv in MFMA k order
  unsigned short* VTl = VTh + HD * AT_VS;
  const int C = heads * HD;
  const int b = blockIdx.z, h = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int nchunk = (M + 15) >> 4;  // 16-wide kv chunks of the second product (7 at M = 100)

  // ---- stage K (row-major) and V (transposed, permuted kv order, zero beyond M)
  const float* kvb = kv + (long)b * M * 2 * C + h * HD;
  for (int i = tid; i < M * (HD / 4); i += 256) {
    const int row = i >> 4, c4 = i & 15;
    const float4 v = *reinterpret_cast<const float4*>(kvb + (long)row * 2 * C + c4 * 4);
    const float a[4] = {v.x * AT_KV_SCALE, v.y * AT_KV_SCALE, v.z * AT_KV_SCALE, v.w * AT_KV_SCALE};
    float hh[4], ll[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float c = __builtin_amdgcn_fmed3f(a[e], -65504.f, 65504.f);
      hh[e] = (float)(_Float16)c;
      ll[e] = c - hh[e];
    }
    *reinterpret_cast<uint2*>(Kh + row * AT_KS + c4 * 4) = make_uint2(at_pack(hh[0], hh[1]), at_pack(hh[2], hh[3]));
    *reinterpret_cast<uint2*>(Kl + row * AT_KS + c4 * 4) = make_uint2(at_pack(ll[0], ll[1]), at_pack(ll[2], ll[3]));
  }
  for (int i = tid; i < nchunk * 16 * (HD / 4); i += 256) {
    const int row = i >> 4, c4 = i & 15;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < M) v = *reinterpret_cast<const float4*>(kvb + (long)row * 2 * C + C + c4 * 4);
    const float a[4] = {v.x * AT_KV_SCALE, v.y * AT_KV_SCALE, v.z * AT_KV_SCALE, v.w * AT_KV_SCALE};
    const int o = row & 15;
    const int pos = (row & ~15) + 8 * ((o >> 2) & 1) + (o & 3) + 4 * (o >> 3);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float c = __builtin_amdgcn_fmed3f(a[e], -65504.f, 65504.f);
      const _Float16 hv = (_Float16)c;
      const _Float16 lv = (_Float16)(c - (float)hv);
      VTh[(c4 * 4 + e) * AT_VS + pos] = __builtin_bit_cast(unsigned short, hv);
      VTl[(c4 * 4 + e) * AT_VS + pos] = __builtin_bit_cast(unsigned short, lv);
    }
  }

  __syncthreads();
  for (int qt = 0; qt < QT; ++qt) {
  // ---- this lane's query row (clamped; out-of-range rows are computed but not stored): B fragments of S^T = K Q^T
  const int q0 = (blockIdx.x * QT + qt) * 128 + wave * 32;
  if (q0 >= N) break;  // wave-uniform; no barrier below
  const int qrow = q0 + l31;
  const int qr = qrow < N ? qrow : N - 1;
  const float* qp = q + ((long)b * N + qr) * C + h * HD + 8 * hi;
  at_u32x4 qh[4], ql[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 v0 = *reinterpret_cast<const float4*>(qp + 16 * t), v1 = *reinterpret_cast<const float4*>(qp + 16 * t + 4);
    // d^-0.5 = 1/8 times AT_Q_SCALE = 64 (exact; undone on the exp2 constant below): q d^-0.5 is typically < 2^-3, where the unscaled low part of the split
    // would be an fp16 subnormal (absolute error 2^-25, sb_split.h); at 8 q it is a normal number down to |q| = 2^-6
    const float a[8] = {v0.x * AT_QS, v0.y * AT_QS, v0.z * AT_QS, v0.w * AT_QS, v1.x * AT_QS, v1.y * AT_QS, v1.z * AT_QS, v1.w * AT_QS};
    at_split8(a, qh[t], ql[t]);
  }

  // ---- S^T[kv][q] * 16: kv blocks of 32 rows (rows past M read a clamped row and are masked below)
  f32x16 sacc[4];
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
        const at_u32x4 kh = *reinterpret_cast<const at_u32x4*>(Kh + krow * AT_KS + 16 * t + 8 * hi);
        const at_u32x4 kl = *reinterpret_cast<const at_u32x4*>(Kl + krow * AT_KS + 16 * t + 8 * hi);
        sacc[c] = at_mfma(kh, ql[t], sacc[c]);
        sacc[c] = at_mfma(kl, qh[t], sacc[c]);
        sacc[c] = at_mfma(kh, qh[t], sacc[c]);
      }
    }
  }

  // ---- softmax over kv for query column (lane & 31); rows held by this lane: kv = 32 c + (r & 3) + 8 (r >> 2) + 4 hi.
  //      The accumulators hold 16 x 64 x the scores: the factor rides on the exp2 constant.
  constexpr float L2E = 1.4426950408889634f / (AT_KV_SCALE * AT_Q_SCALE);
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
  const float mx2 = mx * L2E - AT_P_EXP;
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
  const float inv = 1.0f / (sum * AT_KV_SCALE);  // also undoes the scale of V

  // ---- O^T[d][q] * 16 = sum_kv V^T[d][kv] P^T[kv][q]: chunk cc = registers 8 (cc & 1) .. + 7 of block cc >> 1
  f32x16 oacc[2];
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
      at_u32x4 ph, pl;
      at_split8(pe, ph, pl);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const at_u32x4 vh = *reinterpret_cast<const at_u32x4*>(VTh + (32 * j + l31) * AT_VS + 16 * cc + 8 * hi);
        const at_u32x4 vl = *reinterpret_cast<const at_u32x4*>(VTl + (32 * j + l31) * AT_VS + 16 * cc + 8 * hi);
        oacc[j] = at_mfma(vh, pl, oacc[j]);
        oacc[j] = at_mfma(vl, ph, oacc[j]);
        oacc[j] = at_mfma(vh, ph, oacc[j]);
      }
    }
  }

  // ---- store: lane holds O[q = lane&31][d = 32 j + (r & 3) + 8 (r >> 2) + 4 hi]; 4 consecutive d per float4
  if (qrow < N) {
    const size_t o0 = ((size_t)b * N + qrow) * C + h * HD;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(oacc[j][4 * g] * inv, oacc[j][4 * g + 1] * inv, oacc[j][4 * g + 2] * inv, oacc[j][4 * g + 3] * inv);
        const size_t o = o0 + 32 * j + 8 * g + 4 * hi;
        if (out) *reinterpret_cast<float4*>(out + o) = v;
        if (out_sb) store_sb4(out_sb, sb_plane, o, v);
      }
  }
  }  // query tiles
}

static int g_attn_variant = -1;  // PF_ATTN_VARIANT: 1 = split-f16 MFMA (default), 0 = exact fp32 MFMA
void launch_sr_attention_variant(int variant, const float* q, const float* kv, float* out, int B, int N, int M, int heads, hipStream_t s, unsigned short* out_sb, size_t sb_plane) {
  const dim3 grid((N + 127) / 128, heads, B);
  if (variant == 1) {
    const size_t lds = ((size_t)2 * M * AT_KS + (size_t)2 * HD * AT_VS) * sizeof(unsigned short);
    // query tiles per block: doubled while the grid keeps >= 300 blocks (sweep profiles/r02_tune_attn_qt.txt: stage 1 best at 4, stages 2 / 3 at 2; PF_ATTN_QT overrides)
    static const int qt_env = [] { const char* e = getenv("PF_ATTN_QT"); return e ? atoi(e) : 0; }();
    const int tiles = (N + 127) / 128;
    int QT = 1;
    if (qt_env > 0) QT = qt_env;
    else while (QT < 8 && QT * 2 <= tiles && (long)((tiles + 2 * QT - 1) / (2 * QT)) * heads * B >= 300) QT *= 2;
    const dim3 gridq((tiles + QT - 1) / QT, heads, B);
    hipLaunchKernelGGL(sr_attention_f16_kernel, gridq, dim3(256), lds, s, q, kv, out, out_sb, sb_plane, N, M, heads, QT);
  } else {
    const size_t lds = (size_t)M * (K_ROW + HD) * sizeof(float);
    hipLaunchKernelGGL(sr_attention_kernel, grid, dim3(256), lds, s, q, kv, out, out_sb, sb_plane, N, M, heads);
  }
}
void launch_sr_attention(const float* q, const float* kv, float* out, int B, int N, int M, int heads, hipStream_t s, unsigned short* out_sb, size_t sb_plane) {
  if (g_attn_variant == -1) {
    const char* e = getenv("PF_ATTN_VARIANT");
    g_attn_variant = e ? atoi(e) : 1;
  }
  launch_sr_attention_variant(g_attn_variant, q, kv, out, B, N, M, heads, s, out_sb, sb_plane);
}

}  // namespace pf
