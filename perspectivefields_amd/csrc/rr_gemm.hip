// Row-resident split-f16 GEMM for the small-M linear layers (MiT stages 2-4, ConvNeXt stages 2-4: 3 200 - 51 200 rows; mix_transformers.py:26-29,80-88,
// convnext.py:34-38,93) and the convs that ARE linear layers on gathered rows (kernel == stride, no padding: the MiT spatial-reduction convs
// mix_transformers.py:84-88 and the ConvNeXt down-sampling convs convnext.py:95-101).
//
// Why a second GEMM kernel (profiles/r03_candidates.md, r03_sb_ablate.txt): on these shapes the LDS-tiled kernel (igemm_sb_impl.h) runs its 64 x 64 tiles at 13 %
// MFMA-busy and removing ALL of its global loads gains only 20 %.  Its 32 x 32 wave tiles read 4 KB of fragments from LDS per 3 MFMAs (1365 B per MFMA against a budget
// of 128 B/clk x 8 clk = 1024), it passes two barriers per 6 MFMAs, and every n-tile re-fetches and re-splits the A rows (N / 64 = 5-20 times per element).
// Here (the layout of cnx_mlp.hip, as a plain layer):
//   * a WAVE owns 32 rows.  Per K chunk of KC channels the rows are fetched with line-coalesced loads, turned into the MFMA layout through a wave-private 4.6 KB of
//     LDS (no block barrier on the A side), split ONCE for all the NSUB x 32 output columns of the block and kept in registers as split-f16 fragments;
//   * the weights stream through LDS: one unit = (32 output channels) x (KC) as hi / lo fp16 fragments in MFMA order (a wave-wide ds_read_b128 is one contiguous
//     KB: conflict-free), packed that way at finalize (rr_pack_weights) so that the LDS-DMA (global_load_lds_dwordx4) is a linear copy, into a ring of three units,
//     two ahead; one barrier per unit = per 3 KC / 16 MFMAs of every wave (30 at KC = 160).  LDS traffic: 2 KB per 3 MFMAs = 683 B per MFMA;
//   * transposed products (weights = A operand, rows = B operand): in the 32 x 32 C/D layout a lane then holds 16 outputs of ITS OWN row in groups of four
//     consecutive channels -- the epilogue (weight scale, fused-LayerNorm correction, bias, activation, residual) is per lane, stores are float4;
//   * K is walked chunk by chunk with all NSUB accumulators live, so any K that is a multiple of KC works; a fused input LayerNorm (ConvParams::ln) accumulates its
//     row statistics over the chunks and is applied in the epilogue (y = rstd (acc - mean colsum) + bias, as in igemm_sb_impl.h).
// Block = 4 waves = 128 rows x (NSUB x 32) columns; grid = row blocks x column parts, XCD-aware (the column parts of a row block share an L2).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "igemm_common.h"
#include "sb_split.h"

namespace pf {

typedef _Float16 rr_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 rr_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rr_f16x8, a), __builtin_bit_cast(rr_f16x8, b), c, 0, 0, 0);
}

// compile-time loop: body(std::integral_constant<int, 0>) ... body(<N - 1>) -- the sub-chunk loop below contains wave barriers (convergent), which `#pragma unroll`
// leaves rolled; a rolled loop would index the register arrays dynamically and push them to scratch
template <class F, int... I>
__device__ __forceinline__ void rr_static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>()), ...); }

template <int KC, int NSUB, bool LNF>
__global__ __launch_bounds__(256, 2) void rr_gemm_kernel(const ConvParams p) {
  constexpr int S = KC / 16;         // 16-deep MFMA steps per chunk
  constexpr int Q = KC / 32;         // 32-channel sub-chunks per chunk (one 128-byte line of every row each)
  constexpr int UNIT = S * 2 * 512;  // ushorts of one weight unit: [s][plane][lane][8]
  constexpr int NBUF = 3;
  constexpr int DPT = S / 2;         // 16-byte DMA pieces per thread per unit (S * 2 * 64 pieces / 256 threads)
  constexpr int RS = 36;             // floats per row of the transposition buffer: 144 B, 16-byte slot index 9 r + c -> 16 consecutive rows hit 16 distinct slots
  static_assert(KC % 32 == 0, "chunk = whole 32-channel sub-chunks");
  __shared__ __attribute__((aligned(16))) unsigned short smem[NBUF * UNIT];
  __shared__ __attribute__((aligned(16))) float tsm[4 * 32 * RS];  // per wave: 32 rows x 32 channels, fp32

  const ConvPtrs& P = p.g[0];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int K = p.KH * p.KWCp;     // weight row length; conv_rr_ok: KWCp == KW * C1 (no row padding)
  const int kwc = p.KWCp;          // contiguous input floats per kernel row
  const int nsubs = p.Cout >> 5;
  const int nparts = (nsubs + NSUB - 1) / NSUB;
  const int nmb = (p.M + 127) >> 7;
  const int t = xcd_tile_index(nmb * nparts);
  const int part = t % nparts, mblk = t / nparts;
  const int j0 = part * NSUB;
  const int nsub = min(NSUB, nsubs - j0);
  const int mw = mblk * 128 + wave * 32;  // first row of this wave
  const int m = mw + l31;
  // ---- A side.  LOADS are coalesced: in pass i lane L fetches 16 bytes (piece L & 7) of row 8 i + (L >> 3) -- eight lanes cover one whole 128-byte line of a row's
  // 32-channel sub-chunk (v1 of this kernel let every lane read its own row: 64 lines touched per instruction, 16 bytes used of each, and four waves' 80 KB of
  // half-used lines thrashed the 32 KB L1 -- profiles/r03_tune_rr_v1.txt: slower than the LDS tiles on every shape).  The wave then turns the sub-chunk around through
  // its private 4.6 KB of LDS (no block barrier: one wave's LDS operations execute in order) into the MFMA layout: lane (row l31, half hi) holds channels
  // 16 s' + 8 hi + (0 .. 7) of the sub-chunk's two 16-deep steps.
  const float* xr[4];
  {
    const int hw = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mc = min(mw + 8 * i + (lane >> 3), p.M - 1);  // rows past the end: a valid row, never stored
      const int b = mc / hw, r = mc - b * hw, oy = r / p.Wo, ox = r - oy * p.Wo;
      xr[i] = P.x + ((size_t)(b * p.H + oy * p.stride) * p.W + (size_t)ox * p.stride) * p.C1 + (lane & 7) * 4;
    }
  }
  const size_t xky = (size_t)p.W * p.C1;  // floats between kernel rows of the gathered patch
  float* const tb = tsm + wave * (32 * RS);
  float* const tw = tb + (lane >> 3) * RS + (lane & 7) * 4;  // + 8 i RS
  const float* const tr = tb + l31 * RS + 8 * hi;            // + 16 s'

  const int nkc = K / KC;
  const int U = nkc * nsub;  // weight units this block walks: (chunk, column subtile), subtile fastest

  // ---- weights: unit (subtile j, chunk kc) = KC / 16 steps x (hi, lo) fragments, packed in MFMA fragment order at finalize (rr_pack_weights: [j][kc][s][plane][lane][8]):
  // the LDS-DMA is a linear copy, every 1 KB instruction reads 1 KB of contiguous global memory.  Piece e = tid + 256 i of the unit = fragment wave + 4 i, lane.
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const char* const wbase = reinterpret_cast<const char*>(P.w_rr) + (size_t)tid * 16;
  auto dma = [&](int u, int buf) {
    const int kc = u / nsub, jj = u - kc * nsub;
    const char* src = wbase + ((size_t)(j0 + jj) * nkc + kc) * (UNIT * 2);
#pragma unroll
    for (int i = 0; i < DPT; ++i) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((buf * UNIT + (wave + 4 * i) * 512) * 2));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + 4096 * i), "s"(dst) : "memory");
    }
  };

  u32x4 xh[S], xl[S];
  float piv = 0.f, s1 = 0.f, s2 = 0.f;
  auto load_rows = [&](int kc) {
    const int k0 = kc * KC;
    const int ky = k0 / kwc, off = k0 - ky * kwc;
    const size_t o = (size_t)ky * xky + off;
    u32x4 v[Q * 4];  // (a native vector type: an array of float4 structs that is only copied, never read by field, stays an alloca -- 336 bytes of scratch per thread)
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[q * 4 + i] = *reinterpret_cast<const u32x4*>(xr[i] + o + 32 * q);
    rr_static_for([&](auto qc) {
      constexpr int q = decltype(qc)::value;
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(tw + 8 * i * RS) = v[q * 4 + i];
      __builtin_amdgcn_wave_barrier();  // compiler fence: the reads below are other lanes' writes (same wave: the LDS queue keeps them in order)
      float4 a[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) a[h] = *reinterpret_cast<const float4*>(tr + 16 * (h >> 1) + 4 * (h & 1));
      __builtin_amdgcn_wave_barrier();
      if constexpr (LNF) {
        if (kc == 0 && q == 0) {  // pivot = mean of the row's first 32 channels (LayerNorm is shift invariant; the shifted row carries no large common offset into the sums)
          float sm = 0.f;
#pragma unroll
          for (int h = 0; h < 4; ++h) sm += (a[h].x + a[h].y) + (a[h].z + a[h].w);
          piv = (sm + __shfl_xor(sm, 32)) * (1.0f / 32.0f);
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          a[h] = make_float4(a[h].x - piv, a[h].y - piv, a[h].z - piv, a[h].w - piv);
          s1 += (a[h].x + a[h].y) + (a[h].z + a[h].w);
          s2 = fmaf(a[h].x, a[h].x, fmaf(a[h].y, a[h].y, fmaf(a[h].z, a[h].z, fmaf(a[h].w, a[h].w, s2))));
        }
      }
#pragma unroll
      for (int sp = 0; sp < 2; ++sp) {
        uint2 h0, l0, h1, l1;
        split4_f16(a[2 * sp], h0, l0);
        split4_f16(a[2 * sp + 1], h1, l1);
        xh[2 * q + sp] = u32x4{h0.x, h0.y, h1.x, h1.y};
        xl[2 * q + sp] = u32x4{l0.x, l0.y, l1.x, l1.y};
      }
    }, std::make_integer_sequence<int, Q>());
  };

  f32x16 acc[NSUB];
#pragma unroll
  for (int j = 0; j < NSUB; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  dma(0, 0);
  if (U > 1) dma(1, 1);
  load_rows(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int u = 0;
  for (int kc = 0; kc < nkc; ++kc) {
#pragma unroll
    for (int jj = 0; jj < NSUB; ++jj) {
      if (jj < nsub) {  // block-uniform
        // unit u + 2 into the buffer last read in step u - 1 (every wave is past the barrier that ended it)
        const bool more = u + 2 < U;
        if (more) dma(u + 2, (u + 2) % NBUF);
        const unsigned short* wb = smem + (u % NBUF) * UNIT + lane * 8;
        // two accumulators (even / odd k steps): consecutive MFMAs never wait for each other's result.  The fragments of step pair s + 2 are read BEFORE the six
        // MFMAs of pair s are issued (two register sets): with one or two waves per SIMD nothing else hides the LDS latency (hipcc otherwise reads a pair's
        // fragments right in front of its MFMAs and waits -- 40 % of the unit's time on the launches that give a CU a single block).
        f32x16 accb;
#pragma unroll
        for (int e = 0; e < 16; ++e) accb[e] = 0.f;
        u32x4 fr[2][4];
#pragma unroll
        for (int f = 0; f < 4; ++f) fr[0][f] = *reinterpret_cast<const u32x4*>(wb + f * 512);
#pragma unroll
        for (int s = 0; s < S; s += 2) {
          constexpr int dummy = 0; (void)dummy;
          const int cur = (s >> 1) & 1;
          if (s + 2 < S) {
#pragma unroll
            for (int f = 0; f < 4; ++f) fr[cur ^ 1][f] = *reinterpret_cast<const u32x4*>(wb + (2 * (s + 2) + f) * 512);
          }
          __builtin_amdgcn_sched_barrier(0);  // keep the reads in front of the MFMAs
          acc[jj] = rr_mfma(fr[cur][0], xl[s], acc[jj]);
          accb = rr_mfma(fr[cur][2], xl[s + 1], accb);
          acc[jj] = rr_mfma(fr[cur][1], xh[s], acc[jj]);
          accb = rr_mfma(fr[cur][3], xh[s + 1], accb);
          acc[jj] = rr_mfma(fr[cur][0], xh[s], acc[jj]);
          accb = rr_mfma(fr[cur][2], xh[s + 1], accb);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[jj][e] += accb[e];
        // unit u + 1 (issued one step ago) has landed for this wave; the DMA issued in this step may stay in flight
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DPT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++u;
      }
    }
    if (kc + 1 < nkc) load_rows(kc + 1);  // the other resident block's waves cover this wave's load latency
  }

  // ---- epilogue: this lane's 16 outputs of its row per subtile, float4 per group g: n = 32 (j0 + jj) + 8 g + 4 hi + (0 .. 3)
  if (m >= p.M) return;
  float mu = 0.f, rs = 1.f;
  if constexpr (LNF) {
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    const float invK = 1.0f / (float)K;
    mu = s1 * invK;
    rs = 1.0f / sqrtf(fmaxf(fmaf(-mu, mu, s2 * invK), 0.f) + p.ln_eps);
  }
  float* yrow = P.y + (size_t)m * p.ldy;
  const float* rrow = P.res1 ? P.res1 + (size_t)m * p.ldy : nullptr;
#pragma unroll
  for (int jj = 0; jj < NSUB; ++jj) {
    if (jj < nsub) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * (j0 + jj) + 8 * g + 4 * hi;
        const float4 iv = *reinterpret_cast<const float4*>(P.w_h16_inv_scale + n);
        float4 v = make_float4(acc[jj][4 * g] * iv.x, acc[jj][4 * g + 1] * iv.y, acc[jj][4 * g + 2] * iv.z, acc[jj][4 * g + 3] * iv.w);
        if constexpr (LNF) {
          const float4 cs = *reinterpret_cast<const float4*>(P.ln_colsum + n);
          v.x = rs * fmaf(-mu, cs.x, v.x); v.y = rs * fmaf(-mu, cs.y, v.y); v.z = rs * fmaf(-mu, cs.z, v.z); v.w = rs * fmaf(-mu, cs.w, v.w);
        }
        if (P.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(P.bias + n);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (p.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (p.act == ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        if (rrow) {
          const float4 r = *reinterpret_cast<const float4*>(rrow + n);
          v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (p.post_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(yrow + n) = v;
      }
    }
  }
}

// K chunk held in registers: the largest of 160 / 128 / 96 / 64 that divides a kernel row (KW * C1 contiguous input floats)
int rr_chunk_of(int kwc) { return kwc % 160 == 0 ? 160 : (kwc % 128 == 0 ? 128 : (kwc % 96 == 0 ? 96 : (kwc % 64 == 0 ? 64 : 0))); }
static int rr_chunk(const ConvParams& p) { return rr_chunk_of(p.KW * p.C1); }
// Host side: the split-f16 weight planes [2][Cout][K] (split_f16x2) in the kernel's DMA order [j = Cout / 32][kc = K / KC][s = KC / 16][plane][lane][8]:
// value = plane[32 j + (lane & 31)][kc KC + 16 s + 8 (lane >> 5) + e].  Cout % 32 == 0, K % KC == 0.
std::vector<unsigned short> rr_pack_weights(const std::vector<unsigned short>& planes, int Cout, int K, int KC) {
  const size_t n = (size_t)Cout * K;
  std::vector<unsigned short> o(2 * n);
  const int nkc = K / KC, S = KC / 16;
  size_t w = 0;
  for (int j = 0; j < Cout / 32; ++j)
    for (int kc = 0; kc < nkc; ++kc)
      for (int s = 0; s < S; ++s)
        for (int pl = 0; pl < 2; ++pl)
          for (int lane = 0; lane < 64; ++lane) {
            const size_t src = (size_t)pl * n + (size_t)(32 * j + (lane & 31)) * K + (size_t)kc * KC + 16 * s + 8 * (lane >> 5);
            for (int e = 0; e < 8; ++e) o[w++] = planes[src + e];
          }
  return o;
}
// variant 0: NSUB = 5 (160 columns per block), 1: NSUB = 4 (128)
bool conv_rr_ok(const ConvParams& p, int variant) {
  if (variant < 0 || variant > 1 || p.nterms != NT_F16X3 || p.groups != 1 || p.C2 != 0 || p.ups || p.nchw_out) return false;  // (a split-K request is dropped by launch_conv_sb, as for the halo tiles)
  const ConvPtrs& q = p.g[0];
  if (!q.x || q.x_sb || !q.y || q.y_sb || q.head_kind || q.bias_tab || q.res2 || !q.w_rr || !q.w_h16_inv_scale) return false;
  if (p.pad != 0 || p.KH != p.KW || p.stride != p.KH || (p.H % p.stride) != 0 || (p.W % p.stride) != 0) return false;  // a linear layer on (gathered) rows
  if ((p.Cout & 31) != 0 || (p.C1 & 3) != 0 || p.KWCp != p.KW * p.C1 || p.ldy != p.Cout || rr_chunk(p) == 0) return false;
  if (p.ln && (p.KH != 1 || !q.ln_colsum)) return false;
  return true;
}

template <int KC, int NSUB>
static void launch_rr_cfg(const ConvParams& p, hipStream_t s) {
  const int nparts = ((p.Cout >> 5) + NSUB - 1) / NSUB, nmb = (p.M + 127) >> 7;
  const dim3 grid(nmb * nparts), block(256);
  if (p.ln) hipLaunchKernelGGL((rr_gemm_kernel<KC, NSUB, true>), grid, block, 0, s, p);
  else      hipLaunchKernelGGL((rr_gemm_kernel<KC, NSUB, false>), grid, block, 0, s, p);
}
template <int NSUB>
static void launch_rr_n(const ConvParams& p, hipStream_t s) {
  switch (rr_chunk(p)) {
    case 160: launch_rr_cfg<160, NSUB>(p, s); break;
    case 128: launch_rr_cfg<128, NSUB>(p, s); break;
    case 96: launch_rr_cfg<96, NSUB>(p, s); break;
    default: launch_rr_cfg<64, NSUB>(p, s); break;
  }
}
void launch_conv_rr(const ConvParams& p, int variant, hipStream_t s) {
  if (variant == 0) launch_rr_n<5>(p, s);
  else launch_rr_n<4>(p, s);
}

}  // namespace pf
