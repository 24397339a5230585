// Fused ConvNeXt block MLP for gfx950:  y += ls * pwconv2( GELU( pwconv1( LayerNorm(d) ) ) )   (convnext.py:49-56: norm, pwconv1, act,
// pwconv2, gamma; the residual add of :58).  The 4C-wide hidden map never exists outside registers.
//
// Why a kernel of its own: at the large ConvNeXt stages the two 1x1 layers are memory-bound GEMMs (M = 204 800 rows, K or N = 96: the hidden
// map is 314 MB written and 314 MB read back per block and step), each at 45-60 % of its HBM bound.  Here a WAVE owns 32 rows for the whole
// block:
//   * its rows (minus a per-row pivot; LayerNorm is shift invariant) sit in registers as split-f16 B-operand fragments, loaded once;
//     their mean / rstd come from the same registers (lane pair reduction);
//   * the hidden units are produced 32 at a time by GEMM 1 in TRANSPOSED form (weights are the MFMA's A operand, rows its B operand): in
//     the 32x32 C/D layout a lane then holds 16 hidden values of ITS OWN row, exactly what the B operand of GEMM 2 needs -- after the
//     LayerNorm correction, bias, erf-GELU and the fp16 split they feed GEMM 2 straight from registers (the k index of GEMM 2 is a fixed
//     permutation of the hidden index; the host packs W2 in that order).  No LDS round trip, no shuffles, no barrier between the GEMMs;
//   * GEMM 2 accumulates y^T (channels x rows) in registers over all hidden chunks; the epilogue adds bias + residual and stores float4s.
// The weights stream through LDS in fragment order (a wave-wide ds_read_b128 of a fragment is one contiguous KB: conflict-free), one
// 32-hidden-unit chunk (W1 rows + W2 columns) per step, double-buffered: one barrier per chunk, shared by the block's 4 waves (128 rows).
// Contractions: the split-f16 scheme of igemm_sb_impl.h (a ~ ah + al, al = fp16(a - ah) unscaled, w = wh + wl, three fp16 MFMAs per product into one fp32
// accumulator, per-output-channel power-of-two weight scale).  Same mathematics as LayerNorm-fused pwconv1 + pwconv2 (ConvParams::ln).
#include <stdlib.h>

#include <type_traits>

#include "igemm_common.h"
#include "sb_split.h"

namespace pf {

typedef _Float16 cm_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 cm_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cm_f16x8, a), __builtin_bit_cast(cm_f16x8, b), c, 0, 0, 0);
}

// Packed weights (cnx_mlp_pack in engine.hip), per 32-hidden-unit chunk t, all fp16 in MFMA fragment order (lane-major, 8 values per lane):
//   W1 part: [s = C/16 steps][plane hi / lo][lane][8]   value = W1s[32 t + (lane & 31)][ (lane >> 5) C/2 + 8 s + e ]
//   W2 part: [q = C/32][u = 2][plane][lane][8]          value = W2s[32 q + (lane & 31)][ 32 t + 16 u + (e & 3) + 8 (e >> 2) + 4 (lane >> 5) ]
// tab: inv1[H], cs1[H] (column sums of the gamma-folded W1), b1[H] (bias + W1 beta), inv2[C], b2[C]  (H = 4 C)
template <int C, int DIAG = 0 /*tuning builds (ablation): 1 no GELU, 2 no weight DMA inside the loop, 3 no MFMA, 5 no s_setprio*/>
__global__ __launch_bounds__(256, C <= 96 ? 3 : 1) void cnx_mlp_kernel(const float* __restrict__ d, float* __restrict__ y, const unsigned short* __restrict__ wpk,
                                                         const float* __restrict__ tab, int M, float eps, unsigned* sat, float sat_limit) {
  constexpr int H = 4 * C, S1 = C / 16, Q = C / 32, NCH = H / 32;
  constexpr bool PRIO = DIAG != 5;
  constexpr int CH1 = S1 * 2 * 512, CH2 = Q * 2 * 2 * 512;  // ushorts per chunk part (one fragment = 64 lanes x 8 = 512 ushorts)
  constexpr int CHUNK = CH1 + CH2;
  constexpr int F4_PER_THREAD = CHUNK * 2 / 16 / 256;        // 16-byte pieces of a chunk per thread
  static_assert(CHUNK * 2 % (16 * 256) == 0, "chunk must be a whole number of 16-byte pieces per thread");
  constexpr int TAB = 3 * H;  // floats staged in LDS (the per-hidden-unit tables, read every chunk); inv2 / b2 are read from global memory once, in the epilogue:
                              // 2 x 24 KB + 4.5 KB = 53 760 bytes keeps THREE blocks per CU (12 waves, 156 VGPRs each) for C = 96
  __shared__ __attribute__((aligned(16))) unsigned short smem[2 * CHUNK + 2 * TAB];
  float* tabs = reinterpret_cast<float*>(smem + 2 * CHUNK);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int m = blockIdx.x * 128 + wave * 32 + l31;
  const bool mok = m < M;

  {  // per-channel tables -> LDS: all loads in flight at once (a strided copy loop would be TAB / 1024 dependent round trips per block)
    constexpr int TAB4 = TAB / 4, NTL = (TAB4 + 255) / 256;
    static_assert(TAB % 4 == 0, "tables are copied as float4");
    float4 tv[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) tv[j] = reinterpret_cast<const float4*>(tab)[min(tid + 256 * j, TAB4 - 1)];
#pragma unroll
    for (int j = 0; j < NTL; ++j)
      if (tid + 256 * j < TAB4) reinterpret_cast<float4*>(tabs)[tid + 256 * j] = tv[j];
  }

  // ---- weights: chunk t -> LDS buffer by LDS-DMA (global_load_lds_dwordx4, 16 bytes per lane).  The global image is already in fragment order,
  // so the copy is linear: piece tid + 256 j of the chunk goes to the same piece of the buffer = wave-uniform base + lane x 16, exactly the
  // DMA's addressing.  No staging registers, no ds_write pass.  Issued through inline asm: hipcc otherwise waits vmcnt(0) for the DMA in front of
  // the first ds_read that follows it (same __shared__ array), i.e. before the MFMAs it is supposed to hide behind.  hipcc does not count an asm
  // load: the explicit s_waitcnt vmcnt(0) in front of the barrier that ends a chunk is what orders the DMA of the next chunk before its reads.
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  auto dma_w = [&](int t, int buf) {
    const char* src = reinterpret_cast<const char*>(wpk + (size_t)t * CHUNK) + tid * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * CHUNK * 2 + wave * 1024));  // wave-uniform LDS byte address
#pragma unroll
    for (int j = 0; j < F4_PER_THREAD; ++j) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + 4096 * j), "s"(dst + 4096u * j) : "memory");
    }
  };
  dma_w(0, 0);

  // ---- this lane's half of its row: channels hi C/2 ... hi C/2 + C/2 - 1, as S1 fragments of 8 (k slot (s, e) <-> channel hi C/2 + 8 s + e)
  u32x4 xh[S1], xl[S1];
  float mu, rs;
  {
    const float4* src = reinterpret_cast<const float4*>(d + (size_t)min(m, M - 1) * C + hi * (C / 2));  // rows past the end: a valid row, never stored
    float4 v[2 * S1];
#pragma unroll
    for (int j = 0; j < 2 * S1; ++j) v[j] = src[j];
    // pivot = the row's mean (first pass over the registers; an approximate mean is enough, the statistics below are taken of the shifted row): a single
    // channel as pivot would turn an outlier channel into a common offset of the whole shifted row
    float piv = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * S1; ++j) piv += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    piv = (piv + __shfl_xor(piv, 32)) * (1.0f / C);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * S1; ++j) {
      v[j] = make_float4(v[j].x - piv, v[j].y - piv, v[j].z - piv, v[j].w - piv);
      s1 += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      s2 = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, s2))));
    }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    mu = s1 * (1.0f / C);
    rs = 1.0f / sqrtf(fmaxf(fmaf(-mu, mu, s2 * (1.0f / C)), 0.f) + eps);
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      uint2 h0, l0, h1, l1;
      split4_f16(v[2 * s], h0, l0);
      split4_f16(v[2 * s + 1], h1, l1);
      xh[s] = u32x4{h0.x, h0.y, h1.x, h1.y};
      xl[s] = u32x4{l0.x, l0.y, l1.x, l1.y};
      split_f16_mfma_pad(xl[s]);  // register-direct MFMA operand: sb_split.h
    }
  }

  f32x16 acc2[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[q][e] = 0.f;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // tables + chunk 0 in LDS

  // one chunk; BUF is a compile-time constant (the loop below is unrolled by two) so that the compiler can see that the DMA target (buffer
  // 1 - BUF) and the fragment reads (buffer BUF) are disjoint -- with a run-time buffer index it waits for the DMA before the first ds_read
  auto step = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    if (t + 1 < NCH && DIAG != 2) dma_w(t + 1, 1 - BUF);  // lands during this chunk's MFMAs; that buffer was last read in step t - 1 (barrier passed)
    const unsigned short* wb = smem + BUF * CHUNK + lane * 8;
    // GEMM 1 (transposed): acc1[hidden 32 x rows 32] = W1 chunk (A operand) x rows (B operand); smallest partial products first.
    // Two accumulators (even / odd k steps): consecutive MFMAs never wait for each other's result.
    // s_setprio around the MFMA clusters: the co-resident waves are in different phases (another block's wave is in its 440-instruction
    // VALU epilogue); priority lets the wave that has MFMAs to issue get its issue slot, the VALU stream fills the rest (+2 %)
    f32x16 acc1, acc1b;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc1[e] = acc1b[e] = 0.f;
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < S1; s += 2) {
      const u32x4 wh0 = *reinterpret_cast<const u32x4*>(wb + (s * 2) * 512);
      const u32x4 wl0 = *reinterpret_cast<const u32x4*>(wb + (s * 2 + 1) * 512);
      const u32x4 wh1 = *reinterpret_cast<const u32x4*>(wb + (s * 2 + 2) * 512);
      const u32x4 wl1 = *reinterpret_cast<const u32x4*>(wb + (s * 2 + 3) * 512);
      if (DIAG == 3) { acc1[0] += __uint_as_float(wh0.x ^ wl0.y ^ xl[s].x ^ wh1.x ^ wl1.y ^ xl[s + 1].x); continue; }
      acc1 = cm_mfma(wh0, xl[s], acc1);
      acc1b = cm_mfma(wh1, xl[s + 1], acc1b);
      acc1 = cm_mfma(wl0, xh[s], acc1);
      acc1b = cm_mfma(wl1, xh[s + 1], acc1b);
      acc1 = cm_mfma(wh0, xh[s], acc1);
      acc1b = cm_mfma(wh1, xh[s + 1], acc1b);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc1[e] += acc1b[e];
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    // LayerNorm correction + bias + GELU on this lane's 16 hidden values of its own row, then the fp16 split: B operand of GEMM 2
    u32x4 hh[2], hl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float4 hv[2];
#pragma unroll
      for (int gg = 0; gg < 2; ++gg) {
        const int g = 2 * u + gg;
        const int hid = 32 * t + 8 * g + 4 * hi;
        const float4 iv = *reinterpret_cast<const float4*>(tabs + hid);
        const float4 cs = *reinterpret_cast<const float4*>(tabs + H + hid);
        const float4 bb = *reinterpret_cast<const float4*>(tabs + 2 * H + hid);
        float4 a = make_float4(acc1[4 * g] * iv.x, acc1[4 * g + 1] * iv.y, acc1[4 * g + 2] * iv.z, acc1[4 * g + 3] * iv.w);
        a.x = rs * fmaf(-mu, cs.x, a.x) + bb.x; a.y = rs * fmaf(-mu, cs.y, a.y) + bb.y; a.z = rs * fmaf(-mu, cs.z, a.z) + bb.z; a.w = rs * fmaf(-mu, cs.w, a.w) + bb.w;
        hv[gg] = DIAG == 1 ? make_float4(acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]) : make_float4(gelu_erf(a.x), gelu_erf(a.y), gelu_erf(a.z), gelu_erf(a.w));
      }
      uint2 h0, l0, h1, l1;
      split4_f16(hv[0], h0, l0);
      split4_f16(hv[1], h1, l1);
      hh[u] = u32x4{h0.x, h0.y, h1.x, h1.y};
      hl[u] = u32x4{l0.x, l0.y, l1.x, l1.y};
      split_f16_mfma_pad(hl[u]);  // register-direct MFMA operand: sb_split.h
    }
    // GEMM 2 (transposed): acc2[q][channels 32 x rows 32] += W2 chunk (A operand) x hidden (B operand)
    if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {  // the Q accumulators innermost: consecutive MFMAs are independent
      u32x4 wh[Q], wl[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        wh[q] = *reinterpret_cast<const u32x4*>(wb + CH1 + ((q * 2 + u) * 2) * 512);
        wl[q] = *reinterpret_cast<const u32x4*>(wb + CH1 + ((q * 2 + u) * 2 + 1) * 512);
      }
      if (DIAG == 3) {
#pragma unroll
        for (int q = 0; q < Q; ++q) acc2[q][0] += __uint_as_float(wh[q].x ^ wl[q].y ^ hl[u].x ^ hh[u].y);
        continue;
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) acc2[q] = cm_mfma(wh[q], hl[u], acc2[q]);
#pragma unroll
      for (int q = 0; q < Q; ++q) acc2[q] = cm_mfma(wl[q], hh[u], acc2[q]);
#pragma unroll
      for (int q = 0; q < Q; ++q) acc2[q] = cm_mfma(wh[q], hh[u], acc2[q]);
    }
    if (PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of chunk t + 1 has landed ...
    __syncthreads();                                  // ... and so has everybody's; every wave is done with buffer BUF
  };
  static_assert(NCH % 2 == 0, "chunk loop is unrolled by two");
  for (int t = 0; t < NCH; t += 2) {
    step(t, std::integral_constant<int, 0>());
    step(t + 1, std::integral_constant<int, 1>());
  }

  // ---- epilogue: y[m][n] += acc2 * inv2 + b2, float4 per (q, g): n = 32 q + 8 g + 4 hi + (0..3)
  if (!mok) return;
  float* yrow = y + (size_t)m * C;
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = 32 * q + 8 * g + 4 * hi;
      const float4 iv = *reinterpret_cast<const float4*>(tab + 3 * H + n);
      const float4 bb = *reinterpret_cast<const float4*>(tab + 3 * H + C + n);
      const float4 r = *reinterpret_cast<const float4*>(yrow + n);
      float4 v;
      v.x = fmaf(acc2[q][4 * g], iv.x, bb.x) + r.x; v.y = fmaf(acc2[q][4 * g + 1], iv.y, bb.y) + r.y;
      v.z = fmaf(acc2[q][4 * g + 2], iv.z, bb.z) + r.z; v.w = fmaf(acc2[q][4 * g + 3], iv.w, bb.w) + r.w;
      if (sat) sat_watch4(sat, sat_limit, v.x, v.y, v.z, v.w);  // the residual stream feeds the next block's depthwise conv / LayerNorm-fused pwconv1 (ConvParams::sat)
      *reinterpret_cast<float4*>(yrow + n) = v;
    }
}

// C = 192 (stage 2): 96 + 96 fragment / accumulator registers leave one wave per SIMD; r02 measured it slower than the two GEMMs it replaces (old operand split),
// with the r03 split it is ahead end to end (1477-1481 vs 1471-1473 img/s, two runs each on one box, profiles/r03_candidates.md): on by default, PF_CNX_MLP_192=0 restores the GEMM pair
bool cnx_mlp_supported(int C) { return C == 96 || C == 192; }
bool cnx_mlp_preferred(int C) {
  const char* e = getenv("PF_CNX_MLP_192");  // read per call (weight build time only): a process may create engines of both kinds
  const int with192 = e ? atoi(e) : 1;
  return C == 96 || (C == 192 && with192);
}

void launch_cnx_mlp(const float* d, float* y, const unsigned short* wpk, const float* tab, long M, int C, float eps, hipStream_t s, unsigned* sat, float sat_limit) {
  const dim3 grid((unsigned)((M + 127) / 128)), block(256);
#ifdef PF_TUNING_BUILD
  static const int diag = [] { const char* e = getenv("PF_CNX_DIAG"); return e ? atoi(e) : 0; }();
  if (C == 96 && diag == 1) { hipLaunchKernelGGL((cnx_mlp_kernel<96, 1>), grid, block, 0, s, d, y, wpk, tab, (int)M, eps, sat, sat_limit); return; }
  if (C == 96 && diag == 2) { hipLaunchKernelGGL((cnx_mlp_kernel<96, 2>), grid, block, 0, s, d, y, wpk, tab, (int)M, eps, sat, sat_limit); return; }
  if (C == 96 && diag == 3) { hipLaunchKernelGGL((cnx_mlp_kernel<96, 3>), grid, block, 0, s, d, y, wpk, tab, (int)M, eps, sat, sat_limit); return; }
  if (C == 96 && diag == 5) { hipLaunchKernelGGL((cnx_mlp_kernel<96, 5>), grid, block, 0, s, d, y, wpk, tab, (int)M, eps, sat, sat_limit); return; }
#endif
  if (C == 96) hipLaunchKernelGGL((cnx_mlp_kernel<96>), grid, block, 0, s, d, y, wpk, tab, (int)M, eps, sat, sat_limit);
  else if (C == 192) hipLaunchKernelGGL((cnx_mlp_kernel<192>), grid, block, 0, s, d, y, wpk, tab, (int)M, eps, sat, sat_limit);
}

}  // namespace pf
