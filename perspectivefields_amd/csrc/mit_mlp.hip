// Fused MiT block Mlp for gfx950:  y = x + fc2( GELU( dwconv3x3( fc1( LayerNorm(x) ) ) ) )   (mix_transformers.py:49-56 Mlp.forward, :200 the
// residual, :497-508 DWConv), for the two large stages (C = 64 @80x80, C = 128 @40x40).  The 4C-wide hidden map never goes to HBM.
//
// As three kernels (LayerNorm-fused fc1, depthwise 3x3 + GELU, fc2) the block moves the hidden map four times (210 MB each at stage 1 and batch 32)
// and every one of the three is bound by that traffic.  Here a block of 4 waves owns a TY x TX patch of pixels of one image:
//   * GEMM 1 runs on the patch PLUS its one-pixel halo (HY x HX pixels, RT1 row tiles of 32): the rows (minus a per-row pivot) live in registers as
//     split-f16 fragments for the whole kernel, their LayerNorm statistics come from the same registers;
//   * per chunk of 32 hidden units: GEMM 1 (transposed form: weights = MFMA A operand) -> LayerNorm correction + bias in registers -> the hidden
//     values of the halo pixels to LDS (fp32; pixels outside the image are ZERO: the depthwise conv pads the hidden map, not the input) -> barrier ->
//     every thread computes the 3x3 depthwise sum + bias + erf-GELU for HPT hidden units of one interior pixel from nine LDS rows, splits the result
//     into fp16 hi / lo and writes it as the B operand of GEMM 2 -> barrier -> GEMM 2 accumulates y^T (channels x pixels) in registers;
//   * weights of a chunk (W1 rows, W2 columns, depthwise taps, per-hidden-unit tables) stream through LDS in fragment order by LDS-DMA, double-buffered;
//     the DMA of chunk t + 1 is issued after the first barrier of chunk t and waited for before its second one;
//   * epilogue: y = acc * inv_scale + bias + x for the patch's pixels (float4 per lane: transposed accumulators).
// x and y must be different buffers (neighbouring blocks read each other's rows as halo): the engine ping-pongs the token stream.
// Contractions: the split-f16 scheme of igemm_sb_impl.h (three fp16 MFMAs per product, fp32-class accuracy).
#include <stdlib.h>

#include <type_traits>

#include "igemm_common.h"
#include "sb_split.h"

namespace pf {

typedef _Float16 mm_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mm_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(mm_f16x8, a), __builtin_bit_cast(mm_f16x8, b), c, 0, 0, 0);
}

// Chunk layout (mit_mlp_pack in engine.hip), CHUNK_BYTES per 32 hidden units, fp16 fragments lane-major (64 lanes x 8 values = 1 KB each):
//   [0, S1 * 2 KB)                      W1: [s][plane hi / lo][lane][8] = W1s[32 t + (lane & 31)][(lane >> 5) C/2 + 8 s + e]      (S1 = C / 16)
//   [.., + Q * 2 * 2 KB)                W2: [q][u][plane][lane][8]      = W2s[32 q + (lane & 31)][32 t + 16 u + 8 (lane >> 5) + e]  (Q = C / 32)
//   then 13 x 32 floats                 inv1, cs1, b1 (LayerNorm-folded fc1), depthwise taps [9][32], depthwise bias   (chunk padded to a multiple of 1 KB)
// tab2: inv2[C], b2[C]
template <int C> struct MitMlpCfg {
  static constexpr int S1 = C / 16, Q = C / 32;
  static constexpr int W1_BYTES = S1 * 2 * 1024, W2_BYTES = Q * 2 * 2 * 1024, TAB_FLOATS = 13 * 32;
  static constexpr int USED_BYTES = W1_BYTES + W2_BYTES + TAB_FLOATS * 4;       // multiple of 16
  static constexpr int CHUNK_BYTES = ((USED_BYTES + 1023) / 1024) * 1024;      // stride of a chunk in global memory and of a buffer in LDS
};

// SB (r06, C = 128): ONE weight buffer instead of two -- the W1 part of a chunk is dead after GEMM 1 and its W2 part is not needed before GEMM 2, so at barrier (A) of chunk t
// the block requests W2 of chunk t (into the W2 region GEMM 2 of chunk t - 1 has left) and W1 + tables of chunk t + 1 (into the W1 region GEMM 1 of chunk t has left; the
// tables, which the depthwise phase of chunk t still reads, alternate between two 1.6 KB slots).  Same DMA volume, same request point, same single wait in front of barrier
// (B) as the double-buffered form; LDS 95 -> 63 KB, i.e. TWO blocks per CU at C = 128 (240 VGPRs fit two waves per SIMD).
template <int C, int TY, int TX, bool SB = false>
__global__ __launch_bounds__(256, (C <= 64 || SB) ? 2 : 1) void mit_mlp_kernel(const float* __restrict__ x, float* __restrict__ y, const unsigned short* __restrict__ wpk,
                                                                      const float* __restrict__ tab2, int B, int Hs, int Ws, float eps, unsigned* sat, float sat_limit) {
  typedef MitMlpCfg<C> Cfg;
  constexpr int H = 4 * C, S1 = Cfg::S1, Q = Cfg::Q, NCH = H / 32;
  constexpr int HY = TY + 2, HX = TX + 2, NHALO = HY * HX, RT1 = (NHALO + 31) / 32, NINT = TY * TX, RT2 = NINT / 32;
  constexpr int T1 = (RT1 + 3) / 4;                 // GEMM-1 row tiles per wave (wave w: tiles w, w + 4)
  constexpr int SP = NINT / 32;                      // depthwise phase: a thread owns SP adjacent pixels of a patch row x 4 hidden units (32 strips x 8 groups)
  constexpr int HS = 36;                            // floats per hidden row in LDS (144 B: 16 consecutive rows hit distinct 16-byte bank groups)
  static_assert(NINT % 32 == 0 && RT2 * Q == 8 && TX % SP == 0 && TY * (TX / SP) == 32 && NCH % 2 == 0, "geometry");
  constexpr int CHUNK = Cfg::CHUNK_BYTES;
  constexpr int HBUF_BYTES = RT1 * 32 * HS * 4, H2_BYTES = NINT * 64 * 2;
  constexpr int TABB = Cfg::TAB_FLOATS * 4;
  constexpr int WREG = SB ? Cfg::W1_BYTES + Cfg::W2_BYTES + 2 * TABB : 2 * CHUNK;     // SB: [W1 | W2 | tables of even chunks | tables of odd chunks]
  __shared__ __attribute__((aligned(16))) unsigned char smem[WREG + HBUF_BYTES + H2_BYTES];
  float* Hbuf = reinterpret_cast<float*>(smem + WREG);                          // [RT1 * 32][HS] hidden values of the halo pixels (one chunk)
  unsigned short* H2 = reinterpret_cast<unsigned short*>(smem + WREG + HBUF_BYTES);  // [2 planes][NINT][32] fp16, 64-byte rows, XOR piece swizzle

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int tilesX = (Ws + TX - 1) / TX, tilesY = (Hs + TY - 1) / TY;
  int bid = blockIdx.x;
  const int tx = bid % tilesX; bid /= tilesX;
  const int ty = bid % tilesY; bid /= tilesY;
  const int b = bid;
  const int y0 = ty * TY, x0 = tx * TX;
  const float* xb = x + (size_t)b * Hs * Ws * C;

  // ---- weight chunks by LDS-DMA (inline asm: hipcc would wait for the DMA in front of the next ds_read, cnx_mlp.hip)
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  auto dma_w = [&](int t, int buf) {
    const char* src = reinterpret_cast<const char*>(wpk) + (size_t)t * CHUNK + tid * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(buf * CHUNK + wave * 1024));
#pragma unroll
    for (int j = 0; j < (Cfg::USED_BYTES + 4095) / 4096; ++j) {
      if (4096 * (j + 1) <= Cfg::USED_BYTES || tid * 16 + 4096 * j < Cfg::USED_BYTES) {  // last round: only the lanes that still have a piece
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src + 4096 * j), "s"(dst + 4096u * j) : "memory");
      }
    }
  };
  // SB: one piece of a chunk (nbytes from byte `from` of chunk t) to LDS offset `to`
  auto dma_piece = [&](int t, int from, int nbytes, int to) {
    const char* src = reinterpret_cast<const char*>(wpk) + (size_t)t * CHUNK + from + tid * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(to + wave * 1024));
    for (int j = 0; 4096 * j < nbytes; ++j) {
      if (tid * 16 + 4096 * j < nbytes) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src + 4096 * j), "s"(dst + 4096u * j) : "memory");
      }
    }
  };
  auto dma_w1_tab = [&](int t) {   // W1 + tables of chunk t
    dma_piece(t, 0, Cfg::W1_BYTES, 0);
    dma_piece(t, Cfg::W1_BYTES + Cfg::W2_BYTES, TABB, Cfg::W1_BYTES + Cfg::W2_BYTES + (t & 1) * TABB);
  };
  auto dma_w2 = [&](int t) { dma_piece(t, Cfg::W1_BYTES, Cfg::W2_BYTES, Cfg::W1_BYTES); };
  if constexpr (SB) dma_w1_tab(0); else dma_w(0, 0);

  // ---- this wave's GEMM-1 rows (halo pixels): split-f16 fragments + LayerNorm statistics, held for the whole kernel
  u32x4 xh[T1][S1], xl[T1][S1];
  float mu[T1], rs[T1];
  bool rok[T1];
#pragma unroll
  for (int k = 0; k < T1; ++k) {
    const int tile = wave + 4 * k;
    const int r = tile * 32 + l31;                       // halo pixel index
    const int hy = r / HX, hx = r - hy * HX;
    const int py = y0 - 1 + hy, px = x0 - 1 + hx;
    rok[k] = tile < RT1 && r < NHALO && (unsigned)py < (unsigned)Hs && (unsigned)px < (unsigned)Ws;
    const int cy = min(max(py, 0), Hs - 1), cx = min(max(px, 0), Ws - 1);  // a valid row for the loads; its hidden values are zeroed below
    const float4* src = reinterpret_cast<const float4*>(xb + ((size_t)cy * Ws + cx) * C + hi * (C / 2));
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
    mu[k] = s1 * (1.0f / C);
    rs[k] = 1.0f / sqrtf(fmaxf(fmaf(-mu[k], mu[k], s2 * (1.0f / C)), 0.f) + eps);
#pragma unroll
    for (int s = 0; s < S1; ++s) {
      uint2 h0, l0, h1, l1;
      split4_f16(v[2 * s], h0, l0);
      split4_f16(v[2 * s + 1], h1, l1);
      xh[k][s] = u32x4{h0.x, h0.y, h1.x, h1.y};
      xl[k][s] = u32x4{l0.x, l0.y, l1.x, l1.y};
      split_f16_mfma_pad(xl[k][s]);  // register-direct MFMA operand: sb_split.h
    }
  }

  // GEMM-2 sub-tiles of this wave: s = 2 wave, 2 wave + 1 -> (row tile s / Q, channel tile s % Q); two accumulators
  f32x16 acc2[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
  const int rt2 = (2 * wave) / Q, q0 = (2 * wave) % Q;  // both sub-tiles share the row tile (Q is even)

  // depthwise phase: thread -> (strip of SP pixels of one patch row, hidden units 4 hg .. 4 hg + 3 of the chunk).  The nine taps of the four hidden
  // units are read once per thread (9 float4) and the 3 x (SP + 2) neighbourhood of the strip once (instead of 9 + 9 float4 PER PIXEL with one pixel per
  // thread): 27 instead of 72 LDS reads per thread and chunk at SP = 4
  const int hg = tid & 7, strip = tid >> 3;
  const int sy = strip / (TX / SP), sx0 = (strip % (TX / SP)) * SP;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // chunk 0 in LDS

  auto step = [&](int t, auto bufc) {
    constexpr int BUF = decltype(bufc)::value;
    const unsigned char* cb = smem + (SB ? 0 : BUF * CHUNK);
    const unsigned short* w1f = reinterpret_cast<const unsigned short*>(cb) + lane * 8;
    const unsigned short* w2f = reinterpret_cast<const unsigned short*>(cb + Cfg::W1_BYTES) + lane * 8;
    const float* tb = reinterpret_cast<const float*>(cb + Cfg::W1_BYTES + Cfg::W2_BYTES + (SB ? BUF * TABB : 0));  // inv1[32], cs1[32], b1[32], taps [9][32], dw bias [32]
    // ---- GEMM 1 (transposed): hidden[32] x rows[32] per owned row tile, then LayerNorm correction + bias, zero outside the image, -> Hbuf
#pragma unroll
    for (int k = 0; k < T1; ++k) {
      if (wave + 4 * k < RT1) {  // wave-uniform
        f32x16 a1;
#pragma unroll
        for (int e = 0; e < 16; ++e) a1[e] = 0.f;
#pragma unroll
        for (int s = 0; s < S1; ++s) {
          const u32x4 wh = *reinterpret_cast<const u32x4*>(w1f + (s * 2) * 512);
          const u32x4 wl = *reinterpret_cast<const u32x4*>(w1f + (s * 2 + 1) * 512);
          a1 = mm_mfma(wh, xl[k][s], a1);
          a1 = mm_mfma(wl, xh[k][s], a1);
          a1 = mm_mfma(wh, xh[k][s], a1);
        }
        float* hrow = Hbuf + ((wave + 4 * k) * 32 + l31) * HS + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int hl = 8 * g + 4 * hi;  // hidden unit of the chunk
          const float4 iv = *reinterpret_cast<const float4*>(tb + hl), cs = *reinterpret_cast<const float4*>(tb + 32 + hl), bb = *reinterpret_cast<const float4*>(tb + 64 + hl);
          float4 a;
          a.x = rs[k] * fmaf(-mu[k], cs.x, a1[4 * g] * iv.x) + bb.x; a.y = rs[k] * fmaf(-mu[k], cs.y, a1[4 * g + 1] * iv.y) + bb.y;
          a.z = rs[k] * fmaf(-mu[k], cs.z, a1[4 * g + 2] * iv.z) + bb.z; a.w = rs[k] * fmaf(-mu[k], cs.w, a1[4 * g + 3] * iv.w) + bb.w;
          if (!rok[k]) a = make_float4(0.f, 0.f, 0.f, 0.f);
          *reinterpret_cast<float4*>(hrow + 8 * g) = a;
        }
      }
    }
    __syncthreads();  // (A) hidden values of the whole halo tile in Hbuf; every wave is past GEMM 2 of the previous chunk
    if constexpr (SB) {
      dma_w2(t);                           // the W2 region was last read by GEMM 2 of chunk t - 1; needed behind barrier (B)
      if (t + 1 < NCH) dma_w1_tab(t + 1);  // the W1 region by GEMM 1 of this chunk; the tables go to the other slot
    } else {
      if (t + 1 < NCH) dma_w(t + 1, 1 - BUF);  // that buffer was last read by chunk t - 1
    }
    // ---- depthwise 3x3 + bias + GELU for SP pixels x 4 hidden units; fp16 split -> H2 (B operand of GEMM 2)
    {
      typedef float mm_f2 __attribute__((ext_vector_type(2)));
      float4 wt[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) wt[k] = *reinterpret_cast<const float4*>(tb + 96 + k * 32 + 4 * hg);
      const float4 bv = *reinterpret_cast<const float4*>(tb + 96 + 9 * 32 + 4 * hg);
      float4 hv[3][SP + 2];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int xx = 0; xx < SP + 2; ++xx) hv[ky][xx] = *reinterpret_cast<const float4*>(Hbuf + ((sy + ky) * HX + sx0 + xx) * HS + 4 * hg);
#pragma unroll
      for (int pp = 0; pp < SP; ++pp) {
        mm_f2 a0 = {bv.x, bv.y}, a1 = {bv.z, bv.w};
#pragma unroll
        for (int c = 0; c < 3; ++c)    // tap order of the stand-alone kernel (elem.hip): (ky, kx) = (0, c), (1, c), (2, c)
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const float4 h4 = hv[ky][pp + c], w4 = wt[ky * 3 + c];
            a0 = __builtin_elementwise_fma(mm_f2{h4.x, h4.y}, mm_f2{w4.x, w4.y}, a0);
            a1 = __builtin_elementwise_fma(mm_f2{h4.z, h4.w}, mm_f2{w4.z, w4.w}, a1);
          }
        const float4 gv = make_float4(gelu_erf(a0.x), gelu_erf(a0.y), gelu_erf(a1.x), gelu_erf(a1.y));
        uint2 h0, l0;
        split4_f16(gv, h0, l0);
        const int dp = sy * TX + sx0 + pp;
        unsigned short* d = H2 + dp * 32 + ((hg >> 1) ^ ((dp >> 2) & 3)) * 8 + (hg & 1) * 4;
        *reinterpret_cast<uint2*>(d) = h0;
        *reinterpret_cast<uint2*>(d + NINT * 32) = l0;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's part of chunk t + 1
    __syncthreads();  // (B) H2 complete; weights of chunk t + 1 visible to every wave
    // ---- GEMM 2 (transposed): channels[32] x pixels[32], two channel tiles of one row tile per wave
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int prow = rt2 * 32 + l31;
      const int piece = (2 * u + hi) ^ ((prow >> 2) & 3);
      const u32x4 hh = *reinterpret_cast<const u32x4*>(H2 + prow * 32 + piece * 8);
      const u32x4 hl = *reinterpret_cast<const u32x4*>(H2 + NINT * 32 + prow * 32 + piece * 8);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const u32x4 wh = *reinterpret_cast<const u32x4*>(w2f + (((q0 + j) * 2 + u) * 2) * 512);
        const u32x4 wl = *reinterpret_cast<const u32x4*>(w2f + (((q0 + j) * 2 + u) * 2 + 1) * 512);
        acc2[j] = mm_mfma(wh, hl, acc2[j]);
        acc2[j] = mm_mfma(wl, hh, acc2[j]);
        acc2[j] = mm_mfma(wh, hh, acc2[j]);
      }
    }
  };
  for (int t = 0; t < NCH; t += 2) {
    step(t, std::integral_constant<int, 0>());
    step(t + 1, std::integral_constant<int, 1>());
  }

  // ---- epilogue: y[pixel][n] = acc * inv2 + b2 + x[pixel][n]; lane = pixel l31 of the row tile, float4 per (channel tile, group g)
  const int ip = rt2 * 32 + l31;
  const int ipy = ip / TX, ipx = ip - ipy * TX;
  const int oy = y0 + ipy, ox = x0 + ipx;
  if (oy >= Hs || ox >= Ws) return;
  const size_t o = ((size_t)b * Hs * Ws + (size_t)oy * Ws + ox) * C;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = 32 * (q0 + j) + 8 * g + 4 * hi;
      const float4 iv = *reinterpret_cast<const float4*>(tab2 + n), bb = *reinterpret_cast<const float4*>(tab2 + C + n);
      const float4 r = *reinterpret_cast<const float4*>(x + o + n);
      float4 v;
      v.x = fmaf(acc2[j][4 * g], iv.x, bb.x) + r.x; v.y = fmaf(acc2[j][4 * g + 1], iv.y, bb.y) + r.y;
      v.z = fmaf(acc2[j][4 * g + 2], iv.z, bb.z) + r.z; v.w = fmaf(acc2[j][4 * g + 3], iv.w, bb.w) + r.w;
      if (sat) sat_watch4(sat, sat_limit, v.x, v.y, v.z, v.w);  // the token stream feeds the next block's LayerNorm-fused layers RAW (ConvParams::sat)
      *reinterpret_cast<float4*>(y + o + n) = v;
    }
}

bool mit_mlp_supported(int C) { return C == 64 || C == 128; }
// the engine uses it at stage 1 (C = 64: 162 us vs ~260 us as three kernels); the C = 128 instantiation (one block per CU: 96 KB of LDS, 240 VGPRs) measured
// 175 us vs ~165 us in r02 (profiles/r02_mit_mlp.md) and stayed off until r06, when the same-box A/B at the final kernels read +0.6 % (B = 32), +0.7 % (B = 64), +1.1 % (B = 8), +0.7 % (B = 4),
// -0.4 % at B = 2 and -1.1 % at B = 1 (25 blocks per image): on from a batch of `with128` images up (Engine::mit_mlp128, PF_MIT_MLP_128; profiles/r06_mit_mlp128_ab.log)
bool mit_mlp_preferred(int C, int with128) { return C == 64 || (C == 128 && with128 > 0); }
int mit_mlp_chunk_bytes(int C) { return C == 64 ? MitMlpCfg<64>::CHUNK_BYTES : MitMlpCfg<128>::CHUNK_BYTES; }

void launch_mit_mlp(const float* x, float* y, const unsigned short* wpk, const float* tab2, int B, int Hs, int Ws, int C, float eps, hipStream_t s, unsigned* sat, float sat_limit) {
  if (C == 64) {
    const dim3 grid((unsigned)(B * ((Hs + 7) / 8) * ((Ws + 15) / 16)));
    hipLaunchKernelGGL((mit_mlp_kernel<64, 8, 16>), grid, dim3(256), 0, s, x, y, wpk, tab2, B, Hs, Ws, eps, sat, sat_limit);
  } else if (C == 128) {
    const dim3 grid((unsigned)(B * ((Hs + 7) / 8) * ((Ws + 7) / 8)));
    const char* e = getenv("PF_MIT_MLP_SB");   // read per launch (four per forward): tests switch it inside one process
    const int sb = e ? atoi(e) : 1;            // 0: the double-buffered form (one block per CU)
    if (sb) hipLaunchKernelGGL((mit_mlp_kernel<128, 8, 8, true>), grid, dim3(256), 0, s, x, y, wpk, tab2, B, Hs, Ws, eps, sat, sat_limit);
    else hipLaunchKernelGGL((mit_mlp_kernel<128, 8, 8>), grid, dim3(256), 0, s, x, y, wpk, tab2, B, Hs, Ws, eps, sat, sat_limit);
  }
}

}  // namespace pf
