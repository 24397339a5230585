// The two 7 x 7 convolutions that read the normalised IMAGE (3 channels, stored as NHWC4): the low-level encoder of the decoders -- conv 7x7 / stride 2 / pad 3 -> 64 with
// eval-mode BatchNorm folded + ReLU (perspectivefields.py:70-83) -- and MiT's first overlapping patch embedding -- conv 7x7 / stride 4 / pad 3 -> 64 followed by LayerNorm
// eps 1e-5 (mix_transformers.py:205-246) -- as one specialised kernel.
//
// Why not the implicit-GEMM tiles: K = 7 x 7 x 3 = 147 is seven k-steps of a 64 x 64 tile whose prologue and epilogue dominate (83 TF and 1.1 TB/s on the 819 200-pixel
// low-level map: 0.185 ms for 46 GFLOP and 210 MB; the patch embedding needs a second launch for its LayerNorm).  Here the product is TRANSPOSED as in attn_block.hip: the
// 64 x 224 weight matrix (k = (ky, kx padded to 8, channel padded to 4): 14 chunks of 16) lives in LDS as MFMA A-operand fragments (56 KB, staged once per block for
// many pixel tiles), a wave's 32 output pixels are the B operand -- lane (pixel l & 31, half l >> 5) loads, per chunk, the 32 contiguous bytes of two horizontally adjacent
// taps (all 28 loads of a tile are in flight before the first is used; out-of-image taps are buffer loads past the range: zeros, no branch) -- and in the C/D layout a lane
// then owns 32 of ITS pixel's 64 channels: scale, bias, ReLU or the whole LayerNorm (one shuffle with lane ^ 32) happen in registers and the row is stored once.
#include <stdlib.h>

#include <vector>

#include "host_pack.h"
#include "sb_split.h"

namespace pf {

namespace {

typedef float s7_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 s7_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int s7_u32x4 __attribute__((ext_vector_type(4)));

constexpr int S7_N = 64;                        // output channels
constexpr int S7_CH = 14;                       // chunks of 16 contraction values: ky (7) x [kx 0..3 | kx 4..7] x 4 channels
constexpr int S7_WBYTES = S7_CH * 2 * 2 * 1024; // [chunk][n tile 2][plane 2] fragments of 1 KB
constexpr int S7_TAB = 4 * S7_N;                // inverse scale, bias, LayerNorm gamma, beta

__device__ __forceinline__ void s7_split8(const float (&a)[8], s7_u32x4& h, s7_u32x4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split2_f16(a[2 * e], a[2 * e + 1], hh[e], ll[e]);
  h = s7_u32x4{hh[0], hh[1], hh[2], hh[3]};
  l = s7_u32x4{ll[0], ll[1], ll[2], ll[3]};
  split_f16_mfma_pad(l);  // register-direct MFMA operand: sb_split.h
}
__device__ __forceinline__ s7_f32x16 s7_mfma(const s7_u32x4 a, const s7_u32x4 b, const s7_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(s7_f16x8, a), __builtin_bit_cast(s7_f16x8, b), c, 0, 0, 0);
}

}  // namespace

template <bool LN>
__global__ __launch_bounds__(512, 2) void stem7x7_kernel(const Stem7Args p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s7[];
  const unsigned char* Wf = smem_s7;
  const float* tabs = reinterpret_cast<const float*>(smem_s7 + S7_WBYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  for (int i = tid; i < S7_WBYTES / 16; i += 512) reinterpret_cast<s7_u32x4*>(smem_s7)[i] = reinterpret_cast<const s7_u32x4*>(p.wfr)[i];
  for (int i = tid; i < S7_TAB / 4; i += 512) reinterpret_cast<float4*>(const_cast<float*>(tabs))[i] = reinterpret_cast<const float4*>(p.tab)[i];
  __syncthreads();
  const float* t_inv = tabs, *t_bias = tabs + S7_N, *t_g = tabs + 2 * S7_N, *t_b = tabs + 3 * S7_N;
  const long Mtot = (long)p.B * p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((long)p.B * p.H * p.W * 16), 0x00020000);

  for (int qt = 0; qt < p.QT; ++qt) {
    const long m0 = (((long)blockIdx.x * p.QT + qt) * 8 + wave) * 32;
    if (m0 >= Mtot) break;  // wave-uniform; no barrier below
    const long m = m0 + l31;
    const long mc = m < Mtot ? m : Mtot - 1;  // pixels past the end: a valid pixel, computed and not stored
    const int bimg = (int)(mc / (p.Ho * p.Wo));
    const int rem = (int)(mc - (long)bimg * p.Ho * p.Wo);
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - 3, ix0 = ox * p.stride - 3 + 2 * hi;
    // ---- all 28 requests of the tile: chunk c = (ky = c >> 1, kx = 4 (c & 1) + 2 hi, + 1): two adjacent input pixels = 32 contiguous bytes
    s7_u32x4 v[S7_CH][2];
#pragma unroll
    for (int c = 0; c < S7_CH; ++c) {
      const int iy = iy0 + (c >> 1), ix = ix0 + 4 * (c & 1);
      const bool rowok = (unsigned)iy < (unsigned)p.H;
      const unsigned base = (unsigned)(((bimg * p.H + iy) * p.W + ix) * 16);
      v[c][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, rowok && (unsigned)ix < (unsigned)p.W ? base : 0x80000000u, 0, 0);
      v[c][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, rowok && (unsigned)(ix + 1) < (unsigned)p.W ? base + 16u : 0x80000000u, 0, 0);   // (kx = 7: its weights are zero)
    }
    s7_f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
    for (int c = 0; c < S7_CH; ++c) {
      const float a[8] = {__uint_as_float(v[c][0].x), __uint_as_float(v[c][0].y), __uint_as_float(v[c][0].z), __uint_as_float(v[c][0].w),
                          __uint_as_float(v[c][1].x), __uint_as_float(v[c][1].y), __uint_as_float(v[c][1].z), __uint_as_float(v[c][1].w)};
      s7_u32x4 bh, bl;
      s7_split8(a, bh, bl);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        const s7_u32x4 wh = *reinterpret_cast<const s7_u32x4*>(Wf + (((c * 2 + nt) * 2 + 0) * 1024) + lane * 16);
        const s7_u32x4 wl = *reinterpret_cast<const s7_u32x4*>(Wf + (((c * 2 + nt) * 2 + 1) * 1024) + lane * 16);
        acc[nt] = s7_mfma(wh, bl, acc[nt]);
        acc[nt] = s7_mfma(wl, bh, acc[nt]);
        acc[nt] = s7_mfma(wh, bh, acc[nt]);
      }
    }
    // ---- epilogue in registers: lane owns channels 32 nt + 8 g + 4 hi + e of its pixel
    float4 y[2][4];
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = 32 * nt + 8 * g + 4 * hi;
        const float4 iv = *reinterpret_cast<const float4*>(t_inv + n), bb = *reinterpret_cast<const float4*>(t_bias + n);
        float4 w = make_float4(fmaf(acc[nt][4 * g], iv.x, bb.x), fmaf(acc[nt][4 * g + 1], iv.y, bb.y), fmaf(acc[nt][4 * g + 2], iv.z, bb.z), fmaf(acc[nt][4 * g + 3], iv.w, bb.w));
        if (p.relu) w = make_float4(fmaxf(w.x, 0.f), fmaxf(w.y, 0.f), fmaxf(w.z, 0.f), fmaxf(w.w, 0.f));
        y[nt][g] = w;
        if constexpr (LN) s += (w.x + w.y) + (w.z + w.w);
      }
    if constexpr (LN) {   // two passes over the registers like F.layer_norm; the other 32 channels of the pixel sit in lane ^ 32
      s += __shfl_xor(s, 32, 64);
      const float mu = s * (1.0f / S7_N);
      float ss = 0.f;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          y[nt][g] = make_float4(y[nt][g].x - mu, y[nt][g].y - mu, y[nt][g].z - mu, y[nt][g].w - mu);
          ss = fmaf(y[nt][g].x, y[nt][g].x, fmaf(y[nt][g].y, y[nt][g].y, fmaf(y[nt][g].z, y[nt][g].z, fmaf(y[nt][g].w, y[nt][g].w, ss))));
        }
      ss += __shfl_xor(ss, 32, 64);
      const float rs = 1.0f / sqrtf(ss * (1.0f / S7_N) + p.ln_eps);
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 32 * nt + 8 * g + 4 * hi;
          const float4 gm = *reinterpret_cast<const float4*>(t_g + n), be = *reinterpret_cast<const float4*>(t_b + n);
          y[nt][g] = make_float4(fmaf(y[nt][g].x * rs, gm.x, be.x), fmaf(y[nt][g].y * rs, gm.y, be.y), fmaf(y[nt][g].z * rs, gm.z, be.z), fmaf(y[nt][g].w * rs, gm.w, be.w));
        }
    }
    if (m < Mtot) {
      float* yp = p.y + m * S7_N + 4 * hi;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (p.sat) sat_watch4(p.sat, p.sat_limit, y[nt][g].x, y[nt][g].y, y[nt][g].z, y[nt][g].w);
          *reinterpret_cast<float4*>(yp + 32 * nt + 8 * g) = y[nt][g];
        }
    }
  }
}

bool stem7x7_supported(int Cin, int Cout, int K, int stride, int pad) { return Cin == 3 && Cout == S7_N && K == 7 && pad == 3 && (stride == 2 || stride == 4); }

void launch_stem7x7(const Stem7Args& a, int num_cus, hipStream_t s) {
  Stem7Args p = a;
  const long Mtot = (long)p.B * p.Ho * p.Wo, tiles = (Mtot + 31) / 32;
  int QT = (int)((tiles + 16L * num_cus - 1) / (16L * num_cus));   // two blocks of eight waves per CU (four waves per SIMD cover the gather's latency), one round
  QT = QT < 1 ? 1 : (QT > 32 ? 32 : QT);
  p.QT = QT;
  const size_t lds = S7_WBYTES + S7_TAB * sizeof(float);
  const dim3 grid((unsigned)((tiles + 8L * QT - 1) / (8L * QT)));
  if (p.ln) hipLaunchKernelGGL(stem7x7_kernel<true>, grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL(stem7x7_kernel<false>, grid, dim3(512), lds, s, p);
}

// Host side: w [64][3][7][7] (x out_scale[n] when given: the folded BatchNorm) -> Ws[n][k], k = (ky * 8 + kx) * 4 + c (kx = 7 and c = 3: zeros), split-f16 planes
// (host_pack.h split_f16x2: per-output-channel power-of-two scale), fragments [chunk 14][n tile 2][plane 2][lane 64][8]: lane (l31, hi), element j = Ws[32 nt + l31][16 c + 8 hi + j];
// tab = inverse scales, bias, LayerNorm gamma, beta (zeros when there is no LayerNorm)
void stem7x7_pack(const float* w, const double* out_scale, const float* bias, const float* ln_g, const float* ln_b, std::vector<unsigned short>* wfr, std::vector<float>* tab) {
  constexpr int K = S7_CH * 16;
  std::vector<float> ws((size_t)S7_N * K, 0.f);
  for (int n = 0; n < S7_N; ++n)
    for (int c = 0; c < 3; ++c)
      for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx)
          ws[(size_t)n * K + (ky * 8 + kx) * 4 + c] = (float)((double)w[(((size_t)n * 3 + c) * 7 + ky) * 7 + kx] * (out_scale ? out_scale[n] : 1.0));
  const pf_host::F16Planes pl = pf_host::split_f16x2(ws, S7_N);
  const size_t n_all = ws.size();
  wfr->assign(S7_WBYTES / 2, 0);
  for (int c = 0; c < S7_CH; ++c)
    for (int nt = 0; nt < 2; ++nt)
      for (int plane = 0; plane < 2; ++plane)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j)
            (*wfr)[((((size_t)c * 2 + nt) * 2 + plane) * 64 + lane) * 8 + j] = pl.planes[plane * n_all + (size_t)(32 * nt + (lane & 31)) * K + 16 * c + 8 * (lane >> 5) + j];
  tab->assign(S7_TAB, 0.f);
  for (int n = 0; n < S7_N; ++n) {
    (*tab)[n] = pl.inv_scale[n];
    (*tab)[S7_N + n] = bias ? bias[n] : 0.f;
    (*tab)[2 * S7_N + n] = ln_g ? ln_g[n] : 0.f;
    (*tab)[3 * S7_N + n] = ln_b ? ln_b[n] : 0.f;
  }
}

}  // namespace pf
