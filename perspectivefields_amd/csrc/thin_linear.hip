// y = x W^T + b (+ residual) for 128 -> 128 linear layers over many rows -- the q and output projections of the MiT stage-2 blocks (mix_transformers.py:110, :137-138;
// 51 200 token rows at batch 32) -- in the transposed, register-epilogue form of attn_block.hip / stem7.hip.
//
// On the implicit-GEMM tiles these layers are 27 us launches that move 52 - 79 MB: 1600 blocks of 64 x 64 with four k-steps each, i.e. prologue and epilogue.  Here the
// whole 128 x 128 weight matrix sits in LDS as MFMA A-operand fragments (64 KB per block of 8 waves, two blocks per CU), a wave's 32 rows are the B operand, loaded in the
// ACCUMULATOR layout (lane (row l & 31, half l >> 5) holds channels 32 t + 8 g + 4 hi + e: the weights' contraction index is packed in that permuted order, see
// attn_block.hip), so that the residual -- the same rows of x for the output projection -- is read at exactly the positions the lane's accumulators hold, and bias,
// residual, saturation watch and the 16-byte stores happen on the accumulators.
#include <stdlib.h>

#include <vector>

#include "host_pack.h"
#include "sb_split.h"

namespace pf {

namespace {

typedef float tl_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 tl_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int tl_u32x4 __attribute__((ext_vector_type(4)));

constexpr int TL_C = 128;
constexpr int TL_WBYTES = 8 * 4 * 2 * 1024;   // [chunk 8][n tile 4][plane 2] fragments of 1 KB
constexpr int TL_TAB = 2 * TL_C;              // inverse scales, bias

__device__ __forceinline__ void tl_split8(const float (&a)[8], tl_u32x4& h, tl_u32x4& l) {
  unsigned hh[4], ll[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split2_f16(a[2 * e], a[2 * e + 1], hh[e], ll[e]);
  h = tl_u32x4{hh[0], hh[1], hh[2], hh[3]};
  l = tl_u32x4{ll[0], ll[1], ll[2], ll[3]};
  split_f16_mfma_pad(l);  // register-direct MFMA operand: sb_split.h
}
__device__ __forceinline__ tl_f32x16 tl_mfma(const tl_u32x4 a, const tl_u32x4 b, const tl_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(tl_f16x8, a), __builtin_bit_cast(tl_f16x8, b), c, 0, 0, 0);
}

}  // namespace

template <bool RES>
__global__ __launch_bounds__(512, 2) void thin128_kernel(const ThinLinArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tl[];
  const unsigned char* Wf = smem_tl;
  const float* tabs = reinterpret_cast<const float*>(smem_tl + TL_WBYTES);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  for (int i = tid; i < TL_WBYTES / 16; i += 512) reinterpret_cast<tl_u32x4*>(smem_tl)[i] = reinterpret_cast<const tl_u32x4*>(p.wfr)[i];
  for (int i = tid; i < TL_TAB / 4; i += 512) reinterpret_cast<float4*>(const_cast<float*>(tabs))[i] = reinterpret_cast<const float4*>(p.tab)[i];
  __syncthreads();
  const float* t_inv = tabs, *t_bias = tabs + TL_C;

  for (int qt = 0; qt < p.QT; ++qt) {
    const long m0 = (((long)blockIdx.x * p.QT + qt) * 8 + wave) * 32;
    if (m0 >= p.M) break;  // wave-uniform; no barrier below
    const long m = m0 + l31;
    const long mc = m < p.M ? m : p.M - 1;  // rows past the end: a valid row, computed and not stored
    const size_t off = (size_t)mc * TL_C + 4 * hi;
    float4 xr[4][4];   // channels 32 t + 8 g + 4 hi .. + 3
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) xr[t][g] = *reinterpret_cast<const float4*>(p.x + off + 32 * t + 8 * g);
    float4 rr[RES ? 4 : 1][4];
    if constexpr (RES) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) rr[t][g] = *reinterpret_cast<const float4*>(p.res + off + 32 * t + 8 * g);
    }
    tl_f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nt][e] = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {   // chunk c = 2 t + gp: the registers of g = 2 gp, 2 gp + 1
        const float a[8] = {xr[t][2 * gp].x, xr[t][2 * gp].y, xr[t][2 * gp].z, xr[t][2 * gp].w, xr[t][2 * gp + 1].x, xr[t][2 * gp + 1].y, xr[t][2 * gp + 1].z, xr[t][2 * gp + 1].w};
        tl_u32x4 bh, bl;
        tl_split8(a, bh, bl);
        const int c = 2 * t + gp;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          const tl_u32x4 wh = *reinterpret_cast<const tl_u32x4*>(Wf + (((c * 4 + nt) * 2 + 0) * 1024) + lane * 16);
          const tl_u32x4 wl = *reinterpret_cast<const tl_u32x4*>(Wf + (((c * 4 + nt) * 2 + 1) * 1024) + lane * 16);
          acc[nt] = tl_mfma(wh, bl, acc[nt]);
          acc[nt] = tl_mfma(wl, bh, acc[nt]);
          acc[nt] = tl_mfma(wh, bh, acc[nt]);
        }
      }
    if (m < p.M) {
      float* yp = p.y + off;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = 32 * nt + 8 * g + 4 * hi;
          const float4 iv = *reinterpret_cast<const float4*>(t_inv + n), bb = *reinterpret_cast<const float4*>(t_bias + n);
          float4 w = make_float4(fmaf(acc[nt][4 * g], iv.x, bb.x), fmaf(acc[nt][4 * g + 1], iv.y, bb.y), fmaf(acc[nt][4 * g + 2], iv.z, bb.z), fmaf(acc[nt][4 * g + 3], iv.w, bb.w));
          if constexpr (RES) { w.x += rr[nt][g].x; w.y += rr[nt][g].y; w.z += rr[nt][g].z; w.w += rr[nt][g].w; }
          if (p.sat) sat_watch4(p.sat, p.sat_limit, w.x, w.y, w.z, w.w);
          *reinterpret_cast<float4*>(yp + 32 * nt + 8 * g) = w;
        }
    }
  }
}

bool thin128_supported(int K, int N) { return K == TL_C && N == TL_C; }

void launch_thin128(const ThinLinArgs& a, int num_cus, hipStream_t s) {
  ThinLinArgs p = a;
  const long tiles = (p.M + 31) / 32;
  int QT = (int)((tiles + 16L * num_cus - 1) / (16L * num_cus));   // two blocks of eight waves per CU, one round
  QT = QT < 1 ? 1 : (QT > 32 ? 32 : QT);
  p.QT = QT;
  const size_t lds = TL_WBYTES + TL_TAB * sizeof(float);
  {  // 65 KB of dynamic LDS: the attribute, once per device of the process
    static bool done[64] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !done[dev])
      done[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(thin128_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess &&
                  hipFuncSetAttribute(reinterpret_cast<const void*>(thin128_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) == hipSuccess;
  }
  const dim3 grid((unsigned)((tiles + 8L * QT - 1) / (8L * QT)));
  if (p.res) hipLaunchKernelGGL(thin128_kernel<true>, grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL(thin128_kernel<false>, grid, dim3(512), lds, s, p);
}

// Host side: w [128][128] as split-f16 planes (host_pack.h split_f16x2: per-output-channel power-of-two scale) in fragment order with the PERMUTED contraction index:
// fragment (chunk c, n tile, plane), lane (l31, hi), element j = Ws[32 nt + l31][16 c + 8 (j >> 2) + 4 hi + (j & 3)]; tab = inverse scales, bias
void thin128_pack(const float* w, const float* bias, std::vector<unsigned short>* wfr, std::vector<float>* tab) {
  const pf_host::F16Planes pl = pf_host::split_f16x2(std::vector<float>(w, w + (size_t)TL_C * TL_C), TL_C);
  const size_t n_all = (size_t)TL_C * TL_C;
  wfr->assign(TL_WBYTES / 2, 0);
  for (int c = 0; c < 8; ++c)
    for (int nt = 0; nt < 4; ++nt)
      for (int plane = 0; plane < 2; ++plane)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int n = 32 * nt + (lane & 31), k = 16 * c + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
            (*wfr)[((((size_t)c * 4 + nt) * 2 + plane) * 64 + lane) * 8 + j] = pl.planes[plane * n_all + (size_t)n * TL_C + k];
          }
  tab->resize(TL_TAB);
  for (int n = 0; n < TL_C; ++n) { (*tab)[n] = pl.inv_scale[n]; (*tab)[TL_C + n] = bias ? bias[n] : 0.f; }
}

}  // namespace pf
