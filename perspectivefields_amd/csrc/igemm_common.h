// Shared device helpers of the implicit-GEMM kernels (igemm.hip: exact fp32 MFMA; igemm_sb.hip: split-bf16 MFMA).
#pragma once
#include "pf_kernels.h"
#include "sb_split.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int BK = 32;            // K step in elements
static constexpr unsigned OOB = 0x80000000u;  // beyond any buffer (< 2 GiB): hardware range check returns 0

// erf-GELU, branch-free: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7) on v_rcp_f32 / v_exp_f32; in fp32 as close to
// the exact GELU as the libm erff form (max |error| 4.6e-7 vs 4.5e-7 on [-12, 12]) at less than half the instructions
__device__ __forceinline__ float gelu_erf(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(x * x * -1.4426950408889634f);
  return 0.5f * v * (1.0f + copysignf(fmaf(-p, e, 1.0f), v));
}

__device__ __forceinline__ float4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// XCD-aware tile order: hardware places block b on XCD b % 8; give each XCD a contiguous run of tiles so that
// neighbouring m-tiles (shared halo rows) and the n-tiles of one m-tile (same A rows) meet in one L2.
__device__ __forceinline__ int xcd_tile_index(int nblk) {
  const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Epilogue for accumulators in the 32x32 MFMA C/D layout (col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)),
// identical for the fp32 and the bf16 instructions.  The layout gives each lane ONE float per store (row stride
// between registers), which makes a direct epilogue store-issue bound: stage the tile through LDS (free after the
// K loop) in chunks of WM*32 rows and write it back row-major -- every thread then moves 16 bytes per instruction,
// fully coalesced, and the bias / residual operands are read as float4 as well.
// Rows of the block tile are normally consecutive output pixels (m0 + local row); a kernel that tiles the output
// spatially passes the patch instead: local row -> (oy0 + row / tx, ox0 + row % tx) of image b, rows outside the map are skipped.
struct Tile2D { int b, oy0, ox0, tx, odd_shift; };  // odd_shift: cyclic column shift of the odd patch rows (LDS bank layout of the halo kernel)
template <int BM, int BN, int WM, int WN, int SM, int SN, int NT, int SMEM_FLOATS>
__device__ __forceinline__ void epilogue_nhwc(const ConvParams& p, const ConvPtrs& P, f32x16 (&acc)[SM][SN], float* Cs, int m0, int n0,
                                              const Tile2D* t2 = nullptr, const float* oscale = nullptr /*[Cout] factor on the accumulators (split-f16 weight scale)*/,
                                              const float* ln_stat = nullptr /*LDS [BM][2]: (mean, rstd) of the block's input rows -- fused input LayerNorm (ConvParams::ln)*/) {
  constexpr int CROW = BN + 4;            // floats per staged row (keeps 16-B alignment, shifts banks)
  constexpr int CH_ROWS = WM * 32;        // rows per chunk: subtile row i of every wave row
  constexpr int F4_PER_ROW = BN / 4;
  static_assert(CH_ROWS * CROW <= SMEM_FLOATS, "epilogue chunk must fit the operand buffers");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  const int wave_m = wave / WN;
  const int wn0 = (wave % WN) * (SN * 32);
  const int HoWo = p.Ho * p.Wo;
  const bool vec_ok = (p.Cout & 3) == 0 && (p.ldy & 3) == 0;
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    __syncthreads();  // previous chunk fully written back / K loop finished reading the operand tiles
#pragma unroll
    for (int j = 0; j < SN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        Cs[(wave_m * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi) * CROW + wn0 + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    for (int idx = tid; idx < CH_ROWS * F4_PER_ROW; idx += NT) {
      const int row_l = idx / F4_PER_ROW, cq = idx - row_l * F4_PER_ROW;
      const int lm0 = (row_l >> 5) * (SM * 32) + i * 32 + (row_l & 31);
      int m = m0 + lm0;
      const int n = n0 + cq * 4;
      if (t2) {
        const int lm = m - m0, ry = lm / t2->tx, rx = lm - ry * t2->tx;
        const int oy = t2->oy0 + ry, ox = t2->ox0 + ((ry & 1) ? (rx + t2->odd_shift) % t2->tx : rx);
        if (oy >= p.Ho || ox >= p.Wo) continue;
        m = (t2->b * p.Ho + oy) * p.Wo + ox;
      }
      if (m >= p.M || n >= p.Cout) continue;
      float4 v = *reinterpret_cast<const float4*>(Cs + row_l * CROW + cq * 4);
      const float* bsrc = P.bias;
      if (P.bias_tab) {  // position-dependent bias of a folded (Linear -> zero-padded 3x3) pair: 3x3 border cases
        const int rem = m % HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int cy = oy == 0 ? 0 : (oy == p.Ho - 1 ? 2 : 1), cx = ox == 0 ? 0 : (ox == p.Wo - 1 ? 2 : 1);
        bsrc = P.bias_tab + (cy * 3 + cx) * p.Cout;
      }
      const long o = (long)m * p.ldy + n;
      if (vec_ok) {
        if (oscale) { const float4 sc = *reinterpret_cast<const float4*>(oscale + n); v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
        if (ln_stat) {  // y = rstd (x W' - mean colsum) (+ bias below): LayerNorm of the input rows, gamma / beta folded into W' / bias
          const float mu = ln_stat[2 * lm0], rs = ln_stat[2 * lm0 + 1];
          const float4 cs = *reinterpret_cast<const float4*>(P.ln_colsum + n);
          v.x = rs * fmaf(-mu, cs.x, v.x); v.y = rs * fmaf(-mu, cs.y, v.y); v.z = rs * fmaf(-mu, cs.z, v.z); v.w = rs * fmaf(-mu, cs.w, v.w);
        }
        if (bsrc) { const float4 bb = *reinterpret_cast<const float4*>(bsrc + n); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
        if (p.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        else if (p.act == ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        if (P.res1) { const float4 q = *reinterpret_cast<const float4*>(P.res1 + o); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        if (P.res2) { const float4 q = *reinterpret_cast<const float4*>(P.res2 + o); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
        if (p.post_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if constexpr (BN == 32) {
          if (P.head_kind) {  // block-uniform.  The 8 lanes idx % 8 = 0..7 hold the 32 channels of one pixel (same row -> same branch)
            const int sub = cq;  // channels 4 sub .. 4 sub + 3
            const float4 w0 = reinterpret_cast<const float4*>(P.head_w)[sub];
            float d0 = head_dot4(v, w0), d1 = 0.f;
            if (P.head_kind == 1) d1 = head_dot4(v, reinterpret_cast<const float4*>(P.head_w)[8 + sub]);
#pragma unroll
            for (int sh = 4; sh > 0; sh >>= 1) { d0 += __shfl_xor(d0, sh, 8); d1 += __shfl_xor(d1, sh, 8); }
            if (sub == 0) {
              const int b_img = m / HoWo;
              const int r = m - b_img * HoWo;
              if (P.head_kind == 1) {
                d0 += P.head_b[0]; d1 += P.head_b[1];
                const float nrm = fmaxf(sqrtf(fmaf(d1, d1, __fmul_rn(d0, d0))), 1e-12f);  // F.normalize eps (same expression as pred_regression_kernel)
                d0 /= nrm; d1 /= nrm;
                P.head_out[((long)b_img * 2) * HoWo + r] = d0;
                P.head_out[((long)b_img * 2 + 1) * HoWo + r] = d1;
                if (P.head_pn) *reinterpret_cast<float2*>(P.head_pn + (long)m * 4) = make_float2(d0, d1);
              } else {
                d0 = fminf(fmaxf(d0 + P.head_b[0], -1.f), 1.f);
                P.head_out[m] = d0;
                if (P.head_pn) *reinterpret_cast<float2*>(P.head_pn + (long)m * 4 + 2) = make_float2(d0, 0.f);
              }
            }
            continue;
          }
        }
        if (P.y) *reinterpret_cast<float4*>(P.y + o) = v;
        if (P.y_sb) store_sb4(P.y_sb, p.y_sb_plane, (size_t)o, v);  // split once here instead of per (tap, n-tile) in the consumer
      } else {  // ragged channel count: scalar tail
        const float vv[4] = {v.x, v.y, v.z, v.w};
        for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
          float x = vv[e] * (oscale ? oscale[n + e] : 1.f) + (bsrc ? bsrc[n + e] : 0.f);
          if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
          else if (p.act == ACT_GELU) x = gelu_erf(x);
          if (P.res1) x += P.res1[o + e];
          if (P.res2) x += P.res2[o + e];
          if (p.post_relu) x = fmaxf(x, 0.f);
          P.y[o + e] = x;
        }
      }
    }
  }
}

// Direct epilogue for TRANSPOSED accumulators.  With the MFMA operands swapped (weights as the instruction's A operand, pixels as B)
// the 32x32 C/D layout puts GEMM row m = lane & 31 and, per register group g = r >> 2, the four CONSECUTIVE output channels
// n = 8 g + 4 (lane >> 5) + (r & 3) into one lane: every lane owns float4s of its own output row and can apply scale / LayerNorm
// correction / bias / activation / residuals and store 16 bytes straight from registers -- no LDS staging, no barriers (the LDS form
// above costs 2 barriers, 16 SN ds_write_b32 and SN/.. ds_read_b128 per lane and 32-row chunk).  A store instruction covers 32 rows x
// 32 bytes; the four groups of a subtile complete each row's 128-byte line.  Same operation order as epilogue_nhwc: bit-identical.
template <int SM, int SN>
__device__ __forceinline__ void epilogue_direct(const ConvParams& p, const ConvPtrs& P, f32x16 (&acc)[SM][SN], int mw0 /*first row of this wave's tile*/,
                                                int nw0 /*first column*/, int ml0 /*mw0 - m0: row inside the block tile*/, const float* oscale, const float* ln_stat) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
  const int HoWo = p.Ho * p.Wo;
  const bool vec_ok = (p.Cout & 3) == 0 && (p.ldy & 3) == 0;
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    const int m = mw0 + i * 32 + l31;
    if (m >= p.M) continue;
    float mu = 0.f, rs = 1.f;
    if (ln_stat) { mu = ln_stat[2 * (ml0 + i * 32 + l31)]; rs = ln_stat[2 * (ml0 + i * 32 + l31) + 1]; }
    const float* bsrc = P.bias;
    if (P.bias_tab) {
      const int rem = m % HoWo, oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int cy = oy == 0 ? 0 : (oy == p.Ho - 1 ? 2 : 1), cx = ox == 0 ? 0 : (ox == p.Wo - 1 ? 2 : 1);
      bsrc = P.bias_tab + (cy * 3 + cx) * p.Cout;
    }
#pragma unroll
    for (int j = 0; j < SN; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nw0 + j * 32 + 8 * g + 4 * hi;
        if (n >= p.Cout) continue;
        float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        const long o = (long)m * p.ldy + n;
        if (vec_ok) {
          if (oscale) { const float4 sc = *reinterpret_cast<const float4*>(oscale + n); v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
          if (ln_stat) {
            const float4 cs = *reinterpret_cast<const float4*>(P.ln_colsum + n);
            v.x = rs * fmaf(-mu, cs.x, v.x); v.y = rs * fmaf(-mu, cs.y, v.y); v.z = rs * fmaf(-mu, cs.z, v.z); v.w = rs * fmaf(-mu, cs.w, v.w);
          }
          if (bsrc) { const float4 bb = *reinterpret_cast<const float4*>(bsrc + n); v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w; }
          if (p.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          else if (p.act == ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
          if (P.res1) { const float4 q = *reinterpret_cast<const float4*>(P.res1 + o); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
          if (P.res2) { const float4 q = *reinterpret_cast<const float4*>(P.res2 + o); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
          if (p.post_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (P.y) *reinterpret_cast<float4*>(P.y + o) = v;
          if (P.y_sb) store_sb4(P.y_sb, p.y_sb_plane, (size_t)o, v);
        } else {  // ragged channel count: scalar tail
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 4 && n + e < p.Cout; ++e) {
            float x = vv[e] * (oscale ? oscale[n + e] : 1.f) + (bsrc ? bsrc[n + e] : 0.f);
            if (p.act == ACT_RELU) x = fmaxf(x, 0.f);
            else if (p.act == ACT_GELU) x = gelu_erf(x);
            if (P.res1) x += P.res1[o + e];
            if (P.res2) x += P.res2[o + e];
            if (p.post_relu) x = fmaxf(x, 0.f);
            P.y[o + e] = x;
          }
        }
      }
  }
}

}  // namespace pf
