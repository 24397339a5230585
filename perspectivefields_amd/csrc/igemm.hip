// Implicit-GEMM convolution / GEMM for gfx950 on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD, bitwise an fmaf chain).
//
// Replaces the ATen ops the reference runs for: dense 3x3 convs of the decoder heads
// (gravity_head.py:70-116, decode_head.py:234-240), every nn.Linear of MiT-B3 / the
// head MLPs / ConvNeXt pwconvs (mix_transformers.py:26-29,80-83; decode_head.py:49;
// convnext.py:34-38), the strided patch-embed / spatial-reduction / stem / downsample
// convs (mix_transformers.py:217-223,88; convnext.py:93,100; perspectivefields.py:73-75).
//
// Layout: A = activations NHWC (K index = (ky, kx, ci), ci fastest -> for one ky the
// (kx, ci) span is CONTIGUOUS in memory, so the gather is plain 16-byte loads with a
// per-row bounds predicate); B = weights packed [Cout][KH][KWCp] (K contiguous).
// Tiling: 256 threads = 4 waves; block tile BM x BN x 32; both operand tiles are staged
// K-contiguous in LDS with a 36-float row stride (conflict-free ds_read_b128, see
// DESIGN.md); each lane reads 4 consecutive k per ds_read_b128, lanes 0-31 taking
// k = 8t..8t+3 and lanes 32-63 k = 8t+4..8t+7 so that one 128-bit read feeds 4 MFMAs
// (the K order is permuted identically for A and B, which a dot product permits).
// Global->LDS is register-staged and double-buffered: loads of tile t+1 are issued
// before the MFMAs of tile t, one barrier per K step.
#include "pf_kernels.h"

namespace pf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr int BK = 32;
static constexpr int LDS_ROW = BK + 4;  // floats; 36*i mod 64 = 4*(9i mod 16): 16 rows hit 16 distinct 16-B slots

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

template <int BM, int BN, int WM, int WN, bool SMALLC, bool NCHW>
__global__ __launch_bounds__(256) void igemm_kernel(const ConvParams p) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int SM = BM / (WM * 32);
  constexpr int SN = BN / (WN * 32);
  constexpr int A_ROWS = BM / 32;
  constexpr int B_ROWS = BN / 32;
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_ROW];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDS_ROW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  // XCD-aware tile order: hardware places block b on XCD b % 8; give each XCD a
  // contiguous run of tiles so that neighbouring m-tiles (shared halo rows) and the
  // n-tiles of one m-tile (same A rows) meet in one L2.
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int nblk = tilesM * tilesN;
  int t;
  {
    const int b = blockIdx.x, q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int m0 = (t / tilesN) * BM;
  const int n0 = (t % tilesN) * BN;

  // ---- staging geometry: thread -> (row r + 32 i, float4 column c4)
  const int c4 = tid & 7;
  const int r0 = tid >> 3;
  long a_base[A_ROWS];
  int a_iy0[A_ROWS], a_ix0[A_ROWS];
  bool a_ok[A_ROWS];
  const int HoWo = p.Ho * p.Wo;
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + r0 + 32 * i;
    a_ok[i] = m < p.M;
    const int mm = a_ok[i] ? m : 0;
    const int b = mm / HoWo;
    const int rem = mm - b * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    a_iy0[i] = oy * p.stride - p.pad;
    a_ix0[i] = ox * p.stride - p.pad;
    a_base[i] = ((long)b * p.H + a_iy0[i]) * p.W + a_ix0[i];
  }
  const long wrow = (long)p.KH * p.KWCp;

  float4 a_reg[A_ROWS], b_reg[B_ROWS];
  const int nJ = p.KWCp / BK;
  const int nK = p.KH * nJ;

  auto load_tiles = [&](int it) {
    const int ky = it / nJ;
    const int j0 = (it - ky * nJ) * BK;
    const int j = j0 + c4 * 4;
    int kx;
    if (SMALLC) kx = j / p.Cin; else kx = j0 / p.Cin;
    const int ci = j - kx * p.Cin;
    const bool jok = j < p.KWC;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;
      const bool ok = a_ok[i] && jok && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const long pix = a_base[i] + (long)ky * p.W + kx;
      const float* src = (ci < p.C1) ? p.x + pix * p.C1 + ci : p.x2 + pix * p.C2 + (ci - p.C1);
      a_reg[i] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
      const int n = n0 + r0 + 32 * i;
      b_reg[i] = (n < p.Cout) ? *reinterpret_cast<const float4*>(p.w + (long)n * wrow + (long)ky * p.KWCp + j)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tiles = [&](int buf) {
    float* Ad = As + buf * BM * LDS_ROW;
    float* Bd = Bs + buf * BN * LDS_ROW;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<float4*>(Ad + (r0 + 32 * i) * LDS_ROW + c4 * 4) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<float4*>(Bd + (r0 + 32 * i) * LDS_ROW + c4 * 4) = b_reg[i];
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

  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int it = 0; it < nK; ++it) {
    const int buf = it & 1;
    if (it + 1 < nK) load_tiles(it + 1);
    const float* Ab = As + buf * BM * LDS_ROW + (wm0 + l31) * LDS_ROW + 4 * hi;
    const float* Bb = Bs + buf * BN * LDS_ROW + (wn0 + l31) * LDS_ROW + 4 * hi;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float4 af[SM], bf[SN];
#pragma unroll
      for (int i = 0; i < SM; ++i) af[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDS_ROW + kk * 8);
#pragma unroll
      for (int j = 0; j < SN; ++j) bf[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDS_ROW + kk * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < SM; ++i) {
          const float av = e == 0 ? af[i].x : e == 1 ? af[i].y : e == 2 ? af[i].z : af[i].w;
#pragma unroll
          for (int j = 0; j < SN; ++j) {
            const float bv = e == 0 ? bf[j].x : e == 1 ? bf[j].y : e == 2 ? bf[j].z : bf[j].w;
            if (NCHW) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, av, acc[i][j], 0, 0, 0);
            else      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    if (it + 1 < nK) store_tiles(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  if (!NCHW) {
#pragma unroll
    for (int j = 0; j < SN; ++j) {
      const int n = n0 + wn0 + j * 32 + l31;
      const bool nok = n < p.Cout;
      const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
#pragma unroll
      for (int i = 0; i < SM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (nok && m < p.M) {
            float v = acc[i][j][r] + bias;
            if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == ACT_GELU) v = gelu_erf(v);
            const long o = (long)m * p.ldy + n;
            if (p.res1) v += p.res1[o];
            if (p.res2) v += p.res2[o];
            if (p.post_relu) v = fmaxf(v, 0.f);
            p.y[o] = v;
          }
        }
      }
    }
  } else {
    // transposed accumulators: col (lane) = pixel, row (register) = output channel
#pragma unroll
    for (int i = 0; i < SM; ++i) {
      const int m = m0 + wm0 + i * 32 + l31;
      const bool mok = m < p.M;
      const int mm = mok ? m : 0;
      const int b = mm / HoWo;
      const int pix = mm - b * HoWo;
#pragma unroll
      for (int j = 0; j < SN; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + wn0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (mok && n < p.Cout) {
            float v = acc[i][j][r] + (p.bias ? p.bias[n] : 0.f);
            if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == ACT_GELU) v = gelu_erf(v);
            p.y[((long)b * p.Cout + n) * HoWo + pix] = v;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
struct TileCfg { int bm, bn; const char* name; float eff; int blocks_per_cu; };
static const TileCfg kTiles[] = {
    {128, 128, "128x128", 1.00f, 2},
    {128, 64, "128x64", 0.95f, 2},
    {64, 64, "64x64", 0.88f, 4},
    {128, 32, "128x32", 0.85f, 3},
    {64, 128, "64x128", 0.93f, 2},
};
int conv_num_tiles() { return (int)(sizeof(kTiles) / sizeof(kTiles[0])); }
const char* conv_tile_name(int id) { return (id >= 0 && id < conv_num_tiles()) ? kTiles[id].name : "auto"; }

template <int BM, int BN, int WM, int WN>
static void launch_cfg(const ConvParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  const dim3 grid(tilesM * tilesN), block(256);
  const bool smallc = (p.Cin % BK) != 0;
  if (p.nchw_out) {
    if (smallc) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, true>), grid, block, 0, s, p);
    else        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, false, true>), grid, block, 0, s, p);
  } else {
    if (smallc) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, true, false>), grid, block, 0, s, p);
    else        hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, false, false>), grid, block, 0, s, p);
  }
}

static int pick_tile(const ConvParams& p) {
  // cost = rounds of resident blocks x padded tile work / per-tile efficiency
  int best = 0;
  double best_cost = 1e300;
  for (int id = 0; id < conv_num_tiles(); ++id) {
    const TileCfg& c = kTiles[id];
    const long tm = (p.M + c.bm - 1) / c.bm, tn = (p.Cout + c.bn - 1) / c.bn;
    const long blocks = tm * tn;
    const long slots = 256L * c.blocks_per_cu;
    const long rounds = (blocks + slots - 1) / slots;
    // partial last round still costs a full tile time; blend with the ideal to avoid cliffs
    const double tile_t = (double)c.bm * c.bn / c.eff;
    const double cost = 0.5 * rounds * tile_t * c.blocks_per_cu + 0.5 * (double)blocks * tile_t / 256.0;
    if (cost < best_cost) { best_cost = cost; best = id; }
  }
  return best;
}

void launch_conv_tile(const ConvParams& p, int tile_id, hipStream_t s) {
  if (tile_id < 0 || tile_id >= conv_num_tiles()) tile_id = pick_tile(p);
  switch (tile_id) {
    case 0: launch_cfg<128, 128, 2, 2>(p, s); break;
    case 1: launch_cfg<128, 64, 2, 2>(p, s); break;
    case 2: launch_cfg<64, 64, 2, 2>(p, s); break;
    case 3: launch_cfg<128, 32, 4, 1>(p, s); break;
    default: launch_cfg<64, 128, 2, 2>(p, s); break;
  }
}
void launch_conv(const ConvParams& p, hipStream_t s) { launch_conv_tile(p, -1, s); }

}  // namespace pf
