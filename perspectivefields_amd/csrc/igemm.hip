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
#include <stdlib.h>

#include "igemm_common.h"

namespace pf {

static constexpr int LDS_ROW = BK + 4;  // floats; 36*i mod 64 = 4*(9i mod 16): 16 rows hit 16 distinct 16-B slots

// MODE: 0 = Cin % 32 == 0, single source tensor; 1 = Cin == 4 (padded 3-channel inputs); 2 = channel-concat of two tensors
template <int BM, int BN, int WM, int WN, int MODE, bool NCHW, bool PIPE>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(const ConvParams p) {
  constexpr bool SMALLC = MODE == 1;
  constexpr int NT = WM * WN * 64;      // threads per block
  constexpr int RPP = NT / 8;           // tile rows staged per pass (8 threads x float4 = one 32-float row)
  constexpr int SM = BM / (WM * 32);
  constexpr int SN = BN / (WN * 32);
  constexpr int A_ROWS = BM / RPP;
  constexpr int B_ROWS = BN / RPP;
  static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
  __shared__ __attribute__((aligned(16))) float smem[2 * (BM + BN) * LDS_ROW];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDS_ROW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int nblk1 = tilesM * tilesN;
  int t = xcd_tile_index(nblk1 * p.groups);
  // grouped launch: group 1 = a second problem of identical shape (the other decoder head)
  const bool g1 = t >= nblk1;
  if (g1) t -= nblk1;
  const ConvPtrs& P = g1 ? p.g[1] : p.g[0];
  const int m0 = (t / tilesN) * BM;
  const int n0 = (t % tilesN) * BN;

  // ---- staging geometry: thread -> (row r0 + RPP i, float4 column c4).  Everything that depends on
  // the row is computed once: byte offset of the tap-(0,0) pixel and a bit mask of the taps that fall
  // inside the image (bit ky*KW+kx; KH*KW <= 64).  The K loop then needs one add, one shift and one
  // select per row; out-of-image taps and padded rows read offset OOB, for which the buffer range
  // check returns zeros (no branches, no per-tap compares).
  const int c4 = tid & 7;
  const int r0 = tid >> 3;
  const int HoWo = p.Ho * p.Wo;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x2 ? P.x2 : P.x), 0, P.x2 ? p.x2_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.w), 0, p.w_bytes, 0x00020000);
  int a_off1[A_ROWS], a_off2[A_ROWS];
  unsigned long long a_mask[A_ROWS];
#pragma unroll
  for (int i = 0; i < A_ROWS; ++i) {
    const int m = m0 + r0 + RPP * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    const int b = mm / HoWo;
    const int rem = mm - b * HoWo;
    const int oy = rem / p.Wo;
    const int ox = rem - oy * p.Wo;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    const int pix = (b * p.H + iy0) * p.W + ix0;
    a_off1[i] = pix * p.C1 * 4 + (SMALLC ? 0 : c4 * 16);
    a_off2[i] = pix * p.C2 * 4 + (SMALLC ? 0 : c4 * 16);
    unsigned long long mk = 0;
    if (ok)
      for (int ky = 0; ky < p.KH; ++ky)
        for (int kx = 0; kx < p.KW; ++kx)
          if ((unsigned)(iy0 + ky) < (unsigned)p.H && (unsigned)(ix0 + kx) < (unsigned)p.W) mk |= 1ull << (ky * p.KW + kx);
    a_mask[i] = mk;
  }
  unsigned b_off[B_ROWS];
#pragma unroll
  for (int i = 0; i < B_ROWS; ++i) {
    const int n = n0 + r0 + RPP * i;
    b_off[i] = n < p.Cout ? (unsigned)(n * p.KH * p.KWCp + c4 * 4) * 4u : OOB;
  }

  float4 a_reg[A_ROWS], b_reg[B_ROWS];
  const int nJ = p.KWCp / BK;
  const int nK = p.KH * nJ;

  // Branch-free (one basic block per K step, so the scheduler may interleave it with the MFMAs): all
  // decisions are selects on wave-uniform values.
  auto load_tiles = [&](int it) {
    const int ky = it / nJ;
    const int j0 = (it - ky * nJ) * BK;
    if (SMALLC) {
      // Cin == 4: a float4 is one pixel; this thread's tap kx differs per thread
      const int j = j0 + c4 * 4;
      const int kx = j >> 2;
      const bool jok = j < p.KWC;
      const int toff = (ky * p.W + kx) * 16;
      const int bit = ky * p.KW + kx;
#pragma unroll
      for (int i = 0; i < A_ROWS; ++i) {
        const bool ok = jok && ((a_mask[i] >> bit) & 1ull);
        a_reg[i] = buf_load16(rx, ok ? (unsigned)(a_off1[i] + toff) : OOB);
      }
    } else {
      // Cin % 32 == 0: the whole 32-float chunk lies in one tap and one source tensor (all uniform)
      const int kx = j0 / p.Cin;
      const int ci0 = j0 - kx * p.Cin;
      const int bit = ky * p.KW + kx;
      if (MODE == 2) {
        const bool first = ci0 < p.C1;
        const int toff = ((ky * p.W + kx) * (first ? p.C1 : p.C2) + (first ? ci0 : ci0 - p.C1)) * 4;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
          const unsigned off = ((a_mask[i] >> bit) & 1ull) ? (unsigned)((first ? a_off1[i] : a_off2[i]) + toff) : OOB;
          const float4 v1 = buf_load16(rx, first ? off : OOB);
          const float4 v2 = buf_load16(rx2, first ? OOB : off);
          a_reg[i] = make_float4(v1.x + v2.x, v1.y + v2.y, v1.z + v2.z, v1.w + v2.w);  // the other one is an OOB zero
        }
      } else {
        const int toff = ((ky * p.W + kx) * p.C1 + ci0) * 4;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) a_reg[i] = buf_load16(rx, ((a_mask[i] >> bit) & 1ull) ? (unsigned)(a_off1[i] + toff) : OOB);
      }
    }
    const unsigned woff = (unsigned)(ky * p.KWCp + j0) * 4u;
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) b_reg[i] = buf_load16(rw, b_off[i] == OOB ? OOB : b_off[i] + woff);
  };
  auto store_tiles = [&](int buf) {
    float* Ad = As + buf * BM * LDS_ROW;
    float* Bd = Bs + buf * BN * LDS_ROW;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<float4*>(Ad + (r0 + RPP * i) * LDS_ROW + c4 * 4) = a_reg[i];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<float4*>(Bd + (r0 + RPP * i) * LDS_ROW + c4 * 4) = b_reg[i];
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

  auto mfma_step = [&](const float* Ab, const float* Bb, int kk) {
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
  };

  for (int it = 0; it < nK; ++it) {
    const int buf = it & 1;
    const float* Ab = As + buf * BM * LDS_ROW + (wm0 + l31) * LDS_ROW + 4 * hi;
    const float* Bb = Bs + buf * BN * LDS_ROW + (wn0 + l31) * LDS_ROW + 4 * hi;
    if (PIPE) {
      // Software-pipelined order, pinned with scheduling fences: the MFMAs of the first quarter start right
      // after the barrier; the address arithmetic + global loads of the next tile are issued behind them (they
      // execute while the matrix pipe is busy); the LDS stores of the next tile go in front of the last quarter
      // so that only the barrier itself separates the last MFMA of this tile from the first of the next.
      constexpr int NM = SM * SN * 4;  // MFMAs per quarter
      const int nxt = it + 1 < nK ? it + 1 : it;  // last step: harmless reload into the buffer nobody reads again
      mfma_step(Ab, Bb, 0);
      __builtin_amdgcn_sched_barrier(0);
      load_tiles(nxt);
      mfma_step(Ab, Bb, 1);
      // interleave: fragment reads first, then one MFMA followed by a slice of the address math / loads
      __builtin_amdgcn_sched_group_barrier(0x100, SM + SN, 0);
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(Ab, Bb, 2);
      __builtin_amdgcn_sched_barrier(0);
      store_tiles(buf ^ 1);
      mfma_step(Ab, Bb, 3);
      __builtin_amdgcn_sched_group_barrier(0x100, SM + SN, 0);
#pragma unroll
      for (int g = 0; g < NM; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    } else {
      if (it + 1 < nK) load_tiles(it + 1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) mfma_step(Ab, Bb, kk);
      if (it + 1 < nK) store_tiles(buf ^ 1);
    }
    __syncthreads();
  }

  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  if (!NCHW) {
    epilogue_nhwc<BM, BN, WM, WN, SM, SN, NT, 2 * (BM + BN) * LDS_ROW>(p, P, acc, smem, m0, n0);
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
            float v = acc[i][j][r] + (P.bias ? P.bias[n] : 0.f);
            if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (p.act == ACT_GELU) v = gelu_erf(v);
            P.y[((long)b * p.Cout + n) * HoWo + pix] = v;
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
struct TileCfg { int bm, bn; const char* name; float eff; int blocks_per_cu; };
// eff = intrinsic throughput (TFLOP/s, all CUs busy, quantisation divided out) measured with scripts/tune_conv.py;
// "p" = software-pipelined issue order.  Ids >= kNumF32 are the split-bf16 tiles of igemm_sb.hip.
static const TileCfg kTiles[] = {
    {128, 128, "128x128", 114.f, 2},    // plain order, kept for A/B
    {64, 64, "64x64", 108.f, 4},        // plain order, kept for A/B
    {128, 128, "128x128p", 120.f, 2},
    {128, 64, "128x64p", 112.f, 2},
    {64, 64, "64x64p", 115.f, 4},
    {128, 32, "128x32p", 101.f, 3},
    {64, 128, "64x128p", 110.f, 2},
    {128, 256, "128x256p", 109.f, 1},
    {256, 128, "256x128w8p", 115.f, 1},
    {256, 256, "256x256w8p", 122.f, 1},
};
static constexpr int kNumF32 = (int)(sizeof(kTiles) / sizeof(kTiles[0]));
int conv_num_tiles() { return kNumF32 + conv_sb_num_tiles(); }
int conv_tile_bm(int id) { return id < kNumF32 ? kTiles[id].bm : conv_sb_tile_bm(id - kNumF32); }
int conv_tile_bn(int id) { return id < kNumF32 ? kTiles[id].bn : conv_sb_tile_bn(id - kNumF32); }
const char* conv_tile_name(int id) {
  if (id < 0 || id >= conv_num_tiles()) return "auto";
  return id < kNumF32 ? kTiles[id].name : conv_sb_tile_name(id - kNumF32);
}
bool conv_tile_is_sb(int id) { return id >= kNumF32 && id < conv_num_tiles(); }
bool conv_tile_usable(const ConvParams& p, int id) {
  if (id < 0 || id >= conv_num_tiles()) return false;
  if (id >= kNumF32) return conv_sb_eligible(p) && conv_sb_tile_ok(p, id - kNumF32);
  if (p.ups || p.g[0].head_kind || p.ln) return false;  // fused up-sampling / prediction head / input LayerNorm: split tiles only
  for (int g = 0; g < p.groups; ++g)  // the exact-fp32 kernel reads fp32 inputs; split-plane output only through the NHWC epilogue
    if (!p.g[g].x || (p.C2 > 0 && !p.g[g].x2) || (p.nchw_out && (!p.g[g].y || p.g[g].y_sb))) return false;
  return true;
}

template <int BM, int BN, int WM, int WN, bool PIPE = false>
static void launch_cfg(const ConvParams& p, hipStream_t s) {
  const int tilesM = (p.M + BM - 1) / BM, tilesN = (p.Cout + BN - 1) / BN;
  const dim3 grid(tilesM * tilesN * p.groups), block(WM * WN * 64);
  const int mode = (p.Cin % BK) != 0 ? 1 : (p.C2 > 0 ? 2 : 0);
  if (p.nchw_out) {
    if (mode == 1)      hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 1, true, PIPE>), grid, block, 0, s, p);
    else if (mode == 2) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 2, true, PIPE>), grid, block, 0, s, p);
    else                hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 0, true, PIPE>), grid, block, 0, s, p);
  } else {
    if (mode == 1)      hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 1, false, PIPE>), grid, block, 0, s, p);
    else if (mode == 2) hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 2, false, PIPE>), grid, block, 0, s, p);
    else                hipLaunchKernelGGL((igemm_kernel<BM, BN, WM, WN, 0, false, PIPE>), grid, block, 0, s, p);
  }
}

static int g_forced_tile = -2;  // -2: not read yet; PF_CONV_TILE=<id> forces one tile config (tuning aid)

// static cost model over the fp32 tiles (used for shapes the autotuner has not seen)
int pick_tile(const ConvParams& p) {
  if (g_forced_tile == -2) {
    const char* e = getenv("PF_CONV_TILE");
    g_forced_tile = e ? atoi(e) : -1;
  }
  if (g_forced_tile >= 0 && conv_tile_usable(p, g_forced_tile)) return g_forced_tile;
  // the split kernels (3 or 6 MFMAs at the 16x bf16 / fp16 rate) beat the exact-fp32 MFMA wherever they can run
  if (conv_sb_eligible(p) || !conv_tile_usable(p, 2)) return kNumF32 + conv_sb_default_tile(p);
  // MFMA-bound model: a CU works through its share of the blocks at the tile's intrinsic rate, so
  // time ~ ceil(blocks / 256 CUs) x (tile MFMA time + fill latency hidden by the resident blocks)
  int best = 2;
  double best_cost = 1e300;
  for (int id = 2; id < kNumF32; ++id) {
    const TileCfg& c = kTiles[id];
    if (c.bn > 32 && p.Cout <= 32) continue;
    const long tm = (p.M + c.bm - 1) / c.bm, tn = (p.Cout + c.bn - 1) / c.bn;
    const long blocks = tm * tn * p.groups;
    const long per_cu = (blocks + 255) / 256;
    const double K = (double)p.KH * p.KWCp;
    const double t_mfma = 2.0 * c.bm * c.bn * K / (c.eff * 1e6 / 256.0);
    const double cost = (double)per_cu * (t_mfma + 4.0 / c.blocks_per_cu);
    if (cost < best_cost) { best_cost = cost; best = id; }
  }
  return best;
}

int conv_default_tile(const ConvParams& p) { return pick_tile(p); }

void launch_conv_tile(const ConvParams& p, int tile_id, hipStream_t s) {
  if (!conv_tile_usable(p, tile_id)) tile_id = pick_tile(p);
  if (tile_id >= kNumF32) { launch_conv_sb(p, tile_id - kNumF32, s); return; }
  switch (tile_id) {
    case 0: launch_cfg<128, 128, 2, 2>(p, s); break;
    case 1: launch_cfg<64, 64, 2, 2>(p, s); break;
    case 2: launch_cfg<128, 128, 2, 2, true>(p, s); break;
    case 3: launch_cfg<128, 64, 2, 2, true>(p, s); break;
    case 4: launch_cfg<64, 64, 2, 2, true>(p, s); break;
    case 5: launch_cfg<128, 32, 4, 1, true>(p, s); break;
    case 6: launch_cfg<64, 128, 2, 2, true>(p, s); break;
    case 7: launch_cfg<128, 256, 2, 2, true>(p, s); break;
    case 8: launch_cfg<256, 128, 4, 2, true>(p, s); break;
    default: launch_cfg<256, 256, 2, 4, true>(p, s); break;
  }
}
void launch_conv(const ConvParams& p, hipStream_t s) { launch_conv_tile(p, -1, s); }

}  // namespace pf
