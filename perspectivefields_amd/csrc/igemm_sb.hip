// Split-bf16 implicit GEMM, fp32-accurate form (6 partial products): tile table, eligibility and dispatch by precision.
// Kernel: igemm_sb_impl.h.
#include <algorithm>

#include "igemm_sb_impl.h"

namespace pf {

// ---------------------------------------------------------------------------------------
struct SbCfg { int bm, bn, pfd; const char* name; };
// "f2"/"f3" = register prefetch ring 2 / 3 K steps deep (small tiles: latency-bound K loops).
// (A software-pipelined variant -- K step 16, two LDS buffers, the next tile's split + LDS stores interleaved with this
// tile's MFMAs in the same wave -- was built and measured 10-30 % SLOWER on every shape, profiles/r01_tune_conv_sbd_negative.txt:
// twice the barriers and fragment-read latencies per K, while the 2-3 co-resident blocks already overlap the phases.)
static const SbCfg kSb[] = {{128, 128, 1, "sb128x128"}, {64, 64, 1, "sb64x64"}, {128, 64, 1, "sb128x64"}, {256, 128, 1, "sb256x128w8"},
                             {128, 256, 1, "sb128x256w8"}, {128, 32, 1, "sb128x32"}, {256, 256, 1, "sb256x256w8"},
                             {64, 64, 2, "sb64x64f2"}, {64, 64, 3, "sb64x64f3"}, {128, 64, 2, "sb128x64f2"}, {128, 32, 2, "sb128x32f2"},
                             {128, 128, 2, "sb128x128f2"},
#ifdef PF_TUNING_BUILD
                             // ablation / scheduling forms of four linear tiles (igemm_sb_impl.h SB_ABL_PARAM): "sbA<mask>_*" wrong results by construction, timing only
                             {64, 64, 1, "sbA16_64x64"}, {64, 64, 1, "sbA32_64x64"}, {64, 64, 1, "sbA48_64x64"}, {64, 64, 1, "sbA1_64x64"},
                             {64, 64, 3, "sbA16_64x64f3"}, {64, 64, 3, "sbA32_64x64f3"}, {64, 64, 3, "sbA48_64x64f3"}, {64, 64, 3, "sbA1_64x64f3"},
                             {128, 128, 1, "sbA16_128x128"}, {128, 128, 1, "sbA32_128x128"}, {128, 128, 1, "sbA48_128x128"}, {128, 128, 1, "sbA1_128x128"},
                             {256, 128, 1, "sbA16_256x128w8"}, {256, 128, 1, "sbA32_256x128w8"}, {256, 128, 1, "sbA48_256x128w8"}, {256, 128, 1, "sbA1_256x128w8"},
#endif
                             // "sbh": 3x3 / stride 1 convs with an LDS-staged input halo tile, 8 x 16 output patch per block (igemm_sbh.hip)
                             {128, 128, 0, "sbh128x128"}, {128, 64, 0, "sbh128x64"}, {128, 32, 0, "sbh128x32"}, {256, 64, 0, "sbh256x64w8"},
#ifdef PF_TUNING_BUILD
                             // split-f16 scheme only: "d" = weights double-buffered in LDS (one barrier per tap), "t3" = weights of a kernel row per step
                             {128, 128, 0, "sbhd128x128"}, {128, 64, 0, "sbhd128x64"}, {128, 32, 0, "sbhd128x32"}, {256, 64, 0, "sbhd256x64w8"},
                             {256, 64, 0, "sbh256x64w8t3"}, {128, 64, 0, "sbh128x64t3"},
#endif
                             {256, 32, 0, "sbh256x32"},  // split-f16 scheme: 16 x 16 patch, N = 32, 4 waves
#ifndef PF_TUNING_BUILD
                             {256, 64, 0, "sbh256x64"},  // split-f16 scheme: 16 x 16 patch, N = 64, 4 waves as 4 x 1: wave tile 64 x 64 (conv_fuse_conv0: fused x2 up-sampling needs a 4-wave tile)
#endif
#ifdef PF_TUNING_BUILD
                             {256, 256, 0, "sbh256x256w8"}, {256, 256, 0, "sbhd256x256w8"},  // split-f16 scheme: 16 x 16 patch x all 256 output channels, 8 waves (rejected)
                             // ablation forms of sbh256x64w8 (igemm_sbh.hip SBH_ABL_PARAM): wrong results by construction, timing only
                             {256, 64, 0, "sbhA1"}, {256, 64, 0, "sbhA2"}, {256, 64, 0, "sbhA3"}, {256, 64, 0, "sbhA4"}, {256, 64, 0, "sbhA8"}, {256, 64, 0, "sbhA48"},
                             {256, 64, 0, "sbhA12"}, {256, 64, 0, "sbhA11"}, {256, 64, 0, "sbhA15"}, {256, 64, 0, "sbhA63"},
                             {256, 64, 0, "sbhAa"}, {256, 64, 0, "sbhAb"}, {256, 64, 0, "sbhLA0"}, {256, 64, 0, "sbhLA2"}, {256, 64, 0, "sbhLAbf"}, {256, 64, 0, "sbhLAbf0"}, {256, 64, 0, "sbhDMA"}, {256, 64, 0, "sbhREG"},
                             {256, 128, 0, "sbh256x128w8"}, {256, 128, 0, "sbh256x128w8u"}, {256, 64, 0, "sbh256x64"},  // 16 x 16 patch x 128 channels, 8 waves: wave tile 64 x 64 (683 B of LDS fragment reads per MFMA instead of 1024); "u" = no 128-VGPR cap
#endif
                             {256, 64, -1, "wino256x64d"},  // Winograd F(2x2, 3x3): wino256x64c with one MFMA + <= 4 VALU instructions per hand-placed slot; the LAST BUT ONE tile
                             {256, 64, -1, "wino256x64c"},  // Winograd F(2x2, 3x3), 4 waves, B-operand fragments computed in registers (no transformed operand in LDS); always the LAST tile
};
#ifdef PF_TUNING_BUILD
static constexpr int kFirstH = 12 + 16;  // index of the first "sbh" tile (behind the linear tiles' tuning forms)
#else
static constexpr int kFirstH = 12;  // index of the first "sbh" tile
#endif
int conv_sb_num_tiles() { return (int)(sizeof(kSb) / sizeof(kSb[0])); }
const char* conv_sb_tile_name(int id) { return kSb[id].name; }
int conv_sb_tile_bm(int id) { return kSb[id].bm; }
int conv_sb_tile_bn(int id) { return kSb[id].bn; }

bool conv_sb_eligible(const ConvParams& p) {
  const bool f16 = p.nterms == NT_F16X3;
  // stems (3 input channels padded to 4): split-f16 scheme, fp32 input, kernel rows of at most 8 taps (one 32-float chunk per row)
  const bool stem = p.Cin == 4 && p.C2 == 0 && f16 && p.KWCp == BK && p.KH * p.KW <= 64 && !p.g[0].x_sb && (p.groups == 1 || !p.g[1].x_sb);
  if (p.nchw_out || ((p.Cin % BK) != 0 && !stem)) return false;
  for (int g = 0; g < p.groups; ++g) {
    const ConvPtrs& q = p.g[g];
    if (f16 ? (!q.w_h16 || !q.w_h16_inv_scale) : !q.w_sb) return false;
    if (q.x_sb) {  // split-plane input: its plane format (sb_split.h) must be the one this scheme multiplies
      if (((p.x_sb_plane & SB_FMT_F16) != 0) != f16) return false;
      if (p.C2 > 0 && (!q.x2_sb || ((p.x2_sb_plane & SB_FMT_F16) != 0) != f16)) return false;
    } else {
      if (!q.x || (p.C2 > 0 && !q.x2)) return false;
    }
    if (!q.y && !q.y_sb && !q.head_kind) return false;
  }
  return true;
}

bool conv_sbh_ok(const ConvParams& p);                                   // igemm_sbh.hip

// Static choice for shapes the tile table (tuned/gfx950_tiles.txt) does not hold; follows what the per-shape tuning picks
// (profiles/r01_tune_conv_*.txt): halo tiles for 3x3 / stride-1 convs on maps of 40^2 and more; otherwise the largest tile
// that still gives every CU about two blocks, with a register prefetch ring (f2 / f3) when the K loop is deep.
int conv_sb_default_tile(const ConvParams& p) {
  const long K = (long)p.KH * p.KWCp;
  if (p.ups) return kFirstH + (p.Cout <= 32 ? 2 : (p.Cout <= 64 ? 1 : 0));  // the only tiles that interpolate while staging
  if (conv_sbh_ok(p) && p.Ho >= 40 && p.Wo >= 40) {
    if (p.Cout <= 32) return kFirstH + 2;
    if (p.Cout <= 64) return kFirstH + 1;
    return p.Ho >= 80 ? kFirstH + 3 : kFirstH + 1;
  }
  if (p.Cout <= 32) return K >= 512 ? 10 : 5;
  auto blocks = [&](int bm, int bn) { return ((long)(p.M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn) * p.groups; };
  if (p.Cout >= 128 && blocks(256, 128) >= 768) return 3;
  if (p.Cout >= 128 && blocks(128, 128) >= 512) return K >= 1024 ? 11 : 0;
  if (blocks(128, 64) >= 512) return K >= 1024 ? 9 : 2;
  return K >= 2048 ? 8 : (K >= 1024 ? 7 : 1);
}


void launch_conv_sbf(const ConvParams& p, int sb_tile, hipStream_t s);  // igemm_sbf.hip (split-f16)

void launch_conv_sbh(const ConvParams& p, int h_tile, hipStream_t s);

bool conv_sbh_tile_ok(const ConvParams& p, int h_tile);                  // igemm_sbh.hip
bool conv_sb_tile_ok(const ConvParams& p, int sb_tile) {
  if (sb_tile >= conv_sb_num_tiles() - 2) return conv_wino_ok(p);  // "wino256x64d", "wino256x64c"
#ifdef PF_TUNING_BUILD
  if (sb_tile >= 12 && sb_tile < kFirstH)  // tuning forms of the linear tiles: split-f16 scheme, one fp32 input, plain epilogue
    return p.nterms == NT_F16X3 && !p.ln && p.C2 == 0 && !p.g[0].x_sb && p.Cin != 4 && (p.Cin % BK) == 0 && !p.ups && !p.g[0].head_kind;
#endif
  if (p.g[0].head_kind && (kSb[sb_tile].bn != 32 || p.Cout != 32)) return false;  // the fused prediction head lives in the BN = 32 epilogue
  if (p.ln) {  // fused input LayerNorm: linear tiles, 1x1, fp32 rows, the whole row inside one block's K loop
    if (sb_tile >= kFirstH || p.KH != 1 || p.KW != 1 || p.stride != 1 || p.pad != 0 || p.C2 > 0 || (p.Cin % BK) != 0 || (p.Cout & 3) || p.splitk > 1 || p.ups) return false;
    for (int g = 0; g < p.groups; ++g)
      if (!p.g[g].ln_colsum || p.g[g].x_sb || !p.g[g].x) return false;
    return true;
  }
  if (sb_tile >= kFirstH) return conv_sbh_tile_ok(p, sb_tile - kFirstH);
  if (p.Cin == 4) return kSb[sb_tile].bm <= 128 && kSb[sb_tile].bn <= 128;  // stem form: built for the 4-wave tiles
  return !p.ups;
}

static int g_splitk_env = -2;  // PF_SPLITK=0 disables split-K, N > 1 forces the factor (tuning aid)
// shape-only part (the engine's workspace dry run uses it before any pointer exists): deep K (>= 32 steps of 32) and at most 300
// tiles of 64x64, i.e. barely one block per CU -- enough slices for ~1024 blocks, each keeping >= 8 K steps
int conv_splitk_shape(long M, int Cout, int KH, int KWCp, int groups) {
  if (g_splitk_env == -2) { const char* e = getenv("PF_SPLITK"); g_splitk_env = e ? atoi(e) : -1; }
  if (g_splitk_env == 0 || (Cout & 3)) return 1;
  const int nK = KH * (KWCp / BK);
  const long blocks64 = ((M + 63) / 64) * ((Cout + 63) / 64) * groups;
  if (nK < 32 || blocks64 > 300) return 1;
  int S = g_splitk_env > 1 ? g_splitk_env : (int)std::min<long>(8, 1024 / blocks64);
  while (S > 1 && nK / S < 8) --S;
  return S;
}
int conv_splitk_factor(const ConvParams& p) {
  if (p.nchw_out || p.ups || p.ln || (p.Cin % BK) != 0 || !conv_sb_eligible(p)) return 1;
  for (int g = 0; g < p.groups; ++g)
    if (p.g[g].head_kind || p.g[g].bias_tab || p.g[g].res2 || p.g[g].y_sb || !p.g[g].y) return 1;
  return conv_splitk_shape(p.M, p.Cout, p.KH, p.KWCp, p.groups);
}

// y = post(act(oscale * sum_s partial[s] + bias) + res1), 4 outputs per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float4* __restrict__ partial, int S, long mn4, int n4, const float4* __restrict__ oscale,
                                                            const float4* __restrict__ bias, int act, const float4* __restrict__ res1, int post_relu, float4* __restrict__ y) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < mn4; i += (long)gridDim.x * 256) {
    float4 v = partial[i];
    for (int k = 1; k < S; ++k) { const float4 q = partial[(long)k * mn4 + i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    const int c = (int)(i % n4);
    if (oscale) { const float4 q = oscale[c]; v.x *= q.x; v.y *= q.y; v.z *= q.z; v.w *= q.w; }
    if (bias) { const float4 q = bias[c]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    if (act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (act == ACT_GELU) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
    if (res1) { const float4 q = res1[i]; v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
    if (post_relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    y[i] = v;
  }
}

void launch_conv_sb(const ConvParams& p0, int sb_tile, hipStream_t s) {
  ConvParams p = p0;
  // split-K only on the linear tiles, with scratch from the caller and 16-byte rows
  if (p.splitk > 1 && (sb_tile >= 12 /* halo tiles and, in tuning builds, the linear tiles' tuning forms */ || p.ln || !p.g[0].partial || (p.groups > 1 && !p.g[1].partial) || (p.Cout & 3) || p.ldy != p.Cout)) p.splitk = 1;
  struct Reduce { const ConvParams& p; hipStream_t s; ~Reduce() {
    if (p.splitk <= 1) return;
    const long mn4 = (long)p.M * p.Cout / 4;
    const unsigned blocks = (unsigned)std::min<long>((mn4 + 255) / 256, 4096);
    for (int g = 0; g < p.groups; ++g) {
      const ConvPtrs& q = p.g[g];
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(q.partial), p.splitk, mn4, p.Cout / 4,
                         reinterpret_cast<const float4*>(p.nterms == NT_F16X3 ? q.w_h16_inv_scale : nullptr), reinterpret_cast<const float4*>(q.bias), p.act,
                         reinterpret_cast<const float4*>(q.res1), p.post_relu, reinterpret_cast<float4*>(q.y));
    }
  } } reduce_after{p, s};
  if (sb_tile >= conv_sb_num_tiles() - 2) {
    if (conv_wino_ok(p)) { launch_conv_wino(p, s, conv_sb_num_tiles() - 1 - sb_tile); return; }
    sb_tile = conv_sb_default_tile(p);
  }
  if (sb_tile >= kFirstH) {
    if (conv_sbh_tile_ok(p, sb_tile - kFirstH)) { launch_conv_sbh(p, sb_tile - kFirstH, s); return; }
    sb_tile = conv_sb_default_tile(p);
  }
  if (p.nterms == NT_F16X3) launch_conv_sbf(p, sb_tile, s);
  else launch_conv_sb_nt<6>(p, sb_tile, s);
}

}  // namespace pf
