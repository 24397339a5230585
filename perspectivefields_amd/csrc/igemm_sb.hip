// Split-bf16 implicit GEMM, fp32-accurate form (6 partial products): tile table, eligibility and dispatch by precision.
// Kernel: igemm_sb_impl.h.
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
                             // "sbh": 3x3 / stride 1 convs with an LDS-staged input halo tile, 8 x 16 output patch per block (igemm_sbh.hip)
                             {128, 128, 0, "sbh128x128"}, {128, 64, 0, "sbh128x64"}, {128, 32, 0, "sbh128x32"}, {256, 64, 0, "sbh256x64w8"},
#ifdef PF_TUNING_BUILD
                             // split-f16 scheme only: "d" = weights double-buffered in LDS (one barrier per tap), "t3" = weights of a kernel row per step
                             {128, 128, 0, "sbhd128x128"}, {128, 64, 0, "sbhd128x64"}, {128, 32, 0, "sbhd128x32"}, {256, 64, 0, "sbhd256x64w8"},
                             {256, 64, 0, "sbh256x64w8t3"}, {128, 64, 0, "sbh128x64t3"},
#endif
};
static constexpr int kFirstH = 12;  // index of the first "sbh" tile
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


void launch_conv_sb3(const ConvParams& p, int sb_tile, hipStream_t s);  // igemm_sb3.hip
void launch_conv_sb1(const ConvParams& p, int sb_tile, hipStream_t s);  // igemm_sb1.hip
void launch_conv_sbf(const ConvParams& p, int sb_tile, hipStream_t s);  // igemm_sbf.hip (split-f16)

void launch_conv_sbh(const ConvParams& p, int h_tile, hipStream_t s);

bool conv_sbh_tile_ok(const ConvParams& p, int h_tile);                  // igemm_sbh.hip
bool conv_sb_tile_ok(const ConvParams& p, int sb_tile) {
  if (p.g[0].head_kind && (kSb[sb_tile].bn != 32 || p.Cout != 32)) return false;  // the fused prediction head lives in the BN = 32 epilogue
  if (sb_tile >= kFirstH) return conv_sbh_tile_ok(p, sb_tile - kFirstH);
  if (p.Cin == 4) return kSb[sb_tile].bm <= 128 && kSb[sb_tile].bn <= 128;  // stem form: built for the 4-wave tiles
  return !p.ups;
}

void launch_conv_sb(const ConvParams& p, int sb_tile, hipStream_t s) {
  if (sb_tile >= kFirstH) {
    if (conv_sbh_tile_ok(p, sb_tile - kFirstH)) { launch_conv_sbh(p, sb_tile - kFirstH, s); return; }
    sb_tile = conv_sb_default_tile(p);
  }
  if (p.nterms == NT_F16X3) launch_conv_sbf(p, sb_tile, s);
  else if (p.nterms == 3) launch_conv_sb3(p, sb_tile, s);
  else if (p.nterms == 1) launch_conv_sb1(p, sb_tile, s);
  else launch_conv_sb_nt<6>(p, sb_tile, s);
}

}  // namespace pf
