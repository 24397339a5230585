// Winograd F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 convolutions of the decoders (decode_head.py:224-256 ResidualConvUnit, gravity_head.py:139-176), split-f16
// arithmetic: 16 "position" GEMMs of K = Cin instead of 9 taps x 4 pixels -- 2.25x fewer MFMAs than the halo kernel (igemm_sbh.hip) for the same output.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      d: 4 x 4 input tile (2 x 2 outputs + halo), g: 3 x 3 filter
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]    A^T = [1 1 1 0; 0 1 -1 -1]
//
// U = G g G^T is made once on the host (fp64), scaled per output channel by a power of two and split into two fp16 planes like every split-f16 weight (engine.hip
// split_f16x2), and stored in MFMA FRAGMENT ORDER so that a wave streams it from L2 straight into registers (no LDS, as rb_gemm.hip does).
// V = B^T d B is adds only (fp32), then the 2-way fp16 split without its clamp; |V| <= 4 max|d|, so this kernel's window ends at 65504 / 4 (conv_wino_ok's caller).
//
// Block = 16 x 16 output pixels of one image (8 x 8 tiles = 64 GEMM columns) x 64 output channels x all 16 positions, FOUR waves -- one per SIMD, 512 registers each:
//   wave w owns row xi = w of the transformed tile (positions 4w .. 4w + 3: 256 accumulator registers); M^T = U V^T, i.e. the WEIGHTS are the MFMA's A operand (rows =
//   output channels) -- four consecutive accumulator registers are then four consecutive channels of one tile: 16-byte LDS / global accesses in the epilogue.
//   The B operand (V) never exists in LDS: each wave computes the fragments it multiplies in registers from the raw halo tile (wino4c_f2x2_kernel's comment).
//   Epilogue (shared): A over nu in registers (4 positions -> 2 values), one 128 KB exchange between the waves through LDS, thread = (tile, 4 channels) applies A over
//   xi, the weight scale, bias / activation / residuals (the same order as epilogue_nhwc) and stores 4 pixels x 16 bytes.
// Two kernels: wino256x64c (the transform interleaved with the MFMAs by sched_group_barrier) and wino256x64d (the same data flow, every instruction of the chunk loop
// placed by hand; the engine's default).  The round's earlier forms -- 8 waves with V through LDS, 4 waves with V through LDS -- and two later ones are in
// profiles/r05_rejected/ with their measurements in profiles/r05_winograd.md.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "igemm_common.h"
#include "sb_split.h"

namespace pf {

namespace {

typedef _Float16 wf16x8 __attribute__((ext_vector_type(8)));

constexpr int W_TY = 8, W_TX = 8;             // tiles per block
constexpr int W_PY = 2 * W_TY, W_PX = 2 * W_TX;  // output pixels per block
constexpr int W_HY = W_PY + 2, W_HX = W_PX + 2;  // input halo
constexpr int W_NPIX = W_HY * W_HX;           // 324
constexpr int W_BN = 64;                      // output channels per block
constexpr int W_KC = 16;                      // input channels per chunk
constexpr int M_POS = 64 * 128;               // epilogue: bytes of one position: 64 tiles x 32 fp32
constexpr int W4_NT = 256;                    // four waves
constexpr int RAW4_F4 = (W_NPIX * 4 + W4_NT - 1) / W4_NT;  // 16-byte halo elements per thread and chunk (6; the sixth for 16 threads only)

__device__ __forceinline__ float4 f4add(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(const float4 a, const float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

}  // namespace

// s_memtime stamps of block 17's waves (STAMP instantiation only: PF_WINO_STAMPS=1 in pf_op_conv2d_bench; a stamp drains lgkmcnt, i.e. it perturbs the step it sits in)
#define WINO4_STAMP(i) do { if constexpr (STAMP) { if (blockIdx.x == 17 && lane == 0) p.stamps[wave * 128 + (i)] = __builtin_readcyclecounter(); } } while (0)
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

typedef float wf2 __attribute__((ext_vector_type(2)));
struct WF4 { wf2 lo, hi; };  // four channels as two packed pairs: the transform's adds are v_pk_add_f32 (plain operand forms only: no op_sel)
// 2-way fp16 split WITHOUT the clamp of split2_f16 (sb_split.h): one v_cvt_pk_f16_f32 + two v_fma_mix per pair instead of also two v_med3 -- the transform's VALU
// instructions are paid in MFMA issue time (profiles/r05_winograd.md).  |v| > 65504 then gives inf / NaN instead of saturating: the inputs of a Winograd layer must
// stay below 65504 / 4 (PerspectiveFields.check_range and precision = "auto" hold them to that limit; the engine's saturation counter watches their producers)
__device__ __forceinline__ void split2_f16_nc(const wf2 v, unsigned& h, unsigned& l) {
  const sb_h2 hv = {(_Float16)v.x, (_Float16)v.y};
  h = __builtin_bit_cast(unsigned, hv);
  asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(v.x), "v"(h));
  asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(v.y), "v"(h));
}

// Epilogue of the 4-wave forms: wave w holds row xi = w of the transformed output (acc[nu][cout sub-tile][tile sub-tile]).
// (bimg2, oy2, ox2)[h]: HALF (the half-patch form of wino4d): image and output origin of tile rows 4 h .. 4 h + 3; otherwise entry 0 is the square patch's origin
template <bool STAMP, bool HALF = false>
__device__ __forceinline__ void wino4_epilogue(const ConvParams& p, const ConvPtrs& P, f32x16 (&acc)[4][2][2], unsigned char* wsm, const int (&bimg2)[2], const int (&oy2)[2],
                                               const int (&ox2)[2], int n0) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  // item = (tile, channel quad): 64 x 16 = 1024 items, 4 per thread; item = tid + 256 it -> the SAME channel quad (tid & 15) for all four: scale / bias once.
  // Every global operand (scale, bias, the residuals of all 16 output pixels) is requested BEFORE the LDS reads and the arithmetic: one load latency per block, not one
  // per item (the first form of this loop spent 12 000 cycles here -- profiles/r05_winograd.md)
  const int act = p.act, post_relu = p.post_relu;
  const int e_c4 = tid & 15, n = n0 + e_c4 * 4;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 sc = *reinterpret_cast<const float4*>(P.w_wino_inv + n);
  float4 bb = zero4;
  if (P.bias) bb = *reinterpret_cast<const float4*>(P.bias + n);
  const bool has_r1 = P.res1 != nullptr, has_r2 = P.res2 != nullptr;
  float4 q1[4][2][2], q2[4][2][2];
  unsigned oo[4][2][2];  // element offsets (every activation of the engine stays below 2 GiB)
  bool okp[4][2][2];
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e_tile = (tid >> 4) + 16 * it, e_ty = e_tile >> 3, e_tx = e_tile & 7;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int eh = HALF ? e_ty >> 2 : 0;
        const int oy = oy2[eh] + 2 * (HALF ? e_ty & 3 : e_ty) + a, ox = ox2[eh] + 2 * e_tx + b;
        okp[it][a][b] = oy < p.Ho && ox < p.Wo;
        oo[it][a][b] = HALF && !okp[it][a][b] ? 0u : (unsigned)(((bimg2[eh] * p.Ho + oy) * p.Wo + ox) * p.ldy + n);
        q1[it][a][b] = zero4; q2[it][a][b] = zero4;
      }
  }
  if (has_r1 || has_r2) {  // block-uniform
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
          if (has_r1 && okp[it][a][b]) q1[it][a][b] = *reinterpret_cast<const float4*>(P.res1 + oo[it][a][b]);
          if (has_r2 && okp[it][a][b]) q2[it][a][b] = *reinterpret_cast<const float4*>(P.res2 + oo[it][a][b]);
        }
  }
  // ---- epilogue: A over nu in registers (this wave's row xi: 4 positions -> 2 values per accumulator element), then the four rows meet in LDS
  //      Es[xi][b][tile 64][64 channels] fp32 = 128 KB (row pitch 256 B, 16-byte pieces XOR-swizzled by the tile index)
  __syncthreads();
  unsigned char* const Es = wsm;
#pragma unroll
  for (int ns = 0; ns < 2; ++ns)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const int tile = m * 32 + l31;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float4 s0, s1;
        {
          const f32x16 &a0 = acc[0][ns][m], &a1 = acc[1][ns][m], &a2 = acc[2][ns][m], &a3 = acc[3][ns][m];
          s0 = make_float4((a0[4 * j] + a1[4 * j]) + a2[4 * j], (a0[4 * j + 1] + a1[4 * j + 1]) + a2[4 * j + 1], (a0[4 * j + 2] + a1[4 * j + 2]) + a2[4 * j + 2], (a0[4 * j + 3] + a1[4 * j + 3]) + a2[4 * j + 3]);
          s1 = make_float4((a1[4 * j] - a2[4 * j]) - a3[4 * j], (a1[4 * j + 1] - a2[4 * j + 1]) - a3[4 * j + 1], (a1[4 * j + 2] - a2[4 * j + 2]) - a3[4 * j + 2], (a1[4 * j + 3] - a2[4 * j + 3]) - a3[4 * j + 3]);
        }
        const int piece = (ns * 8 + 2 * j + hi) ^ (tile & 15);  // 16 pieces of 16 bytes per 256-byte row
        unsigned char* dstp = Es + (wave * 2) * (64 * 256) + tile * 256 + piece * 16;
        *reinterpret_cast<float4*>(dstp) = s0;
        *reinterpret_cast<float4*>(dstp + 64 * 256) = s1;
      }
    }
  __syncthreads();
  WINO4_STAMP(2);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e_tile = (tid >> 4) + 16 * it;
    const unsigned char* srcp = Es + e_tile * 256 + ((e_c4 ^ (e_tile & 15)) * 16);
    float4 sb[4][2];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int b = 0; b < 2; ++b) sb[xi][b] = *reinterpret_cast<const float4*>(srcp + (xi * 2 + b) * (64 * 256));
    float4 yv[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      yv[0][b] = f4add(f4add(sb[0][b], sb[1][b]), sb[2][b]);
      yv[1][b] = f4sub(f4sub(sb[1][b], sb[2][b]), sb[3][b]);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float4 w = yv[a][b];
        w.x = fmaf(w.x, sc.x, bb.x); w.y = fmaf(w.y, sc.y, bb.y); w.z = fmaf(w.z, sc.z, bb.z); w.w = fmaf(w.w, sc.w, bb.w);
        if (act == ACT_RELU) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
        else if (act == ACT_GELU) { w.x = gelu_erf(w.x); w.y = gelu_erf(w.y); w.z = gelu_erf(w.z); w.w = gelu_erf(w.w); }
        const float4 r1 = q1[it][a][b], r2 = q2[it][a][b];
        w.x += r1.x; w.y += r1.y; w.z += r1.z; w.w += r1.w;
        w.x += r2.x; w.y += r2.y; w.z += r2.z; w.w += r2.w;
        if (post_relu) { w.x = fmaxf(w.x, 0.f); w.y = fmaxf(w.y, 0.f); w.z = fmaxf(w.z, 0.f); w.w = fmaxf(w.w, 0.f); }
        if (okp[it][a][b]) {
          if (p.sat) sat_watch4(p.sat, p.sat_limit, w.x, w.y, w.z, w.w);
          *reinterpret_cast<float4*>(P.y + oo[it][a][b]) = w;
        }
      }
  }
  WINO4_STAMP(3);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// "wino256x64c": the 4-wave form WITHOUT the transformed operand in LDS.  The ablation of wino256x64w4 (profiles/r05_winograd.md) put its largest single cost on the
// LDS stores of V (23 % of the launch: 16 stores of 16 bytes per thread and chunk, in a stream that has no second wave to cover them; that form is archived in
// profiles/r05_rejected/wino_8wave_w4.hip.txt).  Here wave w (row xi = w of the
// transformed tile) computes the MFMA B-operand fragments it needs ITSELF, in fragment layout: lane = (tile column l & 31, channel half l >> 5) reads the 2 x 4 input
// pixels its row needs (8 channels each: 16 ds_read_b128 per tile sub-tile) from the raw halo tile, forms the row combination (one packed FMA with a wave-uniform sign),
// the four column combinations and the fp16 split -- 8 channels x 4 positions = the four fragments of (tile sub-tile m, nu = 0..3).  No V stores, no fragment reads,
// no second barrier; the VALU work is the same (every V element is computed exactly once, by the wave that multiplies it).
//   raw halo tile: 18 rows x (18 pixels x 64 B + 32 B pad); the four 16-byte pieces of a pixel are stored XOR-swizzled by ((column >> 1) & 3): conflict-free
//   ds_read_b128 for the stride-2 tile origins (brute-force search, tests/test_host_logic.py); three buffers (chunk c, c + 1 being read, c + 2 being filled): ONE barrier per chunk.
//   software pipeline over the 8 fragments (m, nu) of a chunk: the MFMAs of fragment f run two fragments behind the transform (ring of four fragment registers); a
//   fragment = 6 MFMAs = 3 sub-steps of {MFMA, VALU, MFMA, VALU}; the transform pieces, the LDS reads (8 per sub-step, one fragment ahead of their use), the weight
//   requests (one per sub-step) and the raw-halo staging (one element per sub-step) are placed by hand; sched_barrier between the sub-steps.
namespace {
constexpr int RC_ROW = 18 * 64 + 32;          // 1184 bytes per halo row
constexpr int RC_BYTES = W_HY * RC_ROW;       // 21312
constexpr int WC_SMEM = 16 * M_POS;           // the epilogue's exchange (128 KB) is the largest user; the three raw tiles take 63 936 bytes of it
static_assert(3 * RC_BYTES <= WC_SMEM, "raw tiles fit");
struct V8 { wf2 c[4]; };                      // eight channels
}  // namespace

template <bool STAMP>
__global__ __launch_bounds__(W4_NT, 1) void wino4c_f2x2_kernel(const ConvParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char wsm[WC_SMEM];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int tilesN = p.Cout / W_BN;
  const int tilesX = (p.Wo + W_PX - 1) / W_PX, tilesY = (p.Ho + W_PY - 1) / W_PY;
  const int nblk1 = p.B * tilesY * tilesX * tilesN;
  int t = xcd_tile_index(nblk1 * p.groups);
  const bool g1 = t >= nblk1;
  if (g1) t -= nblk1;
  const ConvPtrs& P = g1 ? p.g[1] : p.g[0];
  const int nt = t % tilesN;
  int mt = t / tilesN;
  const int bx = mt % tilesX; mt /= tilesX;
  const int by = mt % tilesY;
  const int bimg = mt / tilesY;
  const int oy0 = by * W_PY, ox0 = bx * W_PX, n0 = nt * W_BN;
  const int nC = p.Cin / W_KC;

  // ---- raw halo staging: element e = tid + 256 i -> (pixel tid / 4 + 64 i, logical piece tid % 4); LDS offset with the piece swizzle
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, p.x_bytes, 0x00020000);
  unsigned g_off[RAW4_F4];
  int s_off[RAW4_F4];
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) {
    const int pix = (tid >> 2) + 64 * i, c4 = tid & 3;
    const int hy = pix / W_HX, hx = pix - hy * W_HX;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool ok = pix < W_NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    g_off[i] = ok ? (unsigned)(((bimg * p.H + iy) * p.W + ix) * p.Cin * 4 + c4 * 16) : OOB;
    s_off[i] = hy * RC_ROW + hx * 64 + ((c4 ^ ((hx >> 1) & 3)) * 16);
  }
  const bool s_last = tid < 4 * (W_NPIX - 64 * (RAW4_F4 - 1));
  auto raw_soff = [&](int c) { return (c < nC ? c : nC - 1) * (W_KC * 4); };
  u32x4 ra[RAW4_F4];
  auto raw_load1 = [&](int i, int c) { ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, g_off[i], raw_soff(c), 0); };
  auto raw_store1 = [&](int i, int c) {
    if (i + 1 < RAW4_F4 || s_last) *reinterpret_cast<u32x4*>(wsm + (c % 3) * RC_BYTES + s_off[i]) = ra[i];
  };

  // ---- transform operands: lane -> tile column l31 of sub-tile m (tile row ty = l31 / 8 + 4 m, tx = l31 % 8), channels 8 hi .. 8 hi + 7 = logical pieces 2 hi, 2 hi + 1
  // row pair of this wave's xi: xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3  ->  t = xA + sgn xB
  const int rA = wave == 0 ? 0 : (wave == 2 ? 2 : 1), rB = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
  const float sgn_f = wave == 1 ? 1.f : -1.f;
  const wf2 sgn = {sgn_f, sgn_f};
  const int ty0 = l31 >> 3, tx0 = l31 & 7;
  const int t_base = (2 * ty0) * RC_ROW + (2 * tx0) * 64;
  // byte offsets inside a raw tile of this lane's two pieces for pixel columns {0, 1} (swizzle tx & 3) and {2, 3} (swizzle (tx + 1) & 3)
  int t_pc[2][2];
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) t_pc[jp][hf] = t_base + jp * 128 + (((2 * hi + hf) ^ ((tx0 + jp) & 3)) * 16);
  V8 xa[4], xb[4];  // the two pixel rows (4 columns x 8 channels each) of one tile sub-tile
  auto t_read = [&](int c, int m, int part) {  // part 0: row A, part 1: row B (8 ds_read_b128 each)
    const unsigned char* base = wsm + (c % 3) * RC_BYTES + m * (8 * RC_ROW) + (part ? rB : rA) * RC_ROW;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const WF4 v = *reinterpret_cast<const WF4*>(base + t_pc[j >> 1][hf] + (j & 1) * 64);
        if (part) { xb[j].c[2 * hf] = v.lo; xb[j].c[2 * hf + 1] = v.hi; }
        else      { xa[j].c[2 * hf] = v.lo; xa[j].c[2 * hf + 1] = v.hi; }
      }
  };
  V8 tr[4];  // row combination, four pixel columns
  auto t_rows = [&](int j) {
#pragma unroll
    for (int e = 0; e < 4; ++e) tr[j].c[e] = __builtin_elementwise_fma(sgn, xb[j].c[e], xa[j].c[e]);
  };
  u32x4 vf[4][2];  // ring of four fragments: [slot][plane hi / lo]
  V8 vo;           // one column combination (between its two pieces)
  auto t_cols = [&](int nu) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      vo.c[e] = nu == 0 ? tr[0].c[e] - tr[2].c[e] : (nu == 1 ? tr[1].c[e] + tr[2].c[e] : (nu == 2 ? tr[2].c[e] - tr[1].c[e] : tr[1].c[e] - tr[3].c[e]));
  };
  auto t_split = [&](int slot) {
    unsigned h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2_f16_nc(vo.c[e], h[e], l[e]);
    vf[slot][0] = u32x4{h[0], h[1], h[2], h[3]};
    vf[slot][1] = u32x4{l[0], l[1], l[2], l[3]};
  };

  // ---- weights: fragments of positions 4 wave + nu through a buffer resource: lane offset in a VGPR, (chunk, position) in the SGPR offset
  const int w_frags = tilesN * nC * 16;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.w_wino), 0, w_frags * 4096, 0x00020000);
  const int w_s0 = (nt * nC * 16 + 4 * wave) * 4096;
  u32x4 bw[4][2][2];  // [nu][cout sub-tile][plane]
  auto load_w1 = [&](int c, int nu, int piece) {
    const int cc = c < nC ? c : nC - 1;
    bw[nu][piece >> 1][piece & 1] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + piece * 1024, w_s0 + (cc * 16 + nu) * 4096, 0);
  };
  f32x16 acc[4][2][2];  // [nu][cout sub-tile][tile sub-tile]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][b][m][e] = 0.f;
  auto mma_one = [&](int f, int i) {  // MFMA i = 0..5 of fragment f = 4 m + nu: product i / 2 (wh vl, wl vh, wh vh), cout sub-tile i % 2
    const int m = f >> 2, nu = f & 3, t3 = i >> 1, ns = i & 1;
    const int tw = t3 == 1 ? 1 : 0, tv = t3 == 0 ? 1 : 0;
    acc[nu][ns][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf16x8, bw[nu][ns][tw]), __builtin_bit_cast(wf16x8, vf[f & 3][tv]), acc[nu][ns][m], 0, 0, 0);
  };

  // ---- prologue: raw(0), raw(1) -> LDS; raw(2) requested; weights of chunk 0; fragments 0 and 1 of chunk 0
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_load1(i, 0);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_store1(i, 0);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_load1(i, 1);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_store1(i, 1);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_load1(i, 2);
#pragma unroll
  for (int nu = 0; nu < 4; ++nu)
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) load_w1(0, nu, pc);
  __syncthreads();
  t_read(0, 0, 0); t_read(0, 0, 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) t_rows(j);
  t_cols(0); t_split(0);
  t_cols(1); t_split(1);
  WINO4_STAMP(0);

#pragma unroll 1
  for (int c = 0; c < nC; ++c) {
    if (c < 16) WINO4_STAMP(8 + 2 * c);
    // entering: fragments 0, 1 of chunk c are in vf[0], vf[1]; tr = row combinations of (chunk c, m = 0); raw(c), raw(c + 1) in LDS; ra = raw(c + 2) requested
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      const int f = k / 3, ks = k % 3;     // MFMAs of fragment f
      const int g = f + 2;                 // the transform works on fragment g (g >= 8: fragments 0, 1 of chunk c + 1)
      const int gc = g < 8 ? c : c + 1, gf = g & 7, gm = gf >> 2, gnu = gf & 3;
      // ---- chores (memory instructions: placed by hand, at most ~8 LDS reads + 1 store + 1-2 requests per sub-step)
      // raw(c + 2) -> LDS, then raw(c + 3) requested: element i in sub-steps 2 i (store) and 2 i + 1 (request), i = 0..5
      if (k < 2 * RAW4_F4) { if ((k & 1) == 0) raw_store1(k >> 1, c + 2); else raw_load1(k >> 1, c + 3); }
      // pixel rows of the NEXT tile sub-tile, one fragment ahead of the row combination: during the nu = 3 fragment (gnu == 3): row A in its second sub-step, row B in its third
      if (gnu == 3 && ks == 1) t_read(gm == 0 ? gc : gc + 1, gm ^ 1, 0);
      if (gnu == 3 && ks == 2) t_read(gm == 0 ? gc : gc + 1, gm ^ 1, 1);
      // weights of chunk c + 1 for position nu: requested after its last MFMAs of this chunk (fragment 4 + nu), one 16-byte piece per sub-step
      if (f >= 5 && f - 5 < 3) { if (ks < 2) { load_w1(c + 1, f - 5, 2 * ks); load_w1(c + 1, f - 5, 2 * ks + 1); } }
      if (f == 0 && ks < 2) { load_w1(c, 3, 2 * ks); load_w1(c, 3, 2 * ks + 1); }  // position 3: last read in the previous chunk's fragment 7
      // ---- two MFMAs and the transform's piece
      mma_one(f, 2 * ks);
      if (gnu == 0) {              // rows (16 packed FMAs) + first column combination + split: 32 VALU instructions over three sub-steps
        if (ks == 0) { t_rows(0); t_rows(1); }
        if (ks == 1) { t_rows(2); t_rows(3); t_cols(0); }
        if (ks == 2) t_split(g & 3);
      } else {
        if (ks == 0) t_cols(gnu);
        if (ks == 1) t_split(g & 3);
      }
      mma_one(f, 2 * ks + 1);
      SGB(0x008, 1); SGB(0x002, 6); SGB(0x008, 1); SGB(0x002, 6);
      __builtin_amdgcn_sched_barrier(0);
      if (c == 2) WINO4_STAMP(80 + k);
    }
    if (c < 16) WINO4_STAMP(9 + 2 * c);
    __syncthreads();   // raw(c + 2) complete in LDS; raw(c) free
  }
  WINO4_STAMP(1);
  const int bimg2[2] = {bimg, bimg}, oy2[2] = {oy0, oy0}, ox2[2] = {ox0, ox0};
  wino4_epilogue<STAMP>(p, P, acc, wsm, bimg2, oy2, ox2, n0);
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// "wino256x64d": the data flow of wino256x64c with EVERY instruction of the chunk loop placed by hand, one MFMA per slot.  The disassembly of wino256x64c showed what
// its stamps could not: between two MFMAs the compiler had put anything from 0 to 33 VALU instructions (sched_group_barrier does not classify the inline-asm
// v_fma_mix as VALU, the exec-masked store of the last halo element split the body into two scheduling regions, ~48 integer address instructions per chunk, and half
// of the row / column combinations scalarised) -- a VALU cluster longer than the ~32 cycles one MFMA occupies the matrix core delays the next MFMA by the excess, and
// an empty slot hides nothing.  Here:
//   * a chunk = 48 slots {one MFMA, <= 4 VALU instructions (<= 23 cycles), <= 2 memory instructions}, sched_barrier(0) after every slot;
//   * window w (6 slots) multiplies fragment w = (tile sub-tile m, nu) and transforms fragment w + 2: slot 0 the column combination (4 packed), slots 1-4 the fp16
//     split (v_cvt_pk / v_fma_mixlo / v_fma_mixhi of four channel pairs, skewed so that no instruction depends on its predecessor), slot 5 ONE pixel column of the row
//     combination (4 packed FMAs) for the tile sub-tile that needs it next -- its 2 x 2 ds_read_b128 are issued in slots 0 and 1 of the same window;
//   * 22 address instructions per chunk instead of 48: every LDS address is a register + an immediate; two sets of read addresses and the six store addresses walk
//     the ring of raw tiles by one v_add each, in slots where they are idle (a body unrolled by three with immediates only made hipcc hoist 66 "register + constant"
//     addresses out of the loop: 628 spilled dwords; a 48-iteration #pragma unroll exceeds the unroller's budget -- the slots are macro-expanded);
//   * the last (partial) halo element is stored by every thread, the idle ones into a dump area: no branch in the body;
//   * the six halo requests sit right behind the weights of position 3 and as far ahead of the next weights as the staging registers allow: loads return in order,
//     a weight fragment from L2 must not queue behind a halo pixel from HBM.
// Measured (profiles/r05_winograd.md): -3 ... 5 % per launch, +1.3 ... 2.9 % end to end against wino256x64c; the ablations and the three rearrangements that gained nothing.
// Arithmetic and accumulation order are those of wino256x64c: bit-identical results.
// DABL (tuning builds, PF_WINO_ABL=<mask>; WRONG results, timing only): 1 = no LDS reads of the pixel columns, 2 = no raw-halo staging, 4 = no weight requests,
// 8 = no transform arithmetic, 16 = no barrier, 32 = no address updates -- each removed from the chunk loop only.  (Mask 1 also makes the
// transform loop invariant: the compiler hoists it -- read it as "no LDS reads, no transform".)
// HALF: the block's 8 x 8 tiles are TWO independent half patches of 8 x 16 output pixels (tile rows 0..3 / 4..7 = tile sub-tiles m = 0 / 1), each with its own
// (image, row, column) origin and its own 10-row halo in the raw tile (20 rows instead of 18; the column pitch, the piece swizzle and every lane's addresses inside a
// sub-tile are unchanged: sub-tile 1 merely starts 10 rows down instead of 8).  For maps whose height is no multiple of 16 -- the 40 x 40 maps of the decoders: 3 x 3
// patches of 16 x 16 cover 48 x 48 (69 % of the blocks' work is inside the image); 5 x 3 half patches cover 40 x 48 (83 %): 240 blocks per head and channel tile at
// batch 32 instead of 288.  Per-tile arithmetic and accumulation order are untouched: bit-identical to the square form.
template <bool STAMP, int DABL = 0, bool HALF = false>
__global__ __launch_bounds__(W4_NT, 1) void wino4d_f2x2_kernel(const ConvParams p) {
  __shared__ __attribute__((aligned(16))) unsigned char wsm[WC_SMEM];
  constexpr int HROWS = HALF ? 20 : W_HY;          // halo rows in a raw tile
  constexpr int RCB = HROWS * RC_ROW;              // bytes of a raw tile
  constexpr int MSTEP = (HALF ? 10 : 8) * RC_ROW;  // sub-tile 1 below sub-tile 0
  constexpr int NPIX = HROWS * W_HX;
  static_assert(3 * RCB + W4_NT * 16 <= WC_SMEM && (NPIX * 4 + W4_NT - 1) / W4_NT == RAW4_F4, "raw tiles + dump area fit; six halo elements per thread");

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int tilesN = p.Cout / W_BN;
  const int tilesX = (p.Wo + W_PX - 1) / W_PX, tilesY = HALF ? (p.Ho + 7) / 8 : (p.Ho + W_PY - 1) / W_PY;   // HALF: half patches per column of patches
  const int nhalf = p.B * tilesY * tilesX;          // HALF: half patches of one group
  const int nblk1 = (HALF ? (nhalf + 1) / 2 : nhalf) * tilesN;
  int t = xcd_tile_index(nblk1 * p.groups);
  const bool g1 = t >= nblk1;
  if (g1) t -= nblk1;
  const ConvPtrs& P = g1 ? p.g[1] : p.g[0];
  const int nt = t % tilesN;
  const int n0 = nt * W_BN;
  int bimg2[2], oy2[2], ox2[2];   // origin of tile rows 0..3 / 4..7
  if constexpr (HALF) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int hp = 2 * (t / tilesN) + h;
      const bool live = hp < nhalf;
      hp = live ? hp : nhalf - 1;
      const int bx = hp % tilesX, by = (hp / tilesX) % tilesY;
      bimg2[h] = hp / (tilesX * tilesY);
      oy2[h] = live ? by * 8 : (1 << 28);    // a dead half (odd count): every pixel outside the image -> zeros in, nothing out
      ox2[h] = bx * W_PX;
    }
  } else {
    int mt = t / tilesN;
    const int bx = mt % tilesX; mt /= tilesX;
    const int by = mt % tilesY;
    bimg2[0] = bimg2[1] = mt / tilesY;
    oy2[0] = oy2[1] = by * W_PY;
    ox2[0] = ox2[1] = bx * W_PX;
  }
  const int nC = p.Cin / W_KC;

  // ---- raw halo staging: element e = tid + 256 i -> (pixel tid / 4 + 64 i, logical piece tid % 4); st[i]: its LDS address (piece swizzle) in the tile being filled;
  //      the threads without a sixth element store it (zeros) into a dump area behind the three tiles
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, p.x_bytes, 0x00020000);
  unsigned g_off[RAW4_F4];
  int st[RAW4_F4];
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) {
    const int pix = (tid >> 2) + 64 * i, c4 = tid & 3;
    const int hy = pix / W_HX, hx = pix - hy * W_HX;
    const int hh = HALF ? (hy >= 10 ? 1 : 0) : 0;                  // which half's halo this row belongs to
    const int iy = oy2[hh] - 1 + (HALF ? hy - 10 * hh : hy), ix = ox2[hh] - 1 + hx;
    const bool ok = pix < NPIX && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    g_off[i] = ok ? (unsigned)(((bimg2[hh] * p.H + iy) * p.W + ix) * p.Cin * 4 + c4 * 16) : OOB;
    st[i] = pix < NPIX ? hy * RC_ROW + hx * 64 + ((c4 ^ ((hx >> 1) & 3)) * 16) : 3 * RCB + tid * 16;
  }
  auto raw_soff = [&](int c) { return (c < nC ? c : nC - 1) * (W_KC * 4); };
  u32x4 ra[RAW4_F4];
  auto raw_load1 = [&](int i, int c) { ra[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, g_off[i], raw_soff(c), 0); };
  auto raw_store1 = [&](int i, int imm) { *reinterpret_cast<u32x4*>(wsm + st[i] + imm) = ra[i]; };

  // ---- transform operands: lane -> tile column l31 of sub-tile m (tile row ty = l31 / 8 + 4 m, tx = l31 % 8), channels 8 hi .. 8 hi + 7 = logical pieces 2 hi, 2 hi + 1
  // row pair of this wave's xi: xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3  ->  t = xA + sgn xB
  const int rA = wave == 0 ? 0 : (wave == 2 ? 2 : 1), rB = wave == 3 ? 3 : (wave == 2 ? 1 : 2);
  const float sgn_f = wave == 1 ? 1.f : -1.f;
  wf2 sgn = {sgn_f, sgn_f}, neg = {-1.f, -1.f};
  asm volatile("" : "+v"(sgn), "+v"(neg));  // register pairs: the packed forms take no scalar pair / no negated operand here
  const int ty0 = l31 >> 3, tx0 = l31 & 7;
  const int t_base = (2 * ty0) * RC_ROW + (2 * tx0) * 64;
  // LDS addresses of this lane's two pieces for pixel columns {0, 1} (swizzle tx & 3) and {2, 3} (swizzle (tx + 1) & 3), rows A and B, sub-tile 0.  Two sets that
  // walk the ring of raw tiles: X serves the windows that read tile c (0, 1, 3), Y those that read tile c + 1 (2, 4..7); both are advanced by one tile per chunk in
  // slots where they are idle (one v_add each: 22 address instructions per chunk with the six store addresses)
  int X[2][2][2], Y[2][2][2];  // [row A / B][column pair][piece]
#pragma unroll
  for (int jp = 0; jp < 2; ++jp)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const int o = t_base + jp * 128 + (((2 * hi + hf) ^ ((tx0 + jp) & 3)) * 16);
      X[0][jp][hf] = Y[0][jp][hf] = o + rA * RC_ROW;
      X[1][jp][hf] = Y[1][jp][hf] = o + rB * RC_ROW;
    }
  V8 xA, xB;  // ONE pixel column (8 channels) of the two rows: read in slots 0 / 1 of a window, combined in its slot 5
  auto col_read = [&](const int (&S)[2][2][2], int imm, int j, int part) {  // imm: sub-tile offset (compile time)
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      const WF4 v = *reinterpret_cast<const WF4*>(wsm + S[part][j >> 1][hf] + (imm + (j & 1) * 64));
      if (part) { xB.c[2 * hf] = v.lo; xB.c[2 * hf + 1] = v.hi; }
      else      { xA.c[2 * hf] = v.lo; xA.c[2 * hf + 1] = v.hi; }
    }
  };
  // (the packed instructions are written out: left to itself the compiler scalarised half of these combinations -- 6 instructions per slot instead of 4)
  auto pk_fma = [](const wf2 a, const wf2 b, const wf2 c) { wf2 d; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; };
  auto pk_add = [](const wf2 a, const wf2 b) { wf2 d; asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; };
  V8 tr[4];  // row combination, four pixel columns
  auto t_rows = [&](int j) {
#pragma unroll
    for (int e = 0; e < 4; ++e) tr[j].c[e] = pk_fma(sgn, xB.c[e], xA.c[e]);
  };
  V8 vo;  // one column combination
  auto t_cols = [&](int nu) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      vo.c[e] = nu == 0 ? pk_fma(neg, tr[2].c[e], tr[0].c[e]) : (nu == 1 ? pk_add(tr[1].c[e], tr[2].c[e]) : (nu == 2 ? pk_fma(neg, tr[1].c[e], tr[2].c[e]) : pk_fma(neg, tr[3].c[e], tr[1].c[e])));
  };
  unsigned vfh[4][4], vfl[4][4];  // ring of four fragments: [slot][channel pair], planes hi / lo
  auto s_cvt = [&](int slot, int e) {
    const sb_h2 hv = {(_Float16)vo.c[e].x, (_Float16)vo.c[e].y};
    vfh[slot][e] = __builtin_bit_cast(unsigned, hv);
  };
  auto s_lo = [&](int slot, int e) { asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(vfl[slot][e]) : "v"(vo.c[e].x), "v"(vfh[slot][e])); };
  auto s_hi = [&](int slot, int e) { asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(vfl[slot][e]) : "v"(vo.c[e].y), "v"(vfh[slot][e])); };

  // ---- weights: fragments of positions 4 wave + nu through a buffer resource (as in wino256x64c)
  const int w_frags = tilesN * nC * 16;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(P.w_wino), 0, w_frags * 4096, 0x00020000);
  const int w_s0 = (nt * nC * 16 + 4 * wave) * 4096;
  u32x4 bw[4][2][2];  // [nu][cout sub-tile][plane]
  auto load_w1 = [&](int c, int nu, int piece) {
    const int cc = c < nC ? c : nC - 1;
    bw[nu][piece >> 1][piece & 1] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16 + piece * 1024, w_s0 + (cc * 16 + nu) * 4096, 0);
  };
  f32x16 acc[4][2][2];  // [nu][cout sub-tile][tile sub-tile]
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][b][m][e] = 0.f;
  auto mma_one = [&](int f, int i) {  // MFMA i = 0..5 of fragment f = 4 m + nu: product i / 2 (wh vl, wl vh, wh vh), cout sub-tile i % 2
    const int m = f >> 2, nu = f & 3, t3 = i >> 1, ns = i & 1, sl = f & 3;
    const int tw = t3 == 1 ? 1 : 0;
    const u32x4 v = t3 == 0 ? u32x4{vfl[sl][0], vfl[sl][1], vfl[sl][2], vfl[sl][3]} : u32x4{vfh[sl][0], vfh[sl][1], vfh[sl][2], vfh[sl][3]};
    acc[nu][ns][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf16x8, bw[nu][ns][tw]), __builtin_bit_cast(wf16x8, v), acc[nu][ns][m], 0, 0, 0);
  };

  // ---- prologue: raw(0), raw(1) -> LDS; then the requests in the order a chunk of the loop leaves them in (the compiler's s_waitcnt insertion merges the loop's
  //      entry states): raw(2), then the weights of (chunk 0, positions 0..2)
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_load1(i, 0);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_store1(i, 0);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_load1(i, 1);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_store1(i, RCB);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) raw_load1(i, 2);
#pragma unroll
  for (int nu = 0; nu < 3; ++nu)
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) load_w1(0, nu, pc);
#pragma unroll
  for (int i = 0; i < RAW4_F4; ++i) st[i] += 2 * RCB;  // the loop's first chunk fills tile 2
  __syncthreads();
  // rows of (chunk 0, m = 0); fragments 0 and 1 of chunk 0; pixel column 0 of (chunk 0, m = 1)
#pragma unroll
  for (int j = 0; j < 4; ++j) { col_read(X, 0, j, 0); col_read(X, 0, j, 1); t_rows(j); }
#pragma unroll
  for (int nu = 0; nu < 2; ++nu) {
    t_cols(nu);
#pragma unroll
    for (int e = 0; e < 4; ++e) { s_cvt(nu, e); s_lo(nu, e); s_hi(nu, e); }
  }
  col_read(X, MSTEP, 0, 0); col_read(X, MSTEP, 0, 1); t_rows(0);
  WINO4_STAMP(0);

  // entering chunk c: fragments 0, 1 of chunk c in ring slots 0, 1; tr[0] = column 0 of (c, m = 1), tr[1..3] = columns of (c, m = 0); raw(c), raw(c + 1) in LDS;
  // ra = raw(c + 2) requested; weights of chunk c, positions 0..2 requested; X, Y -> tile c; st -> tile c + 2.
  // WD_SLOT(K): slot K of the chunk: window w = K / 6 runs the MFMAs of fragment w and the transform of fragment g = w + 2 (g >= 8: fragments 0, 1 of chunk c + 1).
  // The pixel column a window combines in slot 5 (read in slots 0, 1): nu = 1 -> column 3 of its own sub-tile, nu = 0 / 2 / 3 -> column 0 / 2 / 1 of the NEXT sub-tile;
  // it lies in tile c for windows 0, 1, 3 (address set X) and in tile c + 1 for the others (Y).
#define WD_SLOT(K)                                                                                                                                  \
  {                                                                                                                                                 \
    constexpr int k = (K), w = k / 6, s = k % 6;                                                                                                    \
    constexpr int g = w + 2, gf = g & 7, gm = gf >> 2, gnu = gf & 3;                                                                                \
    constexpr int rj = gnu == 1 ? 3 : (gnu == 0 ? 0 : (gnu == 2 ? 2 : 1));                                                                          \
    constexpr int rm = gnu == 1 ? gm : (gm ^ 1);                                                                                                    \
    constexpr bool useX = w == 0 || w == 1 || w == 3;                                                                                               \
    /* memory instructions of the slot */                                                                                                           \
    if constexpr (s < 2 && (DABL & 1) == 0) { if constexpr (useX) col_read(X, rm * MSTEP, rj, s); else col_read(Y, rm * MSTEP, rj, s); } \
    /* raw(c + 2) -> LDS in slots 2..7 (its requests are a chunk old), then the six requests for raw(c + 3) in slots 10..17: BEHIND the weights of position 3 and as \
       far ahead of the next weights as the registers allow -- loads return in order, a weight fragment from L2 must not queue behind a halo pixel from HBM */    \
    if constexpr ((DABL & 2) == 0) {                                                                                                                \
      if constexpr (w == 0 && s >= 2) raw_store1(s - 2, 0);                                                                                         \
      if constexpr (w == 1 && (s == 2 || s == 3)) raw_store1(s + 2, 0);                                                                             \
      if constexpr (w == 1 && s >= 4) raw_load1(s - 4, c + 3);                                                                                      \
      if constexpr (w == 2 && s >= 2) raw_load1(s, c + 3);                                                                                          \
    }                                                                                                                                               \
    /* weights: position 3 of this chunk in window 0 (last read in window 7 of the previous chunk); positions 0..2 of chunk c + 1 in windows 5..7 */ \
    if constexpr (s >= 2 && w == 0 && (DABL & 4) == 0) load_w1(c, 3, s - 2);                                                                        \
    if constexpr (s >= 2 && w >= 5 && (DABL & 4) == 0) load_w1(c + 1, w - 5, s - 2);                                                                \
    /* the MFMA and the transform's piece */                                                                                                        \
    mma_one(w, s);                                                                                                                                  \
    if constexpr ((DABL & 8) == 0) {                                                                                                                \
      if constexpr (s == 0) t_cols(gnu);                                                                                                            \
      if constexpr (s == 1) { s_cvt(g & 3, 0); s_cvt(g & 3, 1); s_cvt(g & 3, 2); s_lo(g & 3, 0); }                                                  \
      if constexpr (s == 2) { s_hi(g & 3, 0); s_lo(g & 3, 1); s_lo(g & 3, 2); }                                                                     \
      if constexpr (s == 3) { s_cvt(g & 3, 3); s_hi(g & 3, 1); s_lo(g & 3, 3); }                                                                    \
      if constexpr (s == 4) { s_hi(g & 3, 2); s_hi(g & 3, 3); }                                                                                     \
      if constexpr (s == 5) t_rows(rj);                                                                                                             \
    }                                                                                                                                               \
    /* address sets: Y -> tile c + 1 in windows 0, 1 (idle until window 2); X -> tile c + 1 in windows 4, 5 (idle after window 3); the store addresses in window 3 */ \
    if constexpr ((DABL & 32) == 0) {                                                                                                               \
      if constexpr (w < 2 && s >= 2) { Y[(s - 2) >> 1][w][(s - 2) & 1] += d0; }                                                                     \
      if constexpr ((w == 4 || w == 5) && s >= 2) { X[(s - 2) >> 1][w - 4][(s - 2) & 1] += d0; }                                                   \
      if constexpr (w == 3) st[s] += d2;                                                                                                            \
    }                                                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                                                              \
    if (c == 2) WINO4_STAMP(80 + k);                                                                                                                \
  }
#define WD_SLOT6(B) WD_SLOT(B) WD_SLOT((B) + 1) WD_SLOT((B) + 2) WD_SLOT((B) + 3) WD_SLOT((B) + 4) WD_SLOT((B) + 5)
  int ph = 0;  // c % 3
#pragma unroll 1
  for (int c = 0; c < nC; ++c) {
    const int d0 = ph == 2 ? -2 * RCB : RCB;   // tile (c + 1) - tile c
    const int d2 = ph == 0 ? -2 * RCB : RCB;   // tile (c + 3) - tile (c + 2)
    if (c < 16) WINO4_STAMP(8 + 2 * c);
    WD_SLOT6(0) WD_SLOT6(6) WD_SLOT6(12) WD_SLOT6(18) WD_SLOT6(24) WD_SLOT6(30) WD_SLOT6(36) WD_SLOT6(42)
    if (c < 16) WINO4_STAMP(9 + 2 * c);
    ph = ph == 2 ? 0 : ph + 1;
    if constexpr ((DABL & 16) == 0) __syncthreads();   // raw(c + 2) complete in LDS; raw(c) free
  }
#undef WD_SLOT6
#undef WD_SLOT
  WINO4_STAMP(1);
  wino4_epilogue<STAMP, HALF>(p, P, acc, wsm, bimg2, oy2, ox2, n0);
}

// 3x3 / stride 1 / pad 1, split-f16 scheme, one fp32 NHWC input, fp32 NHWC output, Winograd weights present
bool conv_wino_ok(const ConvParams& p) {
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || p.nterms != NT_F16X3 || p.nchw_out || p.ups || p.ln || p.splitk > 1) return false;
  if (p.C2 != 0 || (p.Cin % W_KC) != 0 || (p.Cout % W_BN) != 0 || (p.ldy & 3) != 0) return false;
  for (int g = 0; g < p.groups; ++g) {
    const ConvPtrs& q = p.g[g];
    if (!q.w_wino || !q.w_wino_inv || !q.x || !q.y || q.x_sb || q.y_sb || q.head_kind || q.bias_tab) return false;
  }
  return true;
}

static bool wino_half_geometry(const ConvParams& p) {
  static const int half_env = [] { const char* e = getenv("PF_WINO_HALF"); return e ? atoi(e) : 1; }();
  const int rem = p.Ho % W_PY;
  return half_env && rem >= 1 && rem <= 8;
}
int conv_wino_blocks(const ConvParams& p) {
  const int tilesN = p.Cout / W_BN, tilesX = (p.Wo + W_PX - 1) / W_PX;
  if (wino_half_geometry(p)) return ((p.B * ((p.Ho + 7) / 8) * tilesX + 1) / 2) * tilesN * p.groups;
  return p.B * ((p.Ho + W_PY - 1) / W_PY) * tilesX * tilesN * p.groups;
}

void launch_conv_wino(const ConvParams& p, hipStream_t s, int variant) {
  const int tilesN = p.Cout / W_BN, tilesX = (p.Wo + W_PX - 1) / W_PX, tilesY = (p.Ho + W_PY - 1) / W_PY;
  const dim3 grid(p.B * tilesY * tilesX * tilesN * p.groups);
  if (variant == 1) {  // "wino256x64d"
#ifdef PF_TUNING_BUILD
    static int dabl = -1;
    if (dabl < 0) { const char* e = getenv("PF_WINO_ABL"); dabl = e ? atoi(e) : 0; }
    switch (dabl) {
      case 1: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 1>), grid, dim3(W4_NT), 0, s, p); return;
      case 2: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 2>), grid, dim3(W4_NT), 0, s, p); return;
      case 4: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 4>), grid, dim3(W4_NT), 0, s, p); return;
      case 8: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 8>), grid, dim3(W4_NT), 0, s, p); return;
      case 16: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 16>), grid, dim3(W4_NT), 0, s, p); return;
      case 32: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 32>), grid, dim3(W4_NT), 0, s, p); return;
      case 7: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 7>), grid, dim3(W4_NT), 0, s, p); return;     // MFMAs + transform arithmetic + barrier, no memory instruction
      case 55: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 55>), grid, dim3(W4_NT), 0, s, p); return;   // 7 + 16 + 32: MFMAs + transform arithmetic only
      case 63: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 63>), grid, dim3(W4_NT), 0, s, p); return;   // MFMAs only
      case 15: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 15>), grid, dim3(W4_NT), 0, s, p); return;   // MFMAs + barrier + address updates
      case 6: hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 6>), grid, dim3(W4_NT), 0, s, p); return;      // no halo staging, no weight requests
      default: break;
    }
#endif
    // Half-patch geometry where the square patches would waste half a patch row: 1 <= Ho mod 16 <= 8 (the decoders' 40 x 40 maps; PF_WINO_HALF=0: square patches always)
    if (wino_half_geometry(p) && !p.stamps) {
      const int nhalf = p.B * ((p.Ho + 7) / 8) * tilesX;
      hipLaunchKernelGGL((wino4d_f2x2_kernel<false, 0, true>), dim3(((nhalf + 1) / 2) * tilesN * p.groups), dim3(W4_NT), 0, s, p);
      return;
    }
    if (p.stamps) hipLaunchKernelGGL(wino4d_f2x2_kernel<true>, grid, dim3(W4_NT), 0, s, p);
    else hipLaunchKernelGGL(wino4d_f2x2_kernel<false>, grid, dim3(W4_NT), 0, s, p);
    return;
  }
  // variant 0: "wino256x64c"
  if (p.stamps) hipLaunchKernelGGL(wino4c_f2x2_kernel<true>, grid, dim3(W4_NT), 0, s, p);
  else hipLaunchKernelGGL(wino4c_f2x2_kernel<false>, grid, dim3(W4_NT), 0, s, p);
}

// Host side: packed fp32 weights [Cout][3][KWCp] (k = (kx, ci), ci fastest) -> U = G g G^T per (cout, cin) in fp64, scaled per output channel by a power of two
// (largest |U| of the channel in [2^13, 2^14)), split into wh = fp16(U S), wl = fp16(U S - wh), laid out in the kernel's fragment order.
void wino_pack_weights(const float* packed, int Cout, int Cin, int KWCp, std::vector<unsigned short>* planes, std::vector<float>* inv_scale) {
  const int nC = Cin / W_KC, tilesN = Cout / W_BN;
  planes->assign((size_t)tilesN * nC * 16 * 2048, 0);
  inv_scale->assign(Cout, 1.f);
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  std::vector<double> U((size_t)Cin * 16);
  auto bits16 = [](_Float16 h) { unsigned short u; std::memcpy(&u, &h, 2); return u; };
  for (int n = 0; n < Cout; ++n) {
    double mx = 0.0;
    for (int ci = 0; ci < Cin; ++ci) {
      double g[3][3], t[4][3];
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = packed[((size_t)n * 3 + ky) * KWCp + kx * Cin + ci];
      for (int i = 0; i < 4; ++i)
        for (int kx = 0; kx < 3; ++kx) t[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
          U[(size_t)ci * 16 + i * 4 + j] = u;
          mx = std::max(mx, std::fabs(u));
        }
    }
    int e = 0;
    if (mx > 0.0 && std::isfinite(mx)) { int ex; (void)std::frexp(mx, &ex); e = 14 - ex; }
    e = std::max(-100, std::min(100, e));
    const double S = std::ldexp(1.0, e);
    (*inv_scale)[n] = std::ldexp(1.0f, -e);
    const int nt = n / W_BN, ns = (n % W_BN) / 32, row = n % 32;
    for (int ci = 0; ci < Cin; ++ci) {
      const int c = ci / W_KC, k = ci % W_KC, lane = row + 32 * (k / 8), el = k % 8;
      for (int pos = 0; pos < 16; ++pos) {
        const double us = U[(size_t)ci * 16 + pos] * S;
        const _Float16 hi = (_Float16)us;
        const _Float16 lo = (_Float16)(us - (double)hi);
        const size_t base = (((size_t)nt * nC + c) * 16 + pos) * 2048 + (size_t)(ns * 2) * 512 + (size_t)lane * 8 + el;
        (*planes)[base] = bits16(hi);
        (*planes)[base + 512] = bits16(lo);
      }
    }
  }
}

}  // namespace pf
