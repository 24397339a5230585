// Split-bf16 implicit GEMM for 3x3 / stride 1 / pad 1 convolutions with an LDS-staged INPUT HALO TILE.
//
// igemm_sb_kernel gathers the A operand per tap: every input pixel of a 3x3 conv is fetched from L2 and split into its
// three bf16 parts nine times (once per tap), by every n-tile.  Here a block owns an 8 x 16 patch of output pixels of one
// image (128 GEMM rows) and, per 32-channel chunk, stages the 10 x 18 input halo ONCE (fetch + exact split, 180 rows);
// the nine taps then read shifted windows of that tile straight from LDS: per lane the halo row of its output pixel plus
// a block-uniform tap offset (odd patch rows are rotated by two columns so that the shifted ds_read_b128 stay
// bank-conflict free under the instruction's non-contiguous lane groups).  Per 9 K steps the A side costs 23 KB of loads and 180 x 32 splits instead of 144 KB and
// 1152 x 32; the weight tile (B) is staged per step as before.  The XOR piece swizzle is keyed on the halo row.
// K order is (channel chunk, tap) instead of (tap, channel chunk): same products, different fp32 summation order.
// fp32 operands; the two parity schemes: exact 3-way bf16 split (6 partial products) and 2-way fp16 split (3, the default).
// Measured and not kept (profiles/r01_tune_conv_sbh.txt): 16 x 16 patches with 8 waves (ties the 8 x 16 / 4-wave form on
// 256 -> 256 @80^2, loses elsewhere) and staging the weights of a whole kernel row per barrier pair (TPG = 3: the extra
// LDS costs a resident block, 3-7 % slower).
#include <stdlib.h>

#include "igemm_common.h"
#include "sb_split.h"

namespace pf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ f32x16 mfma16h(const u32x4 a, const u32x4 b, const f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
static constexpr int H_ROW = BK;  // ushorts per LDS row (64 bytes)

// Ablation forms of the halo kernel (tuning builds only; scripts/tune_conv.py tiles "sbhA<mask>", results are WRONG by construction -- they time the
// kernel with one cost removed): 1 = no split arithmetic while staging the halo (raw bits stored), 2 = no wh 2^-11 scaling of the weight fragment,
// 4 = no barriers in the K loop, 8 = half the LDS fragment reads (the second 16-deep chunk of a step reuses the first one's fragments),
// 16 / 32 = no global loads of the halo / of the weights in the K loop (opaque register values instead), 0x100 / 0x200 = the next halo chunk's loads are
// issued in tap step 0 / 2 instead of 4 (a real variant, right results).  The product build has no such parameter: ABL is the constant 0.
#ifdef PF_TUNING_BUILD
#define SBH_ABL_PARAM , int ABL = 0
#else
#define SBH_ABL_PARAM
static constexpr int ABL = 0;
#endif
__device__ __forceinline__ int sbh_piece(int row, int piece) { return piece ^ ((row >> 2) & 3); }

// SCH = 6: exact 3-way bf16 split, 6 MFMAs per product; SCH = NT_F16X3: 2-way fp16 split of the activations (2 LDS planes),
// scaled weights wh + wl from global memory plus wh2 = wh 2^-11 made while staging (3 LDS planes), 3 MFMAs per product.
template <int H_TY, int H_TX /*output patch*/, int BN, int WM, int WN, int MODE, int TPG /*taps whose weights are staged together: 1 or 3 (one kernel row)*/, int SCH,
          bool DB = false /*two LDS buffers for the weights: the next tap's weights are stored while this tap is still being read -> one barrier per tap instead of two*/,
          bool UPS = false /*the first input x is stored at HALF resolution: the conv runs on its bilinear x2 up-sampling (decode_head.py:284-286,
                             gravity_head.py:172), interpolated while the halo tile is staged -- the up-sampled tensor never exists in HBM*/,
          bool ASB = false /*the input comes as the two fp16 planes of the split-f16 scheme (ConvPtrs::x_sb, written by the producing conv's epilogue): the halo
                             staging is a plain 16-byte copy per plane -- no split arithmetic in this kernel (VALU instructions are paid in MFMA issue time,
                             profiles/DESIGN_history_r01_r04.md 4.7), and an element is split once by its producer instead of once per n-tile and halo overlap here.  MODE 0, no UPS.*/
          SBH_ABL_PARAM>
__global__ __launch_bounds__(WM * WN * 64, ((WM * WN == 8 && BN < 256 && (ABL & 0x8000) == 0) || (SCH == NT_F16X3 && TPG == 1 && !DB && (ABL & 0xa000) == 0 && BN <= 64 && !(H_TY == 16 && BN == 64 && WM * WN == 4))) ? 4 : 2) void igemm_sbh_kernel(const ConvParams p) {  // second argument: min waves per SIMD (the DMA-ring forms of the 4-wave tiles with BN <= 64 stay inside 128 VGPRs: four resident blocks; tuning builds: 0x8000 lifts that)
  constexpr int H_HX = H_TX + 2, H_HY = H_TY + 2;  // halo
  constexpr int H_ROWS = H_HX * H_HY;              // 180 halo pixels for 8 x 16, 324 for 16 x 16
  constexpr int BM = H_TY * H_TX;
  constexpr int NT = WM * WN * 64;
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave grid must tile the block");
  constexpr int SM = BM / (WM * 32);
  constexpr int SN = BN / (WN * 32);
  constexpr int RPB = NT / 4;                        // B rows staged per pass
  constexpr int B_ROWS = (BN + RPB - 1) / RPB;
  constexpr int A_F4 = (H_ROWS * 8 + NT - 1) / NT;   // float4 loads per thread per halo chunk (6)
  constexpr int PLANE_A = H_ROWS * H_ROW, PLANE_B = BN * H_ROW;  // ushorts
  constexpr int EPI_USHORTS = 2 * (WM * 32) * (BN + 4);
  constexpr bool F16 = SCH == NT_F16X3;
  // A planes in LDS, B planes in LDS, B planes loaded from global memory.  split-f16: the third weight operand wh2 = wh 2^-11 is made in
  // REGISTERS from the wh fragment (4 v_pk_mul_f16 per fragment -- the VALU has slack, the LDS pipe does not): 2 LDS planes, 2 of every
  // 14 (SN = 1) / 4 of every 20 (SN = 2) fragment reads and a third of the weight-staging LDS writes less than with a staged wh2 plane
  constexpr int NPA = F16 ? 2 : 3, NPB = F16 ? 2 : 3, NPG = F16 ? 2 : 3;
  static_assert(!DB || TPG == 1, "double-buffered weights: one tap per step");
  constexpr int BBUF = NPB * TPG * PLANE_B;  // ushorts of one weight buffer
  // UPS: the half-resolution SOURCE pixels under the halo tile (rows oy0/2 - 1 .. + H_TY/2 + 1, columns likewise; indices clamped
  // into the map), one 32-channel chunk in fp32 -- staged once, every halo pixel then interpolates its 4 neighbours from LDS
  constexpr int S_TY = H_TY / 2 + 2, S_TX = H_TX / 2 + 2, S_PIX = S_TY * S_TX;  // 6 x 10 for an 8 x 16 patch
  constexpr int SRC_USHORTS = UPS ? S_PIX * BK * 2 : 0;
  constexpr int S_F4 = UPS ? (S_PIX * 8 + NT - 1) / NT : 1;  // float4 loads per thread per source chunk (2)
  // DMAW: the weights of a tap go from global memory straight into LDS (global_load_lds_dwordx4, no staging registers, no ds_write) into a ring of buffers ahead of
  // their use, one barrier per tap.  Motivation (profiles/r02_sbh_ablation.md): hipcc sinks the register-staged weight loads of the plain loop to the END of the tap's
  // MFMA phase (it reuses the fragment registers for them under the 128-VGPR cap), so their L2 latency is exposed in every tap: 18 % of the kernel's time on
  // 256 -> 256 @80^2.  Every split-f16 tile with the plain tap loop uses it (r03): THREE buffers (two taps ahead) on the 16 x 16 patch / 8-wave tile of the
  // 256 -> 256 decoder convs, whose two resident blocks per CU fit the 66 KB (0.790 -> 0.762 ms at 256 -> 256 @80^2); TWO buffers (one tap ahead, issued at the
  // start of the step) on the 4-wave tiles, where a third buffer would cost a resident block (profiles/r03_sbh_variants.txt: 128x32 0.563 -> 0.507 ms, 256x32
  // 0.503 -> 0.465, 128x128 0.220 -> 0.204, 128x64 0.214 -> 0.209; +0.8 % end to end once the unscaled low plane had halved the K loop's VALU work).  Results are
  // bit-identical to the register-staged loop (same products, same order).  Tuning builds: ABL bit 0x2000 forces the register-staged loop, 0x1000 / 0x4000 three / two buffers.
  constexpr bool DMAW = SCH == NT_F16X3 && TPG == 1 && !DB && (ABL & 0x2000) == 0;
  constexpr int NBUF = DMAW ? (((ABL & 0x1000) != 0 || ((ABL & 0x4000) == 0 && H_TY == 16 && BN == 64 && WM * WN == 8)) ? 3 : 2) : (DB ? 2 : 1);
  constexpr int OPER_USHORTS = NPA * PLANE_A + NBUF * BBUF + SRC_USHORTS;
  constexpr int SMEM_USHORTS = OPER_USHORTS > EPI_USHORTS ? OPER_USHORTS : EPI_USHORTS;
  __shared__ __attribute__((aligned(16))) unsigned short smem_u[SMEM_USHORTS];
  unsigned short* As = smem_u;                  // [NPA][H_ROWS][H_ROW]
  unsigned short* Bs0 = smem_u + NPA * PLANE_A;  // [DB ? 2 : 1][TPG][NPB][BN][H_ROW]
  float* Ss = reinterpret_cast<float*>(smem_u + NPA * PLANE_A + NBUF * BBUF);  // UPS: [S_PIX][BK] fp32 source tile

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int l31 = lane & 31;
  const int hi = lane >> 5;

  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesX = (p.Wo + H_TX - 1) / H_TX, tilesY = (p.Ho + H_TY - 1) / H_TY;
  const int nblk1 = p.B * tilesY * tilesX * tilesN;
  int t = xcd_tile_index(nblk1 * p.groups);
  const bool g1 = t >= nblk1;
  if (g1) t -= nblk1;
  const ConvPtrs& P = g1 ? p.g[1] : p.g[0];
  const int n0 = (t % tilesN) * BN;
  int mt = t / tilesN;
  const int tx = mt % tilesX; mt /= tilesX;
  const int ty = mt % tilesY;
  const int bimg = mt / tilesY;
  const int oy0 = ty * H_TY, ox0 = tx * H_TX;

  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x), 0, p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.x2 ? P.x2 : P.x), 0, P.x2 ? p.x2_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(F16 ? P.w_h16 : P.w_sb), 0, (F16 ? 2u : 5u) * p.w_sb_plane_bytes, 0x00020000);

  // ---- A halo staging: element e = tid + NT i -> (halo row e / 8, float4 e % 8 of the 32-channel chunk)
  // UPS: x is at half resolution.  a_off1 then addresses this thread's SOURCE-tile elements (s_off), and a_ups[i] packs, for
  // halo element i, the source-tile slots of its four neighbours and the weights: slot00 (bits 0-7), +1 column (bit 8), +1 row
  // (bit 9), lx (bits 10-11), ly (bits 12-13) with 0 -> 0, 1 -> 0.25, 2 -> 0.75, valid (bit 14).  src = (dst + 0.5) / 2 - 0.5
  // clamped at 0, neighbour clamped at the last row / column: exactly the stand-alone upsample2x kernels of elem.hip.
  static_assert(!UPS || (TPG == 1 && !DB && (H_TY % 2) == 0 && (H_TX % 2) == 0), "fused up-sampling: plain tap loop, even patch");
  static_assert(!ASB || (SCH == NT_F16X3 && MODE == 0 && !UPS), "split-plane input: split-f16 scheme, one input, no fused up-sampling");
  // ASB: element e -> (halo row e / 8, piece e % 8): pieces 0-3 = the four 16-byte pieces (8 channels each) of the hi plane's 32-channel chunk, 4-7 = the lo plane's
  const __amdgpu_buffer_rsrc_t rxs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(ASB ? P.x_sb : nullptr), 0,
                                                                       ASB ? (unsigned)(((p.x_sb_plane & ~(size_t)1) + (size_t)p.x_bytes / 4) * 2) : 0u, 0x00020000);
  unsigned a_off1[A_F4], a_off2[MODE == 2 ? A_F4 : 1], a_ups[UPS ? A_F4 : 1], s_off[S_F4];
  const int hs = p.H >> 1, ws = p.W >> 1;
#pragma unroll
  for (int i = 0; i < A_F4; ++i) {
    const int e = tid + NT * i;
    const int hrow = e >> 3, c4 = e & 7;
    const int hy = hrow / H_HX, hx = hrow - hy * H_HX;
    const int iy = oy0 - 1 + hy, ix = ox0 - 1 + hx;
    const bool in_tile = hrow < H_ROWS;
    const bool ok = in_tile && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const int pix = (bimg * p.H + iy) * p.W + ix;
    if (UPS) {
      const int y0 = iy <= 0 ? 0 : ((iy & 1) ? (iy - 1) >> 1 : (iy >> 1) - 1), x0 = ix <= 0 ? 0 : ((ix & 1) ? (ix - 1) >> 1 : (ix >> 1) - 1);
      const unsigned ly = iy <= 0 ? 0u : ((iy & 1) ? 1u : 2u), lx = ix <= 0 ? 0u : ((ix & 1) ? 1u : 2u);
      const int slot = (y0 - (oy0 / 2 - 1)) * S_TX + (x0 - (ox0 / 2 - 1));  // source tile origin (oy0/2 - 1, ox0/2 - 1)
      a_ups[UPS ? i : 0] = (unsigned)(ok ? slot : 0) | (x0 < ws - 1 ? 0x100u : 0u) | (y0 < hs - 1 ? 0x200u : 0u) | (lx << 10) | (ly << 12) | (ok ? 0x4000u : 0u);
      a_off1[i] = OOB;
    } else if (ASB) {  // byte offset inside the plane pair: plane (c4 >> 2) + pixel + 16-byte piece (c4 & 3); the chunk offset (2 bytes per channel) is added per chunk
      a_off1[i] = ok ? (unsigned)((size_t)(c4 >> 2) * (p.x_sb_plane & ~(size_t)1) * 2 + (size_t)pix * p.C1 * 2 + (c4 & 3) * 16) : OOB;
    } else {
      a_off1[i] = ok ? (unsigned)(pix * p.C1 * 4 + c4 * 16) : OOB;
    }
    if (MODE == 2) a_off2[i] = ok ? (unsigned)(pix * p.C2 * 4 + c4 * 16) : OOB;
  }
#pragma unroll
  for (int j = 0; j < S_F4; ++j) {  // source-tile element e = tid + NT j -> (slot e / 8, float4 e % 8); coordinates clamped into the map
    const int e = tid + NT * j, slot = e >> 3, c4 = e & 7;
    const int sy = min(max(oy0 / 2 - 1 + slot / S_TX, 0), hs - 1), sx = min(max(ox0 / 2 - 1 + slot % S_TX, 0), ws - 1);
    s_off[j] = (UPS && slot < S_PIX) ? (unsigned)(((bimg * hs + sy) * ws + sx) * p.C1 * 4 + c4 * 16) : OOB;
  }
  // ---- B staging: thread -> (row rb0 + RPB i, 16-byte piece pc of the 64-byte K chunk), three planes
  const int pc = tid & 3;
  const int rb0 = tid >> 2;
  unsigned b_off[B_ROWS];
#pragma unroll
  for (int i = 0; i < B_ROWS; ++i) {
    const int n = n0 + rb0 + RPB * i;
    b_off[i] = (n < p.Cout && rb0 + RPB * i < BN) ? (unsigned)(n * 3 * p.KWCp + pc * 8) * 2u : OOB;
  }

  float4 ra[A_F4], rb[TPG][B_ROWS][NPG];
  const int nC = p.Cin / BK;  // 32-channel chunks (x first, then x2 when concatenating)

  auto load_a = [&](int c) {  // chunk c (>= nC: nothing, out-of-range offsets)
    const bool live = c < nC;
    const int ci0 = c * BK;
    const bool first = MODE != 2 || ci0 < p.C1;
    const unsigned coff = (unsigned)((first ? ci0 : ci0 - p.C1) * (ASB ? 2 : 4));
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      if (UPS) {
        // an x chunk: the first S_F4 registers carry this thread's part of the half-resolution SOURCE tile (store_a_ups); an x2
        // chunk: the plain halo elements.  Branch-free: the load that does not apply goes to the out-of-range offset (zeros).
        float4 v1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < S_F4) v1 = buf_load16(rx, (live && first && s_off[i < S_F4 ? i : 0] != OOB) ? s_off[i < S_F4 ? i : 0] + coff : OOB);
        if (MODE == 2) {
          const unsigned b2 = a_off2[MODE == 2 ? i : 0];
          const float4 v2 = buf_load16(rx2, (live && !first && b2 != OOB) ? b2 + coff : OOB);
          v1 = make_float4(v1.x + v2.x, v1.y + v2.y, v1.z + v2.z, v1.w + v2.w);
        }
        ra[i] = v1;
        continue;
      }
      const unsigned base = (MODE != 2 || first) ? a_off1[i] : a_off2[MODE == 2 ? i : 0];
      const unsigned off = (live && base != OOB) ? base + coff : OOB;
      if (MODE == 2) {
        const float4 v1 = buf_load16(rx, first ? off : OOB);
        const float4 v2 = buf_load16(rx2, first ? OOB : off);
        ra[i] = make_float4(v1.x + v2.x, v1.y + v2.y, v1.z + v2.z, v1.w + v2.w);
      } else {
        if constexpr ((ABL & 16) != 0) {
          float o = __builtin_bit_cast(float, off | 0x3f000000u);
          asm volatile("" : "+v"(o));  // opaque: the split below stays
          ra[i] = make_float4(o, o, o, o);
        } else {
          ra[i] = buf_load16(ASB ? rxs : rx, off);
        }
      }
    }
  };
  // source tile -> LDS, then every halo element of this thread = bilinear combination of four LDS values -> split -> A planes
  auto store_a_ups = [&]() {
#pragma unroll
    for (int j = 0; j < S_F4; ++j) {
      const int e = tid + NT * j;
      if (e < S_PIX * 8) *reinterpret_cast<float4*>(Ss + e * 4) = ra[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int e = tid + NT * i, hrow = e >> 3, c4 = e & 7;
      if (hrow < H_ROWS) {
        unsigned f = a_ups[UPS ? i : 0];
        asm volatile("" : "+v"(f));  // keep the decode below inside the loop: hoisted, its addresses / weights cost ~40 VGPRs for the whole K loop
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f & 0x4000u) {
          const float* s00 = Ss + (f & 0xffu) * BK + c4 * 4;
          const int dx = (f & 0x100u) ? BK : 0, dy = (f & 0x200u) ? S_TX * BK : 0;
          const unsigned cx = (f >> 10) & 3u, cy = (f >> 12) & 3u;
          const float lx = cx == 0 ? 0.f : (cx == 1 ? 0.25f : 0.75f), ly = cy == 0 ? 0.f : (cy == 1 ? 0.25f : 0.75f);
          const float hx = 1.f - lx, hy = 1.f - ly;
          const float4 v00 = *reinterpret_cast<const float4*>(s00), v01 = *reinterpret_cast<const float4*>(s00 + dx);
          const float4 v10 = *reinterpret_cast<const float4*>(s00 + dy), v11 = *reinterpret_cast<const float4*>(s00 + dx + dy);
          o = make_float4(bilerp2x(v00.x, v01.x, v10.x, v11.x, hx, lx, hy, ly), bilerp2x(v00.y, v01.y, v10.y, v11.y, hx, lx, hy, ly),
                          bilerp2x(v00.z, v01.z, v10.z, v11.z, hx, lx, hy, ly), bilerp2x(v00.w, v01.w, v10.w, v11.w, hx, lx, hy, ly));
        }
        uint2 h, m;
        split4_f16(o, h, m);
        unsigned short* d = As + hrow * H_ROW + sbh_piece(hrow, c4 >> 1) * 8 + (c4 & 1) * 4;
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + PLANE_A) = m;
      }
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
      const int e = tid + NT * i, hrow = e >> 3, c4 = e & 7;  // recomputed (cheaper than six more live registers)
      if (ASB) {
        if (hrow < H_ROWS)  // the planes hold what split4_f16 would compute: copy piece (c4 & 3) of plane (c4 >> 2)
          *reinterpret_cast<float4*>(As + (c4 >> 2) * PLANE_A + hrow * H_ROW + sbh_piece(hrow, c4 & 3) * 8) = ra[i];
        continue;
      }
      if (hrow < H_ROWS) {
        uint2 h, m, l;
        if constexpr (F16 && (ABL & 1) != 0) {
          h = make_uint2(__builtin_bit_cast(unsigned, ra[i].x), __builtin_bit_cast(unsigned, ra[i].y));
          m = make_uint2(__builtin_bit_cast(unsigned, ra[i].z), __builtin_bit_cast(unsigned, ra[i].w));
        } else if (F16) split4_f16(ra[i], h, m);
        else split4(ra[i], h, m, l);
        unsigned short* d = As + hrow * H_ROW + sbh_piece(hrow, c4 >> 1) * 8 + (c4 & 1) * 4;
        *reinterpret_cast<uint2*>(d) = h;
        *reinterpret_cast<uint2*>(d + PLANE_A) = m;
        if (!F16) *reinterpret_cast<uint2*>(d + 2 * PLANE_A) = l;
      }
    }
  };
  auto load_b = [&](int c, int tap0) {  // weights of (chunk c, taps tap0 .. tap0 + TPG - 1); c >= nC: nothing
    const bool live = c < nC;
#pragma unroll
    for (int u = 0; u < TPG; ++u) {
      const int tap = tap0 + u;
      const int ky = tap / 3, kx = tap - 3 * ky;
      const unsigned woff = (unsigned)(ky * p.KWCp + kx * p.Cin + c * BK) * 2u;
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i)
#pragma unroll
        for (int pl = 0; pl < NPG; ++pl)
          if constexpr ((ABL & 32) != 0) {
            float o = __builtin_bit_cast(float, (b_off[i] + woff) | 0x3c003c00u);
            asm volatile("" : "+v"(o));
            rb[u][i][pl] = make_float4(o, o, o, o);
          } else {
            rb[u][i][pl] = buf_load16(rw, (live && b_off[i] != OOB) ? b_off[i] + woff + (unsigned)pl * p.w_sb_plane_bytes : OOB);
          }
    }
  };
  auto store_b = [&](int buf = 0) {
    unsigned short* Bs = Bs0 + (DB ? buf * BBUF : 0);
#pragma unroll
    for (int u = 0; u < TPG; ++u)
#pragma unroll
      for (int i = 0; i < B_ROWS; ++i)
        if (BN % RPB == 0 || rb0 + RPB * i < BN) {
#pragma unroll
          for (int pl = 0; pl < NPG; ++pl)
            *reinterpret_cast<float4*>(Bs + (u * NPB + pl) * PLANE_B + (rb0 + RPB * i) * H_ROW + sbh_piece(rb0 + RPB * i, pc) * 8) = rb[u][i][pl];
        }
  };

  // ---- DMAW: 16-byte piece e = tid + NT j of a tap's weights, e -> (plane, row, LDS slot); the XOR piece swizzle is applied on the GLOBAL side
  // (slot s of row r holds piece s ^ ((r >> 2) & 3)), the LDS side of the DMA is linear: wave w writes the 1 KB at piece index 64 w + NT j
  constexpr int DMA_PIECES = NPB * BN * 4;
  static_assert(!DMAW || DMA_PIECES % NT == 0, "DMA weights: whole rounds");
  constexpr int DMA_R = DMAW ? DMA_PIECES / NT : 1;
  const char* wsrc[DMA_R];
#pragma unroll
  for (int j = 0; j < DMA_R; ++j) {
    const int e = tid + NT * j;
    const int plane = e / (BN * 4), rem = e % (BN * 4), row = rem >> 2, slot = rem & 3;
    const int piece = slot ^ ((row >> 2) & 3);
    const int n = min(n0 + row, p.Cout - 1);  // columns past Cout are never stored: any finite weights do
    wsrc[j] = reinterpret_cast<const char*>(P.w_h16) + (size_t)plane * p.w_sb_plane_bytes + ((size_t)n * 3 * p.KWCp + piece * 8) * 2;
  }
  const unsigned bs_lds = (unsigned)(size_t)(__attribute__((address_space(3))) void*)Bs0;
  auto dma_b = [&](int c, int tap, int buf) {
    const int cc = c < nC ? c : nC - 1;  // past the end: a harmless reload (keeps the vmcnt bookkeeping static)
    const int ky = tap / 3, kx = tap - 3 * ky;
    const unsigned woff = (unsigned)(ky * p.KWCp + kx * p.Cin + cc * BK) * 2u;
#pragma unroll
    for (int j = 0; j < DMA_R; ++j) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(bs_lds + (unsigned)(buf * BBUF * 2 + (wave * 64 + NT * j) * 16));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(wsrc[j] + woff), "s"(dst) : "memory");
    }
  };

  f32x16 acc[SM][SN];
#pragma unroll
  for (int i = 0; i < SM; ++i)
#pragma unroll
    for (int j = 0; j < SN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // GEMM row -> patch pixel: rows 16 k .. 16 k + 15 are patch row k; ODD patch rows are rotated by ODD_SHIFT columns.
  // ds_read_b128 is serviced in the non-contiguous lane groups {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}: with the plain
  // mapping lanes 16-31 sit 18 halo rows (not 16) after lanes 0-15 and two lanes of every group share a bank slot (PMC:
  // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.26-0.34); rotating the odd rows by -2 columns makes every group cover 16
  // distinct residues mod 16 again, for every tap offset (brute-force check in tests/test_host_logic.py).
  constexpr int ODD_SHIFT = H_TX - 2;
  const int wm0 = (wave / WN) * (SM * 32);
  const int wn0 = (wave % WN) * (SN * 32);
  int hb[SM];  // halo row of this lane's output pixel (tap (0,0)) per 32-row subtile
#pragma unroll
  for (int i = 0; i < SM; ++i) {
    const int ml = wm0 + i * 32 + l31;
    const int pr = ml / H_TX, pc0 = ml % H_TX;
    hb[i] = pr * H_HX + ((pr & 1) ? (pc0 + ODD_SHIFT) % H_TX : pc0);
  }
  const unsigned short* Bb0 = Bs0 + (wn0 + l31) * H_ROW;
  const int swz_b = (l31 >> 2) & 3;

  auto compute = [&](int tap, int slot /*position of this tap's weights in the staged group*/, int buf = 0) {
    const unsigned short* Bb = Bb0 + ((DB || DMAW) ? buf * BBUF : 0);
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int toff = ky * H_HX + kx;
    u32x4 af[SM][NPA], bf[SN][3];
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two 16-deep chunks per K step; this lane's 8 k-values = piece 2c + hi
      if ((ABL & 8) == 0 || c == 0) {
#pragma unroll
      for (int i = 0; i < SM; ++i) {
        const int row = hb[i] + toff;
        const unsigned short* ap = As + row * H_ROW + sbh_piece(row, 2 * c + hi) * 8;
#pragma unroll
        for (int pl = 0; pl < NPA; ++pl) af[i][pl] = *reinterpret_cast<const u32x4*>(ap + pl * PLANE_A);
      }
      const int pob = ((2 * c + hi) ^ swz_b) * 8;
#pragma unroll
      for (int j = 0; j < SN; ++j) {
#pragma unroll
        for (int pl = 0; pl < NPB; ++pl) bf[j][pl] = *reinterpret_cast<const u32x4*>(Bb + (slot * NPB + pl) * PLANE_B + j * 32 * H_ROW + pob);
      }
      }
      constexpr int TA[6] = {F16 ? 1 : 2, 0, F16 ? 0 : 1, 1, 0, 0};  // plane of A: l h m m h h | split-f16: al ah ah
      constexpr int TB[6] = {0, F16 ? 1 : 2, F16 ? 0 : 1, 0, 1, 0};  // plane of B: h l m h m h | split-f16: wh wl wh (al is unscaled: no wh 2^-11 operand)
      // (s_setprio(1) around this MFMA cluster, as in cnx_mlp.hip: measured -3.5 % end to end, profiles/r03_validate_call4.log)
#pragma unroll
      for (int t6 = 0; t6 < (F16 ? 3 : 6); ++t6)
#pragma unroll
        for (int i = 0; i < SM; ++i)
#pragma unroll
          for (int j = 0; j < SN; ++j)
            acc[i][j] = mfma16h<F16>(af[i][TA[t6]], bf[j][TB[t6]], acc[i][j]);
    }
  };

  // prologue: halo chunk 0 and the weights of the first tap group -> LDS
  constexpr int NG = 9 / TPG;  // tap groups per chunk
  if constexpr (DMAW) {
    constexpr int AHEAD = NBUF - 1;  // taps of look-ahead
    // VMEM instructions one load_a() issues (out-of-range offsets still issue): the static vmcnt bookkeeping below depends on it
    constexpr int A_LOADS = UPS ? S_F4 + (MODE == 2 ? A_F4 : 0) : (MODE == 2 ? 2 : 1) * A_F4;
    load_a(0);
#pragma unroll
    for (int u = 0; u < AHEAD; ++u) dma_b(0, u, u);
    if (UPS && 0 < p.C1) store_a_ups(); else store_a();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // step q = 9 c + g reads weight buffer q % 3 = g % 3; the DMA of step q + 2 goes into buffer (g + 2) % 3, last read in step q - 1 (every wave is past
    // the barrier that ended it).  End of step q: every wave waits for ITS part of step q + 1's weights (issued one step ago: the younger VMEM
    // instructions -- this step's DMA and the halo loads of this / the previous step -- stay in flight), then one barrier.
    // Two buffers: step q reads buffer q % 2 (`par`: nine taps per chunk, the parity alternates from chunk to chunk), its DMA (tap q + 1) goes into the other one,
    // last read in step q - 1; the wait at the end of the step then covers THIS step's DMA (younger: only this step's halo loads).
    constexpr int LA = NG / 2;
    int par = 0;
    for (int c = 0; c < nC; ++c) {
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        dma_b(g + AHEAD < 9 ? c : c + 1, (g + AHEAD) % 9, NBUF == 3 ? (g + 2) % 3 : (par ^ 1));
        if (g == LA) load_a(c + 1);
        compute(g, 0, NBUF == 3 ? g % 3 : par);
        constexpr int younger = NBUF == 3 ? DMA_R : 0;  // three buffers: this step's DMA is the NEXT step's business
        if (g == LA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger + A_LOADS) : "memory");
        else if (g == LA + 1 && NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger + A_LOADS) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger) : "memory");
        if (g == 8 && c + 1 < nC) {  // every wave has read this chunk's halo
          __syncthreads();
          if (UPS && (c + 1) * BK < p.C1) store_a_ups(); else store_a();  // block-uniform
        }
        __syncthreads();
        par ^= 1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the two look-ahead DMAs past the end: the epilogue reuses the LDS
    __syncthreads();
  } else {
  load_a(0);
  load_b(0, 0);
  if (UPS && 0 < p.C1) store_a_ups(); else store_a();
  store_b();
  __syncthreads();
  }
  if constexpr (DMAW) {
  } else if constexpr (DB) {
    // weights of step q = 9 c + g live in LDS buffer q & 1: while the waves read buffer q & 1, the weights of step q + 1 (loaded
    // at the start of this step) are stored into the other buffer -- nobody reads it since the barrier that ended step q - 1
    int par = 0;
    for (int c = 0; c < nC; ++c) {
#pragma unroll
      for (int g = 0; g < 9; ++g) {
        if (g == 4) load_a(c + 1);
        if (g + 1 < 9) load_b(c, g + 1); else load_b(c + 1, 0);
        compute(g, 0, par);
        store_b(par ^ 1);
        if (g == 8 && c + 1 < nC) { __syncthreads(); store_a(); }  // every wave has read this chunk's halo
        __syncthreads();
        par ^= 1;
      }
    }
  } else
  for (int c = 0; c < nC; ++c) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {  // unrolled: no branch around any load, the s_waitcnt counts stay exact
      if constexpr ((ABL & 0x800) != 0) {  // weights first: the wait in front of store_b then leaves the (younger) halo loads in flight
        if (g + 1 < NG) load_b(c, (g + 1) * TPG); else load_b(c + 1, 0);
      }
      if (g == ((ABL & 0x100) ? 0 : (ABL & 0x200) ? 2 : NG / 2)) load_a(c + 1);  // next halo chunk: in flight during the second half of this one
      if constexpr ((ABL & 0x800) == 0) {
        if (g + 1 < NG) load_b(c, (g + 1) * TPG); else load_b(c + 1, 0);
      }
#pragma unroll
      for (int u = 0; u < TPG; ++u) compute(g * TPG + u, u);
      if constexpr ((ABL & 4) == 0) __syncthreads();  // every wave has read this group's weights (and, in the last group, this chunk's halo)
      store_b();
      if (g == NG - 1 && c + 1 < nC) {
        if (UPS && (c + 1) * BK < p.C1) store_a_ups(); else store_a();  // block-uniform
      }
      if constexpr ((ABL & 4) == 0) __syncthreads();
    }
  }

  const Tile2D t2{bimg, oy0, ox0, H_TX, ODD_SHIFT};
  epilogue_nhwc<BM, BN, WM, WN, SM, SN, NT, SMEM_USHORTS / 2>(p, P, acc, reinterpret_cast<float*>(smem_u), 0, n0, &t2, F16 ? P.w_h16_inv_scale : nullptr);
}

#ifdef PF_TUNING_BUILD
template <int MASK>
static void launch_sbh_abl(const ConvParams& p, hipStream_t s) {  // sbh256x64w8 (the tile of the dominant 256 -> 256 @80^2 launches) with one cost removed
  const int tilesN = (p.Cout + 63) / 64, tilesX = (p.Wo + 15) / 16, tilesY = (p.Ho + 15) / 16;
  const dim3 grid(p.B * tilesY * tilesX * tilesN * p.groups), block(512);
  if (p.nterms != NT_F16X3 || p.C2 > 0 || p.ups || p.g[0].x_sb) return;
  hipLaunchKernelGGL((igemm_sbh_kernel<16, 16, 64, 4, 2, 0, 1, NT_F16X3, false, false, false, MASK>), grid, block, 0, s, p);
}
#endif

template <int H_TY, int H_TX, int BN, int WM, int WN, int TPG = 1, bool DB = false>
static void launch_sbh_cfg(const ConvParams& p, hipStream_t s) {
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesX = (p.Wo + H_TX - 1) / H_TX, tilesY = (p.Ho + H_TY - 1) / H_TY;
  const dim3 grid(p.B * tilesY * tilesX * tilesN * p.groups), block(WM * WN * 64);
  if (p.ups) {  // conv_sbh_tile_ok: split-f16 scheme, 8 x 16 patch / 4-wave tiles, plain tap loop
    if constexpr (WM * WN == 4 && TPG == 1 && !DB) {
      if (p.C2 > 0) hipLaunchKernelGGL((igemm_sbh_kernel<H_TY, H_TX, BN, WM, WN, 2, 1, NT_F16X3, false, true>), grid, block, 0, s, p);
      else          hipLaunchKernelGGL((igemm_sbh_kernel<H_TY, H_TX, BN, WM, WN, 0, 1, NT_F16X3, false, true>), grid, block, 0, s, p);
    }
    return;
  }
  if (p.g[0].x_sb) {  // conv_sbh_tile_ok: split-f16 planes, one input, plain tap loop
    if constexpr (TPG == 1 && !DB) hipLaunchKernelGGL((igemm_sbh_kernel<H_TY, H_TX, BN, WM, WN, 0, 1, NT_F16X3, false, false, true>), grid, block, 0, s, p);
    return;
  }
  if (p.nterms == NT_F16X3) {
    if (p.C2 > 0) hipLaunchKernelGGL((igemm_sbh_kernel<H_TY, H_TX, BN, WM, WN, 2, TPG, NT_F16X3, DB>), grid, block, 0, s, p);
    else          hipLaunchKernelGGL((igemm_sbh_kernel<H_TY, H_TX, BN, WM, WN, 0, TPG, NT_F16X3, DB>), grid, block, 0, s, p);
  } else if constexpr (TPG != 1 || DB) {
    return;  // the t3 / double-buffered forms exist for the split-f16 scheme only (conv_sbh_tile_ok)
  } else {
    if (p.C2 > 0) hipLaunchKernelGGL((igemm_sbh_kernel<H_TY, H_TX, BN, WM, WN, 2, TPG, 6>), grid, block, 0, s, p);
    else          hipLaunchKernelGGL((igemm_sbh_kernel<H_TY, H_TX, BN, WM, WN, 0, TPG, 6>), grid, block, 0, s, p);
  }
}

// 3x3 / stride 1 / pad 1, fp32 operands, fp32-accurate mode, channel counts multiples of 32
bool conv_sbh_ok(const ConvParams& p) {
  if (p.KH != 3 || p.KW != 3 || p.stride != 1 || p.pad != 1 || (p.nterms != 6 && p.nterms != NT_F16X3) || p.nchw_out) return false;
  if ((p.C1 % BK) != 0 || (p.C2 % BK) != 0 || p.KWCp != p.KWC) return false;
  const bool planes = p.g[0].x_sb != nullptr;  // split-plane input: the fp16 planes of the split-f16 scheme, one input, no fused up-sampling
  if (planes && (p.nterms != NT_F16X3 || !(p.x_sb_plane & SB_FMT_F16) || p.C2 > 0 || p.ups)) return false;
  for (int g = 0; g < p.groups; ++g) {
    if (planes ? !p.g[g].x_sb : (!p.g[g].x || (p.C2 > 0 && !p.g[g].x2))) return false;
    if (p.nterms == NT_F16X3 ? (!p.g[g].w_h16 || !p.g[g].w_h16_inv_scale) : !p.g[g].w_sb) return false;
  }
  return true;
}

// tiles 4.. (double-buffered weights "d", weights of a kernel row per step "t3") are built for the split-f16 scheme only
// fused up-sampling (p.ups): split-f16 scheme, tiles 0-2 (8 x 16 patch, 4 waves, plain tap loop)
bool conv_sbh_tile_ok(const ConvParams& p, int h_tile) {
  if (!conv_sbh_ok(p)) return false;
#ifdef PF_TUNING_BUILD
  constexpr int kWide32 = 10, kWide64 = 33;  // "sbh256x32": after the tuning-only tiles; "sbh256x64": last
#else
  constexpr int kWide32 = 4, kWide64 = 5;
#endif
  if (p.g[0].x_sb) return h_tile < 4 || h_tile == kWide32 || h_tile == kWide64;  // plane input: the plain-tap-loop tiles
  if (p.ups) return (h_tile < 3 || h_tile == kWide32 || h_tile == kWide64) && p.nterms == NT_F16X3 && (p.H % 2) == 0 && (p.W % 2) == 0;
#ifdef PF_TUNING_BUILD
  if (h_tile >= 13) return p.nterms == NT_F16X3 && p.C2 == 0;  // ablation forms: one plain fp32 input
#endif
  if (h_tile >= kWide32) return p.nterms == NT_F16X3;  // sbh256x32 (and, in tuning builds, the whole-N tiles sbh256x256w8 / sbhd256x256w8)
#ifdef PF_TUNING_BUILD
  return h_tile < 4 || p.nterms == NT_F16X3;
#else
  return h_tile < 4;
#endif
}

// ids = position among the "sbh" tiles of kSb[] (igemm_sb.hip)
void launch_conv_sbh(const ConvParams& p, int h_tile, hipStream_t s) {
  switch (h_tile) {
    case 0: launch_sbh_cfg<8, 16, 128, 2, 2>(p, s); break;
    case 1: launch_sbh_cfg<8, 16, 64, 2, 2>(p, s); break;
    case 2: launch_sbh_cfg<8, 16, 32, 4, 1>(p, s); break;
#ifndef PF_TUNING_BUILD
    case 4: launch_sbh_cfg<16, 16, 32, 4, 1>(p, s); break;  // "sbh256x32": 16 x 16 patch for the N = 32 layer (twice the MFMAs per barrier)
    case 5: launch_sbh_cfg<16, 16, 64, 4, 1>(p, s); break;  // "sbh256x64": 16 x 16 patch, wave tile 64 x 64, 4 waves (r04: for the fused-up-sampling conv0)
#else
    case 33: launch_sbh_cfg<16, 16, 64, 4, 1>(p, s); break;
    case 10: launch_sbh_cfg<16, 16, 32, 4, 1>(p, s); break;
    // whole N = 256 per block (256 x 256 / 8-wave geometry, wave tile 128 x 64): the halo is staged and split ONCE per patch instead of once
    // per n-tile, 0.58 fragment reads per MFMA instead of 1.17 -- but one block per CU at 256 VGPRs (33 / 52 spilled) with the plain
    // two-barrier tap loop: 258 / 221 TF vs 294 TF for sbh256x64w8 on 256 -> 256 @80^2 (profiles/r02_negative_results.md).  "d" = weights
    // double-buffered in LDS
    case 11: launch_sbh_cfg<16, 16, 256, 2, 4>(p, s); break;
    case 12: launch_sbh_cfg<16, 16, 256, 2, 4, 1, true>(p, s); break;
    // measured, no gain (profiles/r02_negative_results.md)
    case 4: launch_sbh_cfg<8, 16, 128, 2, 2, 1, true>(p, s); break;
    case 5: launch_sbh_cfg<8, 16, 64, 2, 2, 1, true>(p, s); break;
    case 6: launch_sbh_cfg<8, 16, 32, 4, 1, 1, true>(p, s); break;
    case 7: launch_sbh_cfg<16, 16, 64, 4, 2, 1, true>(p, s); break;
    case 8: launch_sbh_cfg<16, 16, 64, 4, 2, 3>(p, s); break;
    case 9: launch_sbh_cfg<8, 16, 64, 2, 2, 3>(p, s); break;
    // ablation forms of sbh256x64w8 (wrong results by construction; timing only): "sbhA<mask>"
    case 13: launch_sbh_abl<1>(p, s); break;
    case 14: launch_sbh_abl<2>(p, s); break;
    case 15: launch_sbh_abl<3>(p, s); break;
    case 16: launch_sbh_abl<4>(p, s); break;
    case 17: launch_sbh_abl<8>(p, s); break;
    case 18: launch_sbh_abl<48>(p, s); break;
    case 19: launch_sbh_abl<12>(p, s); break;
    case 20: launch_sbh_abl<11>(p, s); break;
    case 21: launch_sbh_abl<15>(p, s); break;
    case 22: launch_sbh_abl<63>(p, s); break;
    case 23: launch_sbh_abl<16>(p, s); break;
    case 24: launch_sbh_abl<32>(p, s); break;
    case 25: launch_sbh_abl<0x100>(p, s); break;
    case 26: launch_sbh_abl<0x200>(p, s); break;
    case 27: launch_sbh_abl<0x800>(p, s); break;
    case 28: launch_sbh_abl<0x900>(p, s); break;
    case 29: launch_sbh_abl<0x1000>(p, s); break;
    case 30: launch_sbh_abl<0x2000>(p, s); break;
    case 31: launch_sbh_cfg<16, 16, 128, 4, 2>(p, s); break;          // "sbh256x128w8"
    case 32: {                                                        // "sbh256x128w8u": the same without the 128-VGPR cap (one block of 8 waves per CU, no spills)
      const int tilesN = (p.Cout + 127) / 128, tilesX = (p.Wo + 15) / 16, tilesY = (p.Ho + 15) / 16;
      if (p.nterms == NT_F16X3 && p.C2 == 0 && !p.ups && !p.g[0].x_sb)
        hipLaunchKernelGGL((igemm_sbh_kernel<16, 16, 128, 4, 2, 0, 1, NT_F16X3, false, false, false, 0x8000>), dim3(p.B * tilesY * tilesX * tilesN * p.groups), dim3(512), 0, s, p);
      break;
    }
#endif
    default: launch_sbh_cfg<16, 16, 64, 4, 2>(p, s); break;  // 16 x 16 patch, 8 waves, two blocks per CU: weights staged once per 256 rows
  }
}

}  // namespace pf
