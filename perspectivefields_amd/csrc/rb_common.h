// Row-block GEMM machinery for the small-M layer chains (MiT stages 3 / 4: 3 200 - 12 800 token rows at batch 32; mix_transformers.py:108-141, :49-56, :198-202).
//
// Why another GEMM form.  The LDS tiles of igemm_sb_impl.h run these layers at 13-26 % MFMA-busy: a launch is ONE round of 64 x 64 blocks whose time is the latency of a
// block (prologue, 10-40 K steps of {global load -> split -> ds_write -> barrier -> ds_read -> 6 MFMAs -> barrier}, epilogue), and a transformer block is 11 such launches
// with an HBM / L2 round trip between each pair (profiles/r03_epilogue_batching.md).  Here a BLOCK OWNS 64 TOKEN ROWS OF ONE IMAGE for a whole chain of row-local layers:
//   * the A operand (the block's rows) lives in LDS as split-f16 fragments for the whole layer -- written once by the phase that produces it (LayerNorm, attention,
//     the previous layer's epilogue), read by plain conflict-free ds_read_b128;
//   * the WEIGHTS never touch LDS: every wave owns a fixed set of 32-column tiles of the output and streams exactly those columns' weights from L2 straight into
//     registers, in MFMA fragment order (packed at finalize: one buffer_load_dwordx4 per fragment, 1 KB contiguous per wave), RB_D k-steps ahead of their use.
//     The K loop therefore has NO barrier and no LDS write at all: a wave's step is 6 global fragment loads + 6 LDS fragment reads + 15 MFMAs;
//   * products are TRANSPOSED (weights are the MFMA's A operand, token rows its B operand): in the 32 x 32 C/D layout a lane then owns 4 consecutive output channels of
//     ITS OWN row per register group -- bias / LayerNorm / residual / activation and the float4 stores (or the next layer's A fragments) come straight from registers.
// One wave per SIMD (4 waves, 1 block per CU, up to 512 registers per wave): latency is covered by the register prefetch ring, not by occupancy.
//
// Geometry of a pass (NCT column tiles of 32 = the columns one trip over K produces): wave w owns column tiles w + 4 ci (ci < CTW) for BOTH row tiles and, when
// NCT = 4 CTW + 2, one more tile 4 CTW + (w >> 1) for row tile (w & 1)  ->  320 columns = CTW 2 + extra: 5 accumulators (80 registers) per wave.
// Weight stream (rb_pack_w in engine.hip): [pass][k16 step][column tile][plane hi / lo][lane 64][8 halfs], value = Ws[32 ct + (lane & 31)][16 step + 8 (lane >> 5) + e]
// with the per-output-channel power-of-two scale of the split-f16 scheme (igemm_sb_impl.h, sb_split.h); RB_D zero steps of padding at the end (the ring reads ahead).
// A fragments in LDS: chunk c (16 k values) at c * G::CHS, inside it [row tile][plane][slot = (row & 31) + 32 khalf][8 halfs]; the 64 pad bytes per chunk make the
// staging writes of four threads that hold four different chunks of one row conflict-free.
#pragma once
#include "igemm_common.h"
#include "sb_split.h"

namespace pf {

static constexpr int RB_ROWS = 64;         // token rows per block
static constexpr int RB_D = 4;             // weight prefetch depth in k16 steps (register ring)

typedef _Float16 rb_f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 rb_mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rb_f16x8, a), __builtin_bit_cast(rb_f16x8, b), c, 0, 0, 0);
}

template <int CTW_, bool EXTRA_, int RT_ = 2>
struct RbGeo {
  static constexpr int CTW = CTW_;
  static constexpr bool EXTRA = EXTRA_;
  static constexpr int RT = RT_;                           // row tiles of 32 per block: 2 (64-row blocks) or 1 (32-row blocks: the spatial-reduction branch, 100 rows per image)
  static constexpr int NW = CTW + (EXTRA ? 1 : 0);         // weight fragment pairs (hi, lo) per wave and k16 step
  static constexpr int NACC = RT * CTW + (EXTRA ? 1 : 0);  // 32 x 32 accumulators per wave
  static constexpr int NCT = 4 * CTW + (EXTRA ? 2 : 0);    // column tiles per pass
  static constexpr int COLS = NCT * 32;
  static constexpr int STEP_BYTES = NCT * 2048;            // weight bytes per k16 step of a pass
  static constexpr int CHS = RT * 2048 + 64;               // bytes per k16 chunk of the A fragments in LDS
  // RT == 1: the extra tile 4 CTW + (w >> 1) is computed by BOTH waves of a pair (w & 1 = 0, 1) on the same row tile; only the even wave's copy is used
};

// (Measured and removed, profiles/r04_rb_linear.md: an L2 warm-up of the whole weight stream at kernel start -- one 4-byte LDS-DMA load per 128-byte line, the blocks
// of an XCD covering disjoint slices -- made the layers SLOWER in the pipeline: fc1 49 -> 63 us, proj 20 -> 30 us.  The warm-up loads are younger than the ring's
// first steps, so the first counted vmcnt wait of the K loop also waits for every one of them, i.e. for HBM.)
// Per-wave state of the weight stream: one VGPR offset (lane * 16 + running step offset), NW wave-uniform tile offsets
template <class G>
struct RbW {
  __amdgpu_buffer_rsrc_t rw;
  unsigned voff;          // lane * 16 + bytes of the steps already issued
  unsigned toff[G::NW];   // byte offset of this wave's column tiles inside a step (wave-uniform)
  u32x4 f[RB_D][G::NW][2];
  __device__ __forceinline__ void init(const unsigned short* w, size_t bytes, int wave, int lane) {
    rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(w), 0, (int)(bytes < 0x7fffffffUL ? bytes : 0x7fffffffUL), 0x00020000);
    voff = (unsigned)lane * 16u;
#pragma unroll
    for (int ci = 0; ci < G::NW; ++ci) toff[ci] = (unsigned)__builtin_amdgcn_readfirstlane((ci < G::CTW ? wave + 4 * ci : 4 * G::CTW + (wave >> 1)) * 2048);
  }
  // one fragment of the next step of the stream into ring slot d (advance() once all 2 NW fragments of the step are issued)
  __device__ __forceinline__ void load_frag(int d, int ci, int pl) { f[d][ci][pl] = __builtin_amdgcn_raw_buffer_load_b128(rw, voff + (unsigned)pl * 1024u, toff[ci], 0); }
  __device__ __forceinline__ void advance() { voff += (unsigned)G::STEP_BYTES; }
  __device__ __forceinline__ void load(int d) {
#pragma unroll
    for (int ci = 0; ci < G::NW; ++ci)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) load_frag(d, ci, pl);
    advance();
  }
  __device__ __forceinline__ void prologue() {
#pragma unroll
    for (int d = 0; d < RB_D; ++d) load(d);
  }
};

// A fragments of one k16 chunk: all row tiles (+ for 64-row blocks the extra tile's row tile xr = wave & 1, read again: a run-time register select would cost more
// than two LDS reads)
template <class G>
struct RbA {
  u32x4 a[G::RT][2];   // [row tile][plane]
  u32x4 x[2];          // extra tile's row tile (RT == 2)
  __device__ __forceinline__ void read(const unsigned char* chunk, int lane, int xr) {
#pragma unroll
    for (int rt = 0; rt < G::RT; ++rt)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) a[rt][pl] = *reinterpret_cast<const u32x4*>(chunk + (rt * 2 + pl) * 1024 + lane * 16);
    if constexpr (G::EXTRA && G::RT == 2) {
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) x[pl] = *reinterpret_cast<const u32x4*>(chunk + (xr * 2 + pl) * 1024 + lane * 16);
    }
  }
  __device__ __forceinline__ const u32x4& xa(int pl) const { if constexpr (G::RT == 2) return x[pl]; else return a[0][pl]; }
};

// One k16 step of a wave: the LDS reads of the NEXT step's A fragments first (they land under this step's MFMAs), then the step's MFMAs -- per accumulator the three
// partial products of the split-f16 scheme smallest first (wh al, wl ah, wh ah), MFMAs on one accumulator never adjacent -- with the refill load of every weight
// fragment issued right behind the fragment's last use.  The scheduling fences pin that order: left alone hipcc issues the LDS reads behind the last MFMA (their latency
// is then exposed at the head of the next step) and sinks all refills of an unrolled group to its end (the ring would run one step ahead, not RB_D).
#define RB_FENCE() __builtin_amdgcn_sched_barrier(0)
// timing-only forms of a step (wrong results by construction): ABL & 1 no refill loads, & 2 no MFMAs (operands kept alive), & 4 no A fragment reads
template <class G, int ABL>
__device__ __forceinline__ void rb_step_abl(f32x16 (&acc)[G::NACC], RbW<G>& W, int d, const RbA<G>& A, RbA<G>& An, const unsigned char* next_chunk, int lane, int xr) {
  if constexpr (ABL == 8 || ABL == 16) {  // correct results, other issue orders: 8 = product-major (MFMAs on one accumulator NACC apart), refills at the end of the step;
                                          // 16 = the same without the fence behind the LDS reads
    An.read(next_chunk, lane, xr);
    if constexpr (ABL == 8) RB_FENCE();
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int ci = 0; ci < G::CTW; ++ci)
#pragma unroll
        for (int rt = 0; rt < G::RT; ++rt) acc[G::RT * ci + rt] = rb_mfma(W.f[d][ci][t == 1], A.a[rt][t == 0], acc[G::RT * ci + rt]);
      if constexpr (G::EXTRA) acc[G::RT * G::CTW] = rb_mfma(W.f[d][G::CTW][t == 1], A.xa(t == 0), acc[G::RT * G::CTW]);
    }
#pragma unroll
    for (int ci = 0; ci < G::NW; ++ci) { W.load_frag(d, ci, 0); W.load_frag(d, ci, 1); }
    RB_FENCE();
    W.advance();
    return;
  }
  if constexpr ((ABL & 4) == 0) An.read(next_chunk, lane, xr);
  RB_FENCE();
#pragma unroll
  for (int ci = 0; ci < G::NW; ++ci) {
    if constexpr ((ABL & 2) == 0) {
      if (ci < G::CTW) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int rt = 0; rt < G::RT; ++rt) acc[G::RT * ci + rt] = rb_mfma(W.f[d][ci][t == 1], A.a[rt][t == 0], acc[G::RT * ci + rt]);
      } else {
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[G::RT * G::CTW] = rb_mfma(W.f[d][ci][t == 1], A.xa(t == 0), acc[G::RT * G::CTW]);
      }
    } else {
      asm volatile("" :: "v"(W.f[d][ci][0]), "v"(W.f[d][ci][1]), "v"(A.a[0][0]), "v"(A.a[G::RT - 1][1]));
    }
    if constexpr ((ABL & 1) == 0) { W.load_frag(d, ci, 0); W.load_frag(d, ci, 1); }
    RB_FENCE();
  }
  W.advance();
}
template <class G, int ABL = 0>
__device__ __forceinline__ void rb_step(f32x16 (&acc)[G::NACC], RbW<G>& W, int d, const RbA<G>& A, RbA<G>& An, const unsigned char* next_chunk, int lane, int xr) {
  if constexpr (ABL != 0) { rb_step_abl<G, ABL>(acc, W, d, A, An, next_chunk, lane, xr); return; }
  constexpr int X = G::CTW, XA = G::RT * G::CTW, RT = G::RT;  // the extra tile's fragment pair / accumulator
  An.read(next_chunk, lane, xr);
  RB_FENCE();
#pragma unroll
  for (int ci = 0; ci < G::CTW; ++ci) {
    const u32x4 (&w)[2] = W.f[d][ci];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[RT * ci + rt] = rb_mfma(w[0], A.a[rt][1], acc[RT * ci + rt]);
    if (G::EXTRA && ci == 0) {
      acc[XA] = rb_mfma(W.f[d][X][0], A.xa(1), acc[XA]);
      RB_FENCE();
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[RT * ci + rt] = rb_mfma(w[1], A.a[rt][0], acc[RT * ci + rt]);
    W.load_frag(d, ci, 1);
    RB_FENCE();
    if (G::EXTRA && ci == 0) {
      acc[XA] = rb_mfma(W.f[d][X][1], A.xa(0), acc[XA]);
      W.load_frag(d, X, 1);
      RB_FENCE();
    }
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) acc[RT * ci + rt] = rb_mfma(w[0], A.a[rt][0], acc[RT * ci + rt]);
    W.load_frag(d, ci, 0);
    RB_FENCE();
    if (G::EXTRA && ci == G::CTW - 1) {
      acc[XA] = rb_mfma(W.f[d][X][0], A.xa(0), acc[XA]);
      W.load_frag(d, X, 0);
      RB_FENCE();
    }
  }
  W.advance();
}

// (row tile, column tile) of accumulator idx for this wave; `own` = false for the redundant copy of the extra tile in 32-row blocks
template <class G>
__device__ __forceinline__ void rb_tile_of(int idx, int wave, int& rt, int& ct, bool& own) {
  own = true;
  if (idx < G::RT * G::CTW) { rt = idx % G::RT; ct = wave + 4 * (idx / G::RT); }
  else { rt = G::RT == 2 ? (wave & 1) : 0; ct = 4 * G::CTW + (wave >> 1); own = G::RT == 2 || (wave & 1) == 0; }
}

// 16 consecutive k values (one chunk) of a row -> the row's two 16-byte slots per plane.  v[0..3] = k 0-3, 4-7, 8-11, 12-15
__device__ __forceinline__ void rb_store_chunk(unsigned char* chunk, int row, const float4 (&v)[4]) {
  uint2 h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) split4_f16(v[e], h[e], l[e]);
  unsigned char* d = chunk + (row >> 5) * 2048 + (row & 31) * 16;
  *reinterpret_cast<u32x4*>(d) = u32x4{h[0].x, h[0].y, h[1].x, h[1].y};
  *reinterpret_cast<u32x4*>(d + 512) = u32x4{h[2].x, h[2].y, h[3].x, h[3].y};
  *reinterpret_cast<u32x4*>(d + 1024) = u32x4{l[0].x, l[0].y, l[1].x, l[1].y};
  *reinterpret_cast<u32x4*>(d + 1536) = u32x4{l[2].x, l[2].y, l[3].x, l[3].y};
}

}  // namespace pf
