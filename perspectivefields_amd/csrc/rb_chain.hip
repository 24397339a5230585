// Fused row-block chains of a MiT block with spatial reduction (stage 3: C = 320, sr = 2), built from the parts of rb_common.h / rb_gemm.hip.
//
// rb_srkv_kernel -- the whole key / value branch of Attention.forward (mix_transformers.py:119-127) in ONE launch:
//     kv = Linear_kv( LayerNorm_sr( Conv2d_{2x2, stride 2}( LayerNorm_1(x) ) ) )
// It replaced four dependent launches (LayerNorm-1 over all tokens, the split-K spatial-reduction conv, its reduce, the LayerNorm-fused kv GEMM: 11 + 28 + 10 + 19 us
// at batch 32) on 3 200 reduced rows -- launches that cannot fill the chip and whose intermediates made three HBM round trips.
// A block owns 32 REDUCED tokens of one image (100 per image: blocks of 32 / 32 / 32 / 4; geometry RbGeo<2, true, 1>: one row tile, every wave owns 2 + 1 column
// tiles of the 320-column pass).  The 2 x 2 conv is a GEMM over K = 4 taps x 320 channels; a tap's A rows are the LayerNorm-1'ed SOURCE tokens (2 oy + ky, 2 ox + kx),
// gathered and normalised while they are staged -- every source token belongs to exactly one patch, so LayerNorm-1 is computed once per token here too.  Two taps
// (40 k16 chunks, 84 KB of fragments) are resident at a time; the weight ring runs through the restaging and on into the kv layer (one stream: sr, kv pass 0, pass 1).
// The conv output never leaves the registers: bias, LayerNorm over the 320 channels (row sums exchanged between the four waves through LDS, two passes like
// F.layer_norm), split, fragments of the kv GEMM's A operand.
#include <stdlib.h>

#include "rb_common.h"

namespace pf {

template <class G, bool RES, int ACT>
__device__ __forceinline__ void rb_chain_epilogue_store(float* y, int ldy, const float* tabs_inv, const float* tabs_bias, float* scratch, f32x16 (&acc)[G::NACC], int n0, int m0, int nrows,
                                                        int wave, int lane, unsigned* sat = nullptr, float sat_limit = 65504.f) {
  const int l31 = lane & 31, hi = lane >> 5;
  const int rrow = lane >> 3, c4 = lane & 7;
#pragma unroll
  for (int idx = 0; idx < G::NACC; ++idx) {
    int rt, ct;
    bool own;
    rb_tile_of<G>(idx, wave, rt, ct, own);
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(scratch + l31 * 32 + (((2 * g + hi) ^ (l31 & 7)) << 2)) = make_float4(acc[idx][4 * g], acc[idx][4 * g + 1], acc[idx][4 * g + 2], acc[idx][4 * g + 3]);
    __builtin_amdgcn_wave_barrier();
    const int n = n0 + ct * 32 + c4 * 4;
    const float4 iv = *reinterpret_cast<const float4*>(tabs_inv + n), bb = *reinterpret_cast<const float4*>(tabs_bias + n);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rrow + 8 * i;
      v[i] = *reinterpret_cast<const float4*>(scratch + row * 32 + ((c4 ^ (row & 7)) << 2));
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ml = rt * 32 + rrow + 8 * i;
      const float4 w = make_float4(fmaf(v[i].x, iv.x, bb.x), fmaf(v[i].y, iv.y, bb.y), fmaf(v[i].z, iv.z, bb.z), fmaf(v[i].w, iv.w, bb.w));
      if (own && ml < nrows) {
        if (sat) sat_watch4(sat, sat_limit, w.x, w.y, w.z, w.w);
        *reinterpret_cast<float4*>(y + (size_t)(m0 + ml) * ldy + n) = w;
      }
    }
  }
}

template <int C>
__global__ __launch_bounds__(256, 1) void rb_srkv_kernel(const RbSrKvArgs p) {
  using G = RbGeo<2, true, 1>;
  static_assert(C == 320, "geometry: one 320-column pass");
  constexpr int KC = C / 16;       // k16 chunks per tap (20)
  constexpr int CPT = KC / 4;      // chunks per staging thread
  constexpr int HALF = 2 * KC;     // chunks resident at a time: two taps
  __shared__ __attribute__((aligned(16))) unsigned char As[HALF * G::CHS];
  __shared__ __attribute__((aligned(16))) float tabs[2 * C + 4 * C];   // sr: inv, bias [C]; kv: inv [2C], bias [2C]
  __shared__ __attribute__((aligned(16))) float red[2][4][32];         // LayerNorm row sums: [pass][wave][row]
  __shared__ __attribute__((aligned(16))) float escr[4 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int img = blockIdx.x / p.bpi, j = blockIdx.x - img * p.bpi;
  const int ntok = p.Hr * p.Wr;                       // reduced tokens per image
  const int r0 = j * 32, nrows = min(32, ntok - r0);
  const int Wm = 2 * p.Wr;                            // source map width (H = 2 Hr, W = 2 Wr)

  RbW<G> W;
  W.init(p.w, p.w_bytes, wave, lane);

  // ---- staging of two taps: thread -> (source row r = tid / 4: reduced row r & 31 of tap 2 half + (r >> 5), chunks (tid & 3) + 4 i)
  const int sr_ = tid >> 2, sq = tid & 3;
  const int red_t = r0 + min(sr_ & 31, nrows - 1);    // rows past the block's end: a valid row, never stored
  const int oy = red_t / p.Wr, ox = red_t - oy * p.Wr;
  auto stage_half = [&](int half, bool first) {
    const int tap = 2 * half + (sr_ >> 5), ky = tap >> 1, kx = tap & 1;
    const float* xr = p.x + ((size_t)img * (4 * ntok) + (size_t)(2 * oy + ky) * Wm + 2 * ox + kx) * C;
    float4 v[CPT][4];
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = *reinterpret_cast<const float4*>(xr + 16 * (sq + 4 * i) + 4 * e);
    if (first) {  // everything the block needs first in flight at once
      for (int i = tid; i < C / 4; i += 256) {
        reinterpret_cast<float4*>(tabs)[i] = reinterpret_cast<const float4*>(p.sr_inv)[i];
        reinterpret_cast<float4*>(tabs + C)[i] = reinterpret_cast<const float4*>(p.sr_bias)[i];
      }
      for (int i = tid; i < 2 * C / 4; i += 256) {
        reinterpret_cast<float4*>(tabs + 2 * C)[i] = reinterpret_cast<const float4*>(p.kv_inv)[i];
        reinterpret_cast<float4*>(tabs + 4 * C)[i] = reinterpret_cast<const float4*>(p.kv_bias)[i];
      }
      W.prologue();
      }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) s += (v[i][e].x + v[i][e].y) + (v[i][e].z + v[i][e].w);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float mu = s * (1.0f / C);
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[i][e] = make_float4(v[i][e].x - mu, v[i][e].y - mu, v[i][e].z - mu, v[i][e].w - mu);
        ss = fmaf(v[i][e].x, v[i][e].x, fmaf(v[i][e].y, v[i][e].y, fmaf(v[i][e].z, v[i][e].z, fmaf(v[i][e].w, v[i][e].w, ss))));
      }
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2);
    const float rs = 1.0f / sqrtf(ss * (1.0f / C) + p.ln1_eps);
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4 g = *reinterpret_cast<const float4*>(p.ln1_g + 16 * (sq + 4 * i) + 4 * e), b = *reinterpret_cast<const float4*>(p.ln1_b + 16 * (sq + 4 * i) + 4 * e);
        v[i][e] = make_float4(fmaf(v[i][e].x * rs, g.x, b.x), fmaf(v[i][e].y * rs, g.y, b.y), fmaf(v[i][e].z * rs, g.z, b.z), fmaf(v[i][e].w * rs, g.w, b.w));
      }
      rb_store_chunk(As + ((sr_ >> 5) * KC + sq + 4 * i) * G::CHS, sr_ & 31, v[i]);
    }
  };

  f32x16 acc[G::NACC];
#pragma unroll
  for (int i = 0; i < G::NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  RbA<G> A[2];
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();  // every wave has read the first two taps
    stage_half(half, half == 0);
    __syncthreads();
    A[0].read(As, lane, 0);
#pragma unroll 1
    for (int s = 0; s < HALF; s += RB_D) {
#pragma unroll
      for (int d = 0; d < RB_D; ++d) {
        const int nx = s + d + 1 == HALF ? 0 : s + d + 1;  // the look-ahead past the end reads a stale chunk: re-read after the restaging / the LayerNorm below
        rb_step<G>(acc, W, d, A[d & 1], A[(d + 1) & 1], As + nx * G::CHS, lane, 0);
      }
    }
  }

  // ---- conv output (+ bias) in registers -> LayerNorm over its C channels -> fragments of the kv layer's A operand
  // lane: reduced row l31; accumulator idx, register 4 g + e: channel 32 ct + 8 g + 4 hi + e
  float ps = 0.f;
#pragma unroll
  for (int idx = 0; idx < G::NACC; ++idx) {
    int rt, ct;
    bool own;
    rb_tile_of<G>(idx, wave, rt, ct, own);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ct * 32 + 8 * g + 4 * hi;
      const float4 iv = *reinterpret_cast<const float4*>(tabs + n), bb = *reinterpret_cast<const float4*>(tabs + C + n);
      acc[idx][4 * g] = fmaf(acc[idx][4 * g], iv.x, bb.x); acc[idx][4 * g + 1] = fmaf(acc[idx][4 * g + 1], iv.y, bb.y);
      acc[idx][4 * g + 2] = fmaf(acc[idx][4 * g + 2], iv.z, bb.z); acc[idx][4 * g + 3] = fmaf(acc[idx][4 * g + 3], iv.w, bb.w);
      if (own) ps += (acc[idx][4 * g] + acc[idx][4 * g + 1]) + (acc[idx][4 * g + 2] + acc[idx][4 * g + 3]);
    }
  }
  ps += __shfl_xor(ps, 32);
  if (hi == 0) red[0][wave][l31] = ps;
  __syncthreads();  // also: every wave is done with the conv's A fragments
  const float mu = ((red[0][0][l31] + red[0][1][l31]) + (red[0][2][l31] + red[0][3][l31])) * (1.0f / C);
  float pq = 0.f;
#pragma unroll
  for (int idx = 0; idx < G::NACC; ++idx) {
    int rt, ct;
    bool own;
    rb_tile_of<G>(idx, wave, rt, ct, own);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc[idx][r] -= mu;
      if (own) pq = fmaf(acc[idx][r], acc[idx][r], pq);
    }
  }
  pq += __shfl_xor(pq, 32);
  if (hi == 0) red[1][wave][l31] = pq;
  __syncthreads();
  const float rs = 1.0f / sqrtf(((red[1][0][l31] + red[1][1][l31]) + (red[1][2][l31] + red[1][3][l31])) * (1.0f / C) + p.srn_eps);
#pragma unroll
  for (int idx = 0; idx < G::NACC; ++idx) {
    int rt, ct;
    bool own;
    rb_tile_of<G>(idx, wave, rt, ct, own);
    if (!own) continue;  // wave-uniform
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ct * 32 + 8 * g + 4 * hi;
      const float4 gm = *reinterpret_cast<const float4*>(p.srn_g + n), be = *reinterpret_cast<const float4*>(p.srn_b + n);
      const float4 y = make_float4(fmaf(acc[idx][4 * g] * rs, gm.x, be.x), fmaf(acc[idx][4 * g + 1] * rs, gm.y, be.y), fmaf(acc[idx][4 * g + 2] * rs, gm.z, be.z),
                                   fmaf(acc[idx][4 * g + 3] * rs, gm.w, be.w));
      uint2 h, l;
      split4_f16(y, h, l);
      // channel n .. n + 3 of row l31: chunk n / 16, k half (n / 8) & 1, bytes 8 hi .. 8 hi + 7 of the row's 16-byte slot
      unsigned char* d = As + (2 * ct + (g >> 1)) * G::CHS + (l31 + 32 * (g & 1)) * 16 + 8 * hi;
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + 1024) = l;
    }
  }
  __syncthreads();

  // ---- kv = LN(sr) Wkv^T + b: two passes of 320 columns over the same fragments
  const int m0 = img * ntok + r0;
  A[0].read(As, lane, 0);
#pragma unroll 1
  for (int ps2 = 0; ps2 < 2; ++ps2) {
#pragma unroll
    for (int i = 0; i < G::NACC; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll 1
    for (int s = 0; s < KC; s += RB_D) {
#pragma unroll
      for (int d = 0; d < RB_D; ++d) {
        const int nx = s + d + 1 == KC ? 0 : s + d + 1;
        rb_step<G>(acc, W, d, A[d & 1], A[(d + 1) & 1], As + nx * G::CHS, lane, 0);
      }
    }
    rb_chain_epilogue_store<G, false, ACT_NONE>(p.kv, 2 * C, tabs + 2 * C, tabs + 4 * C, escr + wave * 1024, acc, ps2 * G::COLS, m0, nrows, wave, lane, p.sat, 4094.f);
  }
}

// rb_proj_fc1_kernel -- the seam between the two halves of a transformer block in one launch (mix_transformers.py:199-200, :137-139, :52):
//     x1 = x + proj(attn_out);   hidden = fc1(LayerNorm_2(x1))
// A block owns 64 tokens.  The attention output is staged as fragments, the projection runs (20 steps), its accumulators + bias + residual ARE the new token rows: they
// are stored (through the transposing LDS path: full lines) and, still in registers, normalised -- row sums exchanged between the four waves through LDS, two passes --
// and written back as the fragments of fc1's A operand, which then runs its four 320-column passes.  One weight stream (proj, then fc1); the ring reads on through the
// LayerNorm.  Replaces two launches, one staging prologue and the read-back of x1.
template <int C>
__global__ __launch_bounds__(256, 1) void rb_proj_fc1_kernel(const RbProjFc1Args p) {
  using G = RbGeo<2, true, 2>;
  static_assert(C == 320, "geometry: 320-column passes");
  constexpr int KC = C / 16, CPT = KC / 4, H4 = 4 * C;
  __shared__ __attribute__((aligned(16))) unsigned char As[KC * G::CHS];
  __shared__ __attribute__((aligned(16))) float tabs[2 * C + 2 * H4];   // proj: inv, bias [C]; fc1: inv [4C], bias [4C]
  __shared__ __attribute__((aligned(16))) float red[2][4][64];          // LayerNorm row sums: [pass][wave][row]
  __shared__ __attribute__((aligned(16))) float escr[4 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int img = blockIdx.x / p.bpi, j = blockIdx.x - img * p.bpi;
  const int nrows = min(RB_ROWS, p.tokens - j * RB_ROWS);
  const int m0 = img * p.tokens + j * RB_ROWS;

  RbW<G> W;
  W.init(p.w, p.w_bytes, wave, lane);
  {  // ---- attention output rows -> fragments
    const int r = tid >> 2, q = tid & 3;
    const float* ar = p.attn + (size_t)(m0 + min(r, nrows - 1)) * C;
    float4 v[CPT][4];
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = *reinterpret_cast<const float4*>(ar + 16 * (q + 4 * i) + 4 * e);
    for (int i = tid; i < C / 4; i += 256) {
      reinterpret_cast<float4*>(tabs)[i] = reinterpret_cast<const float4*>(p.proj_inv)[i];
      reinterpret_cast<float4*>(tabs + C)[i] = reinterpret_cast<const float4*>(p.proj_bias)[i];
    }
    for (int i = tid; i < H4 / 4; i += 256) {
      reinterpret_cast<float4*>(tabs + 2 * C)[i] = reinterpret_cast<const float4*>(p.fc1_inv)[i];
      reinterpret_cast<float4*>(tabs + 2 * C + H4)[i] = reinterpret_cast<const float4*>(p.fc1_bias)[i];
    }
    W.prologue();
#pragma unroll
    for (int i = 0; i < CPT; ++i) rb_store_chunk(As + (q + 4 * i) * G::CHS, r, v[i]);
  }
  __syncthreads();

  const int xr = wave & 1;
  f32x16 acc[G::NACC];
#pragma unroll
  for (int i = 0; i < G::NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  RbA<G> A[2];
  A[0].read(As, lane, xr);
#pragma unroll 1
  for (int s = 0; s < KC; s += RB_D) {
#pragma unroll
    for (int d = 0; d < RB_D; ++d) {
      const int nx = s + d + 1 == KC ? 0 : s + d + 1;
      rb_step<G>(acc, W, d, A[d & 1], A[(d + 1) & 1], As + nx * G::CHS, lane, xr);
    }
  }

  // ---- x1 = acc inv + bias + x in the accumulator layout (lane: row rt 32 + l31; register 4 g + e of tile (rt, ct): channel 32 ct + 8 g + 4 hi + e)
  float ps[2] = {0.f, 0.f};
#pragma unroll
  for (int idx = 0; idx < G::NACC; ++idx) {
    int rt, ct;
    bool own;
    rb_tile_of<G>(idx, wave, rt, ct, own);
    const float* xrow = p.x + (size_t)(m0 + min(rt * 32 + l31, nrows - 1)) * C;
    float4 rr[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) rr[g] = *reinterpret_cast<const float4*>(xrow + ct * 32 + 8 * g + 4 * hi);
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ct * 32 + 8 * g + 4 * hi;
      const float4 iv = *reinterpret_cast<const float4*>(tabs + n), bb = *reinterpret_cast<const float4*>(tabs + C + n);
      acc[idx][4 * g] = fmaf(acc[idx][4 * g], iv.x, bb.x) + rr[g].x; acc[idx][4 * g + 1] = fmaf(acc[idx][4 * g + 1], iv.y, bb.y) + rr[g].y;
      acc[idx][4 * g + 2] = fmaf(acc[idx][4 * g + 2], iv.z, bb.z) + rr[g].z; acc[idx][4 * g + 3] = fmaf(acc[idx][4 * g + 3], iv.w, bb.w) + rr[g].w;
      t += (acc[idx][4 * g] + acc[idx][4 * g + 1]) + (acc[idx][4 * g + 2] + acc[idx][4 * g + 3]);
    }
    if (idx < 2 * G::CTW) ps[idx & 1] += t;           // compile-time row tile
    else { if (xr) ps[1] += t; else ps[0] += t; }     // the extra tile's row tile is wave-uniform
  }
  // the new token rows leave as full 128-byte lines (transposing LDS path, as rb_epilogue_store)
  {
    const int rrow = lane >> 3, c4 = lane & 7;
    float* scratch = escr + wave * 1024;
#pragma unroll
    for (int idx = 0; idx < G::NACC; ++idx) {
      int rt, ct;
      bool own;
      rb_tile_of<G>(idx, wave, rt, ct, own);
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(scratch + l31 * 32 + (((2 * g + hi) ^ (l31 & 7)) << 2)) = make_float4(acc[idx][4 * g], acc[idx][4 * g + 1], acc[idx][4 * g + 2], acc[idx][4 * g + 3]);
      __builtin_amdgcn_wave_barrier();
      float4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = rrow + 8 * i;
        v[i] = *reinterpret_cast<const float4*>(scratch + row * 32 + ((c4 ^ (row & 7)) << 2));
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ml = rt * 32 + rrow + 8 * i;
        if (ml < nrows) *reinterpret_cast<float4*>(p.x + (size_t)(m0 + ml) * C + ct * 32 + c4 * 4) = v[i];
      }
    }
  }
  // ---- LayerNorm_2 of the rows held in the accumulators
  ps[0] += __shfl_xor(ps[0], 32); ps[1] += __shfl_xor(ps[1], 32);
  if (hi == 0) { red[0][wave][l31] = ps[0]; red[0][wave][32 + l31] = ps[1]; }
  __syncthreads();  // also: every wave is done with the projection's A fragments
  float mu[2], pq[2] = {0.f, 0.f};
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) mu[rt] = ((red[0][0][rt * 32 + l31] + red[0][1][rt * 32 + l31]) + (red[0][2][rt * 32 + l31] + red[0][3][rt * 32 + l31])) * (1.0f / C);
#pragma unroll
  for (int idx = 0; idx < G::NACC; ++idx) {
    const float m = idx < 2 * G::CTW ? mu[idx & 1] : (xr ? mu[1] : mu[0]);
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[idx][r] -= m; t = fmaf(acc[idx][r], acc[idx][r], t); }
    if (idx < 2 * G::CTW) pq[idx & 1] += t;
    else { if (xr) pq[1] += t; else pq[0] += t; }
  }
  pq[0] += __shfl_xor(pq[0], 32); pq[1] += __shfl_xor(pq[1], 32);
  if (hi == 0) { red[1][wave][l31] = pq[0]; red[1][wave][32 + l31] = pq[1]; }
  __syncthreads();
  float rs[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
    rs[rt] = 1.0f / sqrtf(((red[1][0][rt * 32 + l31] + red[1][1][rt * 32 + l31]) + (red[1][2][rt * 32 + l31] + red[1][3][rt * 32 + l31])) * (1.0f / C) + p.ln2_eps);
#pragma unroll
  for (int idx = 0; idx < G::NACC; ++idx) {
    int rt, ct;
    bool own;
    rb_tile_of<G>(idx, wave, rt, ct, own);
    const float r_ = idx < 2 * G::CTW ? rs[idx & 1] : (xr ? rs[1] : rs[0]);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = ct * 32 + 8 * g + 4 * hi;
      const float4 gm = *reinterpret_cast<const float4*>(p.ln2_g + n), be = *reinterpret_cast<const float4*>(p.ln2_b + n);
      const float4 y = make_float4(fmaf(acc[idx][4 * g] * r_, gm.x, be.x), fmaf(acc[idx][4 * g + 1] * r_, gm.y, be.y), fmaf(acc[idx][4 * g + 2] * r_, gm.z, be.z),
                                   fmaf(acc[idx][4 * g + 3] * r_, gm.w, be.w));
      uint2 h, l;
      split4_f16(y, h, l);
      unsigned char* d = As + (2 * ct + (g >> 1)) * G::CHS + rt * 2048 + (l31 + 32 * (g & 1)) * 16 + 8 * hi;
      *reinterpret_cast<uint2*>(d) = h;
      *reinterpret_cast<uint2*>(d + 1024) = l;
    }
  }
  __syncthreads();

  // ---- hidden = LN2(x1) W1^T + b1: four passes of 320 columns
  A[0].read(As, lane, xr);
#pragma unroll 1
  for (int ps2 = 0; ps2 < 4; ++ps2) {
#pragma unroll
    for (int i = 0; i < G::NACC; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll 1
    for (int s = 0; s < KC; s += RB_D) {
#pragma unroll
      for (int d = 0; d < RB_D; ++d) {
        const int nx = s + d + 1 == KC ? 0 : s + d + 1;
        rb_step<G>(acc, W, d, A[d & 1], A[(d + 1) & 1], As + nx * G::CHS, lane, xr);
      }
    }
    rb_chain_epilogue_store<G, false, ACT_NONE>(p.hidden, H4, tabs + 2 * C, tabs + 2 * C + H4, escr + wave * 1024, acc, ps2 * G::COLS, m0, nrows, wave, lane, p.sat);
  }
}

bool rb_srkv_supported(int C, int sr) { return C == 320 && sr == 2; }

void launch_rb_proj_fc1(const RbProjFc1Args& a, int C, hipStream_t s) {
  const dim3 grid((unsigned)(a.B * a.bpi)), block(256);
  if (C == 320) hipLaunchKernelGGL((rb_proj_fc1_kernel<320>), grid, block, 0, s, a);
}

void launch_rb_srkv(const RbSrKvArgs& a, int C, hipStream_t s) {
  const dim3 grid((unsigned)(a.B * a.bpi)), block(256);
  if (C == 320) hipLaunchKernelGGL((rb_srkv_kernel<320>), grid, block, 0, s, a);
}

}  // namespace pf
