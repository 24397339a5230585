// Row-block linear layers for gfx950 (rb_common.h): y = act(LN?(x) W^T + b) + res on blocks of 64 token rows of one image, weights streamed from L2 straight into
// registers in MFMA fragment order, the block's rows as split-f16 fragments in LDS.  Two input forms:
//   rb_linear_kernel        K <= 512: the whole row block is resident in LDS (optionally LayerNorm'ed while staged: mix_transformers.py:199-200 norm1 / norm2);
//                           N = any number of 320- (or 256- / 384-) column passes over the same resident rows (q / proj: 1, kv: 2, fc1: 4)
//   rb_linear_stream_kernel deep K (fc2, K = 4 C): the rows stream through a two-slot LDS ring in k64 stages, split by the staging threads
// Both are plain replacements of one nn.Linear of MiT stages 3 / 4 (mix_transformers.py:26-29, 80-88); the fused chains (rb_chain.hip) are built from the same parts.
#include <stdlib.h>

#include "rb_common.h"

namespace pf {


// s_memtime stamp of one block's waves (STAMP builds of the kernels only; a stamp drains lgkmcnt, i.e. it perturbs the step it sits in)
#define RB_STAMP(i) do { if constexpr (STAMP) { if (blockIdx.x == 17 && lane == 0) p.stamps[wave * 64 + (i)] = __builtin_readcyclecounter(); } } while (0)
static constexpr int RB_NMAX = 1280;  // widest layer of the resident form (fc1 of stage 3)
// per-channel epilogue constants -> LDS (visible after the kernel's first barrier).  Two halves: the loads are issued with the kernel's other first loads (one
// round trip for everything), the LDS writes after the rows have been staged.  N <= 1280: at most 2 float4 per thread and table.
struct RbTabRegs { float4 iv[2], bb[2]; };
__device__ __forceinline__ void rb_tabs_load(RbTabRegs& t, const float* inv, const float* bias, int N, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int k = min(tid + 256 * i, N / 4 - 1);
    t.iv[i] = reinterpret_cast<const float4*>(inv)[k];
    t.bb[i] = reinterpret_cast<const float4*>(bias)[k];
  }
}
__device__ __forceinline__ void rb_tabs_store(const RbTabRegs& t, float* tabs, int N, int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
    if (tid + 256 * i < N / 4) {
      reinterpret_cast<float4*>(tabs)[tid + 256 * i] = t.iv[i];
      reinterpret_cast<float4*>(tabs + N)[tid + 256 * i] = t.bb[i];
    }
}

// Epilogue of one pass.  In the transposed accumulators a lane owns row (rt * 32 + l31) and, per register group g, channels 32 ct + 8 g + 4 hi .. + 3: stored straight
// from there a wave instruction writes 32 rows x 32 bytes -- 32 partial cache lines, ~400 cycles of store issue each (the 20 stores of a pass took 8 000 cycles, as long
// as 14 K steps: profiles/r04_rb_timeline.md).  So every 32 x 32 tile takes a turn through 4 KB of wave-private LDS (no block barrier: a wave's LDS instructions
// execute in order) and leaves as FULL 128-byte lines, 8 rows per store instruction; bias / residual / activation are applied on the row-major side.
// LDS tile: [32 rows][32 floats], the 16-byte column slot XOR-ed with (row & 7): conflict-free for the column-wise writes and the row-wise reads.
// inv / bias come from an LDS copy made at kernel start ([N] inverse scales, then [N] biases): read from global memory they were 5 dependent L2 round trips per pass.
// RES / ACT are compile-time: with run-time branches around the residual loads and the activation hipcc put an s_waitcnt vmcnt(0) at every join -- each tile then
// waited for the previous tile's stores to be acknowledged (and for the ring's last, useless read-ahead): 7 000 of the 8 000 cycles above.
template <class G, bool RES, int ACT>
__device__ __forceinline__ void rb_epilogue_store(const RbLinArgs& p, const float* tabs, float* scratch /*this wave's 4 KB*/, f32x16 (&acc)[G::NACC], int n0, int m0, int nrows,
                                                  int wave, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
  const int rrow = lane >> 3, c4 = lane & 7;
  float4 rr[RES ? G::NACC : 1][4];
  if constexpr (RES) {  // all residual rows of the pass in flight at once
#pragma unroll
    for (int idx = 0; idx < G::NACC; ++idx) {
      int rt, ct;
      bool own;
      rb_tile_of<G>(idx, wave, rt, ct, own);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        rr[idx][i] = *reinterpret_cast<const float4*>(p.res + (size_t)(m0 + min(rt * 32 + rrow + 8 * i, nrows - 1)) * p.N + n0 + ct * 32 + c4 * 4);
    }
  }
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
    const float4 iv = *reinterpret_cast<const float4*>(tabs + n), bb = *reinterpret_cast<const float4*>(tabs + p.N + n);
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = rrow + 8 * i;
      v[i] = *reinterpret_cast<const float4*>(scratch + row * 32 + ((c4 ^ (row & 7)) << 2));
    }
    __builtin_amdgcn_wave_barrier();  // the next tile's writes stay behind this tile's reads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ml = rt * 32 + rrow + 8 * i;
      float4 w = make_float4(fmaf(v[i].x, iv.x, bb.x), fmaf(v[i].y, iv.y, bb.y), fmaf(v[i].z, iv.z, bb.z), fmaf(v[i].w, iv.w, bb.w));
      if constexpr (ACT == ACT_GELU) w = make_float4(gelu_erf(w.x), gelu_erf(w.y), gelu_erf(w.z), gelu_erf(w.w));
      if constexpr (RES) { w.x += rr[idx][i].x; w.y += rr[idx][i].y; w.z += rr[idx][i].z; w.w += rr[idx][i].w; }
      if (own && ml < nrows) {
        if (p.sat) sat_watch4(p.sat, p.sat_limit, w.x, w.y, w.z, w.w);
        *reinterpret_cast<float4*>(p.y + (size_t)(m0 + ml) * p.N + n) = w;
      }
    }
  }
}

// ---- resident form
template <int K, bool LN, class G, bool RES, int ACT, int ABL = 0, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void rb_linear_kernel(const RbLinArgs p) {
  constexpr int KC = K / 16;          // k16 chunks
  constexpr int CPT = KC / 4;         // chunks per staging thread (thread = row tid / 4, chunks (tid & 3) + 4 i)
  static_assert(KC % 4 == 0 && KC % RB_D == 0, "K must be a multiple of 64");
  __shared__ __attribute__((aligned(16))) unsigned char As[KC * G::CHS];
  __shared__ __attribute__((aligned(16))) float tabs[2 * RB_NMAX];
  __shared__ __attribute__((aligned(16))) float escr[4 * 1024];  // epilogue transposition, 4 KB per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  RB_STAMP(0);
  const int img = blockIdx.x / p.bpi, j = blockIdx.x - img * p.bpi;
  const int nrows = min(RB_ROWS, p.tokens - j * RB_ROWS);
  const int m0 = img * p.tokens + j * RB_ROWS;

  RbW<G> W;
  W.init(p.w, p.w_bytes, wave, lane);
  RbTabRegs tr;

  {  // ---- rows -> (LayerNorm) -> split-f16 fragments
    const int r = tid >> 2, q = tid & 3;
    const float* xr = p.x + (size_t)(m0 + min(r, nrows - 1)) * K;  // rows past the block's end: a valid row, never stored
    float4 v[CPT][4];
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = *reinterpret_cast<const float4*>(xr + 16 * (q + 4 * i) + 4 * e);
    // everything the block needs first is in flight at once: its rows (HBM), the epilogue constants, the first RB_D steps of the weight stream
    rb_tabs_load(tr, p.inv, p.bias, p.N, tid);
    W.prologue();
    RB_STAMP(1);
    if constexpr (LN) {  // two passes over the registers, like F.layer_norm: mean, then the variance of the centred row
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < CPT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += (v[i][e].x + v[i][e].y) + (v[i][e].z + v[i][e].w);
      s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
      const float mu = s * (1.0f / K);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < CPT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[i][e] = make_float4(v[i][e].x - mu, v[i][e].y - mu, v[i][e].z - mu, v[i][e].w - mu);
          ss = fmaf(v[i][e].x, v[i][e].x, fmaf(v[i][e].y, v[i][e].y, fmaf(v[i][e].z, v[i][e].z, fmaf(v[i][e].w, v[i][e].w, ss))));
        }
      ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2);
      const float rs = 1.0f / sqrtf(ss * (1.0f / K) + p.ln_eps);
#pragma unroll
      for (int i = 0; i < CPT; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float4 g = *reinterpret_cast<const float4*>(p.ln_g + 16 * (q + 4 * i) + 4 * e), b = *reinterpret_cast<const float4*>(p.ln_b + 16 * (q + 4 * i) + 4 * e);
          v[i][e] = make_float4(fmaf(v[i][e].x * rs, g.x, b.x), fmaf(v[i][e].y * rs, g.y, b.y), fmaf(v[i][e].z * rs, g.z, b.z), fmaf(v[i][e].w * rs, g.w, b.w));
        }
    }
#pragma unroll
    for (int i = 0; i < CPT; ++i) {
      rb_store_chunk(As + (q + 4 * i) * G::CHS, r, v[i]);
    }
    rb_tabs_store(tr, tabs, p.N, tid);
  }
  RB_STAMP(2);
  __syncthreads();
  RB_STAMP(3);

  const int xr = wave & 1;
  RbA<G> A[2];
  A[0].read(As, lane, xr);
  const int npass = p.N / G::COLS;
#pragma unroll 1
  for (int ps = 0; ps < npass; ++ps) {
    f32x16 acc[G::NACC];
#pragma unroll
    for (int i = 0; i < G::NACC; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll 1
    for (int s = 0; s < KC; s += RB_D) {
#pragma unroll
      for (int d = 0; d < RB_D; ++d) {
        const int nx = s + d + 1 == KC ? 0 : s + d + 1;  // the next pass starts over on the same rows
        rb_step<G, ABL>(acc, W, d, A[d & 1], A[(d + 1) & 1], As + nx * G::CHS, lane, xr);
      }
      RB_STAMP(4 + ps * 8 + s / RB_D);
    }
    rb_epilogue_store<G, RES, ACT>(p, tabs, escr + wave * 1024, acc, ps * G::COLS, m0, nrows, wave, lane);
    RB_STAMP(4 + ps * 8 + 6);
  }
}

// ---- streamed form (K a multiple of 64 L; one pass: N = G::COLS).  Three LDS ring slots of one k64 stage each; the hand-over of stage t + 1 (ds_write + the only
// barrier of the stage) sits in the MIDDLE of stage t, so that the look-ahead fragment reads never wait at a stage boundary: between two barriers a wave reads slots
// t - 1 (its last step) and t, and writes slot t + 1.  The rows come from HBM (the 65 MB hidden map of the Mlp): their loads run L stages (~L microseconds) ahead in L
// register sets -- with one stage of lead the loop was a chain of exposed HBM round trips (profiles/r04_rb_ablation.md: 23 us with an EMPTY step body).
template <class G, int L, bool RES, int ABL = 0, bool STAMP = false>
__global__ __launch_bounds__(256, 1) void rb_linear_stream_kernel(const RbLinArgs p, int K) {
  constexpr int SLOT = 4 * G::CHS;
  __shared__ __attribute__((aligned(16))) unsigned char As[3 * SLOT];
  __shared__ __attribute__((aligned(16))) float tabs[2 * G::COLS];
  __shared__ __attribute__((aligned(16))) float escr[4 * 1024];  // epilogue transposition, 4 KB per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  RB_STAMP(0);
  const int img = blockIdx.x / p.bpi, j = blockIdx.x - img * p.bpi;
  const int nrows = min(RB_ROWS, p.tokens - j * RB_ROWS);
  const int m0 = img * p.tokens + j * RB_ROWS;
  const int T = K / 64;  // stages (a multiple of L)

  RbW<G> W;
  W.init(p.w, p.w_bytes, wave, lane);
  RbTabRegs tr;

  const int r = tid >> 2, q = tid & 3;
  const float* xr_ = p.x + (size_t)(m0 + min(r, nrows - 1)) * K + 16 * q;  // rows past the block's end: a valid row, never stored
  float4 st[L][4];  // stage s lives in set s % L from its loads until its ds_write
  auto stage_load = [&](int t, float4 (&v)[4]) {
    const int tc = min(t, T - 1);  // past the end: re-reads the last stage, never used
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = *reinterpret_cast<const float4*>(xr_ + (size_t)tc * 64 + 4 * e);
  };
  stage_load(0, st[0]);  // everything the block needs first in flight at once: stage 0, the epilogue constants, the weight ring, the next stages
  rb_tabs_load(tr, p.inv, p.bias, p.N, tid);
  W.prologue();
#pragma unroll
  for (int u = 1; u < L; ++u) stage_load(u, st[u]);
  rb_store_chunk(As + q * G::CHS, r, st[0]);
  rb_tabs_store(tr, tabs, p.N, tid);
  stage_load(L, st[0]);
  RB_STAMP(1);
  __syncthreads();
  RB_STAMP(2);

  f32x16 acc[G::NACC];
#pragma unroll
  for (int i = 0; i < G::NACC; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  const int xr = wave & 1;
  RbA<G> A[2];
  A[0].read(As, lane, xr);
  int cur = 0;  // slot of stage t
#pragma unroll 1
  for (int t0 = 0; t0 < T; t0 += L) {
#pragma unroll
    for (int u = 0; u < L; ++u) {  // stage t0 + u; the set of stage t0 + u + 1 is (u + 1) % L
      const int t = t0 + u;
      const int nxt = cur == 2 ? 0 : cur + 1;
      const unsigned char* base = As + cur * SLOT;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        if (d == 2) {  // stage t + 1 (requested L stages ago) -> its slot; the loads of stage t + 1 + L take its registers over
          rb_store_chunk(As + nxt * SLOT + q * G::CHS, r, st[(u + 1) % L]);
          __syncthreads();
          stage_load(t + 1 + L, st[(u + 1) % L]);
          __builtin_amdgcn_sched_barrier(0);
        }
        rb_step<G, ABL>(acc, W, d, A[d & 1], A[(d + 1) & 1], d < 3 ? base + (d + 1) * G::CHS : As + nxt * SLOT, lane, xr);
      }
      cur = nxt;
      RB_STAMP(3 + t);
    }
  }
  RB_STAMP(40);
  rb_epilogue_store<G, RES, ACT_NONE>(p, tabs, escr + wave * 1024, acc, 0, m0, nrows, wave, lane);
  RB_STAMP(41);
}

// ---- launchers
// act: ACT_NONE or ACT_GELU; the streamed form has no activation (fc2)
bool rb_linear_supported(int K, int N) {
  if (K == 320 && N % 320 == 0 && N <= RB_NMAX) return true;      // MiT stage 3: q / proj / kv / fc1
  if (K % 256 == 0 && K > 320 && N == 320) return true;  // fc2 (K = 1280)
  return false;
}

template <bool LN, bool RES, int ACT>
static void rb_launch_res(const RbLinArgs& a, dim3 grid, hipStream_t s) {
  hipLaunchKernelGGL((rb_linear_kernel<320, LN, RbGeo<2, true>, RES, ACT>), grid, dim3(256), 0, s, a);
}
void launch_rb_linear(const RbLinArgs& a, int K, hipStream_t s) {
  const int B = a.M / a.tokens;
  const dim3 grid((unsigned)(B * a.bpi)), block(256);
  using G320 = RbGeo<2, true>;
  static const int abl = [] { const char* e = getenv("PF_RB_ABL"); return e ? atoi(e) : 0; }();  // timing-only ablation forms (wrong results): scripts/tune_rb.py
  if (abl) {
#define RB_ABL_CASE(V) if (abl == V) { if (K == 320) hipLaunchKernelGGL((rb_linear_kernel<320, false, G320, false, ACT_NONE, V>), grid, block, 0, s, a); else hipLaunchKernelGGL((rb_linear_stream_kernel<G320, 2, false, V>), grid, block, 0, s, a, K); return; }
    RB_ABL_CASE(1) RB_ABL_CASE(2) RB_ABL_CASE(3) RB_ABL_CASE(7)
#undef RB_ABL_CASE
  }
  if (a.stamps) {
    if (K == 320) hipLaunchKernelGGL((rb_linear_kernel<320, false, G320, false, ACT_NONE, 0, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((rb_linear_stream_kernel<G320, 2, false, 0, true>), grid, block, 0, s, a, K);
    return;
  }
  const bool ln = a.ln_g != nullptr, res = a.res != nullptr, gelu = a.act == ACT_GELU;
  if (K == 320) {
    if (ln) {
      if (res) { if (gelu) rb_launch_res<true, true, ACT_GELU>(a, grid, s); else rb_launch_res<true, true, ACT_NONE>(a, grid, s); }
      else { if (gelu) rb_launch_res<true, false, ACT_GELU>(a, grid, s); else rb_launch_res<true, false, ACT_NONE>(a, grid, s); }
    } else {
      if (res) { if (gelu) rb_launch_res<false, true, ACT_GELU>(a, grid, s); else rb_launch_res<false, true, ACT_NONE>(a, grid, s); }
      else { if (gelu) rb_launch_res<false, false, ACT_GELU>(a, grid, s); else rb_launch_res<false, false, ACT_NONE>(a, grid, s); }
    }
  } else {
    if (res) hipLaunchKernelGGL((rb_linear_stream_kernel<G320, 2, true>), grid, block, 0, s, a, K);
    else hipLaunchKernelGGL((rb_linear_stream_kernel<G320, 2, false>), grid, block, 0, s, a, K);
  }
}

}  // namespace pf
