// Row-block linear layers for gfx950 (rb_common.h): y = act(LN?(x) W^T + b) + res on blocks of 64 token rows of one image, weights streamed from L2 straight into
// registers in MFMA fragment order, the block's rows as split-f16 fragments in LDS.  Two input forms:
//   rb_linear_kernel        K <= 512: the whole row block is resident in LDS (optionally LayerNorm'ed while staged: mix_transformers.py:199-200 norm1 / norm2);
//                           N = any number of 320- (or 256- / 384-) column passes over the same resident rows (q / proj: 1, kv: 2, fc1: 4)
//   rb_linear_stream_kernel deep K (fc2, K = 4 C): the rows stream through a two-slot LDS ring in k64 stages, split by the staging threads
// Both are plain replacements of one nn.Linear of MiT stages 3 / 4 (mix_transformers.py:26-29, 80-88); the fused chains (rb_chain.hip) are built from the same parts.
#include <stdlib.h>

#include "rb_common.h"

namespace pf {


// epilogue of one pass: lane owns row (rt * 32 + l31) and, per register group g, channels n0 + 32 ct + 8 g + 4 hi .. + 3
template <class G>
__device__ __forceinline__ void rb_epilogue_store(const RbLinArgs& p, f32x16 (&acc)[G::NACC], int n0, int m0, int nrows, int wave, int lane) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int idx = 0; idx < G::NACC; ++idx) {
    int rt, ct;
    rb_tile_of<G>(idx, wave, rt, ct);
    const int ml = rt * 32 + l31;
    const bool ok = ml < nrows;
    const size_t row = (size_t)(m0 + (ok ? ml : 0)) * p.N;
    float4 iv[4], bb[4], rr[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + ct * 32 + 8 * g + 4 * hi;
      iv[g] = *reinterpret_cast<const float4*>(p.inv + n);
      bb[g] = *reinterpret_cast<const float4*>(p.bias + n);
      rr[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (p.res) {
#pragma unroll
      for (int g = 0; g < 4; ++g) rr[g] = *reinterpret_cast<const float4*>(p.res + row + n0 + ct * 32 + 8 * g + 4 * hi);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int n = n0 + ct * 32 + 8 * g + 4 * hi;
      float4 v = make_float4(fmaf(acc[idx][4 * g], iv[g].x, bb[g].x), fmaf(acc[idx][4 * g + 1], iv[g].y, bb[g].y), fmaf(acc[idx][4 * g + 2], iv[g].z, bb[g].z),
                             fmaf(acc[idx][4 * g + 3], iv[g].w, bb[g].w));
      if (p.act == ACT_GELU) v = make_float4(gelu_erf(v.x), gelu_erf(v.y), gelu_erf(v.z), gelu_erf(v.w));
      else if (p.act == ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      v.x += rr[g].x; v.y += rr[g].y; v.z += rr[g].z; v.w += rr[g].w;
      if (ok) *reinterpret_cast<float4*>(p.y + row + n) = v;
    }
  }
}

// ---- resident form
template <int K, bool LN, class G>
__global__ __launch_bounds__(256, 1) void rb_linear_kernel(const RbLinArgs p) {
  constexpr int KC = K / 16;          // k16 chunks
  constexpr int CPT = KC / 4;         // chunks per staging thread (thread = row tid / 4, chunks (tid & 3) + 4 i)
  static_assert(KC % 4 == 0 && KC % RB_D == 0, "K must be a multiple of 64");
  __shared__ __attribute__((aligned(16))) unsigned char As[KC * RB_CHS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / p.bpi, j = blockIdx.x - img * p.bpi;
  const int nrows = min(RB_ROWS, p.tokens - j * RB_ROWS);
  const int m0 = img * p.tokens + j * RB_ROWS;

  RbW<G> W;
  W.init(p.w, p.w_bytes, wave, lane);
  W.prologue();  // the first RB_D steps of the stream fly while the rows are staged

  {  // ---- rows -> (LayerNorm) -> split-f16 fragments
    const int r = tid >> 2, q = tid & 3;
    const float* xr = p.x + (size_t)(m0 + min(r, nrows - 1)) * K;  // rows past the block's end: a valid row, never stored
    float4 v[CPT][4];
#pragma unroll
    for (int i = 0; i < CPT; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[i][e] = *reinterpret_cast<const float4*>(xr + 16 * (q + 4 * i) + 4 * e);
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
      rb_store_chunk(As + (q + 4 * i) * RB_CHS, r, v[i]);
    }
  }
  __syncthreads();

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
        rb_step<G>(acc, W, d, A[d & 1], A[(d + 1) & 1], As + nx * RB_CHS, lane, xr);
      }
    }
    rb_epilogue_store<G>(p, acc, ps * G::COLS, m0, nrows, wave, lane);
  }
}

// ---- streamed form (K a multiple of 64; one pass: N = G::COLS).  Three ring slots of one k64 stage each; the hand-over of stage t + 1 (ds_write + the only barrier of
// the stage) sits in the MIDDLE of stage t, so that the look-ahead fragment reads never wait at a stage boundary: between two barriers a wave reads slots t - 1 (its last
// step) and t, and writes slot t + 1.
template <class G>
__global__ __launch_bounds__(256, 1) void rb_linear_stream_kernel(const RbLinArgs p, int K) {
  constexpr int SLOT = 4 * RB_CHS;
  __shared__ __attribute__((aligned(16))) unsigned char As[3 * SLOT];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.x / p.bpi, j = blockIdx.x - img * p.bpi;
  const int nrows = min(RB_ROWS, p.tokens - j * RB_ROWS);
  const int m0 = img * p.tokens + j * RB_ROWS;
  const int T = K / 64;  // stages

  RbW<G> W;
  W.init(p.w, p.w_bytes, wave, lane);
  W.prologue();

  const int r = tid >> 2, q = tid & 3;
  const float* xr_ = p.x + (size_t)(m0 + min(r, nrows - 1)) * K + 16 * q;  // rows past the block's end: a valid row, never stored
  float4 st[4];
  auto stage_load = [&](int t) {
#pragma unroll
    for (int e = 0; e < 4; ++e) st[e] = *reinterpret_cast<const float4*>(xr_ + (size_t)t * 64 + 4 * e);
  };
  stage_load(0);
  rb_store_chunk(As + q * RB_CHS, r, st);
  stage_load(T > 1 ? 1 : 0);
  __syncthreads();

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
  for (int t = 0; t < T; ++t) {
    const int nxt = cur == 2 ? 0 : cur + 1;
    const unsigned char* base = As + cur * SLOT;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (d == 2) {  // stage t + 1 (in registers since the middle of stage t - 1) -> its slot; the loads of stage t + 2 take the registers over
        rb_store_chunk(As + nxt * SLOT + q * RB_CHS, r, st);
        __syncthreads();
        stage_load(min(t + 2, T - 1));  // past the end: re-reads the last stage, never used
        __builtin_amdgcn_sched_barrier(0);
      }
      rb_step<G>(acc, W, d, A[d & 1], A[(d + 1) & 1], d < 3 ? base + (d + 1) * RB_CHS : As + nxt * SLOT, lane, xr);
    }
    cur = nxt;
  }
  rb_epilogue_store<G>(p, acc, 0, m0, nrows, wave, lane);
}

// ---- launchers
bool rb_linear_supported(int K, int N) {
  if (K == 320 && N % 320 == 0) return true;      // MiT stage 3: q / proj / kv / fc1
  if (K % 64 == 0 && K > 320 && N == 320) return true;  // fc2 (K = 1280)
  return false;
}

void launch_rb_linear(const RbLinArgs& a, int K, hipStream_t s) {
  const int B = a.M / a.tokens;
  const dim3 grid((unsigned)(B * a.bpi)), block(256);
  using G320 = RbGeo<2, true>;
  if (K == 320) {
    if (a.ln_g) hipLaunchKernelGGL((rb_linear_kernel<320, true, G320>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((rb_linear_kernel<320, false, G320>), grid, block, 0, s, a);
  } else {
    hipLaunchKernelGGL((rb_linear_stream_kernel<G320>), grid, block, 0, s, a, K);
  }
}

}  // namespace pf
