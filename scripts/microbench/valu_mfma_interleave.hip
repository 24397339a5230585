// Microbenchmark (round 3): how many VALU instructions hide in the shadow of a v_mfma_f32_32x32x16_f16 when they are INTERLEAVED with the
// MFMAs in program order, as opposed to clustered behind them (valu_mfma_overlap.hip mode 2)?
//   mode 0: NM MFMAs per iteration, nothing else                                  (floor: 32 cycles per MFMA and SIMD at the sustained clock)
//   mode 1: NM*R v_fma_f32 per iteration, nothing else
//   mode 2: NM MFMAs, then NM*R v_fma_f32 (clustered)
//   mode 3: even waves MFMA only, odd waves VALU only (role split over co-resident waves; 2x the waves of mode 2 for the same work per SIMD)
//   mode 4: after EACH MFMA exactly R v_fma_f32, order pinned with asm volatile    (what a hand-scheduled K loop would issue)
//   mode 5: the same from builtins + __builtin_amdgcn_sched_group_barrier          (what igemm_sbh.hip can ask hipcc for)
// R sweeps 1..12; 1, 2, 3 waves per SIMD.  Output: microseconds and 2.4 GHz-cycles per MFMA and SIMD.
// hipcc --offload-arch=gfx950 -O3 valu_mfma_interleave.hip -o valu_mfma_interleave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int NM = 36;

template <int MODE, int R>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (_Float16)(0.001f * (threadIdx.x + e));
    b[e] = (_Float16)(0.002f * (threadIdx.x - e));
  }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = 0.1f * e + threadIdx.x;
  const float c1 = 1.0001f + 1e-7f * threadIdx.x, c2 = 0.5f;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE <= 3) {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
      }
      if (do_v) {
#pragma unroll
        for (int n = 0; n < NM * R; ++n) v[n & 7] = __builtin_fmaf(v[n & 7], c1, c2);
      }
    } else if constexpr (MODE == 4) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
#pragma unroll
        for (int r = 0; r < R; ++r) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(m * R + r) & 7]) : "v"(c1), "v"(c2));
      }
    } else {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; ++r) v[(m * R + r) & 7] = __builtin_fmaf(v[(m * R + r) & 7], c1, c2);
      }
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, R, 0);  // R VALU
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int R>
float run(int blocks, int iters, float* d) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, R>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, R>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a);
  hipEventDestroy(b);
  return ms / 5 * 1000.f;
}

template <int R>
void row(int bpc, int iters, float* d) {
  const int blocks = 256 * bpc;
  const float t0 = run<0, R>(blocks, iters, d), t1 = run<1, R>(blocks, iters, d), t2 = run<2, R>(blocks, iters, d), t3 = run<3, R>(2 * blocks, iters, d), t4 = run<4, R>(blocks, iters, d),
              t5 = run<5, R>(blocks, iters, d);
  const double cyc = 1e-6 * 2.4e9 / (iters * double(NM) * bpc);
  printf("| %d | %2d | %7.1f | %7.1f | %7.1f | %7.1f | %7.1f | %7.1f | %5.1f | %5.1f | %5.1f | %5.1f | %5.1f |\n", bpc, R, t0, t1, t2, t3, t4, t5, t0 * cyc, t2 * cyc, t3 * cyc, t4 * cyc, t5 * cyc);
}

int main() {
  float* d;
  hipMalloc(&d, 4 * 256 * 4096);
  const int iters = 200;
  printf("| waves/SIMD | VALU per MFMA | MFMA only us | VALU only us | clustered us | role split us | interleaved (asm) us | interleaved (sched_group_barrier) us | cyc/MFMA: MFMA only | clustered | role "
         "split | asm | sgb |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n");
  for (int bpc = 1; bpc <= 3; ++bpc) {
    row<1>(bpc, iters, d);
    row<2>(bpc, iters, d);
    row<3>(bpc, iters, d);
    row<4>(bpc, iters, d);
    row<5>(bpc, iters, d);
    row<6>(bpc, iters, d);
    row<8>(bpc, iters, d);
    row<12>(bpc, iters, d);
  }
  return 0;
}
