// Microbenchmark: do VALU work and MFMA work of DIFFERENT waves on one SIMD overlap on gfx950?  (decision aid for cnx_mlp.hip)
//   mode 0: every wave issues NM MFMAs per iteration (4 independent accumulators)
//   mode 1: every wave issues NV v_fma_f32 per iteration (8 independent chains)
//   mode 2: every wave issues both, MFMA cluster then VALU cluster (what a fused epilogue kernel does)
//   mode 3: even waves MFMA only, odd waves VALU only (perfect role split across co-resident waves), same total work as mode 2 at 2x the waves
// hipcc --offload-arch=gfx950 -O3 valu_mfma_overlap.hip -o valu_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE, int NM, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  f16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
  float v[8];
  for (int e = 0; e < 8; ++e) v[e] = 0.1f * e + threadIdx.x;
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < iters; ++it) {
    if (do_m) {
#pragma unroll
      for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m & 3], 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
      for (int n = 0; n < NV; ++n) v[n & 7] = __builtin_fmaf(v[n & 7], 1.0001f, 0.5f);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  for (int e = 0; e < 8; ++e) s += v[e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> float run(int blocks, int iters, float* d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<MODE, 36, 448>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<MODE, 36, 448>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5 * 1000.f;
}
int main() {
  float* d; hipMalloc(&d, 4 * 256 * 4096);
  const int iters = 200;
  for (int bpc = 1; bpc <= 3; ++bpc) {
    const int blocks = 256 * bpc;
    const float t0 = run<0>(blocks, iters, d), t1 = run<1>(blocks, iters, d), t2 = run<2>(blocks, iters, d), t3 = run<3>(2 * blocks, iters, d);
    // per-SIMD cycles per iteration at 1 wave per SIMD per block
    printf("blocks/CU %d: MFMA-only %.1f us | VALU-only %.1f us | both in every wave %.1f us (sum %.1f, max %.1f) | role split over 2x waves %.1f us\n", bpc, t0, t1, t2, t0 + t1,
           t0 > t1 ? t0 : t1, t3);
    printf("   MFMA-only: %.1f cycles per MFMA per SIMD at 2.4 GHz; VALU-only: %.2f cycles per v_fma\n", t0 * 1e-6 * 2.4e9 / (iters * 36.0 * bpc), t1 * 1e-6 * 2.4e9 / (iters * 448.0 * bpc));
  }
  return 0;
}
