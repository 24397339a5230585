// Measures issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (sizing the depthwise kernels' VALU bound).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int PK>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s) {
  f2 a[8]; f2 b = {s, s * 0.5f}, c = {0.25f, 0.125f};
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x + i, (float)i};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (PK) a[i] = __builtin_elementwise_fma(a[i], b, c);
        else { a[i].x = fmaf(a[i].x, b.x, c.x); }
      }
  }
  float r = 0; for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
  float* d; hipMalloc(&d, 256 * 4096 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pk = 0; pk < 2; ++pk) {
    const int iters = 20000, blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves / SIMD
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
      else    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double insts = (double)blocks * 4 * iters * 64;  // wave-level VALU instructions
      const double flops = insts * 64 * 2 * (pk ? 2 : 1);
      printf("%s: %.3f ms, %.1f Ginst/s (wave64), %.1f TFLOP/s, cycles per inst per SIMD at 2.4 GHz: %.2f\n", pk ? "v_pk_fma_f32" : "v_fma_f32", ms,
             insts / ms / 1e6, flops / ms / 1e9, 2.4e9 * 1024 * (ms * 1e-3) / insts);
    }
  }
  return 0;
}
