// Do the matrix cores of gfx950 keep fp16 SUBNORMAL inputs?  (decides whether the split-f16 scheme can carry the low activation plane unscaled and drop the wh 2^-11
// weight operand: scripts/emulate_split.py modes f16x3u / f16x3uf -- kept: same accuracy as today; flushed: outside the tolerances.)
//   hipcc --offload-arch=gfx950 -O2 -o mfma_f16_denorm scripts/microbench/mfma_f16_denorm.hip && ./mfma_f16_denorm
// A[m][k] = a (a subnormal fp16 value), B[k][n] = 1 -> every D[m][n] = 16 a if subnormal inputs are kept, 0 if they are flushed.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, unsigned short abits, unsigned short bbits) {
  const _Float16 a = __builtin_bit_cast(_Float16, abits), b = __builtin_bit_cast(_Float16, bbits);
  f16x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = a; bv[i] = b; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = c[15]; }
  if (threadIdx.x == 63) out[2] = c[7];
}
int main() {
  float* d; hipMalloc(&d, 16);
  struct { const char* name; unsigned short a, b; double expect; } cases[] = {
      {"A = 2^-20 (subnormal), B = 1", 0x0010, 0x3c00, 16.0 * 9.5367431640625e-07},
      {"A = 2^-24 (smallest subnormal), B = 1", 0x0001, 0x3c00, 16.0 * 5.9604644775390625e-08},
      {"A = 1, B = 2^-20 (subnormal)", 0x3c00, 0x0010, 16.0 * 9.5367431640625e-07},
      {"A = 2^-14 (smallest normal), B = 1", 0x0400, 0x3c00, 16.0 * 6.103515625e-05},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c.a, c.b);
    float h[4]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-42s D = %.9g %.9g %.9g  expected %.9g  -> %s\n", c.name, h[0], h[1], h[2], c.expect, h[0] == (float)c.expect ? "KEPT" : (h[0] == 0.f ? "FLUSHED" : "OTHER"));
  }
  return 0;
}
