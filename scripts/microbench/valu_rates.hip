// Issue rate of the VALU instructions the Winograd transform / fp16 split are made of, ONE wave per SIMD (the regime of wino256x64w4 / wino256x64c):
// s_memtime around 256 x 32 independent instructions of one kind (8 independent destination sets), lane 0 of wave 0 reports cycles per instruction.
// hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ __launch_bounds__(256, 1) void rate_kernel(float* out, unsigned long long* cyc) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = 1.0001f, b1 = 0.9999f;
  unsigned h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0, h5 = 0, h6 = 0, h7 = 0;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, q = {b0, b1};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 256; ++it) {
    if (KIND == 0) {  // v_fma_f32
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));)
    } else if (KIND == 1) {  // v_cvt_pk_f16_f32
      REP8(asm volatile("v_cvt_pk_f16_f32 %0, %4, %5\n v_cvt_pk_f16_f32 %1, %5, %6\n v_cvt_pk_f16_f32 %2, %6, %7\n v_cvt_pk_f16_f32 %3, %7, %4\n"
                        : "=v"(h0), "=v"(h1), "=v"(h2), "=v"(h3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    } else if (KIND == 2) {  // v_fma_mixlo_f16 (independent destinations)
      REP8(asm volatile("v_fma_mixlo_f16 %0, %4, 1.0, -%8 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %1, %5, 1.0, -%8 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %2, %6, 1.0, -%8 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %3, %7, 1.0, -%8 op_sel_hi:[0,0,1]\n"
                        : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(h4));)
    } else if (KIND == 3) {  // v_pk_add_f32
      REP8(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
    } else if (KIND == 4) {  // the split as the kernels issue it: cvt_pk -> mixlo -> mixhi, four chains interleaved (12 instructions)
      REP8(asm volatile("v_cvt_pk_f16_f32 %0, %8, %9\n v_cvt_pk_f16_f32 %1, %10, %11\n v_cvt_pk_f16_f32 %2, %12, %13\n v_cvt_pk_f16_f32 %3, %14, %15\n"
                        "v_fma_mixlo_f16 %4, %8, 1.0, -%0 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %5, %10, 1.0, -%1 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %6, %12, 1.0, -%2 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %7, %14, 1.0, -%3 op_sel_hi:[0,0,1]\n"
                        "v_fma_mixhi_f16 %4, %9, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %5, %11, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %6, %13, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %7, %15, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
                        : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
    } else if (KIND == 5) {  // v_med3_f32
      REP8(asm volatile("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b0), "v"(b1));)
    } else if (KIND == 6) {  // v_pk_fma_f32
      REP8(asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(q));)
    } else {  // v_cvt_f16_f32 (scalar conversion) + v_pack_b32_f16
      REP8(asm volatile("v_cvt_f16_f32 %0, %4\n v_cvt_f16_f32 %1, %5\n v_cvt_f16_f32 %2, %6\n v_cvt_f16_f32 %3, %7\n" : "=v"(h0), "=v"(h1), "=v"(h2), "=v"(h3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));)
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + (float)(h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5 ^ h6 ^ h7);
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 64); hipMemset(cyc, 0, 64);
  const char* names[8] = {"v_fma_f32", "v_cvt_pk_f16_f32", "v_fma_mixlo_f16", "v_pk_add_f32", "split chain (4 cvt_pk + 4 mixlo + 4 mixhi)", "v_med3_f32", "v_pk_fma_f32", "v_cvt_f16_f32"};
  const int per_iter[8] = {32, 32, 32, 32, 96, 32, 32, 32};
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(256), 0, 0, out, cyc); hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(256), 0, 0, out, cyc);
    hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(256), 0, 0, out, cyc); hipLaunchKernelGGL(rate_kernel<3>, dim3(256), dim3(256), 0, 0, out, cyc);
    hipLaunchKernelGGL(rate_kernel<4>, dim3(256), dim3(256), 0, 0, out, cyc); hipLaunchKernelGGL(rate_kernel<5>, dim3(256), dim3(256), 0, 0, out, cyc);
    hipLaunchKernelGGL(rate_kernel<6>, dim3(256), dim3(256), 0, 0, out, cyc); hipLaunchKernelGGL(rate_kernel<7>, dim3(256), dim3(256), 0, 0, out, cyc);
    hipDeviceSynchronize();
  }
  unsigned long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  for (int k = 0; k < 8; ++k) printf("%-48s %7.2f s_memtime counts per instruction (one wave per SIMD)\n", names[k], (double)h[k] / (256.0 * per_iter[k]));
  return 0;
}
