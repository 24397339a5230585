// Does the VALU work of the Winograd transform hide behind MFMAs when a SECOND wave shares the SIMD?  (profiles/r05_winograd.md: with one wave per SIMD the times add,
// however evenly the instructions are interleaved -- wino256x64d.)  One block per CU, W waves per SIMD; every wave runs the 6-slot window of wino256x64d
//   {MFMA, 4 v_pk_fma} {MFMA, 3 v_cvt_pk, v_fma_mixlo} {MFMA, mixhi, mixlo, mixlo} {MFMA, v_cvt_pk, mixhi, mixlo} {MFMA, mixhi, mixhi} {MFMA, 4 v_pk_fma}
// (MODE 0), the same window without the VALU instructions (MODE 1) or without the MFMAs (MODE 2); lane 0 of wave 0 reports s_memtime counts per window.
// hipcc --offload-arch=gfx950 -O3 slot_overlap.hip -o slot_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define MF(acc) "v_mfma_f32_32x32x16_f16 %" #acc ", %2, %3, %" #acc "\n"
#define PK4 "v_pk_fma_f32 %4, %8, %4, %5\n v_pk_fma_f32 %5, %8, %5, %6\n v_pk_fma_f32 %6, %8, %6, %7\n v_pk_fma_f32 %7, %8, %7, %4\n"
#define S1 "v_cvt_pk_f16_f32 %9, %13, %14\n v_cvt_pk_f16_f32 %10, %15, %16\n v_cvt_pk_f16_f32 %11, %13, %15\n v_fma_mixlo_f16 %17, %13, 1.0, -%9 op_sel_hi:[0,0,1]\n"
#define S2 "v_fma_mixhi_f16 %17, %14, 1.0, -%9 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %18, %15, 1.0, -%10 op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %19, %13, 1.0, -%11 op_sel_hi:[0,0,1]\n"
#define S3 "v_cvt_pk_f16_f32 %12, %14, %16\n v_fma_mixhi_f16 %18, %16, 1.0, -%10 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_fma_mixlo_f16 %20, %14, 1.0, -%12 op_sel_hi:[0,0,1]\n"
#define S4 "v_fma_mixhi_f16 %19, %15, 1.0, -%11 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n v_fma_mixhi_f16 %20, %16, 1.0, -%12 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n"
template <int MODE, int NT>
__global__ __launch_bounds__(NT, 1) void k(float* out, unsigned long long* cyc, int iters) {
  f16v a0, a1, a2, a3;
  for (int e = 0; e < 16; ++e) { a0[e] = 0.f; a1[e] = 0.f; a2[e] = 0.f; a3[e] = 0.f; }
  u4 A = {threadIdx.x, 1, 2, 3}, B = {4, 5, 6, threadIdx.x};
  f2 p0 = {1.f, 2.f}, p1 = {3.f, 4.f}, p2 = {5.f, 6.f}, p3 = {7.f, 8.f}, q = {0.5f, 0.25f};
  unsigned h0 = 0, h1 = 0, h2 = 0, h3 = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
  float x0 = threadIdx.x, x1 = 1.5f, x2 = 2.5f, x3 = 3.5f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0)
      asm volatile(MF(0) PK4 MF(1) S1 MF(0) S2 MF(1) S3 MF(0) S4 MF(1) PK4
                   : "+a"(a0), "+a"(a1) : "v"(A), "v"(B), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(q), "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(l0), "v"(l1), "v"(l2), "v"(l3));
    else if (MODE == 1)
      asm volatile(MF(0) MF(1) MF(0) MF(1) MF(0) MF(1)
                   : "+a"(a0), "+a"(a1) : "v"(A), "v"(B), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(q), "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(l0), "v"(l1), "v"(l2), "v"(l3));
    else if (MODE == 3)   // MODE 0 with FOUR accumulators in rotation (two windows per iteration): a dependent MFMA is three MFMAs away instead of one
      asm volatile(MF(0) PK4 MF(1) S1 MF(21) S2 MF(22) S3 MF(0) S4 MF(1) PK4 MF(21) PK4 MF(22) S1 MF(0) S2 MF(1) S3 MF(21) S4 MF(22) PK4
                   : "+a"(a0), "+a"(a1) : "v"(A), "v"(B), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(q), "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(l0), "v"(l1), "v"(l2), "v"(l3), "a"(a2), "a"(a3));
    else if (MODE == 4)   // MODE 1 with four accumulators
      asm volatile(MF(0) MF(1) MF(21) MF(22) MF(0) MF(1) MF(21) MF(22) MF(0) MF(1) MF(21) MF(22)
                   : "+a"(a0), "+a"(a1) : "v"(A), "v"(B), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(q), "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(l0), "v"(l1), "v"(l2), "v"(l3), "a"(a2), "a"(a3));
    else
      asm volatile(PK4 S1 S2 S3 S4 PK4
                   : "+a"(a0), "+a"(a1) : "v"(A), "v"(B), "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(q), "v"(h0), "v"(h1), "v"(h2), "v"(h3), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(l0), "v"(l1), "v"(l2), "v"(l3));
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int e = 0; e < 16; ++e) s += a0[e] + a1[e] + a2[e] + a3[e];
  out[blockIdx.x * NT + threadIdx.x] = s + p0.x + p1.y;
  if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE, int NT>
void run(float* out, unsigned long long* cyc, double* counts, double* ns) {   // per window: wave 0's s_memtime counts; whole-launch time / iterations (all waves done)
  const int iters = 20000, win = (MODE == 3 || MODE == 4) ? 2 : 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
  }
  unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  *counts = (double)h / iters / win; *ns = best * 1e6 / iters / win;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 64);
  const char* mode[5] = {"6 MFMA + 20 VALU (window of wino256x64d), 2 acc", "6 MFMA only, 2 accumulators", "20 VALU only", "6 MFMA + 20 VALU, 4 accumulators in rotation", "6 MFMA only, 4 accumulators"};
  double c[5][3], n[5][3];
#define RUN3(M) run<M, 256>(out, cyc, &c[M][0], &n[M][0]); run<M, 512>(out, cyc, &c[M][1], &n[M][1]); run<M, 768>(out, cyc, &c[M][2], &n[M][2]);
  RUN3(0) RUN3(1) RUN3(2) RUN3(3) RUN3(4)
  printf("per window (6 MFMA slots) and wave: s_memtime counts of wave 0 | ns from the whole launch; [ns per window of SIMD throughput = ns / waves per SIMD]\n");
  for (int m = 0; m < 5; ++m)
    printf("%-50s 1 w/SIMD %6.1f cnt %6.1f ns | 2 w/SIMD %6.1f cnt %6.1f ns [%5.1f] | 3 w/SIMD %6.1f cnt %6.1f ns [%5.1f]\n", mode[m], c[m][0], n[m][0], c[m][1], n[m][1], n[m][1] / 2, c[m][2], n[m][2], n[m][2] / 3);
  return 0;
}
