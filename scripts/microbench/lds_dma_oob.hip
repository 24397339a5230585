// What does an out-of-range lane of `buffer_load_dwordx4 ... lds` (LDS-DMA through a buffer resource) write?  256 threads, 16 bytes each, resource of 2048 bytes:
// lanes 128..255 are out of range.  LDS pre-filled with -1.  Prints the first float of lanes 0, 127, 128, 255 and the number of out-of-range floats that are 0 / -1.
// hipcc --offload-arch=gfx950 -O3 lds_dma_oob.hip -o lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* y, int nbytes) {
  __shared__ float sm[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = -1.f;
  __syncthreads();
  const unsigned long long a = (unsigned long long)x;
  i32x4_t r = {__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)), __builtin_amdgcn_readfirstlane(nbytes), 0x00020000};
  int voff = threadIdx.x * 16;
  unsigned ldsb = (unsigned)(unsigned long long)(sm) + 1024u * (threadIdx.x >> 6);
  ldsb = __builtin_amdgcn_readfirstlane(ldsb);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(r), "s"(ldsb) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 256) y[i] = sm[i];
}
int main() {
  float *x, *y, hx[1024], hy[1024];
  for (int i = 0; i < 1024; ++i) hx[i] = 1000.f + i;
  hipMalloc(&x, 4096); hipMalloc(&y, 4096); hipMemcpy(x, hx, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, x, y, 2048);
  hipMemcpy(hy, y, 4096, hipMemcpyDeviceToHost);
  int z = 0, m1 = 0, ok = 0;
  for (int i = 0; i < 512; ++i) ok += hy[i] == hx[i];
  for (int i = 512; i < 1024; ++i) { z += hy[i] == 0.f; m1 += hy[i] == -1.f; }
  printf("in range: %d / 512 floats correct; out of range: %d zero, %d untouched (-1) of 512; lane 0 %.0f lane 127 %.0f lane 128 %.0f lane 255 %.0f\n", ok, z, m1, hy[0], hy[508], hy[512], hy[1020]);
  return 0;
}
