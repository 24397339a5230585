// Packed-fp32 operand forms beside other kernels (profiles/r04_dw7_packed.md): each form runs a dependent chain of packed operations per thread and is compared, bit for
// bit, with the same arithmetic in scalar v_fma_f32 / v_mul_f32 / v_add_f32.  Built as a shared library (hipcc -shared -fPIC --offload-arch=gfx950) and driven by
// pk_opsel_beside.py, which keeps THIS LIBRARY's forward running on another stream in a background thread.
//   form 0: v_pk_fma_f32 c, a, w, c  op_sel:[0,1,0] op_sel_hi:[1,1,1]   (both lanes x w.hi, w = src1)       <- dw7_pk.hip's first form
//   form 1: v_pk_fma_f32 c, w, a, c  op_sel:[1,0,0] op_sel_hi:[1,1,1]   (both lanes x w.hi, w = src0)       <- shipped form
//   form 2: v_pk_fma_f32 c, a, w, c  op_sel_hi:[1,0,1]                  (both lanes x w.lo, w = src1)
//   form 3: v_pk_fma_f32 c, a, w, c  op_sel:[0,1,0] op_sel_hi:[1,0,1]   (lanes swapped: lo x w.hi, hi x w.lo)
//   form 4: v_pk_mul_f32 d, a, w     op_sel:[0,1] op_sel_hi:[1,1]; c += d (scalar adds)
//   form 5: v_pk_add_f32 c, c, w     op_sel:[0,1] op_sel_hi:[1,0]       (the compiler's horizontal-reduction form: c.lo += w.hi, c.hi += w.lo)
//   form 6: v_pk_fma_f32 c, a, w, c2 op_sel:[0,0,1] op_sel_hi:[1,1,0]   (src2 halves swapped), c2 = previous c
//   form 7: v_pk_add_f32 d, x, x     op_sel:[0,1] op_sel_hi:[1,0]       (r05: the SAME register pair twice -- exactly what hipcc emitted 95 times for the horizontal sums of
//                                                                        the LayerNorm code in cnx_mlp / mit_mlp / rb_*: d.lo = x.lo + x.hi, d.hi = x.hi + x.lo)
#include <hip/hip_runtime.h>

typedef float pk2 __attribute__((ext_vector_type(2)));

static __device__ __forceinline__ float sfma(float a, float b, float c) { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); return c; }   // scalar on purpose: the
static __device__ __forceinline__ float smul(float a, float b) { float d; asm volatile("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }            // compiler packs vector
static __device__ __forceinline__ float sadd(float a, float b) { float d; asm volatile("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }            // expressions by itself

template <int FORM>
__global__ __launch_bounds__(256) void pk_form_kernel(const pk2* __restrict__ a_in, const pk2* __restrict__ w_in, pk2* __restrict__ out_pk, pk2* __restrict__ out_ref, int iters) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  pk2 c = a_in[t];
  float r0 = c.x, r1 = c.y;
  for (int i = 0; i < iters; ++i) {
    const pk2 a = a_in[(t + 64 * (i + 1)) & 0xFFFFF];
    const pk2 w = w_in[(t + 64 * i) & 0xFFFFF];
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)  // a dense run of the form under test, like the 7 taps of a kernel row
    if (FORM == 0)      { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(a), "v"(w)); r0 = sfma(a.x, w.y, r0); r1 = sfma(a.y, w.y, r1); }
    else if (FORM == 1) { asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(a), "v"(w)); r0 = sfma(a.x, w.y, r0); r1 = sfma(a.y, w.y, r1); }
    else if (FORM == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(c) : "v"(a), "v"(w)); r0 = sfma(a.x, w.x, r0); r1 = sfma(a.y, w.x, r1); }
    else if (FORM == 3) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(c) : "v"(a), "v"(w)); r0 = sfma(a.x, w.y, r0); r1 = sfma(a.y, w.x, r1); }
    else if (FORM == 4) { pk2 d; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(d) : "v"(a), "v"(w)); c.x = sadd(c.x, d.x); c.y = sadd(c.y, d.y); r0 = sadd(r0, smul(a.x, w.y)); r1 = sadd(r1, smul(a.y, w.y)); }
    else if (FORM == 5) { asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(c) : "v"(w)); r0 = sadd(r0, w.y); r1 = sadd(r1, w.x); }
    else if (FORM == 7) { pk2 d; asm volatile("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(d) : "v"(c)); c.x = smul(d.x, w.x); c.y = sadd(smul(d.y, w.y), a.y);
                          const float t0 = sadd(r0, r1), t1 = sadd(r1, r0); r0 = smul(t0, w.x); r1 = sadd(smul(t1, w.y), a.y); }
    else                { pk2 c2 = c; asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(c) : "v"(a), "v"(w), "v"(c2)); const float n0 = sfma(a.x, w.x, r1), n1 = sfma(a.y, w.y, r0); r0 = n0; r1 = n1; }
  }
  out_pk[t] = c;
  out_ref[t] = pk2{r0, r1};
}

extern "C" int pk_form_launch(int form, const void* a, const void* w, void* out_pk, void* out_ref, int blocks, int iters, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const pk2* A = static_cast<const pk2*>(a); const pk2* W = static_cast<const pk2*>(w); pk2* P = static_cast<pk2*>(out_pk); pk2* R = static_cast<pk2*>(out_ref);
  switch (form) {
    case 0: hipLaunchKernelGGL(pk_form_kernel<0>, dim3(blocks), dim3(256), 0, s, A, W, P, R, iters); break;
    case 1: hipLaunchKernelGGL(pk_form_kernel<1>, dim3(blocks), dim3(256), 0, s, A, W, P, R, iters); break;
    case 2: hipLaunchKernelGGL(pk_form_kernel<2>, dim3(blocks), dim3(256), 0, s, A, W, P, R, iters); break;
    case 3: hipLaunchKernelGGL(pk_form_kernel<3>, dim3(blocks), dim3(256), 0, s, A, W, P, R, iters); break;
    case 4: hipLaunchKernelGGL(pk_form_kernel<4>, dim3(blocks), dim3(256), 0, s, A, W, P, R, iters); break;
    case 5: hipLaunchKernelGGL(pk_form_kernel<5>, dim3(blocks), dim3(256), 0, s, A, W, P, R, iters); break;
    case 6: hipLaunchKernelGGL(pk_form_kernel<6>, dim3(blocks), dim3(256), 0, s, A, W, P, R, iters); break;
    case 7: hipLaunchKernelGGL(pk_form_kernel<7>, dim3(blocks), dim3(256), 0, s, A, W, P, R, iters); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
