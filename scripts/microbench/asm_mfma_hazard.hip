// A VALU write of an MFMA's B operand N wait states ahead of the MFMA, in isolation (profiles/r06_asm_mfma_hazard.md): how many states does the matrix core need
// to see the NEW register contents?  Everything between the write and the MFMA sits in ONE asm statement on fixed registers, so the distance is what the source says
// (hipcc pads nothing inside an asm string).
//
//   B fragment v[20:23] of v_mfma_f32_32x32x16_f16 starts as four dwords of (1.0, 1.0); A = all ones; the write under test turns v23 into (2.0, 2.0):
//     kind 0: v_mov_b32 v23, (2.0, 2.0)
//     kind 1: v_fma_mixhi_f16 v23, 3.0f, 1.0, -(1.0 in the high half)        -- the split's instruction (sb_split.h), v23's low half set to 2.0 well ahead
//   pad between the write and the MFMA:
//     0: nothing   1: s_nop 0   2: s_nop 1   3: s_waitcnt lgkmcnt(0) (nothing outstanding: the state that failed in thin128_kernel<false>)   4: v_nop   5: v_nop, v_nop
//     6: s_waitcnt lgkmcnt(0) behind a ds_read_b32 issued just before the write (a wait that really waits: the state the three passing kernels had)
//   D[i][j] = sum_k B[k][j]: 20 with the new v23, 16 (kind 0) / 18 (kind 1) with the old one.  Output per (kind, pad): lanes x registers that are not 20, of all.
//
// hipcc --offload-arch=gfx950 -O2 asm_mfma_hazard.hip -o /tmp/asm_mfma_hazard && /tmp/asm_mfma_hazard
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define PAD0 ""
#define PAD1 "s_nop 0\n\t"
#define PAD2 "s_nop 1\n\t"
#define PAD3 "s_waitcnt lgkmcnt(0)\n\t"
#define PAD4 "v_nop\n\t"
#define PAD5 "v_nop\n\tv_nop\n\t"

#define BODY(WRITE, PRE, PAD)                                                                                                                                   \
  asm volatile("v_mov_b32 v20, %[old]\n\tv_mov_b32 v21, %[old]\n\tv_mov_b32 v22, %[old]\n\tv_mov_b32 v23, %[old]\n\t"                                           \
               "v_mov_b32 v24, %[one]\n\tv_mov_b32 v25, %[one]\n\tv_mov_b32 v26, %[one]\n\tv_mov_b32 v27, %[one]\n\t"                                           \
               "s_nop 7\n\t"                                                                                                                                    \
               "v_fma_mixlo_f16 v28, %[x], 1.0, -%[old] op_sel_hi:[0,0,1]\n\t" /* 2.0 into the low half of a scratch register (kind 1 copies it into v23 below) */ \
               "s_nop 7\n\t" PRE WRITE PAD                                                                                                                       \
               "v_mfma_f32_32x32x16_f16 v[32:47], v[24:27], v[20:23], 0\n\t"                                                                                     \
               "s_nop 15\n\ts_nop 15\n\t"                                                                                                                        \
               "v_mov_b32 %[o0], v32\n\tv_mov_b32 %[o1], v33\n\tv_mov_b32 %[o2], v39\n\tv_mov_b32 %[o3], v47\n\t"                                               \
               "s_waitcnt lgkmcnt(0)"                                                                                                                            \
               : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3)                                                                                 \
               : [old] "v"(old), [one] "v"(one), [nw] "v"(nw), [x] "v"(x), [la] "v"(lds_addr)                                                                    \
               : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", \
                 "v44", "v45", "v46", "v47", "memory")

#define W_MOV "v_mov_b32 v23, %[nw]\n\t"
// kind 1: v23 = (2.0 low, 1.0 high) first -- well ahead --, then the instruction under test writes the HIGH half: 3.0 - 1.0 = 2.0
#define PRE_MIX "v_and_b32 v23, 0xffff0000, v23\n\tv_and_b32 v28, 0xffff, v28\n\tv_or_b32 v23, v23, v28\n\ts_nop 7\n\t"
#define W_MIX "v_fma_mixhi_f16 v23, %[x], 1.0, -%[old] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
#define PRE_LDS "ds_read_b32 v29, %[la]\n\t"
// kind 2 / 3: FOUR writes in a row (the whole fragment, as the split does), the last one adjacent to the pad: D = 32 with the new fragment
#define W_MOV4 "v_mov_b32 v20, %[nw]\n\tv_mov_b32 v21, %[nw]\n\tv_mov_b32 v22, %[nw]\n\tv_mov_b32 v23, %[nw]\n\t"
#define PRE_MIX4 "v_and_b32 v28, 0xffff, v28\n\tv_and_b32 v20, 0xffff0000, v20\n\tv_or_b32 v20, v20, v28\n\tv_mov_b32 v21, v20\n\tv_mov_b32 v22, v20\n\tv_mov_b32 v23, v20\n\ts_nop 7\n\t"
#define W_MIX4 "v_fma_mixhi_f16 v20, %[x], 1.0, -%[old] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 v21, %[x], 1.0, -%[old] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
               "v_fma_mixhi_f16 v22, %[x], 1.0, -%[old] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 v23, %[x], 1.0, -%[old] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"

// kind 4: the neighbourhood of the pair in thin128_kernel<false>: eight ds_read_b128 issued first, a global load and its vmcnt(0) wait (the LDS reads are long complete by then),
// four v_fma_mixlo_f16 + s_nop 0 + four v_fma_mixhi_f16 on v20..v23, `s_waitcnt lgkmcnt(7)` (PADR = the pad in front of it), the MFMA.  D = 32 with the new fragment.
#define BODY_REPLICA(PADR)                                                                                                                                      \
  asm volatile("v_mov_b32 v24, %[one]\n\tv_mov_b32 v25, %[one]\n\tv_mov_b32 v26, %[one]\n\tv_mov_b32 v27, %[one]\n\t"                                           \
               "ds_read_b128 v[48:51], %[la]\n\tds_read_b128 v[52:55], %[la] offset:16\n\tds_read_b128 v[56:59], %[la] offset:32\n\tds_read_b128 v[60:63], %[la] offset:48\n\t" \
               "ds_read_b128 v[64:67], %[la] offset:64\n\tds_read_b128 v[68:71], %[la] offset:80\n\tds_read_b128 v[72:75], %[la] offset:96\n\tds_read_b128 v[76:79], %[la] offset:112\n\t" \
               "global_load_dwordx4 v[80:83], %[gp], off\n\tglobal_load_dwordx4 v[84:87], %[gp], off offset:32\n\t"                                            \
               "s_waitcnt vmcnt(1)\n\t"                                                                                                                        \
               "v_med3_f32 v80, v80, %[lo], %[hi]\n\tv_med3_f32 v81, v81, %[lo], %[hi]\n\tv_med3_f32 v82, v82, %[lo], %[hi]\n\tv_med3_f32 v83, v83, %[lo], %[hi]\n\t" \
               "s_waitcnt vmcnt(0)\n\t"                                                                                                                        \
               "v_med3_f32 v84, v84, %[lo], %[hi]\n\tv_med3_f32 v85, v85, %[lo], %[hi]\n\tv_med3_f32 v86, v86, %[lo], %[hi]\n\tv_med3_f32 v87, v87, %[lo], %[hi]\n\t" \
               "v_cvt_pk_f16_f32 v88, v80, v81\n\tv_cvt_pk_f16_f32 v89, v82, v83\n\tv_cvt_pk_f16_f32 v90, v84, v85\n\tv_cvt_pk_f16_f32 v91, v86, v87\n\t"         \
               /* v20..v23 <- the fp32 values (3.0): what the registers hold BEFORE the split's asm writes them */                                                 \
               "v_mov_b32 v20, v80\n\tv_mov_b32 v21, v82\n\tv_mov_b32 v22, v84\n\tv_mov_b32 v23, v86\n\t"                                                       \
               "v_fma_mixlo_f16 v20, v80, 1.0, -v88 op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 v21, v82, 1.0, -v89 op_sel_hi:[0,0,1]\n\t"                               \
               "v_fma_mixlo_f16 v22, v84, 1.0, -v90 op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 v23, v86, 1.0, -v91 op_sel_hi:[0,0,1]\n\t"                               \
               "s_nop 0\n\t"                                                                                                                                   \
               "v_fma_mixhi_f16 v20, v81, 1.0, -v88 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 v21, v83, 1.0, -v89 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
               "v_fma_mixhi_f16 v22, v85, 1.0, -v90 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 v23, v87, 1.0, -v91 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t" \
               PADR "s_waitcnt lgkmcnt(7)\n\t"                                                                                                                 \
               "v_mfma_f32_32x32x16_f16 v[32:47], v[24:27], v[20:23], 0\n\t"                                                                                     \
               "s_nop 15\n\ts_nop 15\n\t"                                                                                                                        \
               "v_mov_b32 %[o0], v32\n\tv_mov_b32 %[o1], v33\n\tv_mov_b32 %[o2], v39\n\tv_mov_b32 %[o3], v47\n\t"                                               \
               "s_waitcnt lgkmcnt(0)"                                                                                                                            \
               : [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3)                                                                                 \
               : [one] "v"(one), [la] "v"(lds_addr16), [gp] "v"(gp), [lo] "s"(-65504.f), [hi] "v"(65504.f)                                                       \
               : "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46",  \
                 "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69",  \
                 "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "memory")

// x = 2049 + 1/4 ... no: x = 1.0 + 2^-12: hi = fp16(x) = 1.0, lo = fp16(x - 1.0) = 2^-12: B dword = (lo, lo); D = 16 * 2^-12 * ... ; simpler to compare against the padded run
template <int PADR>
__global__ __launch_bounds__(512, 2) void replica_kernel(const float* __restrict__ g, unsigned* __restrict__ bad, float* __restrict__ ref, int iters, int write_ref) {
  __shared__ __attribute__((aligned(16))) unsigned lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = i;
  __syncthreads();
  const unsigned one = 0x3c003c00u;
  const unsigned lds_addr16 = (unsigned)(size_t)(&lds[(threadIdx.x * 4) & 2047]);
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    const float* gp = g + ((size_t)(blockIdx.x * 512 + threadIdx.x) * 16 + (size_t)it * 8192 * 16) % (1u << 22);
    float o0, o1, o2, o3;
    if (PADR == 0) BODY_REPLICA(""); else BODY_REPLICA("s_nop 1\n\t");
    const size_t idx = ((size_t)it * gridDim.x + blockIdx.x) * 512 + threadIdx.x;
    if (write_ref) ref[idx] = o0 + o1 + o2 + o3;
    else wrong += (ref[idx] != o0 + o1 + o2 + o3);
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int KIND, int PAD>
__global__ __launch_bounds__(256) void hazard_kernel(unsigned* __restrict__ bad, int iters) {
  __shared__ unsigned lds[256];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const unsigned old = 0x3c003c00u, one = 0x3c003c00u, nw = 0x40004000u;  // fp16 pairs (1, 1), (1, 1), (2, 2)
  const float x = 3.0f;
  const unsigned lds_addr = (unsigned)(size_t)(&lds[(threadIdx.x * 7) & 255]);
  unsigned wrong = 0;
  for (int it = 0; it < iters; ++it) {
    float o0, o1, o2, o3;
    if (KIND == 0) {
      if (PAD == 0) BODY(W_MOV, "", PAD0); else if (PAD == 1) BODY(W_MOV, "", PAD1); else if (PAD == 2) BODY(W_MOV, "", PAD2); else if (PAD == 3) BODY(W_MOV, "", PAD3);
      else if (PAD == 4) BODY(W_MOV, "", PAD4); else if (PAD == 5) BODY(W_MOV, "", PAD5); else BODY(W_MOV, PRE_LDS, PAD3);
    } else if (KIND == 2) {
      if (PAD == 0) BODY(W_MOV4, "", PAD0); else if (PAD == 1) BODY(W_MOV4, "", PAD1); else if (PAD == 2) BODY(W_MOV4, "", PAD2); else if (PAD == 3) BODY(W_MOV4, "", PAD3);
      else if (PAD == 4) BODY(W_MOV4, "", PAD4); else if (PAD == 5) BODY(W_MOV4, "", PAD5); else BODY(W_MOV4, PRE_LDS, PAD3);
    } else if (KIND == 3) {
      if (PAD == 0) BODY(W_MIX4, PRE_MIX4, PAD0); else if (PAD == 1) BODY(W_MIX4, PRE_MIX4, PAD1); else if (PAD == 2) BODY(W_MIX4, PRE_MIX4, PAD2); else if (PAD == 3) BODY(W_MIX4, PRE_MIX4, PAD3);
      else if (PAD == 4) BODY(W_MIX4, PRE_MIX4, PAD4); else if (PAD == 5) BODY(W_MIX4, PRE_MIX4, PAD5); else BODY(W_MIX4, PRE_MIX4 PRE_LDS, PAD3);
    } else {
      if (PAD == 0) BODY(W_MIX, PRE_MIX, PAD0); else if (PAD == 1) BODY(W_MIX, PRE_MIX, PAD1); else if (PAD == 2) BODY(W_MIX, PRE_MIX, PAD2); else if (PAD == 3) BODY(W_MIX, PRE_MIX, PAD3);
      else if (PAD == 4) BODY(W_MIX, PRE_MIX, PAD4); else if (PAD == 5) BODY(W_MIX, PRE_MIX, PAD5); else BODY(W_MIX, PRE_MIX PRE_LDS, PAD3);
    }
    const float want = KIND >= 2 ? 32.f : 20.f;
    wrong += (o0 != want) + (o1 != want) + (o2 != want) + (o3 != want);
  }
  if (wrong) atomicAdd(bad, wrong);
}

template <int KIND, int PAD>
static void run(unsigned* d_bad, const char* kind, const char* pad) {
  const int blocks = 2048, iters = 16;
  (void)hipMemset(d_bad, 0, 4);
  hipLaunchKernelGGL((hazard_kernel<KIND, PAD>), dim3(blocks), dim3(256), 0, 0, d_bad, iters);
  unsigned bad = 0;
  (void)hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
  printf("%-16s | %-44s | wrong %10u of %u\n", kind, pad, bad, (unsigned)(blocks * 256 * iters * 4));
}

template <int KIND>
static void run_kind(unsigned* d_bad, const char* kind) {
  run<KIND, 0>(d_bad, kind, "0 states (back to back)");
  run<KIND, 1>(d_bad, kind, "1 state: s_nop 0");
  run<KIND, 3>(d_bad, kind, "1 state: s_waitcnt lgkmcnt(0), nothing pending");
  run<KIND, 6>(d_bad, kind, "1 state: s_waitcnt lgkmcnt(0) behind a ds_read");
  run<KIND, 4>(d_bad, kind, "1 state: v_nop");
  run<KIND, 2>(d_bad, kind, "2 states: s_nop 1");
  run<KIND, 5>(d_bad, kind, "2 states: v_nop, v_nop");
}

int main() {
  unsigned* d_bad = nullptr;
  if (hipMalloc(&d_bad, 4) != hipSuccess) { printf("no device\n"); return 1; }
  printf("write under test | between the write and the MFMA reading v[20:23]  | D elements that saw the OLD register\n");
  run_kind<0>(d_bad, "v_mov_b32");
  run_kind<1>(d_bad, "v_fma_mixhi_f16");
  run_kind<2>(d_bad, "4 x v_mov_b32");
  run_kind<3>(d_bad, "4 x v_fma_mixhi");
  {  // the replica: reference = its padded form, then the form that failed
    const int blocks = 512, iters = 8;
    float *g = nullptr, *ref = nullptr;
    std::vector<float> hg(1u << 22);
    for (size_t i = 0; i < hg.size(); ++i) hg[i] = 1.0f + (float)((i * 2654435761u) >> 20) * (1.0f / 8192.0f);   // 1 .. 1.5 with 13 fraction bits: a non-zero low part
    (void)hipMalloc(&g, hg.size() * 4 + 4096); (void)hipMemcpy(g, hg.data(), hg.size() * 4, hipMemcpyHostToDevice);
    (void)hipMalloc(&ref, (size_t)blocks * 512 * iters * 4);
    hipLaunchKernelGGL((replica_kernel<1>), dim3(blocks), dim3(512), 0, 0, g, d_bad, ref, iters, 1);
    for (int padr = 1; padr >= 0; --padr) {
      (void)hipMemset(d_bad, 0, 4);
      if (padr) hipLaunchKernelGGL((replica_kernel<1>), dim3(blocks), dim3(512), 0, 0, g, d_bad, ref, iters, 0);
      else hipLaunchKernelGGL((replica_kernel<0>), dim3(blocks), dim3(512), 0, 0, g, d_bad, ref, iters, 0);
      unsigned bad = 0;
      (void)hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost);
      printf("%-16s | %-44s | differs %8u of %u (lanes x iterations, against the padded run)\n", "replica", padr ? "s_nop 1, s_waitcnt lgkmcnt(7)" : "s_waitcnt lgkmcnt(7) only (the failing form)", bad, (unsigned)(blocks * 512 * iters));
    }
    (void)hipFree(g); (void)hipFree(ref);
  }
  (void)hipFree(d_bad);
  return 0;
}
