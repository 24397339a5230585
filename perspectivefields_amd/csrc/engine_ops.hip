// Op-level entry points of libpf_hip.so (pf_op_*: one kernel family each, host weights uploaded per call) -- what tests/test_gpu_ops.py and the tuning scripts call.
// Not part of the inference path (the engine, engine.hip, never goes through here).
#include "host_pack.h"

using namespace pf;
using namespace pf_host;

extern "C" {

int pf_op_num_conv_tiles(void) { return conv_num_tiles(); }
const char* pf_op_conv_tile_name(int id) { return conv_tile_name(id); }

int pf_op_conv2d(int device, const float* x, const float* x2, int B, int H, int W, int C1, int C2, const float* hw, const float* hb,
                 int Cout, int KH, int KW, int stride, int pad, int act, const float* res1, const float* res2, int post_relu,
                 int nchw_out, int tile_id, float* y, const uint16_t* x_planes, long x_plane_elems, const uint16_t* x2_planes,
                 long x2_plane_elems, uint16_t* y_planes, long y_plane_elems, int precision_flags, void* stream) {
  const int precision = precision_flags & 15;
  const bool allow_splitk = !(precision_flags & 16);
  if (precision != PF_PRECISION_FP32 && precision != PF_PRECISION_FP32_BF16X6) { g_create_error = "pf_op_conv2d: precision must be 0 (split-f16) or 3 (exact bf16 split)"; return PF_ERR_ARG; }
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  const int Cin = C1 + C2;
  if (Cin % 4 != 0 || (C2 > 0 && C1 % 32 != 0)) { g_create_error = "pf_op_conv2d: Cin must be a multiple of 4 (pad on the host), C1 of 32 when concatenating"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  ConvParams p;
  std::vector<float> packed = pack_conv(hw, Cout, Cin, KH, KW, Cin, nullptr, &p.KWC, &p.KWCp);
  p.g[0].w = tmp.up(packed);
  std::vector<unsigned short> sb;
  if (Cin % 32 == 0 || Cin == 4) {
    sb = split_bf16x3(packed); p.g[0].w_sb = tmp.up_u16(sb);
    const F16Planes f = split_f16x2(packed, Cout);
    p.g[0].w_h16 = tmp.up_u16(f.planes); p.g[0].w_h16_inv_scale = tmp.up(f.inv_scale);
  }
  if (KH == 3 && KW == 3 && stride == 1 && pad == 1 && C2 == 0 && Cin % 16 == 0 && Cout % 64 == 0 && p.KWCp == p.KWC) {  // Winograd form (tiles "wino256x64d" / "wino256x64c")
    std::vector<unsigned short> planes;
    std::vector<float> inv;
    wino_pack_weights(packed.data(), Cout, Cin, p.KWCp, &planes, &inv);
    p.g[0].w_wino = tmp.up_u16(planes); p.g[0].w_wino_inv = tmp.up(inv);
  }
  p.g[0].bias = tmp.up(hb, Cout);
  p.g[0].x = x; p.g[0].x2 = x2; p.g[0].res1 = res1; p.g[0].res2 = res2; p.g[0].y = y;
  p.g[0].x_sb = x_planes; p.g[0].x2_sb = x2_planes; p.g[0].y_sb = y_planes;
  p.x_sb_plane = (size_t)x_plane_elems; p.x2_sb_plane = (size_t)x2_plane_elems; p.y_sb_plane = (size_t)y_plane_elems;
  p.B = B; p.H = H; p.W = W; p.C1 = C1; p.C2 = C2;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.Cout = Cout; p.act = act; p.post_relu = post_relu; p.nchw_out = nchw_out;
  p.nterms = precision == PF_PRECISION_FP32_BF16X6 ? 6 : NT_F16X3;
  // split-plane INPUT: the plane format (bit 0 of x_plane_elems, sb_split.h) fixes the scheme: fp16 planes <-> split-f16, bf16 planes <-> bf16 schemes
  if (x_planes) p.nterms = (x_plane_elems & 1) ? NT_F16X3 : (p.nterms == NT_F16X3 ? 6 : p.nterms);
  p.finish();
  if ((!x && !x_planes) || (!y && !y_planes) || (C2 > 0 && !x2 && !x2_planes)) { g_create_error = "pf_op_conv2d: missing input or output"; return PF_ERR_ARG; }
  // an explicit tile that cannot read / write split planes is an error; with fp32 operands an unusable tile id falls back to the cost model
  if (tile_id >= 0 && !conv_tile_usable(p, tile_id) && (x_planes || y_planes)) { g_create_error = "pf_op_conv2d: tile config cannot run this operand format"; return PF_ERR_ARG; }
  {  // split-K by the engine's rule (scratch for the partial sums from a temporary allocation)
    const int S = allow_splitk ? conv_splitk_factor(p) : 1;
    if (S > 1) {
      void* d = nullptr;
      if (hipMalloc(&d, (size_t)S * p.M * p.Cout * 4) == hipSuccess) { tmp.p.push_back(d); p.g[0].partial = static_cast<float*>(d); p.splitk = S; }
    }
  }
  launch_conv_tile(p, tile_id, s);
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_linear_ln(int device, const float* x, long rows, int K, const float* hw, const float* hb, const float* hgamma, const float* hbeta, float eps, int N,
                    int act, const float* res1, int tile_id, float* y, int precision, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (K % 32 != 0 || N % 4 != 0 || !x || !y || !hw || !hgamma || !hbeta || rows <= 0) { g_create_error = "pf_op_linear_ln: K must be a multiple of 32, N of 4"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  ConvParams p;
  std::vector<float> wf, bf, cs;
  fold_ln_linear(hw, hb, hgamma, hbeta, N, K, &wf, &bf, &cs);
  std::vector<float> packed = pack_conv(wf.data(), N, K, 1, 1, K, nullptr, &p.KWC, &p.KWCp);
  p.g[0].w = tmp.up(packed);
  std::vector<unsigned short> sb = split_bf16x3(packed);
  p.g[0].w_sb = tmp.up_u16(sb);
  const F16Planes f = split_f16x2(packed, N);
  p.g[0].w_h16 = tmp.up_u16(f.planes); p.g[0].w_h16_inv_scale = tmp.up(f.inv_scale);
  p.g[0].bias = tmp.up(bf); p.g[0].ln_colsum = tmp.up(cs);
  p.g[0].x = x; p.g[0].res1 = res1; p.g[0].y = y;
  p.B = 1; p.H = (int)rows; p.W = 1; p.C1 = K; p.C2 = 0; p.KH = p.KW = 1; p.stride = 1; p.pad = 0;
  p.Cout = N; p.act = act; p.post_relu = 0; p.nchw_out = 0;
  p.nterms = precision == PF_PRECISION_FP32_BF16X6 ? 6 : NT_F16X3;
  p.ln = 1; p.ln_eps = eps;
  p.finish();
  if (tile_id >= 0 && !conv_tile_usable(p, tile_id)) { g_create_error = "pf_op_linear_ln: tile config cannot run the fused LayerNorm form"; return PF_ERR_ARG; }
  launch_conv_tile(p, tile_id, s);
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_rb_linear(int device, const float* x, long rows, int tokens, int K, const float* hw, const float* hb, const float* hgamma, const float* hbeta, float eps, int N,
                    int act, const float* res, float* y, int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!rb_linear_supported(K, N) || !x || !y || !hw || !hb || rows <= 0 || tokens <= 0 || rows % tokens != 0 || (hgamma && K != 320) || (K != 320 && act != ACT_NONE) || (act != ACT_NONE && act != ACT_GELU)) {
    g_create_error = "pf_op_rb_linear: (K, N) must be (320, multiple of 320) or (multiple of 256 above 320, 320); rows a multiple of tokens; LayerNorm / GELU only with K = 320";
    return PF_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  std::vector<unsigned short> st;
  std::vector<float> inv;
  rb_pack_w(hw, N, K, 320, &st, &inv);
  RbLinArgs a;
  a.x = x; a.ln_g = hgamma ? tmp.up(std::vector<float>(hgamma, hgamma + K)) : nullptr; a.ln_b = hgamma ? tmp.up(std::vector<float>(hbeta, hbeta + K)) : nullptr; a.ln_eps = eps;
  a.w = tmp.up_u16(st); a.w_bytes = st.size() * 2; a.inv = tmp.up(inv); a.bias = tmp.up(std::vector<float>(hb, hb + N)); a.res = res; a.y = y;
  a.M = (int)rows; a.tokens = tokens; a.bpi = (tokens + 63) / 64; a.N = N; a.act = act;
  launch_rb_linear(a, K, s);
  if (getenv("PF_RB_STAMPS") && !hgamma) {  // timing aid: the s_memtime stamps of block 17's four waves of one more launch, to stderr
    unsigned long long* ds = nullptr;
    if (hipMalloc(&ds, 4 * 64 * 8) == hipSuccess) {
      (void)hipMemsetAsync(ds, 0, 4 * 64 * 8, s);
      RbLinArgs b = a; b.stamps = ds; b.res = nullptr;
      launch_rb_linear(b, K, s);
      std::vector<unsigned long long> hs(4 * 64);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(hs.data(), ds, 4 * 64 * 8, hipMemcpyDeviceToHost);
      for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "rb stamps K=%d N=%d wave %d:", K, N, w);
        for (int i = 1; i < 64; ++i) if (hs[w * 64 + i]) fprintf(stderr, " [%d]%lld", i, (long long)(hs[w * 64 + i] - hs[w * 64]));
        fprintf(stderr, "\n");
      }
      (void)hipFree(ds);
    }
  }
  if (iters > 0 && ms_out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch_rb_linear(a, K, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms_out = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_rb_proj_fc1(int device, const float* attn, float* x, int B, int tokens, int C, const float* proj_w, const float* proj_b, const float* ln2_g, const float* ln2_b,
                      float eps, const float* fc1_w, const float* fc1_b, float* hidden, int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (C != 320 || !attn || !x || !hidden || !proj_w || !proj_b || !ln2_g || !ln2_b || !fc1_w || !fc1_b || B <= 0 || tokens <= 0) {
    g_create_error = "pf_op_rb_proj_fc1: C must be 320; all operands required";
    return PF_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  std::vector<unsigned short> st1, st2;
  std::vector<float> inv1, inv2;
  rb_pack_w(proj_w, C, C, 320, &st1, &inv1);
  rb_pack_w(fc1_w, 4 * C, C, 320, &st2, &inv2);
  st1.resize(st1.size() - (size_t)4 * 10 * 2 * 512);
  st1.insert(st1.end(), st2.begin(), st2.end());
  RbProjFc1Args a;
  a.attn = attn; a.x = x; a.w = tmp.up_u16(st1); a.w_bytes = st1.size() * 2; a.proj_inv = tmp.up(inv1); a.proj_bias = tmp.up(proj_b, C);
  a.ln2_g = tmp.up(ln2_g, C); a.ln2_b = tmp.up(ln2_b, C); a.ln2_eps = eps; a.fc1_inv = tmp.up(inv2); a.fc1_bias = tmp.up(fc1_b, 4 * C); a.hidden = hidden;
  a.B = B; a.tokens = tokens; a.bpi = (tokens + 63) / 64;
  launch_rb_proj_fc1(a, C, s);
  if (iters > 0 && ms_out) {  // timing loop (x keeps accumulating: values are meaningless afterwards)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch_rb_proj_fc1(a, C, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms_out = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_mit_attn64(int device, const float* x, const float* kv, float* y, int B, int N, int M, const float* ln_g, const float* ln_b, float eps, const float* q_w, const float* q_b,
                     const float* p_w, const float* p_b, int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!x || !kv || !y || !ln_g || !ln_b || !q_w || !q_b || !p_w || !p_b || B <= 0 || N <= 0 || !mit_attn64_supported(64, 1, M)) {
    g_create_error = "pf_op_mit_attn64: all operands required; 1 <= kv rows <= 128";
    return PF_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  std::vector<unsigned short> wfr;
  std::vector<float> tab;
  attn64_pack(ln_g, ln_b, q_w, q_b, p_w, p_b, &wfr, &tab);
  MitAttn64Args a;
  a.x = x; a.kv = kv; a.y = y; a.wfr = tmp.up_u16(wfr); a.tab = tmp.up(tab); a.B = B; a.N = N; a.M = M; a.ln_eps = eps;
  int cus = 256;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
  launch_mit_attn64(a, cus, s);
  if (iters > 0 && ms_out) {  // timing loop (with y aliasing x the rows keep accumulating: values are meaningless afterwards)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch_mit_attn64(a, cus, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms_out = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_stem7x7(int device, const float* x, float* y, int B, int H, int W, int stride, const float* w, const float* bias, int relu, const float* ln_g, const float* ln_b, float eps,
                  int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!x || !y || !w || B <= 0 || H <= 0 || W <= 0 || !stem7x7_supported(3, 64, 7, stride, 3) || ((ln_g == nullptr) != (ln_b == nullptr))) {
    g_create_error = "pf_op_stem7x7: x, y, weight required; stride 2 or 4; LayerNorm gamma and beta together";
    return PF_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  std::vector<unsigned short> wfr;
  std::vector<float> tab;
  stem7x7_pack(w, nullptr, bias, ln_g, ln_b, &wfr, &tab);
  Stem7Args a;
  a.x = x; a.y = y; a.wfr = tmp.up_u16(wfr); a.tab = tmp.up(tab); a.B = B; a.H = H; a.W = W; a.stride = stride;
  a.Ho = (H + 6 - 7) / stride + 1; a.Wo = (W + 6 - 7) / stride + 1; a.relu = relu; a.ln = ln_g ? 1 : 0; a.ln_eps = eps;
  int cus = 256;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
  launch_stem7x7(a, cus, s);
  if (iters > 0 && ms_out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch_stem7x7(a, cus, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms_out = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_thin128(int device, const float* x, long rows, const float* w, const float* bias, const float* res, float* y, int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!x || !y || !w || rows <= 0) { g_create_error = "pf_op_thin128: x, y, weight required"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  std::vector<unsigned short> wfr;
  std::vector<float> tab;
  thin128_pack(w, bias, &wfr, &tab);
  ThinLinArgs a;
  a.x = x; a.res = res; a.y = y; a.wfr = tmp.up_u16(wfr); a.tab = tmp.up(tab); a.M = rows;
  int cus = 256;
  { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
  launch_thin128(a, cus, s);
  if (iters > 0 && ms_out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch_thin128(a, cus, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms_out = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_rb_srkv(int device, const float* x, int B, int Hr, int Wr, int C, const float* ln1_g, const float* ln1_b, float eps1, const float* sr_w, const float* sr_b,
                  const float* srn_g, const float* srn_b, float eps2, const float* kv_w, const float* kv_b, float* kv, int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!rb_srkv_supported(C, 2) || !x || !kv || !ln1_g || !ln1_b || !sr_w || !sr_b || !srn_g || !srn_b || !kv_w || !kv_b || B <= 0 || Hr <= 0 || Wr <= 0) {
    g_create_error = "pf_op_rb_srkv: C must be 320 (2 x 2 spatial reduction); all weights required";
    return PF_ERR_ARG;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  int kwc = 0, kwcp = 0;
  const std::vector<float> srp = pack_conv(sr_w, C, C, 2, 2, C, nullptr, &kwc, &kwcp);
  std::vector<unsigned short> st1, st2;
  std::vector<float> inv1, inv2;
  rb_pack_w(srp.data(), C, 2 * kwcp, 320, &st1, &inv1);
  rb_pack_w(kv_w, 2 * C, C, 320, &st2, &inv2);
  st1.resize(st1.size() - (size_t)4 * 10 * 2 * 512);
  st1.insert(st1.end(), st2.begin(), st2.end());
  RbSrKvArgs a;
  a.x = x; a.ln1_g = tmp.up(ln1_g, C); a.ln1_b = tmp.up(ln1_b, C); a.ln1_eps = eps1;
  a.w = tmp.up_u16(st1); a.w_bytes = st1.size() * 2; a.sr_inv = tmp.up(inv1); a.sr_bias = tmp.up(sr_b, C);
  a.srn_g = tmp.up(srn_g, C); a.srn_b = tmp.up(srn_b, C); a.srn_eps = eps2; a.kv_inv = tmp.up(inv2); a.kv_bias = tmp.up(kv_b, 2 * C);
  a.kv = kv; a.B = B; a.Hr = Hr; a.Wr = Wr; a.bpi = (Hr * Wr + 31) / 32;
  launch_rb_srkv(a, C, s);
  if (iters > 0 && ms_out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) launch_rb_srkv(a, C, s);
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, e0, e1);
    *ms_out = t / iters;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_mit_mlp(int device, const float* x, float* y, int B, int Hs, int Ws, int C, const float* w1, const float* b1, const float* lng, const float* lnb, float eps,
                  const float* wdw, const float* bdw, const float* w2, const float* b2, int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!mit_mlp_supported(C) || !x || !y || x == y || B <= 0) { g_create_error = "pf_op_mit_mlp: C must be 64 or 128, x and y different buffers"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  std::vector<unsigned short> wpk;
  std::vector<float> tab2;
  mit_mlp_pack(w1, b1, lng, lnb, wdw, bdw, w2, b2, C, &wpk, &tab2);
  const unsigned short* dw = tmp.up_u16(wpk);
  const float* dt = tmp.up(tab2);
  launch_mit_mlp(x, y, dw, dt, B, Hs, Ws, C, eps, s);
  if (iters > 0 && ms_out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) launch_mit_mlp(x, y, dw, dt, B, Hs, Ws, C, eps, s);
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, a, b);
    *ms_out = t / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_cnx_mlp(int device, const float* d, float* y, long rows, int C, const float* w1, const float* b1, const float* lng, const float* lnb, float eps,
                  const float* w2, const float* b2, const float* ls, int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!cnx_mlp_supported(C) || !d || !y || rows <= 0) { g_create_error = "pf_op_cnx_mlp: C must be 96 or 192"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  std::vector<unsigned short> wpk;
  std::vector<float> tab;
  cnx_mlp_pack(w1, b1, lng, lnb, w2, b2, ls, C, &wpk, &tab);
  const unsigned short* dw = tmp.up_u16(wpk);
  const float* dt = tmp.up(tab);
  launch_cnx_mlp(d, y, dw, dt, rows, C, eps, s);
  if (iters > 0 && ms_out) {  // timing loop (y keeps accumulating: values are meaningless afterwards)
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) launch_cnx_mlp(d, y, dw, dt, rows, C, eps, s);
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, a, b);
    *ms_out = t / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_conv2d_bench(int device, int B, int H, int W, int Cin, int Cout, int K, int stride, int pad, int tile_id, int iters, int fmt_prec, float* ms_out) {
  const int fmt = fmt_prec & 15, precision = fmt_prec >> 4;  // low 4 bits: operand format, upper bits: PF_PRECISION_* of the split tiles
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (Cin % 4 != 0 || iters <= 0 || !ms_out) { g_create_error = "pf_op_conv2d_bench: bad argument"; return PF_ERR_ARG; }
  ConvParams p;
  p.KWC = K * Cin; p.KWCp = roundup(p.KWC, 32);
  p.B = B; p.H = H; p.W = W; p.C1 = Cin; p.C2 = 0; p.KH = K; p.KW = K; p.stride = stride; p.pad = pad;
  p.Cout = Cout; p.act = ACT_RELU; p.post_relu = 0; p.nchw_out = 0;
  p.nterms = precision == PF_PRECISION_FP32_BF16X6 ? 6 : NT_F16X3;
  p.finish();
  const size_t nx = (size_t)B * H * W * Cin, nw = (size_t)Cout * K * p.KWCp, ny = (size_t)p.M * Cout;
  float *dx = nullptr, *dw = nullptr, *dy = nullptr, *db = nullptr;
  unsigned short *dsb = nullptr, *dxs = nullptr, *dys = nullptr, *dh16 = nullptr;
  float* dinv = nullptr;
  TmpDev wino_tmp;                    // Winograd weights (freed below)
  unsigned short* dwino = nullptr;
  float* dwino_inv = nullptr;
  if (hipMalloc(&dx, nx * 4) != hipSuccess || hipMalloc(&dw, nw * 4) != hipSuccess || hipMalloc(&dy, ny * 4) != hipSuccess ||
      hipMalloc(&db, (size_t)Cout * 4) != hipSuccess || hipMalloc(&dsb, nw * 10) != hipSuccess || hipMalloc(&dh16, nw * 4) != hipSuccess ||
      hipMalloc(&dinv, (size_t)Cout * 4) != hipSuccess ||
      (fmt >= 1 && hipMalloc(&dxs, nx * 6) != hipSuccess) || (fmt >= 2 && hipMalloc(&dys, ny * 6) != hipSuccess)) { g_create_error = "pf_op_conv2d_bench: hipMalloc failed"; return PF_ERR_DEVICE; }
  {  // uniform [-1,1) data (never zero-fill a bench: DVFS gives zeros a higher clock); activations filled on the device
    std::vector<float> hw(nw), hb(Cout);
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hw) v = rnd() * 0.05f;
    for (auto& v : hb) v = rnd();
    launch_fill_random(dx, (long)nx, 777u, 1.0f, nullptr);
    (void)hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), (size_t)Cout * 4, hipMemcpyHostToDevice);
    const std::vector<unsigned short> sb = split_bf16x3(hw);
    (void)hipMemcpy(dsb, sb.data(), nw * 10, hipMemcpyHostToDevice);
    const F16Planes f = split_f16x2(hw, Cout);
    (void)hipMemcpy(dh16, f.planes.data(), nw * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dinv, f.inv_scale.data(), (size_t)Cout * 4, hipMemcpyHostToDevice);
    if (K == 3 && stride == 1 && pad == 1 && Cin % 16 == 0 && Cout % 64 == 0 && p.KWCp == p.KWC) {  // Winograd form (tiles "wino256x64d" / "wino256x64c")
      std::vector<unsigned short> planes;
      std::vector<float> inv;
      wino_pack_weights(hw.data(), Cout, Cin, p.KWCp, &planes, &inv);
      wino_tmp.sync_free(nullptr);
      dwino = wino_tmp.up_u16(planes); dwino_inv = wino_tmp.up(inv);
    }
  }
  p.g[0].x = dx; p.g[0].w = dw; p.g[0].bias = db; p.g[0].y = dy;
  if (Cin % 32 == 0 || Cin == 4) { p.g[0].w_sb = dsb; p.g[0].w_h16 = dh16; p.g[0].w_h16_inv_scale = dinv; }
  p.g[0].w_wino = dwino; p.g[0].w_wino_inv = dwino_inv;
  // fmt 1: A operand as split planes (fp32 copy withheld); fmt 2: split planes in and out
  const size_t fbit = p.nterms == NT_F16X3 ? SB_FMT_F16 : 0;  // plane format of the scheme under test (nx, ny are multiples of 4)
  if (fmt >= 1) { launch_split_planes(dx, dxs, nx | fbit, (long)nx, nullptr); p.g[0].x_sb = dxs; p.x_sb_plane = nx | fbit; p.g[0].x = nullptr; }
  if (fmt >= 2) { p.g[0].y_sb = dys; p.y_sb_plane = ny | fbit; p.g[0].y = nullptr; }
  if (!conv_tile_usable(p, tile_id) && tile_id >= 0) { *ms_out = -1.f; (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dy); (void)hipFree(db); (void)hipFree(dsb); (void)hipFree(dxs); (void)hipFree(dys); (void)hipFree(dh16); (void)hipFree(dinv); wino_tmp.sync_free(nullptr); return PF_OK; }
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  launch_conv_tile(p, tile_id, nullptr);
  launch_conv_tile(p, tile_id, nullptr);
  (void)hipEventRecord(a, nullptr);
  for (int i = 0; i < iters; ++i) launch_conv_tile(p, tile_id, nullptr);
  (void)hipEventRecord(b, nullptr);
  (void)hipEventSynchronize(b);
  float t = 0.f;
  (void)hipEventElapsedTime(&t, a, b);
  *ms_out = t / iters;
  if (getenv("PF_WINO_STAMPS") && strncmp(conv_tile_name(tile_id), "wino", 4) == 0) {  // timing aid: s_memtime stamps of block 17's waves of one more launch, to stderr
    unsigned long long* ds = nullptr;
    if (hipMalloc(&ds, 8 * 128 * 8) == hipSuccess) {
      (void)hipMemset(ds, 0, 8 * 128 * 8);
      p.stamps = ds;
      launch_conv_tile(p, tile_id, nullptr);
      p.stamps = nullptr;
      std::vector<unsigned long long> hs(8 * 128);
      (void)hipMemcpy(hs.data(), ds, hs.size() * 8, hipMemcpyDeviceToHost);
      (void)hipFree(ds);
      for (int w = 0; w < 8; ++w) {
        const unsigned long long t0 = hs[w * 128];
        if (!t0) continue;
        fprintf(stderr, "%s stamps %dx%d Cin=%d wave %d:", conv_tile_name(tile_id), H, W, Cin, w);
        for (int i = 1; i < 128; ++i) if (hs[w * 128 + i]) fprintf(stderr, " [%d]%llu", i, hs[w * 128 + i] - t0);
        fprintf(stderr, "\n");
      }
    }
  }
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  (void)hipFree(dx); (void)hipFree(dw); (void)hipFree(dy); (void)hipFree(db); (void)hipFree(dsb); (void)hipFree(dxs); (void)hipFree(dys); (void)hipFree(dh16); (void)hipFree(dinv);
  wino_tmp.sync_free(nullptr);
  return rc;
}

int pf_op_dwconv3x3_bench(int device, int variant, int B, int H, int W, int C, int iters, float* ms_out) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (C % 128 != 0 || iters <= 0 || !ms_out) { g_create_error = "pf_op_dwconv3x3_bench: bad argument"; return PF_ERR_ARG; }
  const size_t n = (size_t)B * H * W * C;
  float *dx = nullptr, *dy = nullptr, *dw = nullptr, *db = nullptr;
  if (hipMalloc(&dx, n * 4) != hipSuccess || hipMalloc(&dy, n * 4) != hipSuccess || hipMalloc(&dw, (size_t)9 * C * 4) != hipSuccess ||
      hipMalloc(&db, (size_t)C * 4) != hipSuccess) { g_create_error = "pf_op_dwconv3x3_bench: hipMalloc failed"; return PF_ERR_DEVICE; }
  {
    std::vector<float> hx(n), hw((size_t)9 * C), hb(C);
    uint32_t st = 777u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return (float)((st >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd() * 0.3f;
    for (auto& v : hb) v = rnd() * 0.1f;
    (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
  }
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  launch_dwconv3x3_gelu_variant(variant, dx, dw, db, dy, B, H, W, C, nullptr);
  (void)hipEventRecord(a, nullptr);
  for (int i = 0; i < iters; ++i) launch_dwconv3x3_gelu_variant(variant, dx, dw, db, dy, B, H, W, C, nullptr);
  (void)hipEventRecord(b, nullptr);
  (void)hipEventSynchronize(b);
  float t = 0.f;
  (void)hipEventElapsedTime(&t, a, b);
  *ms_out = t / iters;
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dw); (void)hipFree(db);
  return rc;
}

int pf_op_layernorm(int device, const float* x, const float* hg, const float* hbeta, float* y, long rows, int C, float eps, uint16_t* y_planes, long plane_elems, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  launch_layernorm(x, tmp.up(hg, C), tmp.up(hbeta, C), y, rows, C, eps, s, y_planes, (size_t)plane_elems);
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_dwconv3x3_gelu(int device, const float* x, const float* hw, const float* hb, float* y, int B, int H, int W, int C, uint16_t* y_planes, long plane_elems, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (C % 128 != 0) { g_create_error = "pf_op_dwconv3x3_gelu: C must be a multiple of 128"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  launch_dwconv3x3_gelu(x, tmp.up(pack_dw(hw, C, 3)), tmp.up(hb, C), y, B, H, W, C, s, y_planes, (size_t)plane_elems);
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_dwconv3x3_gelu_cfg(int device, const float* x, const float* hw, const float* hb, float* y, int B, int H, int W, int C, uint16_t* y_planes, long plane_elems, int variant, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (C % 128 != 0) { g_create_error = "pf_op_dwconv3x3_gelu_cfg: C must be a multiple of 128"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  launch_dwconv3x3_gelu_variant(variant, x, tmp.up(pack_dw(hw, C, 3)), tmp.up(hb, C), y, B, H, W, C, s, y_planes, (size_t)plane_elems);
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_dwconv7x7(int device, const float* x, const float* hw, const float* hb, float* y, int B, int H, int W, int C, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (C % 96 != 0) { g_create_error = "pf_op_dwconv7x7: C must be a multiple of 96"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  launch_dwconv7x7(x, tmp.up(pack_dw(hw, C, 7)), tmp.up(hb, C), y, B, H, W, C, s);
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

int pf_op_dwconv7x7_cfg(int device, const float* x, const float* hw, const float* hb, float* y, int B, int H, int W, int C, int variant, int nc, int nb, int th, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (C % 96 != 0 && C % 32 != 0) { g_create_error = "pf_op_dwconv7x7_cfg: C must be a multiple of 32"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  TmpDev tmp;
  launch_dwconv7x7_cfg(variant, nc, nb, th, x, tmp.up(pack_dw(hw, C, 7)), tmp.up(hb, C), y, B, H, W, C, s);
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  tmp.sync_free(s);
  return rc;
}

// Yardstick beside the depthwise kernels (variant 100): a plain 16-byte-per-lane grid-stride copy of the same bytes -- what this memory system gives ANY launch that
// reads n floats and writes n floats (launch + first byte + drain included), the ceiling a stand-alone HBM-bound kernel of that size can be priced against.
__global__ __launch_bounds__(256) void copy_f4_kernel(const float4* __restrict__ x, float4* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) y[i] = x[i];
}
// PF_DW7_BENCH_COLD=K (K >= 2): the timed launches cycle through K input / output buffer pairs, so that a launch reads an input nobody has touched for K - 1 launches
// (K x 2 x bytes > 512 MB: not in the 256 MiB Infinity Cache, nor in L2) -- "cold"; default: ONE pair, re-read and re-written by every launch -- "warm" (L2 / Infinity
// Cache resident for the 20 - 160 MB maps of the ConvNeXt stages, which is also their state inside the forward: the previous kernel has just written them).
int pf_op_dwconv7x7_bench(int device, int variant, int nc, int nb, int th, int B, int H, int W, int C, int iters, float* ms_out) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (C % 32 != 0 || iters <= 0 || !ms_out) { g_create_error = "pf_op_dwconv7x7_bench: bad argument"; return PF_ERR_ARG; }
  const size_t n = (size_t)B * H * W * C;
  int K = 1;
  if (const char* e = getenv("PF_DW7_BENCH_COLD")) K = std::max(1, std::min(64, atoi(e)));
  std::vector<float*> dx(K, nullptr), dy(K, nullptr);
  float *dw = nullptr, *db = nullptr;
  bool ok = hipMalloc(&dw, (size_t)49 * C * 4) == hipSuccess && hipMalloc(&db, (size_t)C * 4) == hipSuccess;
  for (int k = 0; k < K && ok; ++k) ok = hipMalloc(&dx[k], n * 4) == hipSuccess && hipMalloc(&dy[k], n * 4) == hipSuccess;
  if (!ok) { g_create_error = "pf_op_dwconv7x7_bench: hipMalloc failed"; return PF_ERR_DEVICE; }
  for (int k = 0; k < K; ++k) launch_fill_random(dx[k], (long)n, 777u + k, 1.0f, nullptr);
  launch_fill_random(dw, (long)49 * C, 778u, 0.15f, nullptr);
  launch_fill_random(db, (long)C, 779u, 0.1f, nullptr);
  auto launch = [&](int k) {
    if (variant == 100) hipLaunchKernelGGL(copy_f4_kernel, dim3(256 * 8), dim3(256), 0, nullptr, reinterpret_cast<const float4*>(dx[k]), reinterpret_cast<float4*>(dy[k]), (long)(n / 4));
    else launch_dwconv7x7_cfg(variant, nc, nb, th, dx[k], dw, db, dy[k], B, H, W, C, nullptr);
  };
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int k = 0; k < K; ++k) launch(k);
  (void)hipEventRecord(a, nullptr);
  for (int i = 0; i < iters; ++i) launch(i % K);
  (void)hipEventRecord(b, nullptr);
  (void)hipEventSynchronize(b);
  float t = 0.f;
  (void)hipEventElapsedTime(&t, a, b);
  *ms_out = t / iters;
  rc = hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  for (int k = 0; k < K; ++k) { (void)hipFree(dx[k]); (void)hipFree(dy[k]); }
  (void)hipFree(dw); (void)hipFree(db);
  return rc;
}

int pf_op_sr_attention(int device, const float* q, const float* kv, float* out, int B, int N, int M, int heads, uint16_t* out_planes, long plane_elems, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (M <= 0 || M > 128) { g_create_error = "pf_op_sr_attention: kv length must be in 1..128"; return PF_ERR_ARG; }
  launch_sr_attention(q, kv, out, B, N, M, heads, static_cast<hipStream_t>(stream), out_planes, (size_t)plane_elems);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
}

int pf_op_sr_attention_variant(int device, int variant, const float* q, const float* kv, float* out, int B, int N, int M, int heads, int iters, float* ms_out, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (M <= 0 || M > 128 || (variant != 0 && variant != 1)) { g_create_error = "pf_op_sr_attention_variant: kv length must be in 1..128, variant 0 or 1"; return PF_ERR_ARG; }
  hipStream_t s = static_cast<hipStream_t>(stream);
  launch_sr_attention_variant(variant, q, kv, out, B, N, M, heads, s);
  if (iters > 0 && ms_out) {  // timing loop on the caller's data
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    (void)hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) launch_sr_attention_variant(variant, q, kv, out, B, N, M, heads, s);
    (void)hipEventRecord(b, s);
    (void)hipEventSynchronize(b);
    float t = 0.f;
    (void)hipEventElapsedTime(&t, a, b);
    *ms_out = t / iters;
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  }
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
}

int pf_op_upsample2x(int device, const float* x, float* y, int B, int H, int W, int C, uint16_t* y_planes, long plane_elems, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  launch_upsample2x(x, y, B, H, W, C, static_cast<hipStream_t>(stream), y_planes, (size_t)plane_elems);
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
}

int pf_op_split_bf16(int device, const float* x, long n, uint16_t* planes, long plane_elems, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!x || !planes || n <= 0 || (n & 3) || (plane_elems & ~1L) < n) { g_create_error = "pf_op_split_bf16: n must be a positive multiple of 4 and plane_elems >= n"; return PF_ERR_ARG; }
  launch_split_planes(x, planes, (size_t)plane_elems, n, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
}

int pf_op_merge_bf16(int device, const uint16_t* planes, long plane_elems, long n, float* y, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!y || !planes || n <= 0 || (n & 3) || (plane_elems & ~1L) < n) { g_create_error = "pf_op_merge_bf16: n must be a positive multiple of 4 and plane_elems >= n"; return PF_ERR_ARG; }
  launch_merge_planes(planes, (size_t)plane_elems, y, n, static_cast<hipStream_t>(stream));
  return hipGetLastError() == hipSuccess ? PF_OK : PF_ERR_DEVICE;
}

}  // extern "C"
