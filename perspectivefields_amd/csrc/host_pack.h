// Host-side helpers shared by the engine (engine.hip) and the op-level test / bench entry points (engine_ops.hip): error text, device check, the weight packers
// (conv layout, split planes, fused-MLP / row-block weight streams) and the temporary-upload helper of the op entry points.  Header-only (inline).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pf_hip.h"
#include "pf_kernels.h"

namespace pf_host {
using namespace pf;

extern thread_local std::string g_create_error;  // pf_last_error(NULL): defined in engine.hip

inline int roundup(int v, int m) { return (v + m - 1) / m * m; }

inline std::string fmt(const char* f, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}

inline std::vector<float> pack_conv(const float* w, int Cout, int Cin, int KH, int KW, int CinP, const double* out_scale, int* KWC, int* KWCp) {
  *KWC = KW * CinP;
  *KWCp = roundup(*KWC, 32);
  std::vector<float> o((size_t)Cout * KH * *KWCp, 0.f);
  for (int n = 0; n < Cout; ++n)
    for (int ci = 0; ci < Cin; ++ci)
      for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) {
          double v = w[(((size_t)n * Cin + ci) * KH + ky) * KW + kx];
          if (out_scale) v *= out_scale[n];
          o[((size_t)n * KH + ky) * *KWCp + kx * CinP + ci] = (float)v;
        }
  return o;
}

inline std::vector<float> pack_dw(const float* w, int C, int K) {  // [C][1][K][K] -> [K*K][C]
  std::vector<float> o((size_t)K * K * C);
  for (int c = 0; c < C; ++c)
    for (int k = 0; k < K * K; ++k) o[(size_t)k * C + c] = w[(size_t)c * K * K + k];
  return o;
}

// packed fp32 weights -> 5 bf16 planes [5][n] (igemm_sb_impl.h): exact 3-way truncation split h, m, l (h + m + l == w),
// then round-to-nearest-even bf16(w) and round-to-nearest m (operands of the reduced-precision modes)
inline std::vector<unsigned short> split_bf16x3(const std::vector<float>& w) {
  const size_t n = w.size();
  std::vector<unsigned short> o(5 * n);
  auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
  auto flt = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
  auto rne = [](uint32_t u) { return (uint32_t)((u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u); };
  for (size_t i = 0; i < n; ++i) {
    const float a = w[i];
    const uint32_t hb = bits(a) & 0xffff0000u;
    const float r = a - flt(hb);
    const uint32_t mb = bits(r) & 0xffff0000u;
    const float r2 = r - flt(mb);
    o[i] = (unsigned short)(hb >> 16); o[n + i] = (unsigned short)(mb >> 16); o[2 * n + i] = (unsigned short)(bits(r2) >> 16);
    o[3 * n + i] = (unsigned short)(rne(bits(a)) >> 16);
    o[4 * n + i] = (unsigned short)(rne(bits(r)) >> 16);
  }
  return o;
}

// Split-f16 weight planes (igemm_sb_impl.h, NT_F16X3).  Row n (one output channel, `per_row` packed values) is scaled by
// S_n = 2^e with max|w S_n| in [2^13, 2^14) -- exact, and it keeps the low part wl = fp16(w S - wh) a NORMAL fp16 number
// for every weight down to 2^-16 of the row maximum, and the product plane wh 2^-11 (made on the device) exact down to
// 2^-17 of it.  Planes: [0] wh = fp16_rn(w S), [1] wl = fp16_rn(w S - wh); inv_scale[n] = 1 / S_n undoes the scale in the epilogue.
struct F16Planes { std::vector<unsigned short> planes; std::vector<float> inv_scale; };
inline F16Planes split_f16x2(const std::vector<float>& w, int Cout) {
  const size_t n = w.size(), per_row = n / (size_t)Cout;
  F16Planes o;
  o.planes.resize(2 * n);
  o.inv_scale.resize(Cout);
  auto bits16 = [](_Float16 h) { unsigned short u; std::memcpy(&u, &h, 2); return u; };
  for (int r = 0; r < Cout; ++r) {
    float mx = 0.f;
    for (size_t k = 0; k < per_row; ++k) mx = std::max(mx, std::fabs(w[r * per_row + k]));
    int e = 0;
    if (mx > 0.f && std::isfinite(mx)) { int ex; (void)std::frexp(mx, &ex); e = 14 - ex; }  // mx = m 2^ex, m in [0.5, 1)  ->  mx 2^e in [2^13, 2^14)
    e = std::max(-100, std::min(100, e));
    const float S = std::ldexp(1.0f, e);
    o.inv_scale[r] = std::ldexp(1.0f, -e);
    for (size_t k = 0; k < per_row; ++k) {
      const float ws = w[r * per_row + k] * S;
      const _Float16 hi = (_Float16)ws;
      const _Float16 lo = (_Float16)(ws - (float)hi);
      o.planes[r * per_row + k] = bits16(hi);
      o.planes[n + r * per_row + k] = bits16(lo);
    }
  }
  return o;
}

// LayerNorm folded into the Linear that consumes it (ConvParams::ln), in fp64:
//   Linear(LN(x)) = rstd (x - mean) . (W gamma) + (b + W beta)  ->  W'[n][k] = W[n][k] gamma[k], bias' = b + W beta, colsum[n] = sum_k W'[n][k]
inline void fold_ln_linear(const float* w, const float* b, const float* g, const float* be, int N, int K, std::vector<float>* wf, std::vector<float>* bf, std::vector<float>* cs) {
  wf->resize((size_t)N * K); bf->resize(N); cs->resize(N);
  for (int n = 0; n < N; ++n) {
    double sb = b ? (double)b[n] : 0.0, sc = 0.0;
    for (int k = 0; k < K; ++k) {
      const double wv = w[(size_t)n * K + k];
      const float wg = (float)(wv * (double)g[k]);
      (*wf)[(size_t)n * K + k] = wg;
      sc += (double)wg;  // the sum of the weights the kernel really multiplies
      sb += wv * (double)be[k];
    }
    (*bf)[n] = (float)sb; (*cs)[n] = (float)sc;
  }
}

// Weights of the fused ConvNeXt block MLP (cnx_mlp.hip) in MFMA fragment order, split-f16 scheme (split_f16x2's scaling), fp64 folds:
//   W1'[j][c] = W1[j][c] gamma[c], b1' = b1 + W1 beta, cs1[j] = sum_c W1'[j][c]   (LayerNorm folded, as fold_ln_linear)
//   W2'[n][j] = ls[n] W2[n][j],    b2' = ls[n] b2[n]                                 (layer scale folded, convnext.py:54-55)
// chunk t (32 hidden units): W1 part [s][plane][lane][8] = W1s[32 t + (lane & 31)][(lane >> 5) C/2 + 8 s + e],
//                            W2 part [q][u][plane][lane][8] = W2s[32 q + (lane & 31)][32 t + 16 u + (e & 3) + 8 (e >> 2) + 4 (lane >> 5)]
// tab: inv1[H], cs1[H], b1'[H], inv2[C], b2'[C]
inline void cnx_mlp_pack(const float* w1, const float* b1, const float* g, const float* be, const float* w2, const float* b2, const float* ls, int C,
                  std::vector<unsigned short>* wpk, std::vector<float>* tab) {
  const int H = 4 * C, S1 = C / 16, Q = C / 32, NCH = H / 32;
  std::vector<float> w1f, b1f, cs1;
  fold_ln_linear(w1, b1, g, be, H, C, &w1f, &b1f, &cs1);
  std::vector<float> w2f((size_t)C * H), b2f(C);
  for (int n = 0; n < C; ++n) {
    for (int j = 0; j < H; ++j) w2f[(size_t)n * H + j] = (float)((double)w2[(size_t)n * H + j] * (double)ls[n]);
    b2f[n] = (float)((double)b2[n] * (double)ls[n]);
  }
  const F16Planes p1 = split_f16x2(w1f, H), p2 = split_f16x2(w2f, C);
  const size_t n1 = w1f.size(), n2 = w2f.size();
  const size_t CH1 = (size_t)S1 * 2 * 512, CH2 = (size_t)Q * 2 * 2 * 512, CHUNK = CH1 + CH2;
  wpk->assign((size_t)NCH * CHUNK, 0);
  for (int t = 0; t < NCH; ++t) {
    unsigned short* o = wpk->data() + (size_t)t * CHUNK;
    for (int s = 0; s < S1; ++s)
      for (int pl = 0; pl < 2; ++pl)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int row = 32 * t + (lane & 31), col = (lane >> 5) * (C / 2) + 8 * s + e;
            o[((size_t)(s * 2 + pl) * 64 + lane) * 8 + e] = p1.planes[pl * n1 + (size_t)row * C + col];
          }
    o += CH1;
    for (int q = 0; q < Q; ++q)
      for (int u = 0; u < 2; ++u)
        for (int pl = 0; pl < 2; ++pl)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const int row = 32 * q + (lane & 31), col = 32 * t + 16 * u + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
              o[((size_t)((q * 2 + u) * 2 + pl) * 64 + lane) * 8 + e] = p2.planes[pl * n2 + (size_t)row * H + col];
            }
  }
  tab->resize((size_t)3 * H + 2 * C);
  for (int j = 0; j < H; ++j) { (*tab)[j] = p1.inv_scale[j]; (*tab)[H + j] = cs1[j]; (*tab)[2 * H + j] = b1f[j]; }
  for (int n = 0; n < C; ++n) { (*tab)[3 * H + n] = p2.inv_scale[n]; (*tab)[3 * H + C + n] = b2f[n]; }
}

// Weight stream of a row-block linear layer (rb_common.h): [pass of `cols` output channels][k16 step][column tile][plane hi / lo][lane][8 halfs],
// value = Ws[pass * cols + 32 ct + (lane & 31)][16 step + 8 (lane >> 5) + e] with the per-output-channel power-of-two scale of the split-f16 scheme;
// RB_D = 4 zero steps of padding behind the last pass (the register ring reads ahead).  `w` is [N][K] row-major (K = (ky, kx, ci) for a packed conv).
inline void rb_pack_w(const float* w, int N, int K, int cols, std::vector<unsigned short>* stream, std::vector<float>* inv) {
  std::vector<float> wv(w, w + (size_t)N * K);
  const F16Planes pl = split_f16x2(wv, N);
  const size_t n_all = wv.size();
  const int npass = N / cols, nct = cols / 32, steps = K / 16;
  const size_t step_us = (size_t)nct * 2 * 512;
  stream->assign(((size_t)npass * steps + 4) * step_us, 0);
  for (int ps = 0; ps < npass; ++ps)
    for (int st = 0; st < steps; ++st) {
      unsigned short* o = stream->data() + ((size_t)ps * steps + st) * step_us;
      for (int ct = 0; ct < nct; ++ct)
        for (int p = 0; p < 2; ++p)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e)
              o[((size_t)(ct * 2 + p) * 64 + lane) * 8 + e] = pl.planes[p * n_all + (size_t)(ps * cols + 32 * ct + (lane & 31)) * K + 16 * st + 8 * (lane >> 5) + e];
    }
  *inv = pl.inv_scale;
}

// Weights of the fused MiT block Mlp (mit_mlp.hip), one chunk of mit_mlp_chunk_bytes(C) per 32 hidden units t (layout: the kernel's header):
// LayerNorm (norm2) folded into fc1 as fold_ln_linear, split-f16 planes in MFMA fragment order, depthwise taps [ky * 3 + kx][hidden] + bias.
inline void mit_mlp_pack(const float* w1, const float* b1, const float* g, const float* be, const float* wdw /*[H][1][3][3]*/, const float* bdw, const float* w2, const float* b2,
                  int C, std::vector<unsigned short>* wpk, std::vector<float>* tab2) {
  const int H = 4 * C, S1 = C / 16, Q = C / 32, NCH = H / 32;
  std::vector<float> w1f, b1f, cs1;
  fold_ln_linear(w1, b1, g, be, H, C, &w1f, &b1f, &cs1);
  std::vector<float> w2v(w2, w2 + (size_t)C * H);
  const F16Planes p1 = split_f16x2(w1f, H), p2 = split_f16x2(w2v, C);
  const size_t n1 = w1f.size(), n2 = w2v.size();
  const size_t chunk_us = (size_t)mit_mlp_chunk_bytes(C) / 2, w1_us = (size_t)S1 * 2 * 512, w2_us = (size_t)Q * 2 * 2 * 512;
  wpk->assign((size_t)NCH * chunk_us, 0);
  for (int t = 0; t < NCH; ++t) {
    unsigned short* o = wpk->data() + (size_t)t * chunk_us;
    for (int s = 0; s < S1; ++s)
      for (int pl = 0; pl < 2; ++pl)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e)
            o[((size_t)(s * 2 + pl) * 64 + lane) * 8 + e] = p1.planes[pl * n1 + (size_t)(32 * t + (lane & 31)) * C + (lane >> 5) * (C / 2) + 8 * s + e];
    o += w1_us;
    for (int q = 0; q < Q; ++q)
      for (int u = 0; u < 2; ++u)
        for (int pl = 0; pl < 2; ++pl)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e)
              o[((size_t)((q * 2 + u) * 2 + pl) * 64 + lane) * 8 + e] = p2.planes[pl * n2 + (size_t)(32 * q + (lane & 31)) * H + 32 * t + 16 * u + 8 * (lane >> 5) + e];
    float* tb = reinterpret_cast<float*>(o + w2_us);  // inv1[32], cs1[32], b1[32], taps [9][32], dw bias [32]
    for (int j = 0; j < 32; ++j) {
      const int hd = 32 * t + j;
      tb[j] = p1.inv_scale[hd]; tb[32 + j] = cs1[hd]; tb[64 + j] = b1f[hd];
      for (int k = 0; k < 9; ++k) tb[96 + k * 32 + j] = wdw[(size_t)hd * 9 + k];
      tb[96 + 9 * 32 + j] = bdw[hd];
    }
  }
  tab2->resize((size_t)2 * C);
  for (int n = 0; n < C; ++n) { (*tab2)[n] = p2.inv_scale[n]; (*tab2)[C + n] = b2[n]; }
}

// Pillow's precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR (triangle, support 1) filter, in double, so that
// the integer tables are bit-identical to the ones PIL builds (reference path: perspectivefields.py:45 -> Image.resize).

inline int check_device(int device, std::string* err) {
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) { *err = fmt("no HIP device available (%s); libpf_hip has no CPU fallback", hipGetErrorString(e)); return PF_ERR_DEVICE; }
  if (device < 0 || device >= n) { *err = fmt("device %d out of range (have %d)", device, n); return PF_ERR_ARG; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { *err = "hipGetDeviceProperties failed"; return PF_ERR_DEVICE; }
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { *err = fmt("device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName); return PF_ERR_DEVICE; }
  if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed"; return PF_ERR_DEVICE; }
  return PF_OK;
}

struct TmpDev {  // test-entry-point helper: upload host weights, free on scope exit
  std::vector<void*> p;
  float* up(const float* h, size_t n) {
    if (!h) return nullptr;
    void* d = nullptr;
    if (hipMalloc(&d, n * 4) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    p.push_back(d);
    return static_cast<float*>(d);
  }
  float* up(const std::vector<float>& v) { return up(v.data(), v.size()); }
  unsigned short* up_u16(const std::vector<unsigned short>& v) {
    void* d = nullptr;
    if (hipMalloc(&d, v.size() * 2) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, v.data(), v.size() * 2, hipMemcpyHostToDevice);
    p.push_back(d);
    return static_cast<unsigned short*>(d);
  }
  void sync_free(hipStream_t s) { (void)hipStreamSynchronize(s); for (void* d : p) (void)hipFree(d); p.clear(); }
};

}  // namespace pf_host
