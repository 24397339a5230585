// Internal kernel-launcher interface of libpf_hip.so (gfx950 / CDNA4 only).
// All activations are NHWC in HBM, fp32 or (inputs of the split-bf16 GEMMs) three exact bf16 planes (sb_split.h); "tokens (B,N,C)" of the reference's MiT
// blocks are the same memory as NHWC maps, so none of the reference's ~200
// NCHW<->NLC round trips (mix_transformers.py:117-118,246,458,504-506) exist here.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pf {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2 };
#if defined(__HIPCC__)
// saturation watch of a producer (ConvParams::sat): one v_max3 / v_max / compare per 4 outputs, an atomic only when the window is left.  NaN-aware: fmaxf drops a
// NaN operand, so the sum of the four values is tested too (NaN if any of them is NaN, or inf - inf) -- this is the form of every epilogue that is not at its
// register cap (Winograd, row-block layers, fused block MLPs): a Winograd layer whose clamp-free split (wino.hip split2_f16_nc) met a value beyond its window
// produces inf / NaN outputs, and those are counted HERE, at that layer's own output
__device__ __forceinline__ void sat_watch4(unsigned* sat, float limit, float a, float b, float c, float d) {
  const float mx = fmaxf(fmaxf(fabsf(a), fabsf(b)), fmaxf(fabsf(c), fabsf(d)));
  const float s = (a + b) + (c + d);
  if (!(mx <= limit) || s != s) atomicAdd(sat, 1u);
}
// the same with ONE running maximum per thread (two v_max3_f32 per 4 outputs, no temporaries: the epilogues of the GEMM tiles sit at their register caps) ...
__device__ __forceinline__ float sat_acc4(float amax, float a, float b, float c, float d) {
  return __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(amax, __builtin_fabsf(a)), __builtin_fabsf(b)), __builtin_fmaxf(__builtin_fabsf(c), __builtin_fabsf(d)));
}
// ... and one compare at the kernel's end.  SATURATION ONLY: a NaN output is dropped by fmaxf (+-inf is counted).  A NaN can only reach these tiles from a producer
// that was itself counted (an inf beyond every window, or the Winograd epilogue's NaN test above); non-finite values are otherwise the debug forward's business
__device__ __forceinline__ void sat_flush(unsigned* sat, float limit, float amax) {
  if (sat && amax > limit) atomicAdd(sat, 1u);
}
#endif
// Packed fp32 (v_pk_*_f32) whose LOW lane reads the HIGH half of src1 -- hipcc's horizontal reductions `v_pk_add_f32 d, x, x op_sel:[0,1] op_sel_hi:[1,0]`, packed scalar
// FMAs -- was exact alone and wrong in lanes 48..63 on MI355X while this library's kernels ran on another stream (profiles/r04_dw7_packed.md,
// profiles/r05_pk_opsel_beside.txt), which is what every forward with the deferred ParamNet branch is.  tests/test_host_logic.py scans the built library and allows NO such
// form anywhere.  Two remedies: translation units whose kernels hipcc packed that way (cnx_mlp / mit_mlp / rb_gemm / rb_chain) are compiled without the feature
// (build.py NO_PK_F32_FLAGS: every function of the unit, so inlining stays legal -- a per-kernel attribute left the device helpers outlined and their arrays in
// scratch); a single stand-alone kernel without helpers may carry this attribute instead.
#if defined(__HIP_DEVICE_COMPILE__)
#define PF_NO_PK_F32 __attribute__((target("no-packed-fp32-ops")))
#else
#define PF_NO_PK_F32  // host pass: the x86 target has no such feature
#endif
static constexpr int NT_F16X3 = 23;  // ConvParams::nterms value of the split-f16 scheme
// Split-plane tensors: every `plane` stride argument is (elements between consecutive planes, always even) | format bit:
// 0 = three exact bf16 planes (x == h + m + l), SB_FMT_F16 = the two fp16 planes of the split-f16 scheme (sb_split.h)
static constexpr size_t SB_FMT_F16 = 1;

// Implicit-GEMM convolution / GEMM on v_mfma_f32_32x32x2_f32.
//   y[m][n] = post( act( sum_k A[m][k] * Wp[n][k] + bias[n] ) + res1[m][n] + res2[m][n] )
// m = (b, oy, ox) output pixel, k = (ky, kx, ci) with ci fastest, A gathered on the fly
// from one NHWC tensor or from the channel-concat of two (x: C1 channels, x2: C2).
struct ConvPtrs {
  const float* x = nullptr;
  const float* x2 = nullptr;        // nullptr unless channel-concat input
  const float* w = nullptr;         // packed [Cout][KH][KWCp], KWCp = roundup(KW*Cin, 32), zero padded
  const unsigned short* w_sb = nullptr;  // same weights as 5 bf16 planes [5][Cout][KH][KWCp]: exact split h, m, l (h + m + l == w), then
                                         // round-to-nearest bf16(w) and round-to-nearest m for the reduced-precision modes
  // split-f16 scheme (sb_split.h NT_F16X3): weights scaled per output channel by a power of two (row maximum in
  // [2^13, 2^14)) as two fp16 planes [2][Cout][KH][KWCp]: wh = fp16(w S), wl = fp16(w S - wh); w_h16_inv_scale = 1 / S
  const unsigned short* w_h16 = nullptr;
  const float* w_h16_inv_scale = nullptr;  // [Cout]: applied to the accumulators before the bias
  // Winograd F(2x2, 3x3) form of a 3x3 / stride-1 conv (wino.hip): U = G g G^T, per-channel power-of-two scale, two fp16 planes in MFMA fragment order
  const unsigned short* w_wino = nullptr;
  const float* w_wino_inv = nullptr;       // [Cout]
  const float* bias = nullptr;      // [Cout] or nullptr
  const float* bias_tab = nullptr;  // [9][Cout]: bias per 3x3 border case (folded Linear->conv), overrides bias
  const float* res1 = nullptr;      // [M][Cout] or nullptr (may alias y)
  const float* res2 = nullptr;      // [M][Cout] or nullptr
  float* y = nullptr;               // [M][ldy], or nullptr when only the split planes are wanted
  // Fused regression prediction head (Cout == 32, tiles with BN == 32 only): instead of storing the 32-channel map, the
  // epilogue applies the 1x1 head to every pixel -- kind 1: linear_pred_gravity (32 -> 2) + F.normalize (gravity_head.py:117,
  // 190-193), kind 2: linear_pred_latitude (32 -> 1) + clamp(-1, 1) (latitude_head.py:118,189-192) -- and writes the NCHW
  // API output head_out plus its components of the ParamNet input head_pn ([M] float4: g0, g1, lat, 0; may be nullptr)
  // Fused LayerNorm of the INPUT rows (ConvParams::ln, 1x1 layers whose K loop covers the whole row): the weights carry the LayerNorm
  // gamma (W'[n][k] = W[n][k] gamma[k]), bias = b + W beta, and ln_colsum[n] = sum_k W'[n][k]; the kernel contracts the raw rows,
  // accumulates their mean / variance while staging them and applies  y = rstd (acc - mean colsum) + bias  in the epilogue
  const float* ln_colsum = nullptr; // [Cout]
  float* partial = nullptr;         // split-K: [splitk][M][ldy] raw partial sums (fp32 scratch owned by the caller)
  int head_kind = 0;
  const float* head_w = nullptr;    // [nout][32]
  const float* head_b = nullptr;    // [nout]
  float* head_out = nullptr;        // [B][nout][Ho*Wo]
  float* head_pn = nullptr;
  // split-bf16 activation format (sb_split.h): plane 0 of the same NHWC tensors as three exact bf16 planes.  When x_sb
  // is set the split-bf16 kernel copies its A operand instead of splitting it (x / x2 may then be nullptr).
  const unsigned short* x_sb = nullptr;
  const unsigned short* x2_sb = nullptr;
  unsigned short* y_sb = nullptr;   // additional (or only) output in split planes
};
struct ConvParams {
  ConvPtrs g[2];   // grouped launch: `groups` problems of identical shape (the two decoder heads) in one grid
  int groups = 1;
  int B, H, W, C1, C2, Cin;
  int KH, KW, stride, pad;
  int Ho, Wo, Cout;
  int KWC, KWCp;
  int M;
  int act;        // Act applied to (acc + bias)
  int post_relu;  // relu after the residual adds
  int ldy;        // row stride of y / res1 / res2 in floats (normally Cout)
  int nchw_out;   // 1: store y as [B][Cout][Ho*Wo] (API-visible logits), residuals unsupported
  int splitk = 1; // > 1: K is contracted in `splitk` slices by separate blocks (linear split tiles only) into g[].partial, then reduced with
                  // the epilogue (scale, bias, activation, res1, post_relu) by splitk_reduce_kernel -- deterministic (fixed summation order)
  int ups = 0;    // 1: x is stored at half resolution [B][H/2][W/2][C1]; the conv runs on its bilinear x2 up-sampling, interpolated
                  // while the input halo is staged (3x3 halo tiles of the split-f16 scheme only; x2, if any, is at full resolution)
  int ln = 0;     // 1: LayerNorm over the Cin input channels fused into this 1x1 layer (ConvPtrs::ln_colsum; linear split tiles, fp32 input,
                  // no concat, no split-K): mix_transformers.py:200 (norm2 -> fc1), :123-126 (sr norm -> kv), convnext.py:50-51 (norm -> pwconv1)
  float ln_eps = 0.f;
  int nterms = 6; // split kernels: partial products per element product -- NT_F16X3 (23): 2-way fp16 split, 3 products, fp32-class accuracy;
                  // 6: exact 3-way bf16 split, 6 products (fp32-accurate); 3 (bf16, ~16-bit operands); 1 (plain bf16)
  unsigned x_bytes, x2_bytes, w_bytes;  // buffer sizes for the hardware range check (< 2 GiB each)
  unsigned w_sb_plane_bytes;            // bytes of one bf16 weight plane
  size_t x_sb_plane = 0, x2_sb_plane = 0, y_sb_plane = 0;  // elements between consecutive planes of x_sb / x2_sb / y_sb
  // Always-on saturation watch (engine only; nullptr in the op entry points): every producer of a tensor that a split-f16 contraction will read counts the 16-byte
  // groups of its output with an element beyond sat_limit (or NaN) into *sat -- one compare per group and, in a healthy network, no atomic ever.  The limit is the
  // CONSUMER's window: 65504, 65504 / 4 in front of a Winograd conv (wino.hip), 8188 / 4094 for the attention operands q / kv (attn.hip).
  unsigned* sat = nullptr;
  float sat_limit = 65504.f;
  unsigned long long* stamps = nullptr;  // timing aid (wino.hip, PF_WINO_STAMPS=1 in pf_op_conv2d_bench): s_memtime stamps of block 17, [wave][128]
  // fills the derived fields (Ho, Wo, M, Cin, *_bytes) from the primary ones
  void finish() {
    Cin = C1 + C2;
    Ho = (H + 2 * pad - KH) / stride + 1;
    Wo = (W + 2 * pad - KW) / stride + 1;
    M = B * Ho * Wo;
    x_bytes = (unsigned)((size_t)B * (ups ? H / 2 : H) * (ups ? W / 2 : W) * C1 * 4);
    x2_bytes = (unsigned)((size_t)B * H * W * C2 * 4);
    w_bytes = (unsigned)((size_t)Cout * KH * KWCp * 4);
    w_sb_plane_bytes = (unsigned)((size_t)Cout * KH * KWCp * 2);
    ldy = Cout;
  }
};

void launch_conv(const ConvParams& p, hipStream_t s);
// tile choice is exposed for tests / tuning: -1 = auto
void launch_conv_tile(const ConvParams& p, int tile_id, hipStream_t s);
int conv_num_tiles();
int conv_tile_bm(int tile_id);
int conv_tile_bn(int tile_id);
bool conv_tile_is_sb(int tile_id);
bool conv_tile_usable(const ConvParams& p, int tile_id);
int conv_default_tile(const ConvParams& p);  // static choice (cost model) among the usable tiles  // split-bf16 tiles need pre-split weights and Cin % 32 == 0
// split-bf16 kernel family (igemm_sb.hip)
int conv_sb_num_tiles();
const char* conv_sb_tile_name(int id);
int conv_sb_tile_bm(int id);
int conv_sb_tile_bn(int id);
bool conv_sb_eligible(const ConvParams& p);
bool conv_sb_tile_ok(const ConvParams& p, int sb_tile);  // per-tile restrictions (the pipelined "sbd" tiles: fp32 operands, fp32-accurate mode)
int conv_sb_default_tile(const ConvParams& p);
void launch_conv_sb(const ConvParams& p, int sb_tile, hipStream_t s);
// split-K factor for a launch (1 = none): deep-K shapes whose 64x64 tiling gives too few blocks for 256 CUs (the MiT spatial-
// reduction convs: 3 200 rows, K up to 4 096); the caller allocates g[].partial = splitk * M * ldy floats and sets ConvParams::splitk
int conv_splitk_factor(const ConvParams& p);
int conv_splitk_shape(long M, int Cout, int KH, int KWCp, int groups);  // its shape-only part
const char* conv_tile_name(int tile_id);
// Winograd F(2x2, 3x3) kernel (wino.hip; tiles "wino256x64d" / "wino256x64c" of the split family)
bool conv_wino_ok(const ConvParams& p);
void launch_conv_wino(const ConvParams& p, hipStream_t s, int variant = 0);  // 0: "wino256x64c", 1: "wino256x64d"
int conv_wino_blocks(const ConvParams& p);   // grid size of the shipped Winograd kernel for this launch (all groups; the half-patch geometry where it applies)
void wino_pack_weights(const float* packed /*[Cout][3][KWCp], k = (kx, ci)*/, int Cout, int Cin, int KWCp, std::vector<unsigned short>* planes, std::vector<float>* inv_scale);

// rows x C LayerNorm (biased variance), y may alias x
// Optional split-plane output (sb_split.h) of the elementwise / attention kernels: y_sb != nullptr writes the result as
// three bf16 planes `sb_plane` elements apart (in addition to y, or instead of it when y == nullptr).
void launch_layernorm(const float* x, const float* g, const float* b, float* y, long rows, int C, float eps, hipStream_t s,
                      unsigned short* y_sb = nullptr, size_t sb_plane = 0);

// depthwise 3x3 (pad 1) + bias + exact-erf GELU, NHWC, w packed [9][C]
void launch_dwconv3x3_gelu(const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s,
                           unsigned short* y_sb = nullptr, size_t sb_plane = 0);
void launch_dwconv3x3_gelu_variant(int variant, const float* x, const float* w9c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s,
                                   unsigned short* y_sb = nullptr, size_t sb_plane = 0);
// depthwise 7x7 (pad 3) + bias, NHWC, w packed [49][C]
void launch_dwconv7x7(const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s);
// variant 3 = column-blocked streaming kernel with nc columns per thread (4 / 2), nb row buffers (2 / 3), strips of th rows (0 = automatic); 2 = one column per lane
void launch_dwconv7x7_cfg(int variant, int nc, int nb, int th, const float* x, const float* w49c, const float* bias, float* y, int B, int H, int W, int C, hipStream_t s);

// spatial-reduction attention: q [B][N][C], kv [B][M][2C] (k | v), out [B][N][C]; head_dim 64, M <= 128
void launch_sr_attention(const float* q, const float* kv, float* out, int B, int N, int M, int heads, hipStream_t s,
                         unsigned short* out_sb = nullptr, size_t sb_plane = 0);

// variant 1: split-f16 MFMA (default); 0: exact fp32 MFMA
void launch_sr_attention_variant(int variant, const float* q, const float* kv, float* out, int B, int N, int M, int heads, hipStream_t s,
                                 unsigned short* out_sb = nullptr, size_t sb_plane = 0);

// The attention half of a one-head, 64-channel MiT block as one kernel (attn_block.hip): y = x + proj(softmax((LN1(x) Wq^T + bq) K^T / 8) V); y may alias x
struct MitAttn64Args {
  const float* x = nullptr;            // [B][N][64] token rows
  const float* kv = nullptr;           // [B][M][128] keys | values of the spatially reduced tokens
  float* y = nullptr;                  // [B][N][64]
  const unsigned short* wfr = nullptr; // attn64_pack: q and proj fragments
  const float* tab = nullptr;          // attn64_pack: LayerNorm-1 gamma / beta, inverse scales, biases
  int B = 0, N = 0, M = 0, QT = 1;
  float ln_eps = 1e-6f;
  unsigned* sat = nullptr;             // saturation watch: q against the attention window (8188), y against sat_limit
  float sat_limit = 65504.f;
};
bool mit_attn64_supported(int C, int heads, int kv_rows);
void launch_mit_attn64(const MitAttn64Args& a, int num_cus, hipStream_t s);
void attn64_pack(const float* ln_g, const float* ln_b, const float* q_w, const float* q_b, const float* p_w, const float* p_b, std::vector<unsigned short>* wfr, std::vector<float>* tab);

// The two 7 x 7 convolutions on the normalised image (stem7.hip): conv 7x7 / stride 2 or 4 / pad 3, 3 (stored as 4) -> 64 channels, + bias (folded BatchNorm), + ReLU or
// + LayerNorm over the 64 channels, one kernel
struct Stem7Args {
  const float* x = nullptr;            // [B][H][W][4] normalised image (channel 3 = 0)
  float* y = nullptr;                  // [B][Ho][Wo][64]
  const unsigned short* wfr = nullptr; // stem7x7_pack: weight fragments
  const float* tab = nullptr;          // stem7x7_pack: inverse scales, bias, LayerNorm gamma / beta
  int B = 0, H = 0, W = 0, Ho = 0, Wo = 0, stride = 2, relu = 0, ln = 0, QT = 1;
  float ln_eps = 1e-5f;
  unsigned* sat = nullptr;
  float sat_limit = 65504.f;
};
bool stem7x7_supported(int Cin, int Cout, int K, int stride, int pad);
void launch_stem7x7(const Stem7Args& a, int num_cus, hipStream_t s);
void stem7x7_pack(const float* w, const double* out_scale, const float* bias, const float* ln_g, const float* ln_b, std::vector<unsigned short>* wfr, std::vector<float>* tab);

// y = x W^T + b (+ res) for 128 -> 128 layers over many rows (thin_linear.hip): weights in LDS, rows as the MFMA's B operand, register epilogue; y may alias res
struct ThinLinArgs {
  const float* x = nullptr;            // [M][128]
  const float* res = nullptr;          // [M][128] or nullptr
  float* y = nullptr;                  // [M][128]
  const unsigned short* wfr = nullptr; // thin128_pack
  const float* tab = nullptr;
  long M = 0;
  int QT = 1;
  unsigned* sat = nullptr;
  float sat_limit = 65504.f;
};
bool thin128_supported(int K, int N);
void launch_thin128(const ThinLinArgs& a, int num_cus, hipStream_t s);
void thin128_pack(const float* w, const float* bias, std::vector<unsigned short>* wfr, std::vector<float>* tab);

// bilinear x2 (align_corners=False), NHWC
void launch_upsample2x(const float* x, float* y, int B, int H, int W, int C, hipStream_t s, unsigned short* y_sb = nullptr, size_t sb_plane = 0);

// fp32 [n] <-> three exact bf16 planes (n a multiple of 4)
void launch_split_planes(const float* x, unsigned short* y_sb, size_t sb_plane, long n, hipStream_t s);
void launch_merge_planes(const unsigned short* x_sb, size_t sb_plane, float* y, long n, hipStream_t s);
void launch_fill_random(float* p, long n, unsigned seed, float scale, hipStream_t s);
void launch_range_stats(const float* x, long n, float* out4 /*zeroed: max |x|, sum x^2, saturated, non-finite*/, hipStream_t s);

// input normalisation: uint8 NHWC BGR [B][320][320][3] or fp32 NCHW [B][3][320][320] -> fp32 NHWC4 (x-mean)/std, ch3=0
void launch_prep_u8(const uint8_t* in, float* out, long npix, const float* mean3, const float* std3, hipStream_t s);
void launch_prep_f32_nchw(const float* in, float* out, int B, int HW, const float* mean3, const float* std3, hipStream_t s);

// bit-exact PIL antialiased bilinear resize of one uint8 HxWx3 image to OHxOWx3 (two integer passes; tables from the host)
void launch_resize_u8(const uint8_t* in, int H, int W, uint8_t* tmp, uint8_t* out, int OH, int OW, const int* bh, const int* kh, int ksh,
                      const int* bv, const int* kv, int ksv, hipStream_t s);

// the same for up to ResizeBatch::MAX images in one launch pair (per-image pointers / sizes / tables in the kernel arguments)
struct ResizeBatch {
  static constexpr int MAX = 32;
  int n;
  int H[MAX], W[MAX], ksh[MAX], ksv[MAX];
  const uint8_t* in[MAX];
  uint8_t* tmp[MAX];
  uint8_t* out[MAX];
  const int* bh[MAX]; const int* kh[MAX];
  const int* bv[MAX]; const int* kv[MAX];
};
void launch_resize_batch_u8(const ResizeBatch& rb, int OH, int OW, hipStream_t s);

// regression prediction heads: 1x1 (32->2) + L2-normalise, 1x1 (32->1) + clamp; writes NCHW API outputs and the NHWC4 ParamNet input
void launch_pred_regression(const float* tg, const float* tl, const float* wg, const float* bg, const float* wl, const float* bl,
                            float* pred_g_nchw, float* pred_l_nchw, float* pn_in_nhwc4, int B, int HW, hipStream_t s);

// classification decode: argmax over channels of NCHW logits -> decoded fields (gravity (2,HW), latitude degrees (1,HW))
void launch_decode_cls(const float* logit_g, int ng, const float* logit_l, int nl, float* dec_g, float* dec_l, int B, int HW, hipStream_t s);

// post-process one image: gravity (2,h,w)*scale -> bilinear (H,W) -> normalise; latitude bilinear -> (asin->deg)
void launch_postprocess(const float* g2, const float* l1, int h, int w, float* up_out, float* lat_out, int H, int W, int lat_is_sin, hipStream_t s);

// the same for up to PostBatch::MAX images in one launch (per-image sizes / pointers in the kernel arguments)
struct PostBatch {
  static constexpr int MAX = 32;
  int n;
  int H[MAX], W[MAX];
  float rh[MAX], rw[MAX], sxs[MAX], sys[MAX];  // filled by launch_postprocess_batch
  const float* g2[MAX];
  const float* l1[MAX];
  float* up[MAX];
  float* lat[MAX];
};
void launch_postprocess_batch(PostBatch& pb, int h, int w, int lat_is_sin, hipStream_t s);

// nearest resize NHWC4 (ParamNetConvNextRegress input, param_network.py:197)
void launch_nearest_nhwc4(const float* x, float* y, int B, int H, int W, int Ho, int Wo, hipStream_t s);

// ConvNeXt tail: global average pool -> LN(C) -> Linear(C->nout); out [B][nout]
// Fused ConvNeXt block MLP (cnx_mlp.hip): y += ls * pwconv2(GELU(pwconv1(LayerNorm(d)))) for C = 96 / 192, hidden map in registers only.
// wpk / tab: packed by cnx_mlp_pack (engine.hip)
bool cnx_mlp_supported(int C);
bool cnx_mlp_preferred(int C);  // the stages where the engine uses it
void launch_cnx_mlp(const float* d, float* y, const unsigned short* wpk, const float* tab, long M, int C, float eps, hipStream_t s, unsigned* sat = nullptr, float sat_limit = 65504.f);
// Fused MiT block Mlp (mit_mlp.hip): y = x + fc2(GELU(dwconv3x3(fc1(LayerNorm(x))))) for C = 64 / 128; x and y are different buffers.
// wpk / tab2: packed by mit_mlp_pack (engine.hip), one chunk of mit_mlp_chunk_bytes(C) per 32 hidden units
bool mit_mlp_supported(int C);
bool mit_mlp_preferred(int C, int with128);  // the stages for which the engine builds its weights (with128 = Engine::mit_mlp128: smallest batch that takes the C = 128 form, 0 = never)
int mit_mlp_chunk_bytes(int C);
void launch_mit_mlp(const float* x, float* y, const unsigned short* wpk, const float* tab2, int B, int Hs, int Ws, int C, float eps, hipStream_t s, unsigned* sat = nullptr, float sat_limit = 65504.f);
// Row-block linear layers (rb_gemm.hip, rb_common.h): blocks of 64 token rows of one image, weights streamed from L2 into registers in MFMA fragment order
struct RbLinArgs {
  const float* x;            // [M][K] fp32 rows
  const float* ln_g;         // LayerNorm over K (nullptr: none)
  const float* ln_b;
  float ln_eps;
  const unsigned short* w;   // weight stream (rb_pack_w), N / COLS passes of K / 16 steps
  size_t w_bytes;
  const float* inv;          // [N] inverse weight scales
  const float* bias;         // [N]
  const float* res;          // [M][N] or nullptr (may alias y)
  float* y;                  // [M][N]
  int M, tokens, bpi;        // rows, tokens per image, blocks per image = ceil(tokens / 64)
  int N, act;
  // Always-on saturation watch (engine only; nullptr in the op entry points): every producer of a tensor that a split-f16 contraction will read counts the 16-byte
  // groups of its output with an element beyond sat_limit (or NaN) into *sat -- one compare per group and, in a healthy network, no atomic ever.  The limit is the
  // CONSUMER's window: 65504, 65504 / 4 in front of a Winograd conv (wino.hip), 8188 / 4094 for the attention operands q / kv (attn.hip).
  unsigned* sat = nullptr;
  float sat_limit = 65504.f;
  unsigned long long* stamps = nullptr;  // timing aid (scripts/tune_rb.py, PF_RB_STAMPS=1): s_memtime stamps of block 17, [wave][64]
};
bool rb_linear_supported(int K, int N);
// The key / value branch of a MiT block with 2 x 2 spatial reduction in one launch (rb_chain.hip): kv = Linear(LN_sr(Conv2x2s2(LN_1(x)))), mix_transformers.py:119-127
struct RbSrKvArgs {
  const float* x;             // [B][2 Hr][2 Wr][C] token map (pre-norm1)
  const float* ln1_g; const float* ln1_b; float ln1_eps;
  const unsigned short* w;    // weight stream: the conv as a GEMM over K = (ky, kx, ci) (80 steps), then the kv layer's two 320-column passes (2 x 20 steps), RB_D pad steps
  size_t w_bytes;
  const float* sr_inv; const float* sr_bias;   // [C]
  const float* srn_g; const float* srn_b; float srn_eps;
  const float* kv_inv; const float* kv_bias;   // [2 C]
  float* kv;                  // [B][Hr Wr][2 C]
  int B, Hr, Wr, bpi;         // bpi = ceil(Hr Wr / 32) blocks per image
  unsigned* sat = nullptr;    // saturation watch of kv (ConvParams::sat; the attention kernel's window for k / v: 4094)
};
bool rb_srkv_supported(int C, int sr);
// The seam between the attention half and the Mlp half of a MiT block in one launch (rb_chain.hip): x += proj(attn_out); hidden = fc1(LayerNorm_2(x))
struct RbProjFc1Args {
  const float* attn;          // [M][C] attention output
  float* x;                   // [M][C] token stream: residual in, x1 out (in place)
  const unsigned short* w;    // weight stream: proj (C / 16 steps), then fc1's four 320-column passes, RB_D pad steps
  size_t w_bytes;
  const float* proj_inv; const float* proj_bias;   // [C]
  const float* ln2_g; const float* ln2_b; float ln2_eps;
  const float* fc1_inv; const float* fc1_bias;     // [4 C]
  float* hidden;              // [M][4 C]
  int B, tokens, bpi;         // bpi = ceil(tokens / 64)
  unsigned* sat = nullptr;    // saturation watch of x1 and hidden (ConvParams::sat)
};
void launch_rb_proj_fc1(const RbProjFc1Args& a, int C, hipStream_t s);
void launch_rb_srkv(const RbSrKvArgs& a, int C, hipStream_t s);
void launch_rb_linear(const RbLinArgs& a, int K, hipStream_t s);
void launch_gap_ln_head(const float* x, const float* g, const float* b, const float* w, const float* hb, float* out, int B, int HW, int C, int nout, float eps, hipStream_t s);

// camera parameters {roll, elevation (rad), focal_rel, cx_rel, cy_rel} (device) -> up [2][H][W], latitude [H][W] degrees
// (PanoCam.get_up_general / get_lat_general, utils/panocam.py:451-556)
void launch_fields_from_params(const float* cam5, int H, int W, float* up, float* lat, hipStream_t s);

// ParamNet scalar formulas (param_network.py:62-67): raw [B][nraw] -> [B][8] (layout: include/pf_hip.h)
void launch_paramnet_scalars(const float* raw, int nraw, float* out8, int B, int mode, hipStream_t s);

}  // namespace pf
