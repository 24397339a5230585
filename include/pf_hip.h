/* pf_hip.h -- C ABI of libpf_hip.so, the MI355X (gfx950) implementation of the
 * PerspectiveFields dense-field + ParamNet inference path.
 *
 * The reference (jinlinyi/PerspectiveFields) has no FFI layer: its boundary for this
 * path is the Python method PerspectiveFields.forward (perspective2d/perspectivefields.py:
 * 223-272), reached from .inference() (:194-205) and .inference_batch() (:207-221).
 * The entry points below are what a ctypes binding inside that class binds instead of
 * running the nn.Module tree:
 *
 *   pf_create / pf_load_tensor / pf_finalize_weights   <- PerspectiveFields.__init__ + _init_weights
 *                                                         (perspectivefields.py:122-192): build the
 *                                                         network for a zoo architecture and load
 *                                                         the {"model": state_dict} checkpoint
 *   pf_forward_u8 / pf_forward_f32                     <- forward() up to `results`
 *                                                         (perspectivefields.py:234-254,258): normalise,
 *                                                         backbone (mix_transformers.py:449-485), ll_enc
 *                                                         (:70-83), persformer_heads.inference
 *                                                         (persformer_heads.py:73-81), param_net
 *                                                         (param_network.py:46-69 / 193-221)
 *   pf_postprocess                                     <- persformer_heads.postprocess
 *                                                         (persformer_heads.py:83-101 ->
 *                                                         gravity_head.py:237-261, latitude_head.py:195-219,
 *                                                         utils/utils.py:483-507,114-130,148-162)
 *
 * Conventions: plain pointers and sizes only.  Every `d_` pointer is a DEVICE pointer owned
 * by the caller; `h_` pointers are host memory.  All work is enqueued on the given
 * hipStream_t (passed as void*) and never synchronises.  Return value 0 = success,
 * negative = pf_status; pf_last_error() gives the message.  A handle is bound to one
 * device and is not thread-safe.  A workspace (`d_ws`) belongs to the calls in flight on it: two calls that
 * share one workspace must be ordered by the caller (same stream, or an event between the streams) -- the
 * Python host side does this for its own workspace (Engine._order_scratch).  There is no CPU fallback: every entry point fails with
 * PF_ERR_DEVICE when no gfx950 device is usable.
 */
#ifndef PF_HIP_H
#define PF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pf_engine* pf_handle;

enum pf_status {
  PF_OK = 0,
  PF_ERR_ARG = -1,      /* bad argument (null pointer, bad shape, unknown arch) */
  PF_ERR_DEVICE = -2,   /* no usable HIP device / HIP runtime error */
  PF_ERR_WEIGHTS = -3,  /* missing / unexpected / mis-shaped checkpoint tensor, or forward before finalize */
  PF_ERR_WORKSPACE = -4 /* workspace too small */
};

/* architectures = the three module trees the reference zoo instantiates (perspectivefields.py:86-118) */
enum pf_arch {
  PF_ARCH_PARAMNET_CENTERED = 0,   /* regression heads (2 / 1 ch) + ParamNet (ConvNeXt-T, 5 raw outputs) */
  PF_ARCH_PERSNET_CLS = 1,         /* classification heads (73 / 180 logits), no ParamNet */
  PF_ARCH_PARAMNET_UNCENTERED = 2  /* regression heads + ParamNetConvNextRegress (64x64 input, 5 outputs) */
};

#define PF_NET_SIZE 320  /* the network always runs at 320x320 (every reference YAML: DATALOADER.RESIZE) */
#define PF_PARAMS_STRIDE 8
/* Largest batch one pf_forward_* call accepts: the kernels address each activation with 32-bit byte offsets, and the
 * largest per-head activation is 26.2 MB per image (2 GiB / 26.2 MB = 81).  Larger batches return PF_ERR_ARG; the host
 * layer (PerspectiveFields._run) splits longer image lists into chunks, as the reference's callers may pass any length. */
#define PF_MAX_BATCH 81

const char* pf_version(void);
const char* pf_build_digest(void); /* digest of the sources this library was built from (stale-library check of the loader) */
const char* pf_last_error(pf_handle h); /* h may be NULL: error of the last failed pf_create on this thread */

/* device = PF_DEVICE_NONE: the engine's HOST side only -- checkpoint loading, weight repack / folds / splits (kept in host memory), the workspace dry run and the tile
 * table work; every entry point that needs a GPU returns PF_ERR_DEVICE.  The seam the sanitizer build is tested through (PF_ASAN=1 python -m perspectivefields_amd.build,
 * tests/test_host_asan.py); not a CPU path of the network. */
#define PF_DEVICE_NONE (-1)
int pf_create(pf_handle* out, int device, int arch);
int pf_destroy(pf_handle h);

/* Checkpoint tensors by their reference state_dict key (schema: SURVEY.md appendix B), fp32, C-contiguous.
 * `ll_enc.bn1.num_batches_tracked` is accepted and ignored. */
int pf_load_tensor(pf_handle h, const char* key, const float* h_data, const int64_t* shape, int rank);
/* Strict validation (all keys of the architecture present, none unknown), BatchNorm / layer-scale folding,
 * repack to the kernels' layouts, upload. */
int pf_finalize_weights(pf_handle h);

/* channel counts of the API-visible 320x320 maps and number of raw ParamNet outputs (0 if none) */
int pf_output_info(pf_handle h, int* gravity_channels, int* latitude_channels, int* param_raw_outputs);

int pf_max_batch(void); /* = PF_MAX_BATCH */

/* bytes of scratch pf_forward_* needs for a batch of `batch` images (0 for batch <= 0 or > PF_MAX_BATCH) */
size_t pf_workspace_bytes(pf_handle h, int batch);

/* Forward pass for `batch` images already resized to 320x320.
 *   d_images_u8   : [batch][320][320][3] uint8, BGR (what ResizeTransform.apply_image returns, :201)
 *   d_images_f32  : [batch][3][320][320] fp32 BGR 0..255 (the "image" entries forward() receives, :234)
 *   d_pred_gravity : [batch][Cg][320][320] fp32 -- unit up-vectors (Cg=2) or 73 logits
 *   d_pred_latitude: [batch][Cl][320][320] fp32 -- sin(latitude) in [-1,1] (Cl=1) or 180 logits
 *   d_params       : [batch][PF_PARAMS_STRIDE] fp32 or NULL when the arch has no ParamNet:
 *                    CENTERED  : roll, pitch, vfov (deg), rel_focal, raw x0..x3   (param_network.py:62-67)
 *                    UNCENTERED: raw x0..x4 (roll/90, pitch/90, general_vfov/90, rel_cx, rel_cy), rel_focal (closed form of utils/utils.py:47-91 general_vfov_to_focal,
 *                                fp64 on the device), 0, 0   (param_network.py:204-220)
 */
/* Arithmetic of the dense contractions (everything else is always fp32).
 *   FP32 (default, the parity mode): "split-f16" -- every fp32 activation is split on the fly into two fp16 values
 *     (a ~ ah + al with ah = fp16(a), al = fp16(a - ah) UNSCALED: 22-23 significant bits for |a| >= 2^-3, an absolute 2^-25 per element below -- the matrix cores keep fp16 subnormals), the weights (scaled per output channel by a power of two) into wh + wl,
 *     and a product is three fp16 MFMA partial products (ah wh + ah wl + al wh) accumulated in fp32: per-product error
 *     <= 3 * 2^-22, below the fp32 accumulation noise of the contraction itself (scripts/emulate_split.py).
 *     Activations beyond the fp16 range (|x| > 65504) saturate.
 *   FP32_BF16X6: every operand split EXACTLY into three bf16 values, six bf16 MFMA partial products -- fp32-accurate
 *     for any fp32 input (no range restriction), twice the matrix-core work.
 * NO reduced-precision mode is offered (r05): r01-r04 carried "bf16x3" / "bf16" switches on these same kernels -- 98.9 % argmax agreement for +0.6 ... 4 % speed,
 * i.e. lower accuracy for nothing (profiles/r04_bench_configs.json) -- and a bf16 THROUGHPUT path (bf16 activations in HBM, one MFMA per product, its own tile
 * table) was not built; values 1 and 2 of earlier headers are rejected with PF_ERR_ARG.
 * May be called at any time; it applies to the following forwards (tile choices are tuned per mode). */
#define PF_PRECISION_FP32 0
#define PF_PRECISION_FP32_BF16X6 3
int pf_set_precision(pf_handle h, int mode);

/* Always-on saturation watch of the FP32 (split-f16) mode.  d_counter_u32: a caller-owned, zero-initialised 4-byte device counter (nullptr: off).  Every kernel that
 * WRITES a tensor a split-f16 contraction will read (GEMM / conv epilogues, the fused block MLPs, the Winograd convs) adds to it the number of 16-byte output groups
 * with an element beyond the window of that tensor's consumer (65504; 65504 / 4 = 16376 in front of a Winograd conv; 8188 / 4094 for the attention operands; the
 * input limit of a depthwise conv, from its weights).  Cost: one compare per 4 outputs; the counter only ever grows.  Read it in stream order behind a forward (a
 * 4-byte copy): unchanged = every dense-layer input of that forward was inside the window.
 *
 * THE CONTRACT FOR A CALLER THAT PINS PF_PRECISION_FP32 (mandatory reading since r05, when the 256 -> 256 decoder convs became Winograd F(2x2, 3x3)):
 *   - outside a window the direct tiles SATURATE (the split clamps to +-65504: a finite, wrong result), but the Winograd layers' input transform adds four values
 *     and splits WITHOUT the clamp: an input in (16376, 65504] may, and one beyond certainly does, turn into +-inf / NaN in that layer's output;
 *   - both cases move the counter: the producer of such an input is watched with the 16376 limit, and the Winograd / row-block / fused-MLP epilogues also count NaN
 *     (inf - inf) outputs; the register-capped GEMM tiles count saturation and +-inf only (a NaN reaches them only behind a producer that was already counted);
 *   - so: a forward that left the counter unchanged is inside every window and its results are fp32-class; a forward that moved it MUST be discarded and re-run
 *     with PF_PRECISION_FP32_BF16X6 (no window).  There is no third case, and no output of a counted forward is promised to be finite.
 *   The Python layer does exactly that with `precision="auto"` (the default) and stays in the exact mode afterwards; `precision="fp32"` skips the check -- and the
 *   host synchronisation it costs -- and leaves the counter to the caller (tests/test_gpu_e2e.py::test_pinned_fp32_window_contract).
 * Tensors no kernel can watch (LayerNorm outputs, the register-only hidden maps of the fused MLPs) are bounded from the weights alone: pf_static_window_max returns
 * the largest such bound, scaled so that a value > 65504 means "a static bound exceeds its window". */
int pf_set_saturation_counter(pf_handle h, void* d_counter_u32);
int pf_static_window_max(pf_handle h, float* out);

int pf_forward_u8(pf_handle h, int batch, const uint8_t* d_images_u8, float* d_pred_gravity, float* d_pred_latitude,
                  float* d_params, void* d_workspace, size_t workspace_bytes, void* stream);
int pf_forward_f32(pf_handle h, int batch, const float* d_images_f32, float* d_pred_gravity, float* d_pred_latitude,
                   float* d_params, void* d_workspace, size_t workspace_bytes, void* stream);

/* Deferred ParamNet branch (throughput option for callers that issue forward after forward on one stream; off by default).  With on != 0 the ConvNeXt branch of a
 * forward (param_network.py:46-69, convnext.py:140-152) runs on a stream owned by the engine, behind that forward's decoders, and the call returns without joining it:
 * the NEXT forward's backbone (other images, no dependency) runs beside it -- both are chains of small launches that leave most of the chip idle.  d_pred_gravity /
 * d_pred_latitude are valid in `stream` order as always; d_params of a forward becomes valid in `stream` order once the next pf_forward_* has been issued on that
 * stream (it waits for the branch before its decoders overwrite the branch's input) or after pf_join_params(h, stream).  The caller keeps d_params alive until then. */
int pf_set_defer_params(pf_handle h, int on);
int pf_join_params(pf_handle h, void* stream);

/* Low-latency form of pf_forward_u8 for small batches: the ~430 launches of a forward are captured once per (batch, buffer
 * set) into a hipGraph and replayed with one launch (host launch cost, not GPU time, bounds a batch-1 forward).  Same
 * arguments and results; `stream` must be an explicit stream (not the legacy NULL stream); the graph is re-captured when
 * any buffer pointer changes, so callers keep their buffers (the Python layer does). */
int pf_forward_u8_graph(pf_handle h, int batch, const uint8_t* d_images_u8, float* d_pred_gravity, float* d_pred_latitude,
                        float* d_params, void* d_workspace, size_t workspace_bytes, void* stream);

/* Device-side replacement of ResizeTransform.apply_image (perspectivefields.py:34-46,201): PIL's antialiased BILINEAR
 * resize of one uint8 [H][W][3] image to [320][320][3], bit-identical to PIL (integer two-pass filter; coefficient
 * tables built on the host exactly as Pillow does and cached per extent in the handle -- the first call for a new
 * extent performs a small blocking upload).  d_out may point into the batch tensor given to pf_forward_u8. */
size_t pf_resize_workspace_bytes(int H, int W);
int pf_resize_bilinear_u8(pf_handle h, const uint8_t* d_img, int H, int W, uint8_t* d_out_320, void* d_workspace,
                          size_t workspace_bytes, void* stream);

/* The same for a whole batch in two launches per 32 images (inference_batch's per-image resize loop, :210-216):
 * h_imgs = HOST array of B DEVICE pointers to uint8 [H_i][W_i][3] images, h_hw = HOST array [B][2] of (H_i, W_i);
 * d_out = [B][320][320][3] (the tensor pf_forward_u8 takes).  Workspace: sum over images of roundup(H_i * 320 * 3, 256) + 256 bytes. */
int pf_resize_batch_u8(pf_handle h, int batch, const uint8_t* const* h_imgs, const int32_t* h_hw, uint8_t* d_out_320,
                       void* d_workspace, size_t workspace_bytes, void* stream);

/* Tile configuration of the conv / GEMM launches.  Default: a table keyed by launch shape (loaded with pf_load_tile_table;
 * the Python layer loads the one shipped for gfx950, perspectivefields_amd/tuned/gfx950_tiles.txt) and a static heuristic
 * for shapes the table does not hold -- deterministic, no first-call stall.
 * pf_autotune: explicit one-time tuning for a batch size: a normal forward (same arguments and results as pf_forward_u8;
 * workspace of pf_autotune_workspace_bytes) in which every conv / GEMM launch is additionally timed with each tile
 * configuration on this device (HIP events; this call DOES wait on the stream); the fastest per shape is cached in the
 * handle (pf_save_tile_table writes the cache out).  With PF_AUTOTUNE=1 in the environment the Python layer runs it on the
 * first forward of every new batch size; pf_is_tuned then tells whether a batch size has been tuned.
 * pf_load_tile_table / pf_save_tile_table return the number of entries read / written, or a negative pf_status. */
size_t pf_autotune_workspace_bytes(pf_handle h, int batch);
int pf_autotune(pf_handle h, int batch, const uint8_t* d_images_u8, float* d_pred_gravity, float* d_pred_latitude,
                float* d_params, void* d_workspace, size_t workspace_bytes, void* stream);
int pf_is_tuned(pf_handle h, int batch);
int pf_load_tile_table(pf_handle h, const char* path);
int pf_save_tile_table(pf_handle h, const char* path);

/* Post-process ONE image's 320x320 predictions to its original size (H, W):
 *   d_up_out  : [2][H][W] unit up-vectors (x right, y down), d_lat_out : [H][W] degrees.
 * For the classification arch d_workspace must hold 3*320*320 floats (decoded fields). */
int pf_postprocess(pf_handle h, const float* d_pred_gravity, const float* d_pred_latitude, int H, int W,
                   float* d_up_out, float* d_lat_out, void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- per-kernel-class timing with HIP events on the launch stream (bench.py `roofline`) ----
 * classes: 0 implicit-GEMM conv/GEMM (work = algorithmic FLOPs, 2*M*Cout*KH*KW*Cin), 1 attention (FLOPs),
 * 2 LayerNorm, 3 depthwise3x3+GELU, 4 depthwise7x7, 5 bilinear x2 (work = algorithmic bytes in+out), 6 other,
 * 7 implicit-GEMM launches served by the split-bf16 kernel (class 0 then counts only the exact-fp32 MFMA launches).
 * Between begin and end every launch of a class whose bit is set in class_mask is bracketed by an
 * event pair; pf_profile_end synchronises those events and sums elapsed ms / work / launches per class.
 * class_mask bit 31 (PF_PROFILE_LARGE_ONLY): only launches of >= 200 GFLOP are bracketed (a dozen per B = 32 step: the dominant
 * decoder convs) -- cheap enough for every timed step of a benchmark, and the internal side stream stays on. */
#define PF_PROFILE_CLASSES 8
#define PF_PROFILE_LARGE_ONLY 0x80000000u
int pf_profile_begin(pf_handle h, unsigned class_mask);
/* stop bracketing further launches without synchronising (the window can cover the first steps of a timed loop only) */
int pf_profile_pause(pf_handle h);
int pf_profile_end(pf_handle h, double* ms, double* work, long* launches, int n);
/* per-launch records of the last begin/end window (valid until the next pf_profile_begin); returns the
 * total number of records, fills at most max_records entries; mnk = GEMM view [M, N, K, KH] for class 0 */
int pf_profile_records(pf_handle h, int max_records, int* cat, double* work, float* ms, int* mnk);
/* Component split of the path: inside a FULL per-launch window (class_mask without PF_PROFILE_LARGE_ONLY: one stream, forwards joined) the engine also
 * records an event at the start of every component -- 0 input normalisation + MiT-B3 backbone (mix_transformers.py:449-485), 1 low-level encoder
 * (perspectivefields.py:70-83), 2 both decoder heads + prediction heads (gravity_head.py:139-197, latitude_head.py:138-193), 3 ParamNet (param_network.py:46-69),
 * 4 post-process (pf_postprocess_batch).  Call after pf_profile_end: ms[PF_PROFILE_PHASES] = elapsed stream time per component summed over the window (launch gaps and
 * the profile's own event pairs included); returns the number of marks recorded. */
#define PF_PROFILE_PHASES 5
#define PF_PHASE_BACKBONE 0
#define PF_PHASE_LOW_LEVEL 1
#define PF_PHASE_DECODERS 2
#define PF_PHASE_PARAMNET 3
#define PF_PHASE_POSTPROCESS 4
int pf_profile_phases(pf_handle h, int n, double* ms);

/* ---- Debug forward: localise a numerical problem layer by layer, and check a checkpoint's activations against the split-f16 scheme's window.
 * The reference has no such mode; it replaces "print the tensor after every module" in a PyTorch session (perspectivefields.py:223-272 run by hand).
 *   flags & 1  SHADOW taps: the boundary tensors of the path -- every MiT block output "mit.s<stage>.b<block>" (mix_transformers.py:198-202), the four stage
 *              outputs "c1".."c4" (:457-485), the low-level encoder output "ll" (perspectivefields.py:70-83), both decoders' "dec.<head>.conv0" maps
 *              (gravity_head.py:170-171), the ParamNet input "pn.in", stem "pn.stem" and every ConvNeXt block output "pn.s<stage>.b<block>" (convnext.py:46-59,
 *              140-152) -- are copied, NHWC fp32, into d_tap_buf (pf_debug_tap_bytes(batch) bytes); pf_debug_taps returns names, byte offsets and shapes.
 *   flags & 2  RANGE records: for every tensor that enters a dense contraction (all conv / linear layers incl. the fused block MLPs, attention q / kv) max |x|, sum x^2,
 *              the number of elements beyond the fp16 range (the default precision SATURATES them at 65504) and of non-finite elements; pf_debug_ranges returns them.
 *              A tensor whose rms is below ~2^-5 loses relative accuracy in the default precision (the low part of the split turns subnormal below |x| = 2^-3):
 *              run such a checkpoint with PF_PRECISION_FP32_BF16X6.
 * Same results as pf_forward_u8; synchronises the stream before returning.  The record functions return the total number of records and fill at most max_records. */
size_t pf_debug_tap_bytes(pf_handle h, int batch);
int pf_debug_forward_u8(pf_handle h, int batch, const uint8_t* d_in_u8, float* d_pred_gravity, float* d_pred_latitude, float* d_params, void* d_workspace,
                        size_t workspace_bytes, int flags, void* d_tap_buf, size_t tap_bytes, void* stream);
int pf_debug_taps(pf_handle h, int max_records, char* names /*[max][64]*/, long long* byte_offsets, int* shapes /*[max][4]: B, H, W, C*/);
int pf_debug_ranges(pf_handle h, int max_records, char* names /*[max][96]*/, long long* elems, float* stats /*[max][4]*/);

/* The same for a whole batch in one launch per 32 images (the reference's per-image Python loop,
 * gravity_head.py:244-260 / latitude_head.py:201-218): h_hw = HOST array [B][2] of (H, W); h_up_out / h_lat_out = HOST
 * arrays of B DEVICE pointers ([2][H][W] and [H][W] each).  Classification arch: workspace of B*3*320*320 floats. */
int pf_postprocess_batch(pf_handle h, int batch, const float* d_pred_gravity, const float* d_pred_latitude, const int32_t* h_hw,
                         float* const* h_up_out, float* const* h_lat_out, void* d_workspace, size_t workspace_bytes, void* stream);

/* Row N4: camera parameters -> perspective fields on the device, the step every demo of the reference runs
 * right after inference (utils/utils.py:325-381 -> PanoCam.get_up_general / get_lat_general, utils/panocam.py:451-556).
 * d_cam5 = {roll, pitch (= elevation), both in RADIANS, rel_focal, rel_cx, rel_cy} in device memory (so the ParamNet
 * output can feed it without a host round trip); outputs in the layout of pred_gravity_original /
 * pred_latitude_original: d_up [2][H][W] unit vectors, d_lat [H][W] degrees.  No handle: the op is stateless. */
int pf_fields_from_params(int device, const float* d_cam5, int H, int W, float* d_up, float* d_lat, void* stream);

/* ---- kernel-level entry points (used by the parity tests; same kernels pf_forward runs) ----
 * NHWC fp32 device activations; weights are HOST pointers in the reference's layouts.
 * "planes": the engine's internal split activation formats -- an fp32 tensor stored as planes of 16-bit values, plane k at
 * `planes + k * (plane_elems & ~1)`, each laid out like the fp32 tensor.  Bit 0 of every *_plane_elems argument selects the
 * format: 0 = three bf16 planes (x == h + m + l exactly; read by the bf16 schemes), 1 = two fp16 planes of the split-f16
 * scheme (x ~ hi + lo, lo = fp16(x - hi) unscaled: what that scheme's GEMM computes from fp32 while staging; 4 bytes per element).
 * Producers write them for tensors that only feed GEMMs (PF_SBA=1); every *_planes argument is optional (NULL). */
int pf_op_conv2d(int device, const float* d_x, const float* d_x2, int B, int H, int W, int C1, int C2,
                 const float* h_weight /*[Cout][C1+C2][KH][KW]*/, const float* h_bias /*[Cout] or NULL*/,
                 int Cout, int KH, int KW, int stride, int pad, int act /*0 none 1 relu 2 gelu*/,
                 const float* d_res1, const float* d_res2, int post_relu, int nchw_out, int tile_id /*-1 auto*/,
                 float* d_y /*may be NULL when d_y_planes is given*/,
                 const uint16_t* d_x_planes /*replaces d_x*/, long x_plane_elems, const uint16_t* d_x2_planes, long x2_plane_elems,
                 uint16_t* d_y_planes, long y_plane_elems, int precision /*PF_PRECISION_* (split tiles only); + 16: no split-K (otherwise applied by the engine's rule)*/, void* stream);
/* times `iters` launches of one conv shape on random data with tile config `tile_id` (-1 auto); avg ms per launch.
 * fmt_prec = fmt + 16 * precision; fmt 0: fp32 in / out; 1: input as bf16 planes; 2: input and output as planes
 * (1 / 2: the exact bf16 split); precision = PF_PRECISION_* used by the split tiles */
/* y = act(Linear(LayerNorm(x))) + res1 with the LayerNorm fused into the GEMM (ConvParams::ln: row statistics accumulated while the rows are
 * staged, gamma / beta folded into the weights / bias here on the host) -- the engine's form of mix_transformers.py:200 (norm2 -> fc1),
 * :123-126 (sr norm -> kv) and convnext.py:50-51 (norm -> pwconv1).  K % 32 == 0, N % 4 == 0; tile_id as pf_op_conv2d (linear split tiles only). */
int pf_op_linear_ln(int device, const float* d_x, long rows, int K, const float* h_weight /*[N][K]*/, const float* h_bias, const float* h_gamma, const float* h_beta,
                    float eps, int N, int act, const float* d_res1, int tile_id, float* d_y, int precision, void* stream);
/* One nn.Linear of the MiT stage-3 / 4 blocks in the row-block form (rb_gemm.hip): y = act(Linear(LayerNorm?(x))) + res on blocks of 64 token rows of one image --
 * mix_transformers.py:80-88 (q, kv, proj), :26-29 (fc1, fc2), with :199-200 / :125 (norm1, norm2, attn.norm) applied while the rows are staged when h_gamma != NULL.
 * rows = images x tokens; (K, N) = (320, multiple of 320) or (multiple of 256 above 320, 320); res may alias y.  iters > 0 additionally times `iters` launches. */
int pf_op_rb_linear(int device, const float* d_x, long rows, int tokens, int K, const float* h_weight /*[N][K]*/, const float* h_bias, const float* h_gamma, const float* h_beta,
                    float eps, int N, int act, const float* d_res, float* d_y, int iters, float* ms_out, void* stream);
/* The seam between the attention half and the Mlp half of a MiT block in one launch (rb_chain.hip): x += proj(attn_out), hidden = fc1(LayerNorm_2(x)) --
 * mix_transformers.py:137-139 (proj), :199 (residual), :200 / :52 (norm2 -> fc1).  attn (B tokens, C), x (B tokens, C) read and written, hidden (B tokens, 4C);
 * C = 320; weights in the reference's shapes.  iters > 0 additionally times `iters` launches (x is then garbage). */
int pf_op_rb_proj_fc1(int device, const float* d_attn, float* d_x, int B, int tokens, int C, const float* h_proj_w, const float* h_proj_b, const float* h_ln2_gamma,
                      const float* h_ln2_beta, float eps, const float* h_fc1_w, const float* h_fc1_b, float* d_hidden, int iters, float* ms_out, void* stream);
/* The attention half of a one-head, 64-channel MiT block (stage 1 of MiT-B3) as ONE kernel (attn_block.hip): y = x + proj(softmax((LayerNorm_1(x) Wq^T + bq) K^T / 8) V)
 * -- Block.forward, mix_transformers.py:199, with Attention.forward :108-141 (q :110, softmax :133-134, proj :137-138).  x, y: (B, N, 64) token rows (y may alias x),
 * kv: (B, M, 128) keys | values of the spatially reduced tokens (1 <= M <= 128); weights in the reference's shapes.  iters > 0 additionally times `iters` launches. */
int pf_op_mit_attn64(int device, const float* d_x, const float* d_kv, float* d_y, int B, int N, int M, const float* h_ln1_gamma, const float* h_ln1_beta, float eps,
                     const float* h_q_w, const float* h_q_b, const float* h_proj_w, const float* h_proj_b, int iters, float* ms_out, void* stream);
/* The 7 x 7 convolutions that read the normalised image, as one specialised kernel (stem7.hip): conv 7x7 / stride 2 (the low-level encoder, perspectivefields.py:70-83: eval
 * BatchNorm folded by the caller into weight / bias, relu = 1) or stride 4 (MiT's first patch embedding + its LayerNorm, mix_transformers.py:205-246: ln gamma / beta given),
 * pad 3, 3 -> 64 channels.  x: (B, H, W, 4) NHWC4 with channel 3 = 0; y: (B, Ho, Wo, 64); weight in the reference's shape (64, 3, 7, 7).  iters > 0 also times launches. */
int pf_op_stem7x7(int device, const float* d_x, float* d_y, int B, int H, int W, int stride, const float* h_weight, const float* h_bias, int relu, const float* h_ln_gamma,
                  const float* h_ln_beta, float eps, int iters, float* ms_out, void* stream);
/* y = x W^T + b (+ res) for a 128 -> 128 nn.Linear over many rows (thin_linear.hip; the q / output projections of the MiT stage-2 blocks, mix_transformers.py:110,
 * :137-138): x, res, y (rows, 128) device fp32 (y may alias res), weight (128, 128) / bias (128) host, the reference's shapes.  iters > 0 also times launches. */
int pf_op_thin128(int device, const float* d_x, long rows, const float* h_weight, const float* h_bias, const float* d_res, float* d_y, int iters, float* ms_out, void* stream);
/* The key / value branch of a MiT block with 2 x 2 spatial reduction in one launch (rb_chain.hip): kv = Linear_kv(LayerNorm(Conv2d_2x2s2(LayerNorm_1(x)))),
 * mix_transformers.py:119-127 (norm1 of :199 applied to the gathered source tokens).  x: (B, 2 Hr, 2 Wr, C) NHWC token map, C = 320; weights in the reference's shapes
 * (sr [C][C][2][2], kv [2C][C]); kv out: (B, Hr Wr, 2C).  iters > 0 additionally times `iters` launches. */
int pf_op_rb_srkv(int device, const float* d_x, int B, int Hr, int Wr, int C, const float* h_ln1_gamma, const float* h_ln1_beta, float eps1, const float* h_sr_w,
                  const float* h_sr_b, const float* h_srn_gamma, const float* h_srn_beta, float eps2, const float* h_kv_w, const float* h_kv_b, float* d_kv, int iters,
                  float* ms_out, void* stream);
/* One ConvNeXt block MLP in one kernel (cnx_mlp.hip): y += ls * pwconv2(GELU(pwconv1(LayerNorm(d)))), convnext.py:49-58; C = 96 or 192, weights
 * in the reference's shapes (pwconv1 [4C][C], pwconv2 [C][4C], layer scale ls [C]); y is read (residual) and written.  iters > 0 additionally times
 * `iters` launches (avg ms in *ms_out; y is then garbage). */
int pf_op_cnx_mlp(int device, const float* d_d, float* d_y, long rows, int C, const float* h_w1, const float* h_b1, const float* h_ln_gamma, const float* h_ln_beta,
                  float eps, const float* h_w2, const float* h_b2, const float* h_layer_scale, int iters, float* ms_out, void* stream);
/* One MiT block Mlp in one kernel (mit_mlp.hip): y = x + fc2(GELU(dwconv3x3(fc1(LayerNorm(x))))), mix_transformers.py:49-56,200,497-508; x, y: (B, Hs, Ws, C) NHWC
 * token maps in DIFFERENT buffers, C = 64 or 128; weights in the reference's shapes (fc1 [4C][C], dwconv [4C][1][3][3], fc2 [C][4C]).  iters > 0 additionally
 * times `iters` launches (avg ms in *ms_out). */
int pf_op_mit_mlp(int device, const float* d_x, float* d_y, int B, int Hs, int Ws, int C, const float* h_fc1_w, const float* h_fc1_b, const float* h_ln_gamma,
                  const float* h_ln_beta, float eps, const float* h_dw_w, const float* h_dw_b, const float* h_fc2_w, const float* h_fc2_b, int iters, float* ms_out, void* stream);
int pf_op_conv2d_bench(int device, int B, int H, int W, int Cin, int Cout, int K, int stride, int pad, int tile_id, int iters, int fmt_prec, float* ms_out);
/* fp32 <-> planes in the format selected by bit 0 of plane_elems (the names are historical) */
int pf_op_split_bf16(int device, const float* d_x, long n, uint16_t* d_planes, long plane_elems, void* stream);
int pf_op_merge_bf16(int device, const uint16_t* d_planes, long plane_elems, long n, float* d_y, void* stream);
/* times one depthwise-3x3+GELU launch variant on random data (0 = LDS halo tile, 1-4 = register-window direct, 99 = plain copy; 0, 1, 3, 99: tuning builds only;
 * 1000 + 100 s + 10 t + c: multi-column kernel, block shape s, strip height t, (columns, prefetch) code c -- elem.hip) */
int pf_op_dwconv3x3_bench(int device, int variant, int B, int H, int W, int C, int iters, float* ms_out);
int pf_op_layernorm(int device, const float* d_x, const float* h_gamma, const float* h_beta, float* d_y, long rows, int C, float eps,
                    uint16_t* d_y_planes, long plane_elems, void* stream);
int pf_op_dwconv3x3_gelu(int device, const float* d_x, const float* h_weight /*[C][1][3][3]*/, const float* h_bias, float* d_y, int B, int H, int W, int C,
                         uint16_t* d_y_planes, long plane_elems, void* stream);
/* the same with an explicit kernel variant (pf_op_dwconv3x3_bench's ids; >= 1000: multi-column / prefetching kernel, falls back to the
 * default when the shape does not fit its block) */
int pf_op_dwconv3x3_gelu_cfg(int device, const float* d_x, const float* h_weight, const float* h_bias, float* d_y, int B, int H, int W, int C,
                             uint16_t* d_y_planes, long plane_elems, int variant, void* stream);
int pf_op_dwconv7x7(int device, const float* d_x, const float* h_weight /*[C][1][7][7]*/, const float* h_bias, float* d_y, int B, int H, int W, int C, void* stream);
/* the same with an explicit kernel: variant 3 = column-blocked streaming kernel (nc = 4 / 2 output columns per thread, nb = 2 / 3
 * row buffers, th = rows per strip; 0 = automatic), 2 = one column per lane; and a timing loop on random data (avg ms per launch) */
int pf_op_dwconv7x7_cfg(int device, const float* d_x, const float* h_weight, const float* h_bias, float* d_y, int B, int H, int W, int C,
                        int variant, int nc, int nb, int th, void* stream);
int pf_op_dwconv7x7_bench(int device, int variant, int nc, int nb, int th, int B, int H, int W, int C, int iters, float* ms_out);
int pf_op_sr_attention(int device, const float* d_q, const float* d_kv, float* d_out, int B, int N, int M, int heads,
                       uint16_t* d_out_planes, long plane_elems, void* stream);
/* explicit kernel: variant 1 = split-f16 MFMA (default of the forward), 0 = exact fp32 MFMA; iters > 0 additionally times
 * `iters` launches (avg ms per launch in *ms_out; synchronises) */
int pf_op_sr_attention_variant(int device, int variant, const float* d_q, const float* d_kv, float* d_out, int B, int N, int M, int heads,
                               int iters, float* ms_out, void* stream);
int pf_op_upsample2x(int device, const float* d_x, float* d_y, int B, int H, int W, int C, uint16_t* d_y_planes, long plane_elems, void* stream);
int pf_op_num_conv_tiles(void);
const char* pf_op_conv_tile_name(int tile_id);

#ifdef __cplusplus
}
#endif
#endif /* PF_HIP_H */
