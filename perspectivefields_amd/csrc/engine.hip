// libpf_hip.so: weight store, layer graph and C ABI of the MI355X PerspectiveFields path.
// Host-side C++ here only *orchestrates* hand-written gfx950 kernels (igemm.hip, attn.hip,
// elem.hip); there is no CPU or library (MIOpen / hipBLASLt) fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pf_hip.h"
#include "pf_kernels.h"

using namespace pf;

namespace {

}  // namespace
namespace pf_host { thread_local std::string g_create_error; }
#include "host_pack.h"
using namespace pf_host;
namespace {


// ---- architecture constants (reference: mix_transformers.py:511-524, gravity_head.py:121-137, convnext.py:78-79)
constexpr int NET = PF_NET_SIZE;
static_assert((size_t)PF_MAX_BATCH * PF_NET_SIZE * PF_NET_SIZE * 64 * 4 <= 0x7fffffffull &&
              (size_t)PF_MAX_BATCH * (PF_NET_SIZE / 2) * (PF_NET_SIZE / 2) * 256 * 4 <= 0x7fffffffull,
              "PF_MAX_BATCH: every per-head activation must stay below 2 GiB (32-bit byte offsets, OOB marker 2^31)");
const int MIT_DIMS[4] = {64, 128, 320, 512};
const int MIT_HEADS[4] = {1, 2, 5, 8};
const int MIT_DEPTHS[4] = {3, 4, 18, 3};
const int MIT_SR[4] = {8, 4, 2, 1};
const int MIT_PK[4] = {7, 3, 3, 3}, MIT_PS[4] = {4, 2, 2, 2}, MIT_PP[4] = {3, 1, 1, 1};
constexpr int DEC_EMBED = 768, DEC_FEAT = 256, LL_CH = 64;
const int CNX_DEPTHS[4] = {3, 3, 9, 3};
const int CNX_DIMS[4] = {96, 192, 384, 768};

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
  bool used = false;
};

struct ConvW {
  float* w = nullptr;
  float* b = nullptr;
  float* btab = nullptr;  // [9][Cout] border-case biases of a folded Linear->3x3 pair
  unsigned short* wsb = nullptr;  // weights split exactly into 3 bf16 planes (split-bf16 kernel), when Cin % 32 == 0
  unsigned short* wh16 = nullptr; // split-f16 scheme: per-channel power-of-two scaled weights as two fp16 planes wh, wl
  float* wh16_inv = nullptr;      //   and the inverse scale per output channel
  unsigned short* wwino = nullptr; // Winograd F(2x2, 3x3) form of a 3x3 / stride-1 conv (wino.hip): transformed weights in fragment order
  float* wwino_inv = nullptr;      //   and their inverse scale per output channel
  float* ln_s = nullptr;          // fused input LayerNorm (ConvParams::ln): column sums of the gamma-folded weights; nullptr = no fusion
  float ln_eps = 0.f;
  float sat_limit = 65504.f;      // window of the tensor this layer WRITES, set by its consumer (ConvParams::sat_limit): Winograd conv 65504 / 4, attention q / kv 8188 / 4094,
                                  // a depthwise conv behind it: (65504 - max |bias|) / max_c sum_k |w[k][c]|
  int Cout = 0, Cin = 0 /*padded*/, CinReal = 0, KH = 1, KW = 1, stride = 1, pad = 0, KWC = 0, KWCp = 0;
};
struct LNW { float* g = nullptr; float* b = nullptr; int C = 0; float eps = 1e-6f; };
struct DwW { float* w = nullptr; float* b = nullptr; int C = 0; float in_limit = 65504.f; /* its INPUT may reach this much and the output still fits 65504 */ };

// a linear layer in the row-block form (rb_gemm.hip): weight stream + inverse scales + bias
struct RbLin { unsigned short* w = nullptr; size_t bytes = 0; float* inv = nullptr; float* b = nullptr; int N = 0, K = 0; float sat_limit = 65504.f; };
// the key / value branch of a stage-3 block as one launch (rb_chain.hip): combined weight stream (conv as GEMM, then kv) + the two layers' scales / biases
struct RbSrKv { unsigned short* w = nullptr; size_t bytes = 0; float *sr_inv = nullptr, *sr_b = nullptr, *kv_inv = nullptr, *kv_b = nullptr; };
struct RbProjFc1 { unsigned short* w = nullptr; size_t bytes = 0; };  // combined weight stream of rb_proj_fc1_kernel (scales / biases: rproj, rfc1)
struct MitBlock { LNW n1, n2, srn; RbLin rq, rkv, rproj, rfc1, rfc2; RbSrKv rsrkv; RbProjFc1 rpf; ConvW qln /*q with norm1 folded (used next to rsrkv: no LayerNorm-1 launch)*/; ConvW q, kv, proj, sr, fc1, fc2; DwW dw; unsigned short* mlp_w = nullptr; float* mlp_tab = nullptr; /* fused Mlp (mit_mlp.hip), when built */
                  unsigned short* a64_w = nullptr; float* a64_tab = nullptr; /* fused attention half of a one-head, 64-channel block (attn_block.hip), when built */
                  unsigned short* tq_w = nullptr; float* tq_tab = nullptr; unsigned short* tp_w = nullptr; float* tp_tab = nullptr; /* 128 -> 128 q / proj in the thin form (thin_linear.hip), when built */ };
struct MitStage { ConvW pe; LNW pen, norm; std::vector<MitBlock> blocks; };
struct Head {
  ConvW lin[4], proc[4], fold[4], r1c1[4], r1c2[4], r2c1[4], r2c2[4], conv0, conv1, predcls;
  float* predw = nullptr; float* predb = nullptr; int nout = 0;
};
struct CnxBlock { float out_limit = 65504.f; /* window of the residual stream this block writes: the next block's depthwise conv */ DwW dw; LNW n; ConvW pw1, pw2; unsigned short* mlp_w = nullptr; float* mlp_tab = nullptr; /* fused MLP (cnx_mlp.hip), when built */ };
struct Cnx { ConvW stem, ds[3]; LNW stemn, dsn[3], norm; std::vector<CnxBlock> blocks[4]; float* headw = nullptr; float* headb = nullptr; int nout = 0; };

// Split-bf16 activation tensor (sb_split.h): three exact bf16 planes `plane` elements apart.
struct SbT {
  unsigned short* p = nullptr;
  size_t plane = 0;
};
// An activation as the kernels see it: fp32 NHWC and / or split planes (producers write what their consumers read).
struct Ten {
  float* f = nullptr;
  SbT s;
  Ten() {}
  Ten(float* f_) : f(f_) {}
  Ten(float* f_, SbT s_) : f(f_), s(s_) {}
};

struct Profiler;
// Debug forward (pf_debug_forward_u8): SHADOW taps -- named stage / block boundary tensors copied out for a layer-by-layer comparison with the oracle -- and RANGE
// records -- max |x|, sum x^2, saturated (|x| > 65504) and non-finite counts of every tensor that enters a dense contraction (the split-f16 scheme's window, sb_split.h).
struct DebugSink {
  bool shadow = false, range = false;
  char* buf = nullptr; size_t cap = 0, used = 0; bool overflow = false;  // shadow: caller's device buffer
  struct Tap { std::string name; size_t off; int shape[4]; };          // NHWC
  std::vector<Tap> taps;
  float* stats = nullptr; int max_ranges = 0;                          // range: device [max_ranges][4], zeroed before the forward
  struct Rng { std::string name; long long elems; };
  std::vector<Rng> ranges;
  int seq = 0;  // running number of dense launches
};

struct Ctx {
  hipStream_t s;
  uintptr_t base;
  size_t off = 0, peak = 0;
  bool dry;
  Profiler* prof = nullptr;
  DebugSink* dbg = nullptr;
  bool tuning = false;      // autotune pass: time every tile config per conv shape
  float* tune_scratch = nullptr;  // [max_conv_out] floats, followed by 3 bf16 planes of max_conv_out elements
  size_t tune_scratch_elems = 0;
  size_t max_conv_out = 0;  // elements of the largest (grouped) conv output seen by the dry run = tuning scratch size
  float* alloc(size_t nfloats) {
    const size_t bytes = (nfloats * 4 + 255) & ~(size_t)255;
    const size_t o = off;
    off += bytes;
    if (off > peak) peak = off;
    return reinterpret_cast<float*>(base + o);
  }
  int sb_planes = 3;        // planes per split tensor: 3 (exact bf16 split) or 2 (split-f16 scheme; SbT::plane then carries SB_FMT_F16)
  SbT alloc_sb(size_t elems) {
    SbT t;
    const size_t stride = (elems + 127) & ~(size_t)127;
    t.p = reinterpret_cast<unsigned short*>(alloc((sb_planes * stride * 2 + 3) / 4));
    t.plane = stride | (sb_planes == 2 ? SB_FMT_F16 : 0);
    return t;
  }
  // fp32 and / or split planes, as requested
  Ten ten(size_t elems, bool want_f32, bool want_sb) {
    Ten t;
    if (want_f32) t.f = alloc(elems);
    if (want_sb) t.s = alloc_sb(elems);
    return t;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
};

// Per-kernel-class timing with HIP events on the launch stream (bench.py's `roofline` object):
// every launch of a class is bracketed by an event pair; work = algorithmic FLOPs or bytes.
enum ProfCat { PC_IGEMM = 0, PC_ATTN, PC_LAYERNORM, PC_DW3, PC_DW7, PC_UPSAMPLE, PC_OTHER, PC_IGEMM_SB, PC_COUNT };
enum { PH_BACKBONE = 0, PH_LL = 1, PH_DECODERS = 2, PH_PARAMNET = 3, PH_POST = 4, PH_END = 5 };  // = PF_PHASE_* (include/pf_hip.h)
struct Profiler {
  struct Rec { int cat; double work; hipEvent_t a, b; int m, n, k, kh; float ms; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  size_t used = 0;
  bool on = false;
  unsigned mask = 0xffffffffu;
  double min_work = 0.0;  // > 0: only launches with at least this much algorithmic work are bracketed (class-mask bit 31: the few large launches of a step --
                          // a dozen event records per step cost nothing, so EVERY timed step of bench.py can carry them, with the side stream left on)
  hipEvent_t get() {
    if (used == pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return nullptr; pool.push_back(e); }
    return pool[used++];
  }
  // Phase marks of a FULL per-launch profile (one stream, no deferred branch): an event at the start of every component of the path (pf_profile_phases:
  // SURVEY 8d's component split).  A mark ends the phase before it; PH_END closes a phase without opening one.
  struct Mark { int phase; hipEvent_t e; };
  std::vector<Mark> marks;
  void mark(int phase, hipStream_t s) {
    if (!on || min_work > 0.0) return;
    hipEvent_t e = get();
    if (!e) return;
    (void)hipEventRecord(e, s);
    marks.push_back({phase, e});
  }
  void reset() { recs.clear(); marks.clear(); used = 0; }
  ~Profiler() { for (hipEvent_t e : pool) (void)hipEventDestroy(e); }
};
struct ProfScope {
  Profiler* p; hipStream_t s; hipEvent_t b = nullptr;
  ProfScope(Profiler* pr, hipStream_t st, int cat, double work, int m = 0, int n = 0, int k = 0, int kh = 0) : p(pr), s(st) {
    if (!p || !p->on || !((p->mask >> cat) & 1u) || work < p->min_work) { p = nullptr; return; }
    hipEvent_t a = p->get(); b = p->get();
    if (!a || !b) { p = nullptr; return; }
    (void)hipEventRecord(a, s);
    p->recs.push_back({cat, work, a, b, m, n, k, kh, 0.f});
  }
  ~ProfScope() { if (p) (void)hipEventRecord(b, s); }
};

struct ResizeTable { int ksize = 0; std::vector<int> bounds, kk; int *d_bounds = nullptr, *d_kk = nullptr; };
void resize_coeffs(int in_size, int out_size, ResizeTable* t) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const int ksize = (int)std::ceil(support) * 2 + 1;
  t->ksize = ksize;
  t->bounds.assign((size_t)out_size * 2, 0);
  t->kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  const double ss = 1.0 / filterscale;
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    double ww = 0.0;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    for (int x = 0; x < ksize; ++x) k[x] = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double v = (x + xmin - center + 0.5) * ss;
      if (v < 0.0) v = -v;
      const double w = v < 1.0 ? 1.0 - v : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x)
      if (ww != 0.0) k[x] /= ww;
    for (int x = 0; x < ksize; ++x)
      t->kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << 22)) : (int)(0.5 + k[x] * (1 << 22));
    t->bounds[2 * xx] = xmin;
    t->bounds[2 * xx + 1] = xmax;
  }
}

}  // namespace

struct pf_engine {
  int device = 0;
  int arch = 0;
  bool finalized = false;
  std::string err;
  std::unordered_map<std::string, HostTensor> host;
  std::vector<void*> dev_allocs;
  std::map<int, size_t> ws_cache;
  Profiler prof;
  bool autotune = false;     // PF_AUTOTUNE=1: the first forward of every new batch size times all tile configs per conv shape (a 1-2 s stall).
                             // Default off: tiles come from the shipped table (pf_load_tile_table, tuned/gfx950_tiles.txt) or the static
                             // heuristic -- deterministic, no first-call latency cliff; pf_autotune stays available as an explicit call
  std::map<std::vector<int>, int> tile_cache;  // conv shape (+batch) -> fastest tile config, measured on this device
  std::map<int, bool> tuned_batches;
  std::map<int, ResizeTable> resize_tables;  // input extent -> tables for resizing that extent to NET
  std::map<int, size_t> scratch_off, scratch_elems;
  bool split_bf16 = true;    // PF_SPLIT_BF16=0: exact-fp32 MFMA kernels only (no split-bf16 tiles)
  bool fuse_pred = true;     // PF_FUSE_PRED=0: keep the 32-channel 320x320 maps and run the regression prediction heads as their own kernel
  bool fuse_upsample = true; // PF_FUSE_UPSAMPLE=0: materialise the two largest bilinear x2 maps (160^2 x 256, 320^2 x 64 per head) instead of
                             // interpolating them inside the consuming 3x3 convs' halo staging (split-f16 scheme only)
  bool fuse_ln = true;       // PF_FUSE_LN=0: run every LayerNorm as its own kernel.  Default: a LayerNorm whose only consumers are 1x1 layers (MiT norm2 -> fc1,
                             // sr norm -> kv, stage-4 norm1 -> q / kv; ConvNeXt norm -> pwconv1) is folded into them (ConvParams::ln): gamma / beta go into the
                             // weights / bias at finalize, the row statistics are accumulated by the GEMM's own staging threads
  int thin128 = 25600;       // PF_THIN128 (0 = off, n = from n token rows up; 64 KB of weights per block is a prologue that 1 600 rows -- one image -- do not repay: B = 1 -1 %, B >= 16 +0.2 %): the 128 -> 128 q / output projections of the MiT stage-2 blocks in the transposed, register-epilogue form (thin_linear.hip)
  int stem7 = 1;             // PF_STEM7: the two 7 x 7 convs on the normalised image (low-level encoder; first patch embedding + its LayerNorm) as the specialised kernel of stem7.hip
  unsigned short* ll_s7_w = nullptr; float* ll_s7_tab = nullptr; unsigned short* pe_s7_w = nullptr; float* pe_s7_tab = nullptr;
  int attn64 = 1;            // PF_ATTN64: the attention half of the one-head stage-1 blocks (q, attention, proj, residual) as one kernel (attn_block.hip)
  int s3_split = 1;          // PF_S3_SPLIT: MiT stage 3 on two half-batches / two streams (mit(), "the stage-3 split")
  int side_stream_mode = 1;  // 1 (default): the q projection of a MiT block runs on a second stream next to the sr conv + kv GEMM (both consume LayerNorm-1's
                             // output, attention joins them) -- small launches that each fill a fraction of the chip: +0.6 % (3 x A/B on one box, profiles/
                             // r02_negative_results.md); 2: also the low-level encoder conv next to MiT stage 3 (+0.1 %, noise); PF_SIDE_STREAM=0: one stream.
                             // Never forked while tuning or profiling (per-launch events time one stream)
  hipStream_t side = nullptr, side2 = nullptr;   // side: the q projections; side2: the low-level encoder conv, launched next to MiT stage 3
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_ll = nullptr;
  // Deferred ParamNet (pf_set_defer_params): the ConvNeXt branch of forward i runs on an internal stream behind the decoders of forward i and is NOT joined at the
  // end of the call -- forward i + 1's backbone (other images: no dependency) runs next to it.  Both are chains of small, latency-bound launches that leave most of
  // the chip idle (MiT stage 3: 13 % MFMA-busy; ConvNeXt stages 3-4 likewise), and their kernels co-reside (small LDS, 4 waves).  d_params of forward i is valid in
  // the caller's stream order once the NEXT pf_forward* has been issued on that stream (it waits for the branch before its decoders overwrite the branch's input),
  // or after pf_join_params.  The branch works in its own workspace region (input map + activations), behind the main one.
  bool host_only = false;    // pf_create(.., PF_DEVICE_NONE, ..): the host side only (weight repack / folds / splits, workspace dry run, tile table) with the "device" copies of the
                             // weights in host memory -- the sanitizer build's test seam; every entry point that would touch a GPU fails with PF_ERR_DEVICE
  int defer_params = 0;
  hipStream_t pstream = nullptr;
  hipEvent_t ev_pn_in = nullptr, ev_pn_done = nullptr;
  bool pn_pending = false, in_capture = false;
  int pn_pending_B = 0; uintptr_t pn_pending_base = 0;   // batch and workspace of the pending branch: another batch size lays the regions out differently
  // PF_DEFER_AT = k > 0: the deferred branch of forward i is ISSUED (on pstream) only when forward i + 1 reaches MiT stage k -- next to the stage whose launches leave
  // the most room -- instead of right away (k = 0); pf_join_params / the decoders' wait issue it if no forward came
  int defer_at = 0, defer_prio = 0;
  struct PnLater { bool armed = false; int B = 0; const float* pn = nullptr; float* params = nullptr; Ctx ctx; } pn_later;
  void issue_deferred() {
    if (!pn_later.armed) return;
    pn_later.armed = false;
    paramnet(pn_later.ctx, pn_later.B, pn_later.pn, pn_later.params);
    (void)hipEventRecord(ev_pn_done, pstream);
  }
  std::map<int, size_t> pn_off;   // batch -> byte offset of the ParamNet region in the workspace
  bool pstream_ready() {
    if (pstream) return true;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // lo = least urgent (largest number)
    const int prio = defer_prio > 0 ? hi : (defer_prio < 0 ? lo : 0);
    if ((defer_prio ? hipStreamCreateWithPriority(&pstream, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(&pstream, hipStreamNonBlocking)) != hipSuccess) { pstream = nullptr; return false; }
    if (hipEventCreateWithFlags(&ev_pn_in, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_pn_done, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(pstream); pstream = nullptr; return false;
    }
    return true;
  }
  bool ll_forked = false;
  // Streams are a scarce resource: the runtime multiplexes a process's streams onto a few hardware queues (4 by default), and two streams that share a queue do not
  // overlap -- the engine therefore creates only the streams its mode needs (side2 only for PF_SIDE_STREAM=2); with the null stream, `side` and the ParamNet stream
  // a forward uses three queues (profiles/r04_defer_params.md: with nine streams in one process the deferred branch lost its whole gain).
  bool side_ready() {
    if (side && (side2 || side_stream_mode < 2)) return true;
    if (!side) {
      if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) { side = nullptr; return false; }
      if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev_join, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&ev_ll, hipEventDisableTiming) != hipSuccess) return false;
    }
    if (side_stream_mode >= 2 && !side2 && hipStreamCreateWithFlags(&side2, hipStreamNonBlocking) != hipSuccess) { side2 = nullptr; return false; }
    return true;
  }
  bool can_fork(const Ctx& c) { return side_stream_mode && !c.dry && !c.tuning && (!c.prof || c.prof->min_work > 0.0) && side_ready(); }  // (a full per-launch profile keeps one stream: its event pairs must bracket what they name)
  bool sba_heads = false;    // PF_SBA_HEADS=1: the tensors between the 3x3 convs of the decoders' ResidualConvUnits are written as split-f16 planes by the producing
                             // conv's epilogue (plus fp32 where a residual add reads them) and the halo kernel copies them (igemm_sbh ASB) instead of splitting
                             // every element once per n-tile and halo overlap; split-f16 scheme only
  int rb_chain = 60;         // PF_RB_CHAIN: which linear layers of MiT stage 3 run in the row-block form (rb_gemm.hip) instead of the LDS tiles (igemm_sb) once the batch gives
                             // at least rb_min_blocks row blocks: bit mask 1 q, 2 kv, 4 proj, 8 fc1, 16 fc2, 32 the whole key / value branch as one launch (rb_chain.hip; overrides 2), 64 proj + norm2 + fc1 as one launch (overrides 4, 8) (0 = none)
  int rb_min_blocks = 192;   // default for 256 CUs; pf_create rescales it to 3/4 of the device's CU count
  int num_cus = 256;         // hipDeviceProp_t::multiProcessorCount (partitioned / smaller gfx950 configurations: CPX / DPX modes)
  int wino_min_hw = 20;      // PF_WINO=<n>: 3x3 / stride-1 convs with Cin, Cout multiples of 64 on maps of at least n x n run as Winograd F(2x2, 3x3) (wino.hip; split-f16
                             // scheme only; 0 = never).  Their inputs' window ends at 65504 / 4 (the input transform adds four values).  r06: 20 instead of 40 -- with the
                             // half-patch geometry the 20 x 20 RCU convs run 1.45x faster than on the direct tile (profiles/r06_wino_gate.txt; r05's square patches: equal)
  int wino_min_blocks = 96;  // PF_WINO_MIN_BLOCKS: ... and only launches of at least this many blocks (one block per CU, 512 registers per wave: at batch 1 the 40 x 40 maps
                             // give 64 blocks and the direct tile, with more and smaller blocks, is 18 % faster there)
  int wino_tile = -1;        // tile id of the Winograd kernel in use
  unsigned* d_sat = nullptr; // pf_set_saturation_counter: caller-owned device counter of the always-on saturation watch (ConvParams::sat); nullptr = off
  float static_window_max = 0.f;  // largest STATIC bound of a tensor that never reaches HBM (the hidden maps of the fused block MLPs), pf_static_window_max
  int mit_mlp128 = 4;        // PF_MIT_MLP_128 (0 = never, n = from a batch of n images up): the one-kernel Mlp at MiT stage 2 (mit_mlp.hip, C = 128; 25 blocks per image: B = 1 -1.1 %, 2 -0.4 %, 4 +0.7 %, 8 +1.1 %, 32 +0.6 %)
  bool fuse_mit_mlp = true;  // PF_FUSE_MIT_MLP=0: the Mlp of MiT stages 1 / 2 as LayerNorm-fused fc1 + depthwise 3x3 / GELU + fc2 instead of the one-kernel form
                             // (mit_mlp.hip: hidden map in LDS / registers only); split-f16 scheme only
  bool fuse_cnx_mlp = true;  // PF_FUSE_CNX_MLP=0: ConvNeXt blocks of the 96- and 192-channel stages as LayerNorm-fused pwconv1 + pwconv2 GEMMs instead of
                             // the one-kernel MLP (cnx_mlp.hip, hidden map in registers only); split-f16 scheme only
  bool fold_mlp = true;      // PF_FOLD_MLP=0 keeps Linear(C->768) and conv3x3(768->256) as two kernels
  int nterms = NT_F16X3;     // pf_set_precision: NT_F16X3 = 2-way fp16 split, 3 MFMAs per product (default parity mode); 6 = exact 3-way bf16 split
                             // (fp32-accurate, PF_PRECISION_FP32_BF16X6); 3 = "bf16x3", 1 = "bf16" (reduced precision, not parity modes)
  DebugSink dbg;  // records of the last pf_debug_forward_u8
  DebugSink* debug_sink = nullptr;  // non-null while pf_debug_forward_u8 runs
  // hipGraph replay of small-batch forwards (pf_forward_u8_graph): one graph per (batch, buffer set, precision)
  struct GraphEntry { std::vector<uintptr_t> key; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;
  bool sba = false;          // PF_SBA=1: tensors that only feed GEMMs are stored as split-bf16 planes by their producers (sb_split.h);
                             // measured slower end to end (1.5x the bytes on HBM-bound layers), kept as an option -- DESIGN.md 4.2

  MitStage stages[4];
  ConvW ll;
  Head heads[2];  // 0 gravity, 1 latitude
  Cnx cnx;
  bool has_param = false;
  int param_in = NET;  // ParamNet input resolution
  float mean3[3] = {103.53f, 116.28f, 123.675f};
  float std3[3] = {1.f, 1.f, 1.f};

  int fail(int code, const std::string& m) { err = m; return code; }

  // ------------------------------------------------------------------ weights
  ResizeTable* resize_table(int in_size) {
    auto it = resize_tables.find(in_size);
    if (it != resize_tables.end()) return &it->second;
    ResizeTable& t = resize_tables[in_size];
    resize_coeffs(in_size, NET, &t);
    void *db = nullptr, *dk = nullptr;
    if (hipMalloc(&db, t.bounds.size() * 4) != hipSuccess || hipMalloc(&dk, t.kk.size() * 4) != hipSuccess) return nullptr;
    if (hipMemcpy(db, t.bounds.data(), t.bounds.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dk, t.kk.data(), t.kk.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    dev_allocs.push_back(db); dev_allocs.push_back(dk);
    t.d_bounds = static_cast<int*>(db); t.d_kk = static_cast<int*>(dk);
    return &t;
  }
  float* upload(const std::vector<float>& v) {
    void* d = nullptr;
    if (host_only) {
      d = malloc(v.size() * sizeof(float) + 1);
      if (!d) throw std::string("malloc failed for weights");
      std::memcpy(d, v.data(), v.size() * sizeof(float));
      dev_allocs.push_back(d);
      return static_cast<float*>(d);
    }
    if (hipMalloc(&d, v.size() * sizeof(float)) != hipSuccess) throw std::string("hipMalloc failed for weights");
    if (hipMemcpy(d, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) throw std::string("hipMemcpy H2D failed for weights");
    dev_allocs.push_back(d);
    return static_cast<float*>(d);
  }
  unsigned short* upload_u16(const std::vector<unsigned short>& v) {
    void* d = nullptr;
    if (host_only) {
      d = malloc(v.size() * 2 + 1);
      if (!d) throw std::string("malloc failed for weights");
      std::memcpy(d, v.data(), v.size() * 2);
      dev_allocs.push_back(d);
      return static_cast<unsigned short*>(d);
    }
    if (hipMalloc(&d, v.size() * 2) != hipSuccess) throw std::string("hipMalloc failed for weights");
    if (hipMemcpy(d, v.data(), v.size() * 2, hipMemcpyHostToDevice) != hipSuccess) throw std::string("hipMemcpy H2D failed for weights");
    dev_allocs.push_back(d);
    return static_cast<unsigned short*>(d);
  }
  // packed fp32 weights -> device, plus the split-bf16 planes when the layer is eligible for igemm_sb
  void upload_conv_weights(ConvW& c, const std::vector<float>& packed, int CinP, int Cout) {
    c.w = upload(packed);
    if (split_bf16 && (CinP % 32 == 0 || CinP == 4)) {  // CinP == 4: the 3-channel stems (split-f16 "stem" form of igemm_sb_kernel)
      c.wsb = upload_u16(split_bf16x3(packed));
      const F16Planes f = split_f16x2(packed, Cout);
      c.wh16 = upload_u16(f.planes);
      c.wh16_inv = upload(f.inv_scale);
    }
  }
  const HostTensor& get(const std::string& key, std::initializer_list<int64_t> shape) {
    auto it = host.find(key);
    if (it == host.end()) throw fmt("missing checkpoint tensor '%s'", key.c_str());
    HostTensor& t = it->second;
    if (t.shape != std::vector<int64_t>(shape)) {
      std::string got, want;
      for (auto d : t.shape) got += std::to_string(d) + ",";
      for (auto d : shape) want += std::to_string(d) + ",";
      throw fmt("shape mismatch for '%s': got (%s) expected (%s)", key.c_str(), got.c_str(), want.c_str());
    }
    t.used = true;
    return t;
  }
  ConvW make_conv(const std::string& wkey, const std::string& bkey, int Cout, int Cin, int K, int stride, int pad,
                  const double* out_scale = nullptr, const std::vector<float>* bias_override = nullptr) {
    ConvW c;
    const int CinP = roundup(Cin, 4);
    const HostTensor& w = get(wkey, {Cout, Cin, K, K});
    const std::vector<float> packed = pack_conv(w.data.data(), Cout, Cin, K, K, CinP, out_scale, &c.KWC, &c.KWCp);
    upload_conv_weights(c, packed, CinP, Cout);
    if (bias_override) c.b = upload(*bias_override);
    else if (!bkey.empty()) {
      std::vector<float> b = get(bkey, {Cout}).data;
      if (out_scale) for (int n = 0; n < Cout; ++n) b[n] = (float)(b[n] * out_scale[n]);
      c.b = upload(b);
    }
    c.Cout = Cout; c.Cin = CinP; c.CinReal = Cin; c.KH = c.KW = K; c.stride = stride; c.pad = pad;
    if (wino_min_hw > 0 && split_bf16 && K == 3 && stride == 1 && pad == 1 && CinP % 64 == 0 && Cout % 64 == 0 && c.KWCp == c.KWC) {
      std::vector<unsigned short> planes;
      std::vector<float> inv;
      wino_pack_weights(packed.data(), Cout, CinP, c.KWCp, &planes, &inv);
      c.wwino = upload_u16(planes);
      c.wwino_inv = upload(inv);
    }
    return c;
  }
  // ln_pfx != "" (and fuse_ln): the LayerNorm `ln_pfx` in front of this Linear is folded into it (fold_ln_linear)
  ConvW make_linear(const std::string& pfx, int N, int K, const double* out_scale = nullptr, const std::string& ln_pfx = "", float ln_eps = 0.f) {
    ConvW c;
    const HostTensor& w = get(pfx + ".weight", {N, K});
    std::vector<float> b = get(pfx + ".bias", {N}).data;
    if (!ln_pfx.empty() && fuse_ln && K % 32 == 0 && N % 4 == 0) {
      const std::vector<float>& g = get(ln_pfx + ".weight", {K}).data;
      const std::vector<float>& be = get(ln_pfx + ".bias", {K}).data;
      std::vector<float> wf, bf, cs;
      fold_ln_linear(w.data.data(), b.data(), g.data(), be.data(), N, K, &wf, &bf, &cs);
      b = bf;
      upload_conv_weights(c, pack_conv(wf.data(), N, K, 1, 1, K, nullptr, &c.KWC, &c.KWCp), K, N);
      c.ln_s = upload(cs);
      c.ln_eps = ln_eps;
    } else {
      upload_conv_weights(c, pack_conv(w.data.data(), N, K, 1, 1, K, out_scale, &c.KWC, &c.KWCp), K, N);
      if (out_scale) for (int n = 0; n < N; ++n) b[n] = (float)(b[n] * out_scale[n]);
    }
    c.b = upload(b);
    c.Cout = N; c.Cin = K; c.CinReal = K; c.KH = c.KW = 1; c.stride = 1; c.pad = 0;
    return c;
  }
  // Linear(C -> 768) followed (no nonlinearity) by a zero-padded conv3x3(768 -> 256) is one conv3x3(C -> 256):
  //   W'[o][ci][ky][kx] = sum_e Wp[o][e][ky][kx] * Wl[e][ci]
  // and the Linear bias reaches an output pixel only through the taps that fall inside the map:
  //   bias(y,x)[o] = bp[o] + sum_{valid (ky,kx)} T[o][ky][kx],   T[o][ky][kx] = sum_e Wp[o][e][ky][kx] * bl[e]
  // -> a 9-entry table indexed by the 3x3 border case.  Folded in fp64 (reference: decode_head.py:49-53 +
  // gravity_head.py:70-97,145-166).  8.2x fewer FLOPs for these layers (30.1 -> 3.7 GFLOP per head).
  ConvW make_folded(const std::string& lin, const std::string& proc, int C) {
    const HostTensor& wl = get(lin + ".weight", {DEC_EMBED, C});
    const HostTensor& bl = get(lin + ".bias", {DEC_EMBED});
    const HostTensor& wp = get(proc + ".weight", {DEC_FEAT, DEC_EMBED, 3, 3});
    const HostTensor& bp = get(proc + ".bias", {DEC_FEAT});
    std::vector<float> wf((size_t)DEC_FEAT * C * 9);
    std::vector<double> T((size_t)DEC_FEAT * 9), acc(C);
    std::vector<double> wld(wl.data.begin(), wl.data.end());
    for (int o = 0; o < DEC_FEAT; ++o)
      for (int t = 0; t < 9; ++t) {
        std::fill(acc.begin(), acc.end(), 0.0);
        double tb = 0.0;
        for (int e = 0; e < DEC_EMBED; ++e) {
          const double a = wp.data[((size_t)o * DEC_EMBED + e) * 9 + t];
          const double* row = &wld[(size_t)e * C];
          for (int ci = 0; ci < C; ++ci) acc[ci] += a * row[ci];
          tb += a * (double)bl.data[e];
        }
        for (int ci = 0; ci < C; ++ci) wf[((size_t)o * C + ci) * 9 + t] = (float)acc[ci];
        T[(size_t)o * 9 + t] = tb;
      }
    ConvW c;
    upload_conv_weights(c, pack_conv(wf.data(), DEC_FEAT, C, 3, 3, C, nullptr, &c.KWC, &c.KWCp), C, DEC_FEAT);
    std::vector<float> tab((size_t)9 * DEC_FEAT);
    for (int cy = 0; cy < 3; ++cy)
      for (int cx = 0; cx < 3; ++cx)
        for (int o = 0; o < DEC_FEAT; ++o) {
          double b = bp.data[o];
          for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) {
              const bool vy = !((cy == 0 && ky == 0) || (cy == 2 && ky == 2));  // top row has no tap above, bottom none below
              const bool vx = !((cx == 0 && kx == 0) || (cx == 2 && kx == 2));
              if (vy && vx) b += T[(size_t)o * 9 + ky * 3 + kx];
            }
          tab[(size_t)(cy * 3 + cx) * DEC_FEAT + o] = (float)b;
        }
    c.btab = upload(tab);
    c.b = nullptr;
    c.Cout = DEC_FEAT; c.Cin = C; c.CinReal = C; c.KH = c.KW = 3; c.stride = 1; c.pad = 1;
    return c;
  }
  LNW make_ln(const std::string& pfx, int C, float eps) {
    LNW l;
    l.g = upload(get(pfx + ".weight", {C}).data);
    l.b = upload(get(pfx + ".bias", {C}).data);
    l.C = C; l.eps = eps;
    return l;
  }
  // ---- static bounds of tensors no kernel can watch (LayerNorm outputs, the register-only hidden maps of the fused block MLPs): functions of the checkpoint alone
  void note_static(double bound, float limit) { static_window_max = std::max(static_window_max, (float)std::min(3.0e38, bound * 65504.0 / std::max((double)limit, 1e-30))); }
  double ln_bound(const std::string& pfx, int C) {  // |LN(x)_k| <= |gamma_k| sqrt(C) + |beta_k|
    const std::vector<float>& g = host.at(pfx + ".weight").data;
    const std::vector<float>& b = host.at(pfx + ".bias").data;
    double m = 0.0;
    for (int k = 0; k < C; ++k) m = std::max(m, std::fabs((double)g[k]) * std::sqrt((double)C) + std::fabs((double)b[k]));
    return m;
  }
  // hidden = GELU(dw3x3?(fc1(LN(x)))): max_n { sum_k |W1[n][k]| (|gamma_k| sqrt(C) + |beta_k|) + |b1[n]| }, through the depthwise conv when there is one
  double mlp_hidden_bound(const std::string& ln, const std::string& fc1, const std::string& dw, int C) {
    const std::vector<float>& g = host.at(ln + ".weight").data;
    const std::vector<float>& be = host.at(ln + ".bias").data;
    const std::vector<float>& w = host.at(fc1 + ".weight").data;
    const std::vector<float>& b = host.at(fc1 + ".bias").data;
    const int H = 4 * C;
    double m = 0.0;
    for (int n = 0; n < H; ++n) {
      double a = std::fabs((double)b[n]);
      for (int k = 0; k < C; ++k) a += std::fabs((double)w[(size_t)n * C + k]) * (std::fabs((double)g[k]) * std::sqrt((double)C) + std::fabs((double)be[k]));
      if (!dw.empty()) {
        const std::vector<float>& dwv = host.at(dw + ".weight").data;
        const std::vector<float>& db = host.at(dw + ".bias").data;
        double ws = 0.0;
        for (int t = 0; t < 9; ++t) ws += std::fabs((double)dwv[(size_t)n * 9 + t]);
        a = ws * a + std::fabs((double)db[n]);
      }
      m = std::max(m, a);
    }
    return m;
  }
  DwW make_dw(const std::string& pfx, int C, int K) {
    DwW d;
    const std::vector<float>& wv = get(pfx + ".weight", {C, 1, K, K}).data;
    const std::vector<float>& bv = get(pfx + ".bias", {C}).data;
    d.w = upload(pack_dw(wv.data(), C, K));
    d.b = upload(bv);
    d.C = C;
    // |out| <= sum_k |w_k| max |in| + |b| (and |GELU(x)| <= |x| behind the 3x3): the producer of the input is watched with this limit (ConvW::sat_limit)
    double wsum = 0.0, bmax = 0.0;
    for (int ch = 0; ch < C; ++ch) {
      double a = 0.0;
      for (int k = 0; k < K * K; ++k) a += std::fabs((double)wv[(size_t)ch * K * K + k]);
      wsum = std::max(wsum, a);
      bmax = std::max(bmax, std::fabs((double)bv[ch]));
    }
    d.in_limit = (float)std::min(65504.0, std::max(0.0, 65504.0 - bmax) / std::max(wsum, 1e-30));
    return d;
  }

  RbLin make_rb(const std::string& pfx, int N, int K, int cols) {
    RbLin r;
    std::vector<unsigned short> st;
    std::vector<float> inv;
    rb_pack_w(get(pfx + ".weight", {N, K}).data.data(), N, K, cols, &st, &inv);
    r.w = upload_u16(st);
    r.bytes = st.size() * 2;
    r.inv = upload(inv);
    r.b = upload(get(pfx + ".bias", {N}).data);
    r.N = N; r.K = K;
    return r;
  }

  void build_head(Head& hd, const std::string& name, int nout, bool cls) {
    const std::string p = "persformer_heads." + name + "_head.";
    for (int k = 0; k < 4; ++k) {
      const std::string ks = std::to_string(k + 1);
      if (fold_mlp) {
        hd.fold[k] = make_folded(p + "linear_c" + ks + ".proj", p + "linear_c" + ks + "_proc", MIT_DIMS[k]);
      } else {
        hd.lin[k] = make_linear(p + "linear_c" + ks + ".proj", DEC_EMBED, MIT_DIMS[k]);
        hd.proc[k] = make_conv(p + "linear_c" + ks + "_proc.weight", p + "linear_c" + ks + "_proc.bias", DEC_FEAT, DEC_EMBED, 3, 1, 1);
      }
      const std::string f = p + "fusion" + ks + ".";
      if (k < 3) {
        hd.r1c1[k] = make_conv(f + "resConfUnit1.conv1.weight", f + "resConfUnit1.conv1.bias", DEC_FEAT, DEC_FEAT, 3, 1, 1);
        hd.r1c2[k] = make_conv(f + "resConfUnit1.conv2.weight", f + "resConfUnit1.conv2.bias", DEC_FEAT, DEC_FEAT, 3, 1, 1);
      }
      hd.r2c1[k] = make_conv(f + "resConfUnit2.conv1.weight", f + "resConfUnit2.conv1.bias", DEC_FEAT, DEC_FEAT, 3, 1, 1);
      hd.r2c2[k] = make_conv(f + "resConfUnit2.conv2.weight", f + "resConfUnit2.conv2.bias", DEC_FEAT, DEC_FEAT, 3, 1, 1);
    }
    if (wino_min_hw > 0)  // every 256-channel map of the decoder may be read by a Winograd conv (wino.hip: its input transform adds four values, no clamp)
      for (int k = 0; k < 4; ++k)
        for (ConvW* cw : {&hd.fold[k], &hd.proc[k], &hd.r1c1[k], &hd.r1c2[k], &hd.r2c1[k], &hd.r2c2[k]}) cw->sat_limit = 65504.f / 4.f;
    hd.conv0 = make_conv(p + "conv_fuse_conv0.conv.weight", p + "conv_fuse_conv0.conv.bias", 64, DEC_FEAT + LL_CH, 3, 1, 1);
    hd.conv1 = make_conv(p + "conv_fuse_conv1.conv.weight", p + "conv_fuse_conv1.conv.bias", 32, 64, 3, 1, 1);
    hd.nout = nout;
    const std::string pk = p + "linear_pred_" + name;
    if (cls) {
      hd.predcls = make_conv(pk + ".weight", pk + ".bias", nout, 32, 1, 1, 0);
    } else {
      hd.predw = upload(get(pk + ".weight", {nout, 32, 1, 1}).data);
      hd.predb = upload(get(pk + ".bias", {nout}).data);
    }
  }

  void build() {
    // MiT-B3
    int cin = 3;
    for (int s = 0; s < 4; ++s) {
      const int C = MIT_DIMS[s];
      const std::string pe = "backbone.patch_embed" + std::to_string(s + 1);
      stages[s].pe = make_conv(pe + ".proj.weight", pe + ".proj.bias", C, cin, MIT_PK[s], MIT_PS[s], MIT_PP[s]);
      stages[s].pen = make_ln(pe + ".norm", C, 1e-5f);  // nn.LayerNorm default eps (mix_transformers.py:224)
      if (s == 0 && stem7 && split_bf16 && stem7x7_supported(cin, C, MIT_PK[s], MIT_PS[s], MIT_PP[s])) {   // conv + LayerNorm of the first patch embedding as one kernel (stem7.hip)
        std::vector<unsigned short> wfr;
        std::vector<float> tab;
        stem7x7_pack(get(pe + ".proj.weight", {C, cin, 7, 7}).data.data(), nullptr, get(pe + ".proj.bias", {C}).data.data(), get(pe + ".norm.weight", {C}).data.data(),
                     get(pe + ".norm.bias", {C}).data.data(), &wfr, &tab);
        pe_s7_w = upload_u16(wfr); pe_s7_tab = upload(tab);
      }
      for (int i = 0; i < MIT_DEPTHS[s]; ++i) {
        const std::string b = "backbone.block" + std::to_string(s + 1) + "." + std::to_string(i);
        MitBlock mb;
        mb.n1 = make_ln(b + ".norm1", C, 1e-6f);  // mit_b3 norm_layer eps (:519)
        mb.n2 = make_ln(b + ".norm2", C, 1e-6f);
        // fused LayerNorms (fuse_ln): with spatial reduction norm1 also feeds the sr conv (several pixels per GEMM row) and stays a kernel;
        // without it (stage 4) q and kv are its only consumers.  The sr norm feeds kv only, norm2 feeds fc1 only.
        const bool sr1 = MIT_SR[s] == 1;
        mb.q = make_linear(b + ".attn.q", C, C, nullptr, sr1 ? b + ".norm1" : "", 1e-6f);
        mb.kv = make_linear(b + ".attn.kv", 2 * C, C, nullptr, sr1 ? b + ".norm1" : b + ".attn.norm", sr1 ? 1e-6f : 1e-5f);
        mb.proj = make_linear(b + ".attn.proj", C, C);
        if (MIT_SR[s] > 1) {
          mb.sr = make_conv(b + ".attn.sr.weight", b + ".attn.sr.bias", C, C, MIT_SR[s], MIT_SR[s], 0);
          mb.srn = make_ln(b + ".attn.norm", C, 1e-5f);  // Attention.norm default eps (:89)
        }
        mb.fc1 = make_linear(b + ".mlp.fc1", 4 * C, C, nullptr, b + ".norm2", 1e-6f);
        mb.dw = make_dw(b + ".mlp.dwconv.dwconv", 4 * C, 3);
        mb.fc2 = make_linear(b + ".mlp.fc2", C, 4 * C);
        if (fuse_mit_mlp && mit_mlp_preferred(C, mit_mlp128)) note_static(mlp_hidden_bound(b + ".norm2", b + ".mlp.fc1", b + ".mlp.dwconv.dwconv", C), 65504.f);  // hidden map of the fused Mlp (LDS / registers only)
        mb.q.sat_limit = 8188.f; mb.kv.sat_limit = 4094.f;  // the attention kernel's windows (attn.hip: q x 8, k / v x 16 inside the kernel)
        mb.fc1.sat_limit = mb.dw.in_limit;                   // fc1's output goes through the depthwise 3x3 before fc2 contracts it
        if (rb_chain && split_bf16 && rb_linear_supported(C, C) && MIT_SR[s] > 1) {  // stage 3: the row-block form of the block's linear layers
          mb.rq = make_rb(b + ".attn.q", C, C, 320);
          mb.rkv = make_rb(b + ".attn.kv", 2 * C, C, 320);
          mb.rproj = make_rb(b + ".attn.proj", C, C, 320);
          mb.rfc1 = make_rb(b + ".mlp.fc1", 4 * C, C, 320);
          mb.rfc2 = make_rb(b + ".mlp.fc2", C, 4 * C, 320);
          mb.rq.sat_limit = 8188.f; mb.rkv.sat_limit = 4094.f; mb.rfc1.sat_limit = mb.dw.in_limit;
          {  // proj followed by fc1 as ONE stream (rb_proj_fc1_kernel)
            std::vector<unsigned short> st1, st2;
            std::vector<float> inv1, inv2;
            rb_pack_w(get(b + ".attn.proj.weight", {C, C}).data.data(), C, C, 320, &st1, &inv1);
            rb_pack_w(get(b + ".mlp.fc1.weight", {4 * C, C}).data.data(), 4 * C, C, 320, &st2, &inv2);
            st1.resize(st1.size() - (size_t)4 * 10 * 2 * 512);
            st1.insert(st1.end(), st2.begin(), st2.end());
            mb.rpf.w = upload_u16(st1); mb.rpf.bytes = st1.size() * 2;
          }
          if (rb_srkv_supported(C, MIT_SR[s])) {
            int kwc = 0, kwcp = 0;
            const std::vector<float> srp = pack_conv(get(b + ".attn.sr.weight", {C, C, MIT_SR[s], MIT_SR[s]}).data.data(), C, C, MIT_SR[s], MIT_SR[s], C, nullptr, &kwc, &kwcp);  // [C][ky][kx C + ci]
            std::vector<unsigned short> st1, st2;
            std::vector<float> inv1, inv2;
            rb_pack_w(srp.data(), C, MIT_SR[s] * kwcp, 320, &st1, &inv1);
            rb_pack_w(get(b + ".attn.kv.weight", {2 * C, C}).data.data(), 2 * C, C, 320, &st2, &inv2);
            st1.resize(st1.size() - (size_t)4 * 10 * 2 * 512);  // drop the first stream's read-ahead padding: the kv steps follow directly
            st1.insert(st1.end(), st2.begin(), st2.end());
            mb.rsrkv.w = upload_u16(st1); mb.rsrkv.bytes = st1.size() * 2;
            mb.rsrkv.sr_inv = upload(inv1); mb.rsrkv.sr_b = upload(get(b + ".attn.sr.bias", {C}).data);
            mb.rsrkv.kv_inv = upload(inv2); mb.rsrkv.kv_b = upload(get(b + ".attn.kv.bias", {2 * C}).data);
            mb.qln = make_linear(b + ".attn.q", C, C, nullptr, b + ".norm1", 1e-6f);
            mb.qln.sat_limit = 8188.f;
          }
        }
        if (attn64 && split_bf16 && mit_attn64_supported(C, MIT_HEADS[s], 100) && MIT_SR[s] > 1) {
          std::vector<unsigned short> wfr;
          std::vector<float> tab;
          attn64_pack(get(b + ".norm1.weight", {C}).data.data(), get(b + ".norm1.bias", {C}).data.data(), get(b + ".attn.q.weight", {C, C}).data.data(), get(b + ".attn.q.bias", {C}).data.data(),
                      get(b + ".attn.proj.weight", {C, C}).data.data(), get(b + ".attn.proj.bias", {C}).data.data(), &wfr, &tab);
          mb.a64_w = upload_u16(wfr);
          mb.a64_tab = upload(tab);
        }
        if (thin128 && split_bf16 && thin128_supported(C, C) && MIT_SR[s] > 1) {   // stage 2: q (on norm1's output, which the spatial-reduction conv needs anyway) and proj + residual
          std::vector<unsigned short> wfr;
          std::vector<float> tab;
          thin128_pack(get(b + ".attn.q.weight", {C, C}).data.data(), get(b + ".attn.q.bias", {C}).data.data(), &wfr, &tab);
          mb.tq_w = upload_u16(wfr); mb.tq_tab = upload(tab);
          thin128_pack(get(b + ".attn.proj.weight", {C, C}).data.data(), get(b + ".attn.proj.bias", {C}).data.data(), &wfr, &tab);
          mb.tp_w = upload_u16(wfr); mb.tp_tab = upload(tab);
        }
        if (fuse_mit_mlp && mit_mlp_preferred(C, mit_mlp128)) {
          std::vector<unsigned short> wpk;
          std::vector<float> tab2;
          mit_mlp_pack(get(b + ".mlp.fc1.weight", {4 * C, C}).data.data(), get(b + ".mlp.fc1.bias", {4 * C}).data.data(), get(b + ".norm2.weight", {C}).data.data(),
                       get(b + ".norm2.bias", {C}).data.data(), get(b + ".mlp.dwconv.dwconv.weight", {4 * C, 1, 3, 3}).data.data(),
                       get(b + ".mlp.dwconv.dwconv.bias", {4 * C}).data.data(), get(b + ".mlp.fc2.weight", {C, 4 * C}).data.data(), get(b + ".mlp.fc2.bias", {C}).data.data(), C,
                       &wpk, &tab2);
          mb.mlp_w = upload_u16(wpk);
          mb.mlp_tab = upload(tab2);
        }
        stages[s].blocks.push_back(mb);
      }
      stages[s].norm = make_ln("backbone.norm" + std::to_string(s + 1), C, 1e-6f);
      cin = C;
    }
    // low-level encoder with eval BatchNorm folded (perspectivefields.py:70-83; BN eps 1e-5)
    {
      const std::vector<float>& g = get("ll_enc.bn1.weight", {LL_CH}).data;
      const std::vector<float>& be = get("ll_enc.bn1.bias", {LL_CH}).data;
      const std::vector<float>& mu = get("ll_enc.bn1.running_mean", {LL_CH}).data;
      const std::vector<float>& var = get("ll_enc.bn1.running_var", {LL_CH}).data;
      std::vector<double> sc(LL_CH);
      std::vector<float> bias(LL_CH);
      for (int n = 0; n < LL_CH; ++n) {
        sc[n] = (double)g[n] / std::sqrt((double)var[n] + 1e-5);
        bias[n] = (float)((double)be[n] - (double)mu[n] * sc[n]);
      }
      ll = make_conv("ll_enc.conv1.weight", "", LL_CH, 3, 7, 2, 3, sc.data(), &bias);
      if (stem7 && split_bf16 && stem7x7_supported(3, LL_CH, 7, 2, 3)) {
        std::vector<unsigned short> wfr;
        std::vector<float> tab;
        stem7x7_pack(get("ll_enc.conv1.weight", {LL_CH, 3, 7, 7}).data.data(), sc.data(), bias.data(), nullptr, nullptr, &wfr, &tab);
        ll_s7_w = upload_u16(wfr); ll_s7_tab = upload(tab);
      }
      auto it = host.find("ll_enc.bn1.num_batches_tracked");
      if (it != host.end()) it->second.used = true;
    }
    const bool cls = arch == PF_ARCH_PERSNET_CLS;
    build_head(heads[0], "gravity", cls ? 73 : 2, cls);
    build_head(heads[1], "latitude", cls ? 180 : 1, cls);
    has_param = arch != PF_ARCH_PERSNET_CLS;
    param_in = arch == PF_ARCH_PARAMNET_UNCENTERED ? 64 : NET;
    if (has_param) {
      const std::string p = "param_net.backbone.";
      cnx.nout = 5;
      cnx.stem = make_conv(p + "downsample_layers.0.0.weight", p + "downsample_layers.0.0.bias", CNX_DIMS[0], 3, 4, 4, 0);
      cnx.stemn = make_ln(p + "downsample_layers.0.1", CNX_DIMS[0], 1e-6f);
      for (int i = 1; i < 4; ++i) {
        const std::string d = p + "downsample_layers." + std::to_string(i);
        cnx.dsn[i - 1] = make_ln(d + ".0", CNX_DIMS[i - 1], 1e-6f);
        cnx.ds[i - 1] = make_conv(d + ".1.weight", d + ".1.bias", CNX_DIMS[i], CNX_DIMS[i - 1], 2, 2, 0);
      }
      for (int s = 0; s < 4; ++s) {
        const int C = CNX_DIMS[s];
        for (int j = 0; j < CNX_DEPTHS[s]; ++j) {
          const std::string b = p + "stages." + std::to_string(s) + "." + std::to_string(j);
          CnxBlock cb;
          cb.dw = make_dw(b + ".dwconv", C, 7);
          cb.n = make_ln(b + ".norm", C, 1e-6f);
          cb.pw1 = make_linear(b + ".pwconv1", 4 * C, C, nullptr, b + ".norm", 1e-6f);
          // layer scale gamma folded into pwconv2 (convnext.py:54-55: x = gamma * x)
          const std::vector<float>& gm = get(b + ".gamma", {C}).data;
          std::vector<double> sc(gm.begin(), gm.end());
          cb.pw2 = make_linear(b + ".pwconv2", C, 4 * C, sc.data());
          if (fuse_cnx_mlp && cnx_mlp_preferred(C)) {
            std::vector<unsigned short> wpk;
            std::vector<float> tab;
            cnx_mlp_pack(get(b + ".pwconv1.weight", {4 * C, C}).data.data(), get(b + ".pwconv1.bias", {4 * C}).data.data(), get(b + ".norm.weight", {C}).data.data(),
                         get(b + ".norm.bias", {C}).data.data(), get(b + ".pwconv2.weight", {C, 4 * C}).data.data(), get(b + ".pwconv2.bias", {C}).data.data(), gm.data(), C,
                         &wpk, &tab);
            cb.mlp_w = upload_u16(wpk);
            cb.mlp_tab = upload(tab);
          }
          cnx.blocks[s].push_back(cb);
        }
      }
      // windows of the residual stream: what block j writes is read by block j + 1's depthwise conv (then contracted raw by the LayerNorm-fused pwconv1)
      for (int s = 0; s < 4; ++s) {
        if (s > 0) cnx.ds[s - 1].sat_limit = cnx.blocks[s][0].dw.in_limit;
        for (size_t j = 0; j < cnx.blocks[s].size(); ++j) {
          const float lim = j + 1 < cnx.blocks[s].size() ? cnx.blocks[s][j + 1].dw.in_limit : 65504.f;
          cnx.blocks[s][j].out_limit = lim;
          cnx.blocks[s][j].pw2.sat_limit = lim;
        }
      }
      note_static(ln_bound(p + "downsample_layers.0.1", CNX_DIMS[0]), cnx.blocks[0][0].dw.in_limit);  // the stem's LayerNorm output feeds the first depthwise conv
      for (int s = 0; s < 4; ++s)
        for (int j = 0; j < CNX_DEPTHS[s]; ++j) {
          const std::string b = p + "stages." + std::to_string(s) + "." + std::to_string(j);
          if (cnx.blocks[s][j].mlp_w) note_static(mlp_hidden_bound(b + ".norm", b + ".pwconv1", "", CNX_DIMS[s]), 65504.f);  // hidden map of the fused block MLP (registers only)
        }
      cnx.norm = make_ln(p + "norm", CNX_DIMS[3], 1e-6f);
      cnx.headw = upload(get(p + "head.weight", {cnx.nout, CNX_DIMS[3]}).data);
      cnx.headb = upload(get(p + "head.bias", {cnx.nout}).data);
    }
    for (auto& kv : host)
      if (!kv.second.used) throw fmt("unexpected checkpoint tensor '%s' for this architecture", kv.first.c_str());
    host.clear();
  }

  // ------------------------------------------------------------------ debug forward (shadow taps / range records)
  void tap(Ctx& c, const std::string& name, const float* p, int B, int H, int W, int C) {
    DebugSink* d = c.dbg;
    if (!d || !d->shadow || !p) return;
    const size_t bytes = (size_t)B * H * W * C * 4, off = (d->used + 255) & ~(size_t)255;
    d->taps.push_back({name, off, {B, H, W, C}});
    d->used = off + bytes;
    if (c.dry) return;
    if (d->used > d->cap) { d->overflow = true; return; }
    (void)hipMemcpyAsync(d->buf + off, p, bytes, hipMemcpyDeviceToDevice, c.s);
  }
  void range_in(Ctx& c, const std::string& name, const float* p, size_t elems) {
    DebugSink* d = c.dbg;
    if (!d || !d->range || c.dry || !p || (int)d->ranges.size() >= d->max_ranges) return;
    launch_range_stats(p, (long)elems, d->stats + 4 * d->ranges.size(), c.s);
    d->ranges.push_back({name, (long long)elems});
  }

  // ------------------------------------------------------------------ layer helpers
  struct ConvCall {  // one problem of a (possibly grouped) conv launch
    const ConvW* w; Ten x; Ten y;
    const float* res1 = nullptr; const float* res2 = nullptr; Ten x2 = Ten();
    // fused regression prediction head (ConvPtrs::head_*): kind 1 gravity, 2 latitude
    int head_kind = 0; const float* head_w = nullptr; const float* head_b = nullptr; float* head_out = nullptr; float* head_pn = nullptr;
  };
  // ups: calls[].x is stored at half resolution (H/2 x W/2); the conv runs on its bilinear x2 up-sampling (ConvParams::ups)
  void conv_g(Ctx& c, int ngroups, const ConvCall* calls, int B, int H, int W, int act = ACT_NONE, int post_relu = 0, int C1 = -1, int nchw = 0, int ups = 0) {
    const ConvW& w = *calls[0].w;
    const size_t Ho_ = (H + 2 * w.pad - w.KH) / w.stride + 1, Wo_ = (W + 2 * w.pad - w.KW) / w.stride + 1;
    // split-K scratch (conv_splitk_factor: deep-K launches with too few tiles for 256 CUs -- the MiT spatial-reduction convs); decided
    // from the shape and the call's operand set, so that the workspace dry run takes the same decision
    int splitk = 1;
    {
      bool plain = !nchw && !ups && !w.ln_s && w.Cin % 32 == 0 && w.KWCp > 0;
      for (int g = 0; g < ngroups; ++g)
        if (calls[g].head_kind || calls[g].w->btab || calls[g].res2 || calls[g].y.s.p || !calls[g].y.f || calls[g].x.s.p) plain = false;
      if (plain && split_bf16) splitk = conv_splitk_shape((long)B * Ho_ * Wo_, w.Cout, w.KH, w.KWCp, ngroups);
    }
    const size_t mk_part = c.mark();
    float* part = splitk > 1 ? c.alloc((size_t)splitk * ngroups * B * Ho_ * Wo_ * w.Cout) : nullptr;
    struct Rel { Ctx& c; size_t m; ~Rel() { c.release(m); } } rel{c, mk_part};  // stream order makes the reuse safe
    if (c.dry) {
      c.max_conv_out = std::max(c.max_conv_out, (size_t)ngroups * B * Ho_ * Wo_ * w.Cout);
      return;
    }
    ConvParams p;
    p.groups = ngroups;
    for (int g = 0; g < ngroups; ++g) {
      const ConvW& wg = *calls[g].w;
      ConvPtrs& q = p.g[g];
      q.x = calls[g].x.f; q.x2 = calls[g].x2.f; q.w = wg.w; q.w_sb = wg.wsb; q.w_h16 = wg.wh16; q.w_h16_inv_scale = wg.wh16_inv; q.bias = wg.b; q.bias_tab = wg.btab;
      q.w_wino = wg.wwino; q.w_wino_inv = wg.wwino_inv;
      q.res1 = calls[g].res1; q.res2 = calls[g].res2; q.y = calls[g].y.f;
      q.x_sb = calls[g].x.s.p; q.x2_sb = calls[g].x2.s.p; q.y_sb = calls[g].y.s.p;
      q.ln_colsum = wg.ln_s;
      q.head_kind = calls[g].head_kind; q.head_w = calls[g].head_w; q.head_b = calls[g].head_b; q.head_out = calls[g].head_out; q.head_pn = calls[g].head_pn;
    }
    p.x_sb_plane = calls[0].x.s.plane; p.x2_sb_plane = calls[0].x2.s.plane; p.y_sb_plane = calls[0].y.s.plane;
    p.B = B; p.H = H; p.W = W;
    p.C1 = C1 < 0 ? w.Cin : C1; p.C2 = w.Cin - p.C1;
    p.KH = w.KH; p.KW = w.KW; p.stride = w.stride; p.pad = w.pad;
    p.Cout = w.Cout; p.KWC = w.KWC; p.KWCp = w.KWCp;
    p.act = act; p.post_relu = post_relu; p.nchw_out = nchw;
    p.nterms = nterms;
    p.ups = ups;
    p.ln = w.ln_s ? 1 : 0; p.ln_eps = w.ln_eps;
    p.sat = (c.tuning || c.dry) ? nullptr : d_sat; p.sat_limit = w.sat_limit;
    for (int g = 1; g < ngroups; ++g) p.sat_limit = std::min(p.sat_limit, calls[g].w->sat_limit);
    p.finish();
    if (part) {
      p.splitk = splitk;
      for (int g = 0; g < ngroups; ++g) p.g[g].partial = part + (size_t)g * splitk * p.M * p.Cout;
    }
    int tile = -1;
    {
      // operand formats are part of the key: a split-plane input changes which tile is fastest
      const int prec_code = nterms == NT_F16X3 ? 0 : 3;  // = PF_PRECISION_*
      const int fmt_bits = (calls[0].res1 ? 1 : 0) + (calls[0].res2 ? 2 : 0) + (calls[0].x.s.p ? 4 : 0) + (calls[0].y.s.p ? 8 : 0) + (calls[0].y.f ? 0 : 16) + 32 * prec_code + 128 * ups + 256 * (calls[0].head_kind ? 1 : 0);
      std::vector<int> key = {p.M, p.Cout, p.KH, p.KW, p.Cin, p.stride, p.H, p.W, ngroups, p.nchw_out, fmt_bits + 512 * p.ln, p.act};
      auto it = tile_cache.find(key);
      if (it == tile_cache.end() && p.ln && !(c.tuning && c.tune_scratch)) { key[10] = fmt_bits; it = tile_cache.find(key); }  // table without the fused form: same shape's tile
      if (it != tile_cache.end()) tile = it->second;
      else if (c.tuning && c.tune_scratch) { tile = tune_conv(p, c); tile_cache[key] = tile; }
    }
    if (!conv_tile_usable(p, tile)) tile = conv_default_tile(p);
    // Winograd form where it exists and the map is large enough (the tile table knows the direct tiles only); never while tuning (tune_conv times the direct tiles)
    if (wino_min_hw > 0 && wino_tile >= 0 && !(c.tuning && c.tune_scratch) && p.Ho >= wino_min_hw && p.Wo >= wino_min_hw && conv_tile_usable(p, wino_tile) &&
        conv_wino_blocks(p) >= wino_min_blocks) tile = wino_tile;
    if (c.dbg && c.dbg->range) {
      const int q = c.dbg->seq++;
      for (int g = 0; g < ngroups; ++g) {
        // "[winograd]": the layer runs as Winograd F(2x2, 3x3) -- its input transform adds four values, so its window ends at 65504 / 4 (PerspectiveFields._window_limit)
        const std::string nm = fmt("dense%03d%s %dx%d s%d %s%sM=%d N=%d K=%d", q, ngroups > 1 ? (g ? "[latitude]" : "[gravity]") : "", w.KH, w.KW, w.stride, w.ln_s ? "LN-fused " : "", tile >= 0 && strncmp(conv_tile_name(tile), "wino", 4) == 0 ? "[winograd] " : "", p.M, p.Cout, w.KH * w.KW * w.CinReal);
        range_in(c, nm + " x", calls[g].x.f, (size_t)B * (ups ? H / 2 : H) * (ups ? W / 2 : W) * p.C1);
        if (p.C2 > 0) range_in(c, nm + " x2", calls[g].x2.f, (size_t)B * H * W * p.C2);
      }
    }
    ProfScope ps(c.prof, c.s, conv_tile_is_sb(tile) ? PC_IGEMM_SB : PC_IGEMM, 2.0 * ngroups * p.M * (double)w.Cout * w.KH * w.KW * w.CinReal, p.M * ngroups, w.Cout, w.KH * w.KW * w.CinReal, w.KH);
    launch_conv_tile(p, tile, c.s);
  }
  // Measure, don't guess: run the launch with every tile configuration (outputs redirected to scratch so in-place
  // residual layers are not disturbed) and keep the fastest.  Results are identical across tiles (same K order).
  int tune_conv(const ConvParams& p0, Ctx& c) {
    ConvParams p = p0;
    const size_t out_elems = (size_t)p.M * p.Cout;
    unsigned short* sb_scratch = reinterpret_cast<unsigned short*>(c.tune_scratch + c.tune_scratch_elems);
    for (int g = 0; g < p.groups; ++g) {
      if (p.g[g].y) p.g[g].y = c.tune_scratch + g * out_elems;
      if (p.g[g].y_sb) p.g[g].y_sb = sb_scratch + g * out_elems;
    }
    p.y_sb_plane = (c.tune_scratch_elems & ~(size_t)1) | (p0.y_sb_plane & 1);
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
    int best = -1;
    float best_ms = 1e30f;
    for (int t = 0; t < conv_num_tiles(); ++t) {
      if (!conv_tile_usable(p, t)) continue;
      if (strncmp(conv_tile_name(t), "sbhA", 4) == 0 || strncmp(conv_tile_name(t), "sbhLA", 5) == 0 || strncmp(conv_tile_name(t), "sbhDMA", 6) == 0 || strncmp(conv_tile_name(t), "sbhREG", 6) == 0 || strncmp(conv_tile_name(t), "sbhV", 4) == 0 || strncmp(conv_tile_name(t), "sbA", 3) == 0 || strncmp(conv_tile_name(t), "sbPI_", 5) == 0 || strncmp(conv_tile_name(t), "sbI_", 4) == 0) continue;  // tuning builds: ablation forms (wrong results by construction) are for scripts/tune_conv.py only
      if (strncmp(conv_tile_name(t), "wino", 4) == 0) continue;  // the Winograd forms are chosen by map size (wino_min_hw), never by the table: a tuned-in Winograd tile on a small map would run with the
                                                                 // 65504 / 4 window while the range records tag only `wino_tile` as "[winograd]"
      if (conv_tile_bn(t) > 32 && p.Cout <= 32) continue;
      if ((long)conv_tile_bm(t) * conv_tile_bn(t) > 16L * p.M * p.Cout) continue;  // tile far larger than the problem
      launch_conv_tile(p, t, c.s);  // warm-up (instruction cache, L2 state)
      float ms = 1e30f;
      bool ok = true;
      for (int rep = 0; rep < 2 && ok; ++rep) {  // best of two windows of two launches: one noisy window must not pick the tile
        (void)hipEventRecord(a, c.s);
        launch_conv_tile(p, t, c.s);
        launch_conv_tile(p, t, c.s);
        (void)hipEventRecord(b, c.s);
        ok = hipEventSynchronize(b) == hipSuccess;
        float w = 0.f;
        if (ok) { (void)hipEventElapsedTime(&w, a, b); ms = std::min(ms, w); }
      }
      if (!ok) break;
      if (ms < best_ms) { best_ms = ms; best = t; }
    }
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return best;
  }
  void conv(Ctx& c, const ConvW& w, Ten x, int B, int H, int W, Ten y, int act = ACT_NONE, const float* res1 = nullptr,
            const float* res2 = nullptr, int post_relu = 0, Ten x2 = Ten(), int C1 = -1, int nchw = 0) {
    ConvCall one{&w, x, y, res1, res2, x2};
    conv_g(c, 1, &one, B, H, W, act, post_relu, C1, nchw);
  }
  void gemm(Ctx& c, const ConvW& w, Ten x, long rows, Ten y, int act = ACT_NONE, const float* res1 = nullptr) {
    conv(c, w, x, 1, (int)rows, 1, y, act, res1);
  }
  // y.f may alias x; y may carry fp32, split planes, or both
  void ln(Ctx& c, const LNW& l, const float* x, Ten y, long rows) {
    if (c.dry) return;
    ProfScope ps(c.prof, c.s, PC_LAYERNORM, (4.0 + (y.f ? 4.0 : 0.0) + (y.s.p ? 6.0 : 0.0)) * rows * l.C);
    launch_layernorm(x, l.g, l.b, y.f, rows, l.C, l.eps, c.s, y.s.p, y.s.plane);
  }

  // a linear layer in the row-block form: y = act(LN?(x) W^T + b) + res on blocks of 64 tokens of one image (rb_gemm.hip)
  void rb_linear(Ctx& c, const RbLin& r, const float* x, long M, int tokens, float* y, const LNW* lnw = nullptr, int act = ACT_NONE, const float* res = nullptr) {
    range_in(c, fmt("rb_linear %d->%d%s", r.K, r.N, lnw ? " (LN input)" : ""), x, (size_t)M * r.K);
    if (c.dry) return;
    RbLinArgs a;
    a.x = x; a.ln_g = lnw ? lnw->g : nullptr; a.ln_b = lnw ? lnw->b : nullptr; a.ln_eps = lnw ? lnw->eps : 0.f;
    a.w = r.w; a.w_bytes = r.bytes; a.inv = r.inv; a.bias = r.b; a.res = res; a.y = y;
    a.M = (int)M; a.tokens = tokens; a.bpi = (tokens + 63) / 64; a.N = r.N; a.act = act;
    a.sat = d_sat; a.sat_limit = r.sat_limit;
    ProfScope ps(c.prof, c.s, PC_IGEMM_SB, 2.0 * M * (double)r.N * r.K, (int)M, r.N, r.K, 1);
    launch_rb_linear(a, r.K, c.s);
  }

  // a 128 -> 128 linear layer (+ residual) in the thin form (thin_linear.hip)
  void thin(Ctx& c, const unsigned short* wfr, const float* tab, const float* x, const float* res, float* y, long M, float sat_limit, const char* what) {
    range_in(c, fmt("thin128 %s x", what), x, (size_t)M * 128);
    if (c.dry) return;
    ThinLinArgs a;
    a.x = x; a.res = res; a.y = y; a.wfr = wfr; a.tab = tab; a.M = M; a.sat = d_sat; a.sat_limit = sat_limit;
    ProfScope ps(c.prof, c.s, PC_IGEMM_SB, 2.0 * M * 128.0 * 128.0, (int)M, 128, 128, 1);
    launch_thin128(a, num_cus, c.s);
  }

  // conv 7x7 on the normalised image (+ ReLU or + LayerNorm) as one launch of stem7.hip
  void stem(Ctx& c, const unsigned short* wfr, const float* tab, const float* x4, float* y, int B, int H, int W, int stride, bool relu, bool lnorm, float eps, float sat_limit, const char* what) {
    range_in(c, fmt("stem7x7 s%d %s x", stride, what), x4, (size_t)B * H * W * 4);
    if (c.dry) return;
    Stem7Args a;
    a.x = x4; a.y = y; a.wfr = wfr; a.tab = tab; a.B = B; a.H = H; a.W = W; a.stride = stride; a.Ho = (H + 6 - 7) / stride + 1; a.Wo = (W + 6 - 7) / stride + 1;
    a.relu = relu ? 1 : 0; a.ln = lnorm ? 1 : 0; a.ln_eps = eps; a.sat = lnorm ? nullptr : d_sat; a.sat_limit = sat_limit;
    const long M = (long)B * a.Ho * a.Wo;
    ProfScope ps(c.prof, c.s, PC_IGEMM_SB, 2.0 * M * 64.0 * 147.0, (int)M, 64, 147, 7);
    launch_stem7x7(a, num_cus, c.s);
  }

  // MiT-B3 forward_features (mix_transformers.py:449-485); feats[s] = NHWC stage outputs.
  // With `sba` every tensor that only feeds GEMMs (LayerNorm outputs, attention output, the GELU'd hidden map) is
  // written as split planes by its producer and never exists in fp32.
  void mit(Ctx& c, int B, const float* x0, Ten feats[4], const Ten* llf = nullptr) {
    const bool S = sba;
    Ten cur(const_cast<float*>(x0));
    int H = NET, W = NET;
    for (int s = 0; s < 4; ++s) {
      MitStage& st = stages[s];
      if (!c.dry && pn_later.armed && s + 1 == defer_at) issue_deferred();  // the previous forward's ParamNet branch goes beside this stage
      const int C = MIT_DIMS[s], heads_n = MIT_HEADS[s], sr = MIT_SR[s];
      const int Ho = (H + 2 * MIT_PP[s] - MIT_PK[s]) / MIT_PS[s] + 1, Wo = (W + 2 * MIT_PP[s] - MIT_PK[s]) / MIT_PS[s] + 1;
      const long N = (long)Ho * Wo, M = (long)B * N;
      float* x = c.alloc(M * C);  // token stream, updated in place by the residual epilogues
      // the fused Mlp (mit_mlp.hip) reads the halo rows of neighbouring blocks: it writes a second buffer and the two swap roles
      const bool fused_mlp = !st.blocks.empty() && st.blocks[0].mlp_w && nterms == NT_F16X3 && (C != 128 || B >= mit_mlp128);
      float* xalt = fused_mlp ? c.alloc(M * C) : nullptr;
      const SbT xs = S ? c.alloc_sb(M * C) : SbT();  // split copy of the stage output (next patch embed + decoder)
      if (s == 2 && llf && B >= 4 && side_stream_mode >= 2 && can_fork(c)) {  // PF_SIDE_STREAM=2 (measured: no gain)  // the low-level encoder conv (a full-chip launch of its own) next to the small launches of stages 3 / 4
        (void)hipEventRecord(ev_ll, c.s);  // x0 is long since ready; the event only orders the side stream behind this forward's beginning
        (void)hipStreamWaitEvent(side2, ev_ll, 0);
        Ctx c2 = c;
        c2.s = side2;
        conv(c2, ll, Ten(const_cast<float*>(x0)), B, NET, NET, *llf, ACT_RELU);
        (void)hipEventRecord(ev_ll, side2);
        ll_forked = true;
      }
      if (s == 0 && pe_s7_w && nterms == NT_F16X3 && !c.tuning && cur.f) {
        stem(c, pe_s7_w, pe_s7_tab, cur.f, x, B, H, W, MIT_PS[s], false, true, st.pen.eps, 65504.f, "patch_embed1 + norm");
      } else {
        conv(c, st.pe, cur, B, H, W, Ten(x));
        ln(c, st.pen, x, Ten(x), M);
      }
      const size_t mk = c.mark();
      const int kvh = Ho / sr, kvw = Wo / sr;
      const long Mkv = (long)B * kvh * kvw;
      const Ten xn = c.ten(M * C, !S, S);
      float* qb = c.alloc(M * C);
      const Ten ab = c.ten(M * C, !S, S);
      float* srb = c.alloc(Mkv * C);
      const Ten srn = S ? Ten(nullptr, c.alloc_sb(Mkv * C)) : Ten(srb);  // LN(sr conv): in place, or planes only
      float* kvb = c.alloc(Mkv * 2 * C);
      float* hb = c.alloc(M * 4 * C);
      const Ten h2 = c.ten(M * 4 * C, !S, S);
      // One transformer block on `B` images starting at token row 0 of the given buffers (Block.forward, mix_transformers.py:198-202).  gate_B: the batch the
      // row-block gate is taken for (the whole batch, also when the block is issued for one half of it -- see the stage-3 split below)
      auto one_block = [&](Ctx& c, MitBlock& mb, int blk, int B, long M, long Mkv, float*& x, float*& xalt, const Ten& xn, float* qb, const Ten& ab, float* srb, const Ten& srn,
                           float* kvb, float* hb, const Ten& h2, int gate_B, bool may_fork) {
        // x += proj(attn(LN1(x)))            (Block.forward :199; Attention.forward :108-141)
        // row-block form of the block's linear layers (stage 3 at batch >= ~14): q, kv, proj, fc1, fc2
        // one 64-row block per CU: the form pays only when the last round of blocks nearly fills the 256 CUs (same-box A/Bs, profiles/r04_rb_linear.md: B = 32 -> 224 blocks
        // +1.2 %, B = 64 -> 448 +1.2 %; B = 48 -> 336 -0.4 %, B = 24 -> 168 -0.9 %, B = 16 -> 112 -4.5 %)
        const bool thin_q = mb.tq_w && M >= thin128 && nterms == NT_F16X3 && !S && !c.tuning && xn.f;   // (thin_linear.hip; the same conditions hold for the projection below)
        const bool fuse64 = attn64 && mb.a64_w && nterms == NT_F16X3 && !S && !c.tuning && sr > 1;
        const long rb_blocks = (long)gate_B * ((N + 63) / 64);
        const bool use_rb = rb_chain && mb.rq.w && nterms == NT_F16X3 && !S && !c.tuning && rb_blocks >= rb_min_blocks && (rb_blocks % num_cus == 0 || rb_blocks % num_cus >= num_cus * 3 / 4);
        if (sr > 1 && use_rb && (rb_chain & 32) && mb.rsrkv.w && mb.qln.ln_s && fuse_ln) {
          // key / value branch in ONE launch (LayerNorm-1 of the gathered source tokens, 2 x 2 conv, LayerNorm, kv: rb_chain.hip); q with LayerNorm-1 folded into its
          // GEMM beside it: no LayerNorm-1 launch, no split-K conv + reduce, no normalised map in HBM
          const bool fork = B >= 4 && (may_fork && can_fork(c));
          if (fork) {
            (void)hipEventRecord(ev_fork, c.s);
            (void)hipStreamWaitEvent(side, ev_fork, 0);
            Ctx c2 = c;
            c2.s = side;
            gemm(c2, mb.qln, Ten(x), M, Ten(qb));
            (void)hipEventRecord(ev_join, side);
          } else {
            gemm(c, mb.qln, Ten(x), M, Ten(qb));
          }
          range_in(c, fmt("rb_srkv s%d.b%d x (LN-fused)", s + 1, blk), x, (size_t)M * C);
          if (!c.dry) {
            RbSrKvArgs a;
            a.x = x; a.ln1_g = mb.n1.g; a.ln1_b = mb.n1.b; a.ln1_eps = mb.n1.eps;
            a.w = mb.rsrkv.w; a.w_bytes = mb.rsrkv.bytes; a.sr_inv = mb.rsrkv.sr_inv; a.sr_bias = mb.rsrkv.sr_b;
            a.srn_g = mb.srn.g; a.srn_b = mb.srn.b; a.srn_eps = mb.srn.eps; a.kv_inv = mb.rsrkv.kv_inv; a.kv_bias = mb.rsrkv.kv_b;
            a.kv = kvb; a.B = B; a.Hr = kvh; a.Wr = kvw; a.bpi = (kvh * kvw + 31) / 32; a.sat = d_sat;
            ProfScope ps(c.prof, c.s, PC_IGEMM_SB, 2.0 * Mkv * (double)C * (sr * sr * C + 2 * C), (int)Mkv, 3 * C, sr * sr * C + 2 * C, sr);
            launch_rb_srkv(a, C, c.s);
          }
        } else if (sr > 1 && fuse64) {
          // one head of 64 channels (stage 1): q, the attention core, proj and the residual run as ONE kernel below (attn_block.hip); norm1's output is still needed by
          // the spatial-reduction conv
          ln(c, mb.n1, x, xn, M);
          conv(c, mb.sr, xn, B, Ho, Wo, Ten(srb));
          if (mb.kv.ln_s) {
            gemm(c, mb.kv, Ten(srb), Mkv, Ten(kvb));  // LayerNorm(sr conv) inside the kv GEMM
          } else {
            ln(c, mb.srn, srb, srn, Mkv);
            gemm(c, mb.kv, srn, Mkv, Ten(kvb));
          }
        } else if (sr > 1) {
          const bool fork = B >= 4 && (may_fork && can_fork(c));  // batch 1-3: a launch already under-fills the chip; the event pair would only add latency
          if (use_rb && (rb_chain & 1) && fork) {  // q = LN1(x) Wq with the LayerNorm inside the kernel: independent of the LayerNorm launch below
            (void)hipEventRecord(ev_fork, c.s);
            (void)hipStreamWaitEvent(side, ev_fork, 0);
            Ctx c2 = c;
            c2.s = side;
            rb_linear(c2, mb.rq, x, M, (int)N, qb, &mb.n1);
            (void)hipEventRecord(ev_join, side);
          }
          ln(c, mb.n1, x, xn, M);
          if (use_rb && (rb_chain & 1)) {
            if (!fork) rb_linear(c, mb.rq, x, M, (int)N, qb, &mb.n1);
          } else if (fork) {  // q projection next to sr conv + kv GEMM: both read xn, attention needs both
            (void)hipEventRecord(ev_fork, c.s);
            (void)hipStreamWaitEvent(side, ev_fork, 0);
            Ctx c2 = c;
            c2.s = side;
            if (thin_q) thin(c2, mb.tq_w, mb.tq_tab, xn.f, nullptr, qb, M, mb.q.sat_limit, "q"); else gemm(c2, mb.q, xn, M, Ten(qb));
            (void)hipEventRecord(ev_join, side);
          } else {
            if (thin_q) thin(c, mb.tq_w, mb.tq_tab, xn.f, nullptr, qb, M, mb.q.sat_limit, "q"); else gemm(c, mb.q, xn, M, Ten(qb));
          }
          conv(c, mb.sr, xn, B, Ho, Wo, Ten(srb));
          if (use_rb && (rb_chain & 2)) {  // measured slower than the LDS tile on these 3 200 rows (50 row blocks: 20.6 vs 12.7 us, profiles/r04_rb_linear.md)
            rb_linear(c, mb.rkv, srb, Mkv, kvh * kvw, kvb, &mb.srn);  // LayerNorm(sr conv) while the rows are staged
          } else if (mb.kv.ln_s) {
            gemm(c, mb.kv, Ten(srb), Mkv, Ten(kvb));  // LayerNorm(sr conv) inside the kv GEMM
          } else {
            ln(c, mb.srn, srb, srn, Mkv);
            gemm(c, mb.kv, srn, Mkv, Ten(kvb));
          }
        } else if (mb.q.ln_s && mb.kv.ln_s) {
          if (B >= 4 && (may_fork && can_fork(c))) {                // norm1 inside both of its consumers; q next to kv
            (void)hipEventRecord(ev_fork, c.s);
            (void)hipStreamWaitEvent(side, ev_fork, 0);
            Ctx c2 = c;
            c2.s = side;
            gemm(c2, mb.q, Ten(x), M, Ten(qb));
            (void)hipEventRecord(ev_join, side);
          } else {
            gemm(c, mb.q, Ten(x), M, Ten(qb));
          }
          gemm(c, mb.kv, Ten(x), M, Ten(kvb));
        } else {
          ln(c, mb.n1, x, xn, M);
          gemm(c, mb.q, xn, M, Ten(qb));
          gemm(c, mb.kv, xn, M, Ten(kvb));
        }
        if (!fuse64 && B >= 4 && (sr > 1 || (mb.q.ln_s && mb.kv.ln_s)) && (may_fork && can_fork(c))) (void)hipStreamWaitEvent(c.s, ev_join, 0);
        if (c.dbg && c.dbg->range) {
          if (!fuse64) range_in(c, fmt("attention s%d.b%d q", s + 1, blk), qb, (size_t)M * C);   // (fused: q never leaves the registers; the kernel watches it against the same window)
          range_in(c, fmt("attention s%d.b%d kv", s + 1, blk), kvb, (size_t)Mkv * 2 * C);
        }
        if (fuse64) {
          range_in(c, fmt("mit_attn64 s%d.b%d x (LN-fused)", s + 1, blk), x, (size_t)M * C);
          if (!c.dry) {
            MitAttn64Args a;
            a.x = x; a.kv = kvb; a.y = x; a.wfr = mb.a64_w; a.tab = mb.a64_tab; a.B = B; a.N = (int)N; a.M = kvh * kvw; a.ln_eps = mb.n1.eps; a.sat = d_sat; a.sat_limit = mb.proj.sat_limit;
            // work = q + proj (2 x 2 M C C) + QK^T + PV (4 M C kv)
            ProfScope ps(c.prof, c.s, PC_ATTN, 4.0 * M * C * (double)C + 4.0 * M * C * (kvh * kvw));
            launch_mit_attn64(a, num_cus, c.s);
          }
        }
        const bool pf_fused = !fuse64 && use_rb && (rb_chain & 64) && mb.rpf.w;  // x += proj(attn); hidden = fc1(LN2(x)) in one launch (rb_chain.hip)
        if (!fuse64) {
        if (!c.dry) {
          ProfScope ps(c.prof, c.s, PC_ATTN, 4.0 * M * C * (kvh * kvw));  // QK^T + PV
          launch_sr_attention(qb, kvb, ab.f, B, (int)N, kvh * kvw, heads_n, c.s, ab.s.p, ab.s.plane);
        }
        if (pf_fused) {
          range_in(c, fmt("rb_proj_fc1 s%d.b%d attn", s + 1, blk), ab.f, (size_t)M * C);
          if (!c.dry) {
            RbProjFc1Args a;
            a.attn = ab.f; a.x = x; a.w = mb.rpf.w; a.w_bytes = mb.rpf.bytes; a.proj_inv = mb.rproj.inv; a.proj_bias = mb.rproj.b;
            a.ln2_g = mb.n2.g; a.ln2_b = mb.n2.b; a.ln2_eps = mb.n2.eps; a.fc1_inv = mb.rfc1.inv; a.fc1_bias = mb.rfc1.b; a.hidden = hb;
            a.B = B; a.tokens = (int)N; a.bpi = ((int)N + 63) / 64; a.sat = d_sat;
            ProfScope ps(c.prof, c.s, PC_IGEMM_SB, 2.0 * M * (double)C * 5 * C, (int)M, 5 * C, C, 1);
            launch_rb_proj_fc1(a, C, c.s);
          }
        } else if (use_rb && (rb_chain & 4)) rb_linear(c, mb.rproj, ab.f, M, (int)N, x, nullptr, ACT_NONE, x);
        else if (thin_q && mb.tp_w && ab.f) thin(c, mb.tp_w, mb.tp_tab, ab.f, x, x, M, mb.proj.sat_limit, "proj");
        else gemm(c, mb.proj, ab, M, Ten(x), ACT_NONE, x);
        }  // !fuse64
        // x += fc2(gelu(dwconv(fc1(LN2(x)))))   (:200; Mlp.forward :49-56)
        if (fused_mlp && mb.mlp_w) {                  // norm2 + fc1 + depthwise 3x3 + GELU + fc2 + residual in one kernel, x -> xalt
          range_in(c, fmt("mit_mlp s%d.b%d x (LN-fused)", s + 1, blk), x, (size_t)M * C);
          if (!c.dry) {
            ProfScope ps(c.prof, c.s, PC_IGEMM_SB, 2.0 * 2.0 * M * (double)C * 4 * C, (int)M, C, 8 * C, 9);
            launch_mit_mlp(x, xalt, mb.mlp_w, mb.mlp_tab, B, Ho, Wo, C, mb.n2.eps, c.s, d_sat, 65504.f);
          }
          std::swap(x, xalt);
          return;
        }
        if (pf_fused) {
        } else if (use_rb && (rb_chain & 8)) {
          rb_linear(c, mb.rfc1, x, M, (int)N, hb, &mb.n2);  // norm2 while the rows are staged
        } else if (mb.fc1.ln_s) {
          gemm(c, mb.fc1, Ten(x), M, Ten(hb));        // norm2 inside fc1
        } else {
          ln(c, mb.n2, x, xn, M);
          gemm(c, mb.fc1, xn, M, Ten(hb));
        }
        if (!c.dry) {
          ProfScope ps(c.prof, c.s, PC_DW3, (4.0 + (h2.f ? 4.0 : 0.0) + (h2.s.p ? 6.0 : 0.0)) * M * 4 * C);  // read + write of the hidden map
          launch_dwconv3x3_gelu(hb, mb.dw.w, mb.dw.b, h2.f, B, Ho, Wo, 4 * C, c.s, h2.s.p, h2.s.plane);
        }
        if (use_rb && (rb_chain & 16)) rb_linear(c, mb.rfc2, h2.f, M, (int)N, x, nullptr, ACT_NONE, x);
        else gemm(c, mb.fc2, h2, M, Ten(x), ACT_NONE, x);
            };
      // PF_S3_SPLIT (stage 3 only, row-block form, batch >= 16 and even): the stage's 18 blocks are chains of ~6 dependent launches of <= 224 blocks each, every one
      // a single round whose time is a block's latency (prologue, K loop at one wave per SIMD, epilogue) -- the chip idles in every prologue, epilogue and launch gap.
      // Images are independent, so the batch is cut in two halves that walk the stage on TWO streams (the caller's and `side`, which gives up the q-beside-kv fork
      // for it: no additional hardware queue): two desynchronised chains of half-size launches fill each other's gaps.  Same kernels, same per-image arithmetic:
      // bit-identical results (tests/test_gpu_e2e.py::test_stage3_batch_split_is_bit_identical).
      // Measured (same box, profiles/r06_s3_split.md): batch 64 (two halves of 224 blocks, each a full round of its own) +2.2 %; batch 32 (halves of 112 blocks) -0.3 ... -1.8 %:
      // a half-size launch takes as long as the full one (a launch IS a block's latency), so the split pays only when EACH half still fills the chip -- the gate below
      // is the row-block form's gate taken for the half batch (PF_S3_SPLIT=2 forces the split whenever the whole batch passes it: the batch-32 A/B).
      const long rb_all = (long)B * ((N + 63) / 64), rb_half = rb_all / 2;
      auto rb_gate = [&](long nb) { return nb >= rb_min_blocks && (nb % num_cus == 0 || nb % num_cus >= num_cus * 3 / 4); };
      const bool split = s3_split && s == 2 && sr > 1 && !S && !fused_mlp && B >= 16 && B % 2 == 0 && !c.dry && !c.dbg && !c.tuning && !st.blocks.empty() && rb_chain && nterms == NT_F16X3 &&
                         st.blocks[0].rq.w && rb_gate(rb_all) && (s3_split >= 2 || rb_gate(rb_half)) && can_fork(c);
      int blk = -1;
      if (split) {
        const int Bh = B / 2;
        const long Mh = M / 2, Mkvh = Mkv / 2;
        (void)hipEventRecord(ev_fork, c.s);
        (void)hipStreamWaitEvent(side, ev_fork, 0);
        Ctx c2 = c;
        c2.s = side;
        float* x2 = x + Mh * C;
        float* none = nullptr;
        for (MitBlock& mb : st.blocks) {
          ++blk;
          one_block(c, mb, blk, Bh, Mh, Mkvh, x, none, xn, qb, ab, srb, srn, kvb, hb, h2, B, false);
          one_block(c2, mb, blk, Bh, Mh, Mkvh, x2, none, Ten(xn.f ? xn.f + Mh * C : nullptr), qb + Mh * C, Ten(ab.f + Mh * C), srb + Mkvh * C, Ten(srn.f ? srn.f + Mkvh * C : nullptr),
                    kvb + Mkvh * 2 * C, hb + Mh * 4 * C, Ten(h2.f + Mh * 4 * C), B, false);
        }
        (void)hipEventRecord(ev_join, side);
        (void)hipStreamWaitEvent(c.s, ev_join, 0);
      } else {
        for (MitBlock& mb : st.blocks) {
          if (blk >= 0) tap(c, fmt("mit.s%d.b%d", s + 1, blk), x, B, Ho, Wo, C);  // the previous block's output (token stream, pre stage norm)
          ++blk;
          one_block(c, mb, blk, B, M, Mkv, x, xalt, xn, qb, ab, srb, srn, kvb, hb, h2, B, true);
        }
      }
      c.release(mk);
      if (blk >= 0) tap(c, fmt("mit.s%d.b%d", s + 1, blk), x, B, Ho, Wo, C);
      ln(c, st.norm, x, Ten(x, xs), M);  // stage norm; the normalised map is both the output and the next stage's input (:457-462)
      tap(c, fmt("c%d", s + 1), x, B, Ho, Wo, C);
      feats[s] = Ten(x, xs);
      cur = feats[s];
      H = Ho; W = Wo;
    }
  }

  // Both decoder heads up to their 32-channel 320x320 maps (gravity_head.py:139-173 / latitude_head.py:138-172).
  // The two heads have identical shapes and independent weights, so every conv runs as ONE grouped launch
  // (2x the grid: fewer partially filled last waves, half the launches); their tensors are allocated as
  // [2][B][h][w][C] pairs so the bilinear kernels simply see a batch of 2B.
  // Stored tensors are post-ReLU wherever every consumer applies ReLU first (ResidualConvUnit's in-place
  // ReLU, decode_head.py:242-256): RCU(x) = conv2(relu(conv1(relu x))) + relu x.
  // With `sba` a conv's epilogue writes split planes for the next conv (and fp32 only where a residual add reads it).
  // pred_fused(): the regression heads' 1x1 prediction convs run inside conv_fuse_conv1's epilogue (no 32-channel map in HBM)
  bool pred_fused() const { return fuse_pred && arch != PF_ARCH_PERSNET_CLS && nterms == NT_F16X3 && split_bf16; }
  void heads_fwd(Ctx& c, int B, Ten feats[4], Ten llf, float* t32 /*[2][B][320][320][32], unused when the heads are fused*/, float* pg, float* pl, float* pn) {
    const bool S = sba && nterms != NT_F16X3;  // the 3x3 halo kernels of the decoder stage fp32 inputs: split-f16 planes only in MiT / ConvNeXt
    const bool SR = S || (sba_heads && !sba && nterms == NT_F16X3 && split_bf16);  // ResidualConvUnit chain in split planes (PF_SBA_HEADS: fp16 planes, halo kernel ASB)
    Head& hg = heads[0];
    Head& hl = heads[1];
    // a pair of per-head tensors, contiguous as [2][...] in fp32 and in every split plane
    auto pair = [&](size_t per_head, bool want_f32, bool want_sb, Ten& a, Ten& b) {
      a = Ten(); b = Ten();
      if (want_f32) { a.f = c.alloc(2 * per_head); b.f = a.f + per_head; }
      if (want_sb) { a.s = c.alloc_sb(2 * per_head); b.s = a.s; b.s.p = a.s.p + per_head; }
    };
    // The two largest bilinear x2 maps -- up[0] (160^2 x 256) feeding conv_fuse_conv0 and the 320^2 x 64 map feeding
    // conv_fuse_conv1 (decode_head.py:284-286, gravity_head.py:170-172) -- are never materialised: the consuming 3x3 convs
    // interpolate them while staging their input halo tiles (ConvParams::ups; same expression, bit-identical results).
    const bool fuse_up = fuse_upsample && nterms == NT_F16X3 && split_bf16 && !S;
    Ten up[4][2], qfin[2];
    for (int k = 3; k >= 0; --k) {
      const int h = NET >> (k + 2);
      // up[k] is a residual operand (fp32) for k >= 1 and conv0's input (split planes) for k == 0
      if (k == 0 && fuse_up) pair((size_t)B * h * h * DEC_FEAT, true, false, qfin[0], qfin[1]);  // the 80^2 map conv0 up-samples on the fly
      else pair((size_t)B * 4 * h * h * DEC_FEAT, !(S && k == 0), S && k == 0, up[k][0], up[k][1]);
    }
    for (int k = 3; k >= 0; --k) {
      const int h = NET >> (k + 2);
      const size_t M = (size_t)B * h * h;
      const size_t mk = c.mark();
      Ten p0, p1;
      pair(M * DEC_FEAT, true, SR, p0, p1);
      if (fold_mlp) {
        ConvCall cc[2] = {{&hg.fold[k], feats[k], p0}, {&hl.fold[k], feats[k], p1}};
        conv_g(c, 2, cc, B, h, h, ACT_NONE, 1);                               // relu(_ck), Linear folded into the conv
      } else {
        Ten e0, e1;
        pair(M * DEC_EMBED, !S, S, e0, e1);
        ConvCall l[2] = {{&hg.lin[k], feats[k], e0}, {&hl.lin[k], feats[k], e1}};
        conv_g(c, 2, l, 1, (int)M, 1);                                        // MLP (decode_head.py:49-53)
        ConvCall cc[2] = {{&hg.proc[k], e0, p0}, {&hl.proc[k], e1, p1}};
        conv_g(c, 2, cc, B, h, h, ACT_NONE, 1);                               // relu(_ck)
      }
      Ten o0 = p0, o1 = p1;
      if (k < 3) {                                                            // o = relu(up(prev) + RCU1(_ck))
        Ten t0, t1;
        pair(M * DEC_FEAT, !SR, SR, t0, t1);
        ConvCall a[2] = {{&hg.r1c1[k], p0, t0}, {&hl.r1c1[k], p1, t1}};
        conv_g(c, 2, a, B, h, h, ACT_RELU);
        pair(M * DEC_FEAT, true, SR, o0, o1);
        ConvCall b2[2] = {{&hg.r1c2[k], t0, o0, p0.f, up[k + 1][0].f}, {&hl.r1c2[k], t1, o1, p1.f, up[k + 1][1].f}};
        conv_g(c, 2, b2, B, h, h, ACT_NONE, 1);
      }
      Ten t0, t1, q0, q1;
      pair(M * DEC_FEAT, !SR, SR, t0, t1);
      ConvCall a[2] = {{&hg.r2c1[k], o0, t0}, {&hl.r2c1[k], o1, t1}};
      conv_g(c, 2, a, B, h, h, ACT_RELU);
      if (k == 0 && fuse_up) { q0 = qfin[0]; q1 = qfin[1]; }
      else pair(M * DEC_FEAT, true, false, q0, q1);
      ConvCall b2[2] = {{&hg.r2c2[k], t0, q0, o0.f}, {&hl.r2c2[k], t1, q1, o1.f}};
      conv_g(c, 2, b2, B, h, h);                                              // RCU2 output, raw
      if (!c.dry && !(k == 0 && fuse_up)) {                                   // decode_head.py:284-286
        const Ten& u = up[k][0];
        ProfScope ps(c.prof, c.s, PC_UPSAMPLE, (8.0 + (u.f ? 32.0 : 0.0) + (u.s.p ? 48.0 : 0.0)) * M * DEC_FEAT);
        launch_upsample2x(q0.f, u.f, 2 * B, h, h, DEC_FEAT, c.s, u.s.p, u.s.plane);
      }
      c.release(mk);
    }
    const int h = NET / 2;
    Ten z0, z1, zu0, zu1;
    pair((size_t)B * h * h * 64, true, false, z0, z1);
    ConvCall a[2] = {{&hg.conv0, fuse_up ? qfin[0] : up[0][0], z0, nullptr, nullptr, llf}, {&hl.conv0, fuse_up ? qfin[1] : up[0][1], z1, nullptr, nullptr, llf}};
    conv_g(c, 2, a, B, h, h, ACT_RELU, 0, DEC_FEAT, 0, fuse_up ? 1 : 0);     // cat (and the x2 up-sampling) fused into the A gather (:170-171)
    tap(c, "dec.gravity.conv0", z0.f, B, h, h, 64);
    tap(c, "dec.latitude.conv0", z1.f, B, h, h, 64);
    if (!fuse_up) {
      pair((size_t)B * NET * NET * 64, !S, S, zu0, zu1);
      if (!c.dry) {
        ProfScope ps(c.prof, c.s, PC_UPSAMPLE, (8.0 + (zu0.f ? 32.0 : 0.0) + (zu0.s.p ? 48.0 : 0.0)) * B * h * h * 64);
        launch_upsample2x(z0.f, zu0.f, 2 * B, h, h, 64, c.s, zu0.s.p, zu0.s.plane);
      }
    }
    ConvCall b2[2] = {{&hg.conv1, fuse_up ? z0 : zu0, Ten(t32)}, {&hl.conv1, fuse_up ? z1 : zu1, Ten(t32 ? t32 + (size_t)B * NET * NET * 32 : nullptr)}};
    if (pred_fused()) {
      b2[0].head_kind = 1; b2[0].head_w = hg.predw; b2[0].head_b = hg.predb; b2[0].head_out = pg; b2[0].head_pn = pn;
      b2[1].head_kind = 2; b2[1].head_w = hl.predw; b2[1].head_b = hl.predb; b2[1].head_out = pl; b2[1].head_pn = pn;
    }
    conv_g(c, 2, b2, B, NET, NET, ACT_RELU, 0, -1, 0, fuse_up ? 1 : 0);
  }

  // ConvNeXt-T + heads of the ParamNets (convnext.py:140-152)
  void paramnet(Ctx& c, int B, const float* pn_in, float* d_params) {
    const bool S = sba;
    const float* src = pn_in;
    int H = NET;
    if (param_in != NET) {
      float* small = c.alloc((size_t)B * param_in * param_in * 4);
      if (!c.dry) launch_nearest_nhwc4(pn_in, small, B, NET, NET, param_in, param_in, c.s);
      src = small;
      H = param_in;
    }
    int h = H / 4;
    float* y = c.alloc((size_t)B * h * h * CNX_DIMS[0]);
    tap(c, "pn.in", src, B, H, H, 4);
    conv(c, cnx.stem, Ten(const_cast<float*>(src)), B, H, H, Ten(y));
    ln(c, cnx.stemn, y, Ten(y), (long)B * h * h);
    tap(c, "pn.stem", y, B, h, h, CNX_DIMS[0]);
    for (int s = 0; s < 4; ++s) {
      const int C = CNX_DIMS[s];
      if (s > 0) {
        const long Mi = (long)B * h * h;
        const Ten yn_in = S ? Ten(nullptr, c.alloc_sb(Mi * CNX_DIMS[s - 1])) : Ten(y);  // LN output feeds the 2x2 conv only
        ln(c, cnx.dsn[s - 1], y, yn_in, Mi);
        float* yn = c.alloc((size_t)B * (h / 2) * (h / 2) * C);
        conv(c, cnx.ds[s - 1], yn_in, B, h, h, Ten(yn));
        y = yn;
        h /= 2;
      }
      const long M = (long)B * h * h;
      const size_t mk = c.mark();
      float* d = c.alloc(M * C);
      const Ten dn = S ? Ten(nullptr, c.alloc_sb(M * C)) : Ten(d);
      const Ten hb = c.ten(M * 4 * C, !S, S);
      int blk = -1;
      for (CnxBlock& cb : cnx.blocks[s]) {
        if (blk >= 0) tap(c, fmt("pn.s%d.b%d", s + 1, blk), y, B, h, h, C);
        ++blk;
        if (!c.dry) {
          ProfScope ps(c.prof, c.s, PC_DW7, 8.0 * M * C);
          launch_dwconv7x7(y, cb.dw.w, cb.dw.b, d, B, h, h, C, c.s);
        }
        if (cb.mlp_w && nterms == NT_F16X3) {         // norm + pwconv1 + GELU + pwconv2 + layer scale + residual in one kernel
          range_in(c, fmt("cnx_mlp s%d.b%d d (LN-fused)", s + 1, blk), d, (size_t)M * C);
          if (!c.dry) {
            ProfScope ps(c.prof, c.s, PC_IGEMM_SB, 2.0 * 2.0 * M * (double)C * 4 * C, (int)M, C, 8 * C, 1);
            launch_cnx_mlp(d, y, cb.mlp_w, cb.mlp_tab, M, C, cb.n.eps, c.s, d_sat, cb.out_limit);
          }
          continue;
        }
        if (cb.pw1.ln_s) {
          gemm(c, cb.pw1, Ten(d), M, hb, ACT_GELU);   // block norm inside pwconv1
        } else {
          ln(c, cb.n, d, dn, M);
          gemm(c, cb.pw1, dn, M, hb, ACT_GELU);
        }
        gemm(c, cb.pw2, hb, M, Ten(y), ACT_NONE, y);  // y += gamma * pwconv2(...)  (gamma folded)
      }
      if (blk >= 0) tap(c, fmt("pn.s%d.b%d", s + 1, blk), y, B, h, h, C);
      c.release(mk);
    }
    float* raw = c.alloc((size_t)B * 8);
    if (!c.dry) {
      launch_gap_ln_head(y, cnx.norm.g, cnx.norm.b, cnx.headw, cnx.headb, raw, B, h * h, CNX_DIMS[3], cnx.nout, cnx.norm.eps, c.s);
      launch_paramnet_scalars(raw, cnx.nout, d_params, B, arch == PF_ARCH_PARAMNET_CENTERED ? 0 : 1, c.s);
    }
  }

  void run(Ctx& c, int B, const void* in, bool is_u8, float* pg, float* pl, float* params) {
    if (!c.dry && pn_pending && (B != pn_pending_B || c.base != pn_pending_base)) {
      // a deferred branch works at an offset that depends on ITS batch size (behind that batch's main region): a forward with another batch size, or in another
      // workspace buffer, would lay its main region over it (or leave it behind in a buffer the caller may free) -- join before anything of this forward is issued
      issue_deferred();
      (void)hipStreamWaitEvent(c.s, ev_pn_done, 0);
      pn_pending = false;
    }
    float* x0 = c.alloc((size_t)B * NET * NET * 4);
    if (!c.dry && c.prof) c.prof->mark(PH_BACKBONE, c.s);   // input normalisation + MiT-B3
    if (!c.dry) {
      if (is_u8) launch_prep_u8(static_cast<const uint8_t*>(in), x0, (long)B * NET * NET, mean3, std3, c.s);
      else launch_prep_f32_nchw(static_cast<const float*>(in), x0, B, NET * NET, mean3, std3, c.s);
    }
    Ten feats[4];
    // low-level encoder output: conv0's second (concatenated) input only
    const bool Sh = sba && nterms != NT_F16X3;  // as in heads_fwd: the decoder's halo kernels read fp32
    const Ten llf = c.ten((size_t)B * (NET / 2) * (NET / 2) * LL_CH, !Sh, Sh);
    ll_forked = false;
    mit(c, B, x0, feats, &llf);
    if (!c.dry && c.prof) c.prof->mark(PH_LL, c.s);
    if (ll_forked) (void)hipStreamWaitEvent(c.s, ev_ll, 0);
    else if (ll_s7_w && nterms == NT_F16X3 && !c.tuning && llf.f && !llf.s.p) stem(c, ll_s7_w, ll_s7_tab, x0, llf.f, B, NET, NET, 2, true, false, 0.f, ll.sat_limit, "low-level encoder");
    else conv(c, ll, Ten(x0), B, NET, NET, llf, ACT_RELU);  // BN folded (perspectivefields.py:79-83)
    tap(c, "ll", llf.f, B, NET / 2, NET / 2, LL_CH);
    const bool pf = pred_fused();
    float* tg = pf ? nullptr : c.alloc((size_t)2 * B * NET * NET * 32);
    float* tl = pf ? nullptr : tg + (size_t)B * NET * NET * 32;
    // ParamNet's input map and activations live in their own region behind the main one (run_pn_peak: its size, from the dry run), so that a deferred branch
    // (defer_params) never shares memory with the next forward's backbone / decoders
    Ctx cp = c;
    cp.off = 0; cp.peak = 0;
    if (!c.dry) cp.base = c.base + pn_off[B];
    float* pn = has_param ? cp.alloc((size_t)B * NET * NET * 4) : nullptr;
    if (!c.dry && pn_pending) {  // the previous forward's deferred branch still reads pn / writes its activations: the decoders below overwrite pn
      issue_deferred();
      (void)hipStreamWaitEvent(c.s, ev_pn_done, 0);
      pn_pending = false;
    }
    const size_t mk = c.mark();
    if (!c.dry && c.prof) c.prof->mark(PH_DECODERS, c.s);   // both decoder heads + prediction heads
    heads_fwd(c, B, feats, llf, tg, pg, pl, pn);
    c.release(mk);
    if (arch == PF_ARCH_PERSNET_CLS) {
      // 1x1 convs to 73 / 180 logits, stored NCHW because the logits are API-visible (gravity_head.py:259)
      conv(c, heads[0].predcls, Ten(tg), B, NET, NET, Ten(pg), ACT_NONE, nullptr, nullptr, 0, Ten(), -1, 1);
      conv(c, heads[1].predcls, Ten(tl), B, NET, NET, Ten(pl), ACT_NONE, nullptr, nullptr, 0, Ten(), -1, 1);
      if (!c.dry && c.prof) c.prof->mark(PH_END, c.s);
      return;
    }
    if (!c.dry && !pf)
      launch_pred_regression(tg, tl, heads[0].predw, heads[0].predb, heads[1].predw, heads[1].predb, pg, pl, pn, B, NET * NET, c.s);
    if (!c.dry && c.prof) c.prof->mark(has_param ? PH_PARAMNET : PH_END, c.s);
    if (has_param) {
      const bool defer = defer_params && !c.dry && (!c.prof || c.prof->min_work > 0.0) && !c.dbg && !c.tuning && !in_capture && pstream_ready();
      if (defer) {
        (void)hipEventRecord(ev_pn_in, c.s);
        (void)hipStreamWaitEvent(pstream, ev_pn_in, 0);
        cp.s = pstream;
      }
      if (defer && defer_at > 0) {  // issued by the next forward (mit(), stage defer_at) or by whoever needs the result first
        pn_later.armed = true; pn_later.B = B; pn_later.pn = pn; pn_later.params = params; pn_later.ctx = cp;
        pn_pending = true; pn_pending_B = B; pn_pending_base = c.base;
      } else {
        paramnet(cp, B, pn, params);
        if (defer) {
          (void)hipEventRecord(ev_pn_done, pstream);
          pn_pending = true; pn_pending_B = B; pn_pending_base = c.base;
        }
      }
    }
    if (!c.dry && c.prof && has_param) c.prof->mark(PH_END, c.s);
    run_pn_peak = cp.peak;
    if (cp.max_conv_out > c.max_conv_out) c.max_conv_out = cp.max_conv_out;
  }
  size_t run_pn_peak = 0;

  // with_scratch: plus the target of the tuning launches (largest conv output as fp32 + 3 bf16 planes) -- pf_autotune only
  size_t workspace_bytes(int B, bool with_scratch = false) {
    auto it = ws_cache.find(B);
    if (it == ws_cache.end()) {
      Ctx c{nullptr, 4096, 0, 0, true, nullptr};
      c.sb_planes = nterms == NT_F16X3 ? 2 : 3;
      run(c, B, nullptr, true, nullptr, nullptr, nullptr);
      const size_t main_peak = (c.peak + 255) & ~(size_t)255, total = main_peak + ((run_pn_peak + 255) & ~(size_t)255);
      pn_off[B] = main_peak;
      ws_cache[B] = total + 4096;
      scratch_off[B] = total;
      scratch_elems[B] = c.max_conv_out;
      it = ws_cache.find(B);
    }
    return it->second + ((with_scratch || autotune) ? scratch_elems[B] * 10 + 4096 : 0);
  }
  int forward(int B, const void* in, bool is_u8, float* pg, float* pl, float* params, void* ws, size_t ws_bytes, hipStream_t s, bool tune = false) {
    if (host_only) return fail(PF_ERR_DEVICE, "this engine was created with PF_DEVICE_NONE (host side only): no forward");
    if (!finalized) return fail(PF_ERR_WEIGHTS, "pf_forward called before pf_finalize_weights");
    if (B <= 0 || !in || !pg || !pl || !ws) return fail(PF_ERR_ARG, "pf_forward: null pointer or batch <= 0");
    if (has_param && !params) return fail(PF_ERR_ARG, "pf_forward: d_params is required for a ParamNet architecture");
    // The implicit-GEMM kernels address every activation with 32-bit BYTE offsets and use offset 2^31 as the "reads as
    // zero" marker (igemm_common.h OOB): each per-head activation must stay below 2 GiB.  The largest are the
    // 160x160x256 decoder map and the 320x320x64 map before conv_fuse_conv1: B * 26.2 MB  ->  B <= PF_MAX_BATCH (81).
    if (B > PF_MAX_BATCH) return fail(PF_ERR_ARG, fmt("pf_forward: batch %d exceeds PF_MAX_BATCH = %d (32-bit byte offsets inside one activation); split the batch", B, PF_MAX_BATCH));
    const size_t need = workspace_bytes(B, tune);
    if (ws_bytes < need) return fail(PF_ERR_WORKSPACE, fmt("workspace too small: %zu < %zu bytes", ws_bytes, need));
    if (hipSetDevice(device) != hipSuccess) return fail(PF_ERR_DEVICE, "hipSetDevice failed");
    uintptr_t base = (reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255;
    Ctx c{s, base, 0, 0, false, nullptr};
    c.sb_planes = nterms == NT_F16X3 ? 2 : 3;
    c.prof = prof.on ? &prof : nullptr;
    c.dbg = debug_sink;
    if (tune) {
      c.tuning = true;
      c.tune_scratch = reinterpret_cast<float*>(base + scratch_off[B]);
      c.tune_scratch_elems = scratch_elems[B];
    }
    run(c, B, in, is_u8, pg, pl, params);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(PF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
    return PF_OK;
  }
};

static void tune_cache_load(pf_engine* h);
static void tune_cache_save(pf_engine* h);

// =========================================================================== C ABI

extern "C" {

const char* pf_version(void) { return "pf_hip 0.2 (gfx950; split-f16 / split-bf16 / fp32 MFMA)"; }

#ifndef PF_BUILD_DIGEST
#define PF_BUILD_DIGEST "unknown"
#endif
// sha256 over csrc/ + include/ + flags at build time (perspectivefields_amd/build.py): lets the loader refuse a stale .so
const char* pf_build_digest(void) { return PF_BUILD_DIGEST; }

const char* pf_last_error(pf_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int pf_create(pf_handle* out, int device, int arch) {
  if (!out) { g_create_error = "pf_create: out is NULL"; return PF_ERR_ARG; }
  *out = nullptr;
  if (arch < 0 || arch > 2) { g_create_error = fmt("pf_create: unknown arch %d", arch); return PF_ERR_ARG; }
  const bool host_only = device == PF_DEVICE_NONE;
  if (!host_only) {
    const int rc = check_device(device, &g_create_error);
    if (rc != PF_OK) return rc;
  }
  pf_engine* e = new pf_engine();
  e->device = device;
  e->host_only = host_only;
  e->arch = arch;
  if (const char* v = getenv("PF_FOLD_MLP")) e->fold_mlp = atoi(v) != 0;
  if (const char* v = getenv("PF_FUSE_UPSAMPLE")) e->fuse_upsample = atoi(v) != 0;
  if (const char* v = getenv("PF_FUSE_PRED")) e->fuse_pred = atoi(v) != 0;
  if (const char* v = getenv("PF_FUSE_LN")) e->fuse_ln = atoi(v) != 0;
  if (const char* v = getenv("PF_FUSE_CNX_MLP")) e->fuse_cnx_mlp = atoi(v) != 0;
  if (const char* v = getenv("PF_FUSE_MIT_MLP")) e->fuse_mit_mlp = atoi(v) != 0;
  if (const char* v = getenv("PF_MIT_MLP_128")) e->mit_mlp128 = atoi(v);
  if (const char* v = getenv("PF_WINO")) e->wino_min_hw = atoi(v);
  if (const char* v = getenv("PF_WINO_MIN_BLOCKS")) e->wino_min_blocks = atoi(v);
  {
    const char* wt = getenv("PF_WINO_TILE");  // "wino256x64d" (default: 4 waves, B fragments computed in registers, hand-placed slots), or "wino256x64c" (its compiler-scheduled form)
    for (int t = 0; t < conv_num_tiles(); ++t) if (strcmp(conv_tile_name(t), wt ? wt : "wino256x64d") == 0) e->wino_tile = t;
  }
  if (const char* v = getenv("PF_RB_CHAIN")) e->rb_chain = atoi(v);
  if (const char* v = getenv("PF_DEFER_AT")) e->defer_at = atoi(v);
  if (const char* v = getenv("PF_DEFER_PRIO")) e->defer_prio = atoi(v);
  if (!host_only) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->num_cus = prop.multiProcessorCount;
    e->rb_min_blocks = e->num_cus * 3 / 4;  // "the last round of blocks nearly fills the chip": 192 of 256 CUs
  }
  if (const char* v = getenv("PF_RB_MIN_BLOCKS")) e->rb_min_blocks = atoi(v);
#ifdef PF_TUNING_BUILD
#endif
  if (const char* v = getenv("PF_SIDE_STREAM")) e->side_stream_mode = atoi(v);
  if (const char* v = getenv("PF_S3_SPLIT")) e->s3_split = atoi(v);
  if (const char* v = getenv("PF_ATTN64")) e->attn64 = atoi(v);
  if (const char* v = getenv("PF_STEM7")) e->stem7 = atoi(v);
  if (const char* v = getenv("PF_THIN128")) e->thin128 = atoi(v);
  if (const char* v = getenv("PF_SBA_HEADS")) e->sba_heads = atoi(v) != 0;
  if (const char* v = getenv("PF_AUTOTUNE")) e->autotune = atoi(v) != 0;
  if (const char* v = getenv("PF_SPLIT_BF16")) e->split_bf16 = atoi(v) != 0;
  if (const char* v = getenv("PF_SBA")) e->sba = atoi(v) != 0;
  if (!e->split_bf16) e->sba = false;  // split planes are only read by the split-bf16 kernels
  if (!e->split_bf16 || e->sba) e->fuse_ln = false;  // the fused form lives in the split GEMM kernels and reads fp32 rows
  if (!e->split_bf16 || e->sba) e->fuse_cnx_mlp = false;
  if (!e->split_bf16 || e->sba) e->fuse_mit_mlp = false;

  tune_cache_load(e);
  *out = e;
  return PF_OK;
}

int pf_set_precision(pf_handle h, int mode) {
  if (!h) return PF_ERR_ARG;
  if (mode != PF_PRECISION_FP32 && mode != PF_PRECISION_FP32_BF16X6) return h->fail(PF_ERR_ARG, "pf_set_precision: unknown mode (0 = FP32 split-f16, 3 = FP32_BF16X6; the reduced-precision modes 1 / 2 of earlier versions are gone)");
  if (mode != PF_PRECISION_FP32 && !h->split_bf16) return h->fail(PF_ERR_ARG, "pf_set_precision: this mode needs the split kernels (PF_SPLIT_BF16=0 is set)");
  if (h->pn_pending) { h->issue_deferred(); (void)hipStreamSynchronize(h->pstream); h->pn_pending = false; }  // a deferred ParamNet branch still works in the old layout
  h->ws_cache.clear();  // the split-plane activation format (2 fp16 / 3 bf16 planes) follows the scheme
  h->nterms = mode == PF_PRECISION_FP32_BF16X6 ? 6 : NT_F16X3;
  return PF_OK;
}

int pf_set_saturation_counter(pf_handle h, void* d_counter_u32) {
  if (!h) return PF_ERR_ARG;
  h->d_sat = static_cast<unsigned*>(d_counter_u32);
  return PF_OK;
}
int pf_static_window_max(pf_handle h, float* out) {
  if (!h || !out || !h->finalized) return PF_ERR_ARG;
  *out = h->static_window_max;
  return PF_OK;
}
int pf_set_defer_params(pf_handle h, int on) {
  if (!h) return PF_ERR_ARG;
  h->defer_params = on ? 1 : 0;
  return PF_OK;
}
int pf_join_params(pf_handle h, void* stream) {
  if (!h) return PF_ERR_ARG;
  if (h->pn_pending) h->issue_deferred();
  if (h->pn_pending && hipStreamWaitEvent(static_cast<hipStream_t>(stream), h->ev_pn_done, 0) != hipSuccess) return h->fail(PF_ERR_DEVICE, "pf_join_params: hipStreamWaitEvent failed");
  return PF_OK;
}

int pf_destroy(pf_handle h) {
  if (!h) return PF_ERR_ARG;
  if (h->host_only) {
    for (void* d : h->dev_allocs) free(d);
    delete h;
    return PF_OK;
  }
  (void)hipSetDevice(h->device);
  for (void* d : h->dev_allocs) (void)hipFree(d);
  if (h->dbg.stats) (void)hipFree(h->dbg.stats);
  for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->ev_ll) (void)hipEventDestroy(h->ev_ll);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->side2) (void)hipStreamDestroy(h->side2);
  if (h->pstream) { (void)hipStreamSynchronize(h->pstream); (void)hipStreamDestroy(h->pstream); }
  if (h->ev_pn_in) (void)hipEventDestroy(h->ev_pn_in);
  if (h->ev_pn_done) (void)hipEventDestroy(h->ev_pn_done);
  delete h;
  return PF_OK;
}

int pf_load_tensor(pf_handle h, const char* key, const float* data, const int64_t* shape, int rank) {
  if (!h || !key || (!data && rank > 0) || rank < 0 || rank > 4) return h ? h->fail(PF_ERR_ARG, "pf_load_tensor: bad argument") : PF_ERR_ARG;
  if (h->finalized) return h->fail(PF_ERR_WEIGHTS, "pf_load_tensor after pf_finalize_weights");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < rank; ++i) { if (shape[i] <= 0) return h->fail(PF_ERR_ARG, "pf_load_tensor: non-positive dim"); t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  if (data) t.data.assign(data, data + n);
  h->host[key] = std::move(t);
  return PF_OK;
}

int pf_finalize_weights(pf_handle h) {
  if (!h) return PF_ERR_ARG;
  if (h->finalized) return h->fail(PF_ERR_WEIGHTS, "weights already finalized");
  if (!h->host_only && hipSetDevice(h->device) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipSetDevice failed");
  try {
    h->build();
  } catch (const std::string& m) {
    return h->fail(PF_ERR_WEIGHTS, m);
  }
  h->finalized = true;
  return PF_OK;
}

int pf_output_info(pf_handle h, int* g, int* l, int* p) {
  if (!h) return PF_ERR_ARG;
  const bool cls = h->arch == PF_ARCH_PERSNET_CLS;
  if (g) *g = cls ? 73 : 2;
  if (l) *l = cls ? 180 : 1;
  if (p) *p = cls ? 0 : 5;
  return PF_OK;
}

int pf_max_batch(void) { return PF_MAX_BATCH; }

size_t pf_workspace_bytes(pf_handle h, int batch) {
  if (!h || batch <= 0 || batch > PF_MAX_BATCH) return 0;
  const bool was = h->has_param;
  if (!h->finalized) {  // architecture-only estimate is allowed before weights are loaded
    h->has_param = h->arch != PF_ARCH_PERSNET_CLS;
    h->param_in = h->arch == PF_ARCH_PARAMNET_UNCENTERED ? 64 : NET;
    for (int s = 0; s < 4; ++s)
      if (h->stages[s].blocks.empty()) h->stages[s].blocks.resize(MIT_DEPTHS[s]);
    if (h->has_param)
      for (int s = 0; s < 4; ++s)
        if (h->cnx.blocks[s].empty()) h->cnx.blocks[s].resize(CNX_DEPTHS[s]);
  }
  const size_t n = h->workspace_bytes(batch);
  if (!h->finalized) {
    h->has_param = was;
    for (int s = 0; s < 4; ++s) { h->stages[s].blocks.clear(); h->cnx.blocks[s].clear(); }
    h->ws_cache.clear();
  }
  return n;
}

int pf_forward_u8(pf_handle h, int batch, const uint8_t* in, float* pg, float* pl, float* params, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  return h->forward(batch, in, true, pg, pl, params, ws, ws_bytes, static_cast<hipStream_t>(stream));
}
// The forward of a small batch is ~430 launches of a few microseconds each: host launch cost, not GPU time, sets its
// latency.  Captured once per (batch, buffer set) into a hipGraph and replayed with a single launch.
int pf_forward_u8_graph(pf_handle h, int batch, const uint8_t* in, float* pg, float* pl, float* params, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!s) return h->fail(PF_ERR_ARG, "pf_forward_u8_graph: needs an explicit (non-default) stream: the legacy default stream cannot be captured");
  if (h->prof.on || h->autotune) return h->forward(batch, in, true, pg, pl, params, ws, ws_bytes, s);  // event pairs / tuning launches are not capturable
  const std::vector<uintptr_t> key = {(uintptr_t)batch, (uintptr_t)in, (uintptr_t)pg, (uintptr_t)pl, (uintptr_t)params, (uintptr_t)ws, (uintptr_t)h->nterms};
  hipGraphExec_t exec = nullptr;
  for (auto& g : h->graphs) if (g.key == key) { exec = g.exec; break; }
  if (!exec) {
    if (hipSetDevice(h->device) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipSetDevice failed");
    (void)h->workspace_bytes(batch);  // dry run outside the capture
    if (h->pn_pending) { h->issue_deferred(); (void)hipStreamWaitEvent(s, h->ev_pn_done, 0); h->pn_pending = false; }  // a deferred ParamNet branch is joined outside the capture
    if (hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipStreamBeginCapture failed");
    h->in_capture = true;  // a captured forward keeps the branch on the capture stream
    const int rc = h->forward(batch, in, true, pg, pl, params, ws, ws_bytes, s);
    h->in_capture = false;
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture(s, &graph);
    if (rc != PF_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess || !graph) return h->fail(PF_ERR_DEVICE, fmt("hipStreamEndCapture failed: %s", hipGetErrorString(e)));
    const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e2 != hipSuccess) return h->fail(PF_ERR_DEVICE, fmt("hipGraphInstantiate failed: %s", hipGetErrorString(e2)));
    if (h->graphs.size() >= 16) { (void)hipGraphExecDestroy(h->graphs.front().exec); h->graphs.erase(h->graphs.begin()); }
    h->graphs.push_back({key, exec});
  }
  // a replay, too, overwrites the ParamNet input map (or, with another batch size, the region) a pending deferred branch is still working in
  if (h->pn_pending) { h->issue_deferred(); (void)hipStreamWaitEvent(s, h->ev_pn_done, 0); h->pn_pending = false; }
  if (hipGraphLaunch(exec, s) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipGraphLaunch failed");
  return PF_OK;
}

// Debug forward: the ordinary forward plus SHADOW taps (flags & 1: stage / block boundary tensors copied into d_tap_buf, NHWC fp32) and / or RANGE records (flags & 2:
// statistics of every tensor that enters a dense contraction).  Synchronises the stream before it returns; the records are read with pf_debug_taps / pf_debug_ranges.
size_t pf_debug_tap_bytes(pf_handle h, int batch) {
  if (!h || batch <= 0 || batch > PF_MAX_BATCH || !h->finalized) return 0;
  DebugSink d;
  d.shadow = true;
  Ctx c{nullptr, 4096, 0, 0, true, nullptr};
  c.sb_planes = h->nterms == NT_F16X3 ? 2 : 3;
  c.dbg = &d;
  h->run(c, batch, nullptr, true, nullptr, nullptr, nullptr);
  return d.used + 256;
}
int pf_debug_forward_u8(pf_handle h, int batch, const uint8_t* in, float* pg, float* pl, float* params, void* ws, size_t ws_bytes, int flags, void* d_tap_buf,
                        size_t tap_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((flags & 1) && !d_tap_buf) return h->fail(PF_ERR_ARG, "pf_debug_forward_u8: shadow taps need a buffer of pf_debug_tap_bytes(batch) bytes");
  if (h->prof.on || h->autotune) return h->fail(PF_ERR_ARG, "pf_debug_forward_u8: not inside a profiling window / with autotuning on");
  DebugSink& d = h->dbg;
  if (d.stats) { (void)hipFree(d.stats); }
  d = DebugSink();
  d.shadow = (flags & 1) != 0; d.range = (flags & 2) != 0;
  d.buf = static_cast<char*>(d_tap_buf); d.cap = tap_bytes;
  if (d.range) {
    d.max_ranges = 1024;
    if (hipMalloc(reinterpret_cast<void**>(&d.stats), (size_t)d.max_ranges * 16) != hipSuccess) { d.stats = nullptr; return h->fail(PF_ERR_DEVICE, "pf_debug_forward_u8: hipMalloc failed"); }
    (void)hipMemsetAsync(d.stats, 0, (size_t)d.max_ranges * 16, s);
  }
  h->debug_sink = &d;
  const int rc = h->forward(batch, in, true, pg, pl, params, ws, ws_bytes, s);
  h->debug_sink = nullptr;
  if (rc != PF_OK) return rc;
  if (hipStreamSynchronize(s) != hipSuccess) return h->fail(PF_ERR_DEVICE, "pf_debug_forward_u8: stream synchronisation failed");
  if (d.overflow) return h->fail(PF_ERR_WORKSPACE, fmt("pf_debug_forward_u8: tap buffer too small: %zu < %zu bytes", d.cap, d.used));
  return PF_OK;
}
int pf_debug_taps(pf_handle h, int max_records, char* names /*[max][64]*/, long long* byte_offsets, int* shapes /*[max][4]: B, H, W, C (NHWC)*/) {
  if (!h) return PF_ERR_ARG;
  const DebugSink& d = h->dbg;
  const int n = (int)std::min<size_t>(d.taps.size(), (size_t)(max_records < 0 ? 0 : max_records));
  for (int i = 0; i < n; ++i) {
    if (names) { std::strncpy(names + 64 * i, d.taps[i].name.c_str(), 63); names[64 * i + 63] = 0; }
    if (byte_offsets) byte_offsets[i] = (long long)d.taps[i].off;
    if (shapes) for (int k = 0; k < 4; ++k) shapes[4 * i + k] = d.taps[i].shape[k];
  }
  return (int)d.taps.size();
}
int pf_debug_ranges(pf_handle h, int max_records, char* names /*[max][96]*/, long long* elems, float* stats /*[max][4]: max |x|, sum x^2, saturated, non-finite*/) {
  if (!h) return PF_ERR_ARG;
  const DebugSink& d = h->dbg;
  const int n = (int)std::min<size_t>(d.ranges.size(), (size_t)(max_records < 0 ? 0 : max_records));
  if (n > 0 && stats && d.stats && hipMemcpy(stats, d.stats, (size_t)n * 16, hipMemcpyDeviceToHost) != hipSuccess) return h->fail(PF_ERR_DEVICE, "pf_debug_ranges: hipMemcpy failed");
  for (int i = 0; i < n; ++i) {
    if (names) { std::strncpy(names + 96 * i, d.ranges[i].name.c_str(), 95); names[96 * i + 95] = 0; }
    if (elems) elems[i] = d.ranges[i].elems;
    if (stats && d.stats) {  // the two counters are unsigned integers on the device (range_stats_kernel): hand them out as the floats the interface declares
      for (int k = 2; k < 4; ++k) { uint32_t u; std::memcpy(&u, &stats[4 * i + k], 4); stats[4 * i + k] = (float)u; }
    }
  }
  return (int)d.ranges.size();
}

int pf_forward_f32(pf_handle h, int batch, const float* in, float* pg, float* pl, float* params, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  return h->forward(batch, in, false, pg, pl, params, ws, ws_bytes, static_cast<hipStream_t>(stream));
}

// Optional on-disk tile cache (PF_TUNE_CACHE=<file>): read at pf_create, rewritten after every pf_autotune.
static int tile_table_load(pf_engine* h, const char* path);
static int tile_table_save(pf_engine* h, const char* path);
static void tune_cache_load(pf_engine* h) { if (const char* path = getenv("PF_TUNE_CACHE")) (void)tile_table_load(h, path); }
static void tune_cache_save(pf_engine* h) { if (const char* path = getenv("PF_TUNE_CACHE")) (void)tile_table_save(h, path); }

int pf_autotune(pf_handle h, int batch, const uint8_t* in, float* pg, float* pl, float* params, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  const int rc = h->forward(batch, in, true, pg, pl, params, ws, ws_bytes, static_cast<hipStream_t>(stream), true);
  if (rc == PF_OK) { h->tuned_batches[batch * 32 + h->nterms] = true; tune_cache_save(h); }
  return rc;
}

int pf_is_tuned(pf_handle h, int batch) { return (h && (!h->autotune || h->tuned_batches.count(batch * 32 + h->nterms))) ? 1 : 0; }

size_t pf_autotune_workspace_bytes(pf_handle h, int batch) {
  if (!h || !h->finalized || batch <= 0 || batch > PF_MAX_BATCH) return 0;
  return h->workspace_bytes(batch, true);
}

// Tile table files: one line per conv shape, "<12 key ints> <tile name>" (key = GEMM view, layout and precision of the
// launch).  Entries are keyed by shape and name, so a stale file can only cost speed, never correctness.
static int tile_table_load(pf_engine* h, const char* path) {
  FILE* f = fopen(path, "r");
  if (!f) return -1;
  char name[64];
  int k[12], n = 0;
  while (fscanf(f, "%d %d %d %d %d %d %d %d %d %d %d %d %63s", &k[0], &k[1], &k[2], &k[3], &k[4], &k[5], &k[6], &k[7], &k[8], &k[9], &k[10], &k[11], name) == 13) {
    for (int t = 0; t < conv_num_tiles(); ++t)
      if (std::strcmp(conv_tile_name(t), name) == 0) { h->tile_cache[std::vector<int>(k, k + 12)] = t; ++n; break; }
  }
  fclose(f);
  return n;
}
static int tile_table_save(pf_engine* h, const char* path) {
  FILE* f = fopen(path, "w");
  if (!f) return -1;
  int n = 0;
  for (auto& kv : h->tile_cache) {
    if (kv.second < 0) continue;
    for (int v : kv.first) fprintf(f, "%d ", v);
    fprintf(f, "%s\n", conv_tile_name(kv.second));
    ++n;
  }
  fclose(f);
  return n;
}
int pf_load_tile_table(pf_handle h, const char* path) {
  if (!h || !path) return PF_ERR_ARG;
  const int n = tile_table_load(h, path);
  return n < 0 ? h->fail(PF_ERR_ARG, fmt("pf_load_tile_table: cannot read '%s'", path)) : n;
}
int pf_save_tile_table(pf_handle h, const char* path) {
  if (!h || !path) return PF_ERR_ARG;
  const int n = tile_table_save(h, path);
  return n < 0 ? h->fail(PF_ERR_ARG, fmt("pf_save_tile_table: cannot write '%s'", path)) : n;
}

size_t pf_resize_workspace_bytes(int H, int W) { (void)W; return H > 0 ? (size_t)H * NET * 3 + 256 : 0; }

int pf_resize_bilinear_u8(pf_handle h, const uint8_t* d_img, int H, int W, uint8_t* d_out, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  if (!d_img || !d_out || H <= 0 || W <= 0) return h->fail(PF_ERR_ARG, "pf_resize_bilinear_u8: bad argument");
  if (!ws || ws_bytes < pf_resize_workspace_bytes(H, W)) return h->fail(PF_ERR_WORKSPACE, "pf_resize_bilinear_u8: workspace too small");
  if (hipSetDevice(h->device) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipSetDevice failed");
  ResizeTable* th = h->resize_table(W);  // first use of an extent builds + uploads its table (blocking copy, once)
  ResizeTable* tv = h->resize_table(H);
  if (!th || !tv) return h->fail(PF_ERR_DEVICE, "pf_resize_bilinear_u8: could not upload coefficient tables");
  uint8_t* tmp = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
  launch_resize_u8(d_img, H, W, tmp, d_out, NET, NET, th->d_bounds, th->d_kk, th->ksize, tv->d_bounds, tv->d_kk, tv->ksize, static_cast<hipStream_t>(stream));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return h->fail(PF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
  return PF_OK;
}

int pf_resize_batch_u8(pf_handle h, int B, const uint8_t* const* h_imgs, const int32_t* h_hw, uint8_t* d_out, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  if (B <= 0 || !h_imgs || !h_hw || !d_out) return h->fail(PF_ERR_ARG, "pf_resize_batch_u8: bad argument");
  size_t need = 256;
  for (int i = 0; i < B; ++i) {
    if (!h_imgs[i] || h_hw[2 * i] <= 0 || h_hw[2 * i + 1] <= 0) return h->fail(PF_ERR_ARG, "pf_resize_batch_u8: bad image pointer or size");
    need += ((size_t)h_hw[2 * i] * NET * 3 + 255) & ~(size_t)255;
  }
  if (!ws || ws_bytes < need) return h->fail(PF_ERR_WORKSPACE, fmt("pf_resize_batch_u8: workspace too small: %zu < %zu bytes", ws_bytes, need));
  if (hipSetDevice(h->device) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipSetDevice failed");
  uint8_t* tmp = reinterpret_cast<uint8_t*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
  for (int i0 = 0; i0 < B; i0 += ResizeBatch::MAX) {
    ResizeBatch rb;
    rb.n = std::min(B - i0, (int)ResizeBatch::MAX);
    for (int k = 0; k < rb.n; ++k) {
      const int i = i0 + k, H = h_hw[2 * i], W = h_hw[2 * i + 1];
      ResizeTable* th = h->resize_table(W);  // first use of an extent builds + uploads its table (blocking copy, once)
      ResizeTable* tv = h->resize_table(H);
      if (!th || !tv) return h->fail(PF_ERR_DEVICE, "pf_resize_batch_u8: could not upload coefficient tables");
      rb.H[k] = H; rb.W[k] = W; rb.in[k] = h_imgs[i]; rb.tmp[k] = tmp; rb.out[k] = d_out + (size_t)i * NET * NET * 3;
      rb.bh[k] = th->d_bounds; rb.kh[k] = th->d_kk; rb.ksh[k] = th->ksize;
      rb.bv[k] = tv->d_bounds; rb.kv[k] = tv->d_kk; rb.ksv[k] = tv->ksize;
      tmp += ((size_t)H * NET * 3 + 255) & ~(size_t)255;
    }
    launch_resize_batch_u8(rb, NET, NET, static_cast<hipStream_t>(stream));
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return h->fail(PF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
  return PF_OK;
}

int pf_postprocess(pf_handle h, const float* pg, const float* pl, int H, int W, float* up, float* lat, void* ws, size_t ws_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  if (!pg || !pl || !up || !lat || H <= 0 || W <= 0) return h->fail(PF_ERR_ARG, "pf_postprocess: bad argument");
  if (hipSetDevice(h->device) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipSetDevice failed");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (h->arch == PF_ARCH_PERSNET_CLS) {
    const size_t need = (size_t)3 * NET * NET * 4 + 256;
    if (!ws || ws_bytes < need) return h->fail(PF_ERR_WORKSPACE, fmt("pf_postprocess: classification needs %zu workspace bytes", need));
    float* dg = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    float* dl = dg + 2 * NET * NET;
    launch_decode_cls(pg, 73, pl, 180, dg, dl, 1, NET * NET, s);
    launch_postprocess(dg, dl, NET, NET, up, lat, H, W, 0, s);
  } else {
    launch_postprocess(pg, pl, NET, NET, up, lat, H, W, 1, s);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return h->fail(PF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
  return PF_OK;
}

int pf_postprocess_batch(pf_handle h, int B, const float* pg, const float* pl, const int32_t* hw, float* const* up, float* const* lat,
                         void* ws, size_t ws_bytes, void* stream) {
  if (!h) return PF_ERR_ARG;
  if (B <= 0 || !pg || !pl || !hw || !up || !lat) return h->fail(PF_ERR_ARG, "pf_postprocess_batch: bad argument");
  for (int i = 0; i < B; ++i)
    if (hw[2 * i] <= 0 || hw[2 * i + 1] <= 0 || !up[i] || !lat[i]) return h->fail(PF_ERR_ARG, "pf_postprocess_batch: bad size or output pointer");
  if (hipSetDevice(h->device) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipSetDevice failed");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool cls = h->arch == PF_ARCH_PERSNET_CLS;
  const size_t npx = (size_t)NET * NET;
  float *dg = nullptr, *dl = nullptr;
  h->prof.mark(PH_POST, s);
  if (cls) {  // decode the argmax of all images first (workspace: 3 x 320 x 320 floats per image)
    const size_t need = (size_t)B * 3 * npx * 4 + 256;
    if (!ws || ws_bytes < need) return h->fail(PF_ERR_WORKSPACE, fmt("pf_postprocess_batch: classification needs %zu workspace bytes", need));
    dg = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(ws) + 255) & ~(uintptr_t)255);
    dl = dg + (size_t)B * 2 * npx;
    launch_decode_cls(pg, 73, pl, 180, dg, dl, B, NET * NET, s);
  }
  for (int i0 = 0; i0 < B; i0 += PostBatch::MAX) {
    PostBatch pb;
    pb.n = std::min(B - i0, (int)PostBatch::MAX);
    for (int k = 0; k < pb.n; ++k) {
      const int i = i0 + k;
      pb.H[k] = hw[2 * i]; pb.W[k] = hw[2 * i + 1];
      pb.g2[k] = cls ? dg + (size_t)i * 2 * npx : pg + (size_t)i * 2 * npx;
      pb.l1[k] = cls ? dl + (size_t)i * npx : pl + (size_t)i * npx;
      pb.up[k] = up[i]; pb.lat[k] = lat[i];
    }
    launch_postprocess_batch(pb, NET, NET, cls ? 0 : 1, s);
  }
  h->prof.mark(PH_END, s);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return h->fail(PF_ERR_DEVICE, fmt("kernel launch failed: %s", hipGetErrorString(e)));
  return PF_OK;
}

int pf_fields_from_params(int device, const float* d_cam5, int H, int W, float* d_up, float* d_lat, void* stream) {
  std::string err;
  int rc = check_device(device, &err);
  if (rc != PF_OK) { g_create_error = err; return rc; }
  if (!d_cam5 || !d_up || !d_lat || H <= 0 || W <= 0) { g_create_error = "pf_fields_from_params: bad argument"; return PF_ERR_ARG; }
  launch_fields_from_params(d_cam5, H, W, d_up, d_lat, static_cast<hipStream_t>(stream));
  if (hipGetLastError() != hipSuccess) { g_create_error = "pf_fields_from_params: kernel launch failed"; return PF_ERR_DEVICE; }
  return PF_OK;
}

int pf_profile_begin(pf_handle h, unsigned class_mask) {
  if (!h) return PF_ERR_ARG;
  h->prof.reset();
  h->prof.mask = class_mask & 0x7fffffffu;
  h->prof.min_work = (class_mask & 0x80000000u) ? 2.0e11 : 0.0;  // bit 31: launches of >= 200 GFLOP only (the 256 -> 256 @80^2 convs, conv_fuse_conv0 / conv1 at B = 32)
  h->prof.on = true;
  return PF_OK;
}

int pf_profile_pause(pf_handle h) {
  if (!h) return PF_ERR_ARG;
  h->prof.on = false;  // no host synchronisation: the recorded events stay pending until pf_profile_end
  return PF_OK;
}

int pf_profile_end(pf_handle h, double* ms, double* work, long* launches, int n) {
  if (!h || n < PC_COUNT) return h ? h->fail(PF_ERR_ARG, "pf_profile_end: arrays must hold PF_PROFILE_CLASSES entries") : PF_ERR_ARG;
  h->prof.on = false;
  for (int i = 0; i < n; ++i) { ms[i] = 0; work[i] = 0; launches[i] = 0; }
  if (hipSetDevice(h->device) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipSetDevice failed");
  for (auto& r : h->prof.recs) {
    if (hipEventSynchronize(r.b) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipEventSynchronize failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipEventElapsedTime failed");
    r.ms = t;
    ms[r.cat] += t; work[r.cat] += r.work; launches[r.cat] += 1;
  }
  return PF_OK;  // records stay readable through pf_profile_records until the next pf_profile_begin
}

int pf_profile_records(pf_handle h, int max_records, int* cat, double* work, float* ms, int* mnk /*[max][4]: M, N, K, KH*/) {
  if (!h) return PF_ERR_ARG;
  const int n = (int)std::min<size_t>(h->prof.recs.size(), (size_t)(max_records < 0 ? 0 : max_records));
  for (int i = 0; i < n; ++i) {
    const auto& r = h->prof.recs[i];
    if (cat) cat[i] = r.cat;
    if (work) work[i] = r.work;
    if (ms) ms[i] = r.ms;
    if (mnk) { mnk[4 * i] = r.m; mnk[4 * i + 1] = r.n; mnk[4 * i + 2] = r.k; mnk[4 * i + 3] = r.kh; }
  }
  return (int)h->prof.recs.size();
}

int pf_profile_phases(pf_handle h, int n, double* ms) {
  if (!h || !ms || n < PF_PROFILE_PHASES) return h ? h->fail(PF_ERR_ARG, "pf_profile_phases: the array must hold PF_PROFILE_PHASES entries") : PF_ERR_ARG;
  for (int i = 0; i < n; ++i) ms[i] = 0.0;
  if (hipSetDevice(h->device) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipSetDevice failed");
  const auto& m = h->prof.marks;
  for (size_t i = 0; i + 1 < m.size(); ++i) {
    if (m[i].phase == PH_END) continue;   // the gap between two calls (forward -> post-process) belongs to nobody
    if (hipEventSynchronize(m[i + 1].e) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipEventSynchronize failed");
    float t = 0.f;
    if (hipEventElapsedTime(&t, m[i].e, m[i + 1].e) != hipSuccess) return h->fail(PF_ERR_DEVICE, "hipEventElapsedTime failed");
    ms[m[i].phase] += t;
  }
  return (int)m.size();
}

// ---- kernel-level entry points -------------------------------------------------------------

}  // extern "C"
