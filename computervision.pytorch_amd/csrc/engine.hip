// Graph executor behind the C ABI (include/cvx_engine.h): owns workspaces (activations, raw conv
// outputs, gradient buffers, fp16 weight shadows, gradient slabs), plans the concat-free buffer
// views and the backward write/accumulate modes once, and replays the op list on one HIP stream.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/cvx_engine.h"
#ifdef CVX_WITH_CHAIN  // tuning build only (tools/build_tuning.sh): the release library does not carry conv_chain.hip
#include "conv_chain.h"
#endif
#include "bn_act.h"
#include "conv_igemm.h"
#include <map>
#include <mutex>

#include "misc_ops.h"
#include "stem.h"

// events that only order this process's own streams on one device: no timing, no system-scope fence (the marker packet
// between two main-chain kernels costs ~6.5 us with the default flags -- 53 of them per backward pass)
static unsigned cvx_event_flags() {
  static const bool sysfence = cvx_tune_set("CVX_EVENT_SYSFENCE");
  return hipEventDisableTiming | (sysfence ? 0u : hipEventDisableSystemFence);
}

// ---------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;
void cvx_set_error(const std::string& msg) { g_last_error = msg; }
extern "C" const char* cvx_last_error(void) { return g_last_error.c_str(); }
extern "C" int cvx_abi_version(void) { return CVX_ABI_VERSION; }

namespace {

struct DgClass {
  ConvTap* taps = nullptr;
  bool halo_ok = false;
  int pointwise = 0;
  unsigned long long halo_pos = 0, halo_wt = 0;
  int ntaps = 0, oph = 0, opw = 0, OH2 = 0, OW2 = 0;
  const half_t* gemm_pk = nullptr;  // data-gradient weights in the GEMM-shaped kernel's ring image order (per batch plan), channel tiles of gemm_bn rows
  int gemm_bn = 0, gemm_kc = 0;
  const half_t* tile_pk = nullptr;  // ... and in the row-band kernel's LDS image order (conv_tile.hip), channel blocks of tile_bn rows
  int tile_bn = 0;
};

struct ConvRt {
  // static
  ConvTap* taps_fwd = nullptr;
  bool halo_ok = false;
  int pointwise = 0;
  int std3x3 = 0;
  int std7x7 = 0;
  unsigned long long halo_pos = 0, halo_wt = 0;
  int ntaps = 0;
  DgClass dg[16];
  int ndg = 0;
  long long sh_fwd = 0, sh_dg = -1;  // fp16 shadow offsets (elements)
  int cin_g = 0;                     // gathered input channels (view channels, multiple of 8)
  int cin_pad16 = 0;
  // backward plan
  int in_accum = 0, res_accum = 0;
  // stride-2 data gradient as one GEMM with a pixel-shuffle store (plan_gemm_packs): weights [4 phases x cin][4 window taps x C] in the shadow
  // arena (sh_ps, rebuilt from the transposed shadow after every pack_weights), the 2 x 2 window's tap table, the GEMM kernel's packed image
  long long sh_ps = -1;
  PsPackDesc ps_desc;
  ConvTap* ps_taps = nullptr;
  bool ps_on = false;
  const half_t* ps_pk = nullptr;
  int ps_bn = 0, ps_kc = 0;
  int slab_blk0 = 0, slab_blk1 = 0;  // reducer workgroups [blk0, blk1) of this op in the slab block table
  bool raw16 = false;                // CVX_OPF_RAW_F16 honoured: training keeps the raw output in fp16 in ybuf (no xhat, no fp32 scratch)
  bool stem = false;  // 3 -> Cout 3x3 stride-2 conv on the caller's fp32 images: stem.hip
  // per-batch
  half_t* ybuf = nullptr;   // xhat = (y - mean) * invstd of the training forward, fp16 (operand of the BN backward passes)
  half_t* dybuf = nullptr;  // gradient w.r.t. the raw conv output (kept per layer: the weight-gradient runs on a side stream)
  float *mean = nullptr, *invstd = nullptr, *scale = nullptr, *shift = nullptr;
  long long *stat_fwd = nullptr, *stat_bwd = nullptr;  // fixed-point replica slabs [R][C][2]
  hipEvent_t ev_dy = nullptr;
  long long slab_off = 0;
  int nsplit = 1;
  const half_t* gemm_fwd = nullptr;  // forward weights in the GEMM-shaped kernel's ring image order (per batch plan)
  int gemm_fwd_bn = 0, gemm_fwd_kc = 0;
  const half_t* tile_fwd = nullptr;  // forward weights in the row-band kernel's LDS image order (per batch plan)
  int tile_fwd_bn = 0;
};

struct PoolRt {
  uint8_t* idx = nullptr;
  int in_accum = 0;
};

struct Buf {
  cvx_buf_desc d;
  half_t* act = nullptr;
  half_t* grad = nullptr;
};

}  // namespace

struct cvx_engine {
  int device = 0;
  hipStream_t stream = nullptr;
  std::vector<Buf> bufs;
  std::vector<cvx_op_desc> ops;
  std::vector<ConvRt> conv;  // indexed by op
  std::vector<PoolRt> pool;  // indexed by op
  int image_buf = -1, pred_buf = -1;
  float* params = nullptr;
  float* grads = nullptr;
  float* stats = nullptr;
  int64_t n_params = 0, n_stats = 0;
  float bn_eps = 1e-3f, bn_momentum = 0.03f;
  // static device memory
  std::vector<void*> static_allocs;
  half_t* shadow = nullptr;
  long long shadow_elems = 0;
  half_t* zero_page = nullptr;  // DMA source for conv padding
  PackDesc* d_pack = nullptr;
  BlockRef* d_pack_blocks = nullptr;
  int n_pack_blocks = 0;
  // batch plan
  int planned_batch = 0;
  bool planned_train = false;
  std::vector<void*> batch_allocs;
  int64_t batch_bytes = 0, static_bytes = 0;
  long long* stat_region = nullptr;  // [fwd slabs | bwd slabs] of every conv op, zeroed once per pass
  long long stat_half = 0;           // entries per half
  hipStream_t side = nullptr;    // weight gradients run here, concurrently with the data-gradient chain
  hipStream_t red = nullptr;     // gradient-slab reduction (whole-pass backward): off the main chain as well
  // ops with cvx_op_desc.lane >= 2 (Detect levels 1 and 2: short, latency-bound chains, independent of level 0's) run on
  // this stream, beside the main chain, between one fork and one join per pass
  hipStream_t lane = nullptr;
  hipEvent_t ev_lane_fork = nullptr, ev_lane_join = nullptr, ev_pack = nullptr;
  bool use_lanes = false;
  // FORWARD overlap of the Detect levels with the neck's bottom-up path: level l may start on the lane stream as soon as the op that produces its
  // input has run on the main stream (head_producer[l], plan time), not only after the whole neck
  hipEvent_t ev_head[3] = {nullptr, nullptr, nullptr};
  int head_producer[3] = {-1, -1, -1};
  int head_join_op[3] = {-1, -1, -1};  // BACKWARD: the first main-chain op (highest index) whose backward touches the gradient of level l's input
  bool head_overlap = false;
  float* ytmp_lane = nullptr;         // raw fp32 conv output of the lane's layer in flight
  hipEvent_t ev_red = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool image_nhwc = false;            // a non-stem op reads the image: keep an NHWC fp16 copy (channels padded to 8)
  float* ytmp = nullptr;              // raw fp32 conv output of the layer in flight (training forward), shared by all layers
  const float* last_images = nullptr;  // the caller's images of the last training forward (the stem's weight gradient reads them)
  bool inference_only = false;         // the graph holds ops / epilogues without a backward pass (DLA-34): eval forward only
  int64_t plan_generation = 0;         // bumped whenever plan_batch re-allocates: captured hipGraphs hold raw buffer pointers
  float* slabs = nullptr;
  SlabDesc* d_slab = nullptr;
  BlockRef* d_slab_blocks = nullptr;
  int n_slab_blocks = 0;
  // the slab reduction runs in two parts: everything but the first `tail` conv ops as soon as THEIR weight gradients are
  // done (overlapping the last, largest-image weight gradients on the side stream), then the rest
  ColsumDesc* d_colsum = nullptr;    // bias gradients of every conv+bias op in one reduction (cvx_colsum_multi)
  ColsumBlock* d_colsum_blocks = nullptr;
  int n_colsum = 0, n_colsum_blocks = 0, colsum_max_c = 0;
  BnFoldDesc* d_fold = nullptr;  // eval-mode BN fold table (one entry per BN conv), rebuilt with the batch plan
  int n_fold = 0;
  int slab_tail_blocks = 0;  // reducer workgroups of the first ops (the tail of the backward pass)
  int slab_tail_op = -1;     // op index of the last conv op outside the tail (-1: single reduction)
  hipEvent_t ev_mid = nullptr;
  hipEvent_t ev_seg[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // segmented backward: (main, side) pairs
  int ev_seg_next = 0;
  unsigned long long seed = 0, train_pass = 0;  // dropout: cvx_engine_set_seed, training forwards so far
  bool has_bias_act = false;  // a conv + bias (+ ReLU) epilogue without BatchNorm exists (its scale / shift table entry is needed in training too)
  bool fwd_train_done = false;
  // cvx_engine_keep_shadows: the caller vouches that no parameter changed since this engine's previous forward -- the next forward skips
  // the fp32 -> fp16 weight conversion and the kernels' packed weight images if they were made for this plan and cover this mode
  bool keep_shadows_once = false;
  int shadows_gen = -1;        // plan generation the shadows / packed images were made under
  bool shadows_train = false;  // ... by a training forward (which packs the data-gradient images too)
  bool bwd_slabs_clean = false;  // the backward statistic slabs were zeroed by the training forward and not used since
  int last_batch = 0;
  float* last_pred = nullptr;
  // per-kernel-class profiling with HIP events on the launch stream (bench.py's roofline object)
  bool profile = false;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct ProfRec {
    int cls;
    double flops, bytes;
    int op;
  };
  int cur_op = -1;  // op index the launch loops are at (profile records carry it)
#ifdef CVX_WITH_CHAIN
  // eval-mode fusion groups (conv_chain.hip): a Bottleneck's two 3x3 convs, or a whole Detect level, as one tile-resident launch
  struct FusedGroup {
    int first = 0, last = 0;  // op range the launch replaces
    ChainPlan plan;
    void* mem = nullptr;
  };
  std::vector<FusedGroup> fused;
  std::vector<int> fused_at;              // op index -> group that STARTS there, -1 otherwise
  ChainPackJob* d_chain_jobs = nullptr;   // weight pre-pack jobs of all groups (one launch per forward)
  int n_chain_jobs = 0, chain_max_units = 0;
  bool chain_fusion = false;  // off by default: parity-green but 1-6 % slower than the per-layer kernels at batch 32; cvx_engine_set_fusion
#endif
  std::vector<signed char> sppf3;         // per op: 1 = first of three chained 5x5 max pools (SPPF) that go out as one launch, 2 = the other two
  // GEMM-shaped conv kernel: every routed layer's weights are re-ordered by ONE launch per forward, next to cvx_pack_weights
  half_t* gemm_arena = nullptr;
  GemmPackJob* d_gemm_jobs = nullptr;
  int n_gemm_jobs = 0, n_gemm_fwd_jobs = 0, gemm_blocks = 0, gemm_fwd_blocks = 0;
  // row-band conv kernel (conv_tile.hip): likewise
  half_t* tile_arena = nullptr;
  void* d_tile_jobs = nullptr;
  int n_tile_jobs = 0, n_tile_fwd_jobs = 0, tile_blocks = 0, tile_fwd_blocks = 0;
  struct cvx_bw_state* bw = nullptr;  // backward-pass state (whole-pass and segmented entry points)
  std::vector<ProfRec> prof_recs;
};

void cvx_engine_free_bw(cvx_engine* e);  // defined next to cvx_bw_state

namespace {

int dev_alloc(cvx_engine* e, std::vector<void*>& pool, int64_t& counter, void** out, long long bytes) {
  if (bytes <= 0) bytes = 16;
  bytes = (bytes + 255) & ~255LL;
  CVX_HIP(hipMalloc(out, (size_t)bytes));
  pool.push_back(*out);
  counter += bytes;
  return 0;
}
template <typename T>
int upload(cvx_engine* e, std::vector<void*>& pool, int64_t& counter, T** out, const std::vector<T>& host) {
  void* p = nullptr;
  CVX_TRY(dev_alloc(e, pool, counter, &p, (long long)(host.size() * sizeof(T))));
  if (!host.empty()) CVX_HIP(hipMemcpy(p, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)p;
  return 0;
}
void free_pool(std::vector<void*>& pool) {
  for (void* p : pool) (void)hipFree(p);
  pool.clear();
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

enum { PROF_CONV_FWD = 0, PROF_CONV_DGRAD = 1, PROF_CONV_WGRAD = 2, PROF_BN_FWD = 3, PROF_BN_BWD = 4, PROF_MISC = 5, PROF_SLAB_REDUCE = 6 };

struct ProfScope {  // records a start/stop event pair around the launches issued while it lives
  cvx_engine* e;
  hipStream_t st;
  ProfScope(cvx_engine* eng, int cls, double flops, double bytes, hipStream_t stream = nullptr)
      : e(eng->profile ? eng : nullptr), st(stream ? stream : eng->stream) {
    if (!e) return;
    while (e->ev_pool.size() < e->ev_used + 2) {
      hipEvent_t ev;
      if (hipEventCreate(&ev) != hipSuccess) {
        e = nullptr;
        return;
      }
      e->ev_pool.push_back(ev);
    }
    e->prof_recs.push_back({cls, flops, bytes, e->cur_op});
    slot = e->ev_used;
    e->ev_used += 2;
    (void)hipEventRecord(e->ev_pool[slot], st);
  }
  size_t slot = 0;
  ~ProfScope() {
    if (!e) return;
    (void)hipEventRecord(e->ev_pool[slot + 1], st);
  }
};

double conv_flops(const cvx_op_desc& o, int B) { return 2.0 * B * o.oh * o.ow * o.out.c * (double)(o.k * o.k * o.w_cin); }
double conv_bytes(const cvx_op_desc& o, int B) {
  return 2.0 * B * ((double)o.ih * o.iw * o.w_cin + (double)o.oh * o.ow * o.out.c) + 2.0 * o.out.c * o.k * o.k * o.w_cin;
}

ViewDesc make_view(const cvx_engine* e, const cvx_view& v, bool grad) {
  ViewDesc r{nullptr, 0, 0};
  if (v.buf < 0) return r;
  const Buf& b = e->bufs[v.buf];
  half_t* base = grad ? b.grad : b.act;
  r.ld = b.d.c;
  r.bstride = (long long)b.d.h * b.d.w * b.d.c;
  r.p = base + (long long)v.pix_off * b.d.c + v.coff;
  return r;
}

// activation kind of the BN passes / eval epilogue: 0 SiLU, 1 ReLU, 2 none
int act_kind(const cvx_op_desc& o) { return o.act == CVX_ACT_BN_SILU ? 0 : ((o.act == CVX_ACT_BN_LINEAR || o.act == CVX_ACT_BIAS_LINEAR) ? 2 : 1); }
float dropout_p(const cvx_op_desc& o) { return (float)o.k / 65536.f; }
unsigned long long dropout_seed(const cvx_engine* e, int op) {
  return e->seed * 0x9E3779B97F4A7C15ULL + (unsigned long long)e->train_pass * 1000003ULL + (unsigned long long)op;
}

int build_static(cvx_engine* e) {
  const int nops = (int)e->ops.size();
  e->conv.assign(nops, ConvRt());
  e->pool.assign(nops, PoolRt());
  std::vector<PackDesc> packs;
  std::vector<BlockRef> pblocks;
  long long sh = 0;
  for (int i = 0; i < nops; ++i) {
    const cvx_op_desc& o = e->ops[i];
    CVX_CHECK(o.in.buf >= 0 && o.in.buf < (int)e->bufs.size() && o.out.buf >= 0 && o.out.buf < (int)e->bufs.size(), "op view buffer index");
    CVX_CHECK(o.type >= CVX_OP_CONV && o.type <= CVX_OP_DROPOUT, "unknown op type");
    if (o.type == CVX_OP_CONV && (o.act == CVX_ACT_BIAS_RELU || o.act == CVX_ACT_BIAS_LINEAR) && o.res.buf >= 0) e->inference_only = true;
    if (o.type == CVX_OP_CONV && o.res.buf >= 0 && o.act != CVX_ACT_BIAS) {
      const bool pre = (o.flags & CVX_OPF_RES_PRE_ACT) != 0;
      // the training passes hold relu(z + res), silu(z + res) and silu(z) + res (bn_act.hip); relu(z) + res exists in the eval epilogue only
      if (o.act == CVX_ACT_BN_RELU && !pre) e->inference_only = true;
    }
    if (o.type == CVX_OP_DROPOUT) CVX_CHECK(o.k >= 0 && o.k < 65536, "dropout: k = drop probability in units of 2^-16");
    if (o.type == CVX_OP_CONV && (o.act == CVX_ACT_BIAS_RELU || o.act == CVX_ACT_BIAS_LINEAR)) e->has_bias_act = true;
    if (o.type != CVX_OP_CONV) {
      CVX_CHECK(o.in.c % 8 == 0 && o.in.coff % 8 == 0 && o.out.coff % 8 == 0 && o.in.c == o.out.c, "pool / resample / copy views: equal, 8-aligned channel slices");
      continue;
    }
    ConvRt& c = e->conv[i];
    const int T = o.k * o.k;
    CVX_CHECK(T <= CVX_MAX_TAPS, "kernel too large");
    if (o.in.buf == e->image_buf) {
      // YOLO stem: read as the caller's NCHW fp32 tensor by the fp32 stem kernels; no fp16 NHWC copy of the image exists.
      // Any other first layer (DLA's 7x7): inference-only, through an NHWC fp16 copy with the channels padded to 8.
      c.stem = o.w_cin == 3 && o.k == 3 && o.stride == 2 && o.pad == 1 && o.dil == 1 && o.act == CVX_ACT_BN_SILU && !o.needs_dgrad &&
               o.res.buf < 0 && o.out.c % 16 == 0 && o.out.c <= 80 && o.ih % 2 == 0 && o.iw % 2 == 0;
      if (!c.stem) {  // any other first layer (ResNet's / DLA's 7x7): through an NHWC fp16 copy of the image, channels padded to 8
        CVX_CHECK(o.w_cin == 3 && !o.needs_dgrad, "an op that reads the image needs 3 stored input channels and no data gradient");
        e->image_nhwc = true;
      }
    }
    CVX_CHECK(o.in.c % 8 == 0 && o.out.c % 8 == 0 && o.in.coff % 8 == 0 && o.out.coff % 8 == 0, "conv views must be 8-channel aligned");
    CVX_CHECK(o.w_cin <= o.in.c && o.w_cin > o.in.c - 8, "w_cin vs view channels");
    c.cin_g = o.in.c;
    c.cin_pad16 = round_up(o.in.c, 16);
    c.ntaps = T;
    std::vector<ConvTap> taps(T);
    for (int r = 0; r < o.k; ++r)
      for (int s = 0; s < o.k; ++s) taps[r * o.k + s] = ConvTap{r * o.dil - o.pad, s * o.dil - o.pad, r * o.k + s, 0};
    CVX_TRY(upload(e, e->static_allocs, e->static_bytes, &c.taps_fwd, taps));
    c.halo_ok = cvx_halo_pack_taps(taps.data(), T, &c.halo_pos, &c.halo_wt);
    c.pointwise = cvx_taps_pointwise(taps.data(), T);
    c.std3x3 = cvx_taps_std3x3(taps.data(), T);
    c.std7x7 = cvx_taps_std7x7(taps.data(), T);
    // shadow weights
    PackDesc pd;
    pd.src_off = o.w_off;
    pd.Cout = o.out.c;
    pd.T = T;
    pd.Cin = o.w_cin;
    pd.Cin_pad = o.in.c;
    pd.fwd_off = sh;
    c.sh_fwd = sh;
    sh += (long long)pd.Cout * T * pd.Cin_pad;
    sh = (sh + 7) & ~7LL;
    pd.dg_off = -1;
    if (o.needs_dgrad) {
      pd.dg_off = sh;  // [Cin_pad][T][Cout]: the rows of padded input channels are never written and stay zero (memset below)
      c.sh_dg = sh;
      sh += (long long)pd.Cin_pad * T * pd.Cout;
      sh = (sh + 7) & ~7LL;
      // data-gradient tap classes: one per output phase of the forward stride
      const int S = o.stride;
      CVX_CHECK(S * S <= 16, "stride too large");
      // 3x3 / stride 2 / pad 1 on an even map: every phase's taps lie in the 2 x 2 window (dh, dw in {0, 1}) of dy -- the four phases are
      // four channel blocks of ONE stride-1 GEMM (pixel-shuffle data gradient, see ConvRt::sh_ps)
      bool ps_shape = S == 2 && o.k == 3 && o.pad == 1 && o.dil == 1 && o.ih == 2 * o.oh && o.iw == 2 * o.ow && o.out.c % 32 == 0;
      for (int q = 0; q < 16; ++q) c.ps_desc.wtap[q] = -1;
      for (int ph = 0; ph < S; ++ph)
        for (int pw = 0; pw < S; ++pw) {
          std::vector<ConvTap> dt;
          for (int r = 0; r < o.k; ++r) {
            int nh = ph + o.pad - r * o.dil;
            if (((nh % S) + S) % S != 0) continue;
            for (int s = 0; s < o.k; ++s) {
              int nw = pw + o.pad - s * o.dil;
              if (((nw % S) + S) % S != 0) continue;
              dt.push_back(ConvTap{nh / S, nw / S, r * o.k + s, 0});  // exact division
              if (ps_shape) {
                const int dh = nh / S, dw = nw / S;
                if (dh < 0 || dh > 1 || dw < 0 || dw > 1)
                  ps_shape = false;
                else
                  c.ps_desc.wtap[(ph * 2 + pw) * 4 + dh * 2 + dw] = r * o.k + s;
              }
            }
          }
          DgClass& dc = c.dg[c.ndg++];
          dc.ntaps = (int)dt.size();
          dc.oph = ph;
          dc.opw = pw;
          dc.OH2 = (o.ih - ph + S - 1) / S;
          dc.OW2 = (o.iw - pw + S - 1) / S;
          if (dc.ntaps > 0) CVX_TRY(upload(e, e->static_allocs, e->static_bytes, &dc.taps, dt));
          dc.halo_ok = cvx_halo_pack_taps(dt.data(), dc.ntaps, &dc.halo_pos, &dc.halo_wt);
          dc.pointwise = dc.ntaps > 0 ? cvx_taps_pointwise(dt.data(), dc.ntaps) : 0;
        }
      if (ps_shape) {
        c.sh_ps = sh;
        sh += 16LL * pd.Cin_pad * pd.Cout;
        sh = (sh + 7) & ~7LL;
        c.ps_desc.dg_off = pd.dg_off;
        c.ps_desc.ps_off = c.sh_ps;
        c.ps_desc.cin_pad = pd.Cin_pad;
        c.ps_desc.T = T;
        c.ps_desc.C = pd.Cout;
        std::vector<ConvTap> pt(4);
        for (int tau = 0; tau < 4; ++tau) pt[tau] = ConvTap{tau >> 1, tau & 1, tau, 0};
        CVX_TRY(upload(e, e->static_allocs, e->static_bytes, &c.ps_taps, pt));
      }
    }
    const int total = pd.Cout * T * pd.Cin_pad;
    for (int s0 = 0; s0 < total; s0 += 1024) pblocks.push_back(BlockRef{(int)packs.size(), s0});
    if (pd.dg_off >= 0) {  // transposed copy: 32 x 32 (co, ci) tiles per tap, encoded as negative starts (pack_weights_kernel)
      const int ntile = T * ((pd.Cout + 31) / 32) * ((pd.Cin + 31) / 32);
      for (int q = 0; q < ntile; ++q) pblocks.push_back(BlockRef{(int)packs.size(), -(q + 1)});
    }
    packs.push_back(pd);
  }
  e->shadow_elems = sh;
  void* p = nullptr;
  CVX_TRY(dev_alloc(e, e->static_allocs, e->static_bytes, &p, sh * 2));
  e->shadow = (half_t*)p;
  CVX_HIP(hipMemset(p, 0, (size_t)(sh * 2)));
  CVX_TRY(dev_alloc(e, e->static_allocs, e->static_bytes, &p, 256));
  e->zero_page = (half_t*)p;
  CVX_HIP(hipMemset(p, 0, 256));
  CVX_TRY(upload(e, e->static_allocs, e->static_bytes, &e->d_pack, packs));
  CVX_TRY(upload(e, e->static_allocs, e->static_bytes, &e->d_pack_blocks, pblocks));
  e->n_pack_blocks = (int)pblocks.size();

  // ---- forward plan check: activations are operands of the backward pass, so no slice may be produced twice
  {
    std::vector<std::vector<char>> produced(e->bufs.size());
    for (size_t b = 0; b < e->bufs.size(); ++b) produced[b].assign(e->bufs[b].d.c, 0);
    for (int i = 0; i < nops; ++i) {
      const cvx_op_desc& o = e->ops[i];
      if (e->bufs[o.out.buf].d.kind == CVX_BUF_PRED_F32) continue;
      CVX_CHECK(o.out.coff >= 0 && o.out.coff + o.out.c <= e->bufs[o.out.buf].d.c, "output view exceeds its buffer");
      CVX_CHECK(o.in.coff >= 0 && o.in.coff + o.in.c <= e->bufs[o.in.buf].d.c, "input view exceeds its buffer");
      for (int ch = o.out.coff; ch < o.out.coff + o.out.c; ++ch) {
        CVX_CHECK(!produced[o.out.buf][ch], "forward plan: buffer slice written twice (op " + std::to_string(i) + ")");
        produced[o.out.buf][ch] = 1;
      }
    }
  }
  {  // a graph described without any data gradient (needs_dgrad = 0 throughout) was built for inference
    bool any_dgrad = false;
    for (int i = 0; i < nops; ++i) any_dgrad |= e->ops[i].type == CVX_OP_CONV && e->ops[i].needs_dgrad;
    if (!any_dgrad && nops > 1) e->inference_only = true;
  }
  if (e->inference_only) return 0;  // no backward pass exists for these graphs
  // ---- backward write/accumulate plan -----------------------------------------------------
  std::vector<std::vector<char>> written(e->bufs.size());
  for (size_t b = 0; b < e->bufs.size(); ++b) written[b].assign(e->bufs[b].d.c, 0);
  auto claim = [&](const cvx_view& v, int* accum) -> int {
    int nw = 0;
    for (int ch = v.coff; ch < v.coff + v.c; ++ch) nw += written[v.buf][ch];
    CVX_CHECK(nw == 0 || nw == v.c, "backward plan: gradient slice partially written");
    *accum = nw == v.c;
    for (int ch = v.coff; ch < v.coff + v.c; ++ch) written[v.buf][ch] = 1;
    return 0;
  };
  for (int i = nops - 1; i >= 0; --i) {
    const cvx_op_desc& o = e->ops[i];
    if (e->bufs[o.out.buf].d.kind != CVX_BUF_PRED_F32) {
      for (int ch = o.out.coff; ch < o.out.coff + o.out.c; ++ch)
        CVX_CHECK(written[o.out.buf][ch], "backward plan: an op output has no consumer (op " + std::to_string(i) + ")");
    }
    if (o.type == CVX_OP_CONV) {
      if (o.res.buf >= 0) CVX_TRY(claim(o.res, &e->conv[i].res_accum));
      if (o.needs_dgrad) CVX_TRY(claim(o.in, &e->conv[i].in_accum));
    } else {
      CVX_TRY(claim(o.in, &e->pool[i].in_accum));
    }
  }
  return 0;
}

bool same_view(const cvx_view& a, const cvx_view& b) { return a.buf == b.buf && a.coff == b.coff && a.c == b.c && a.pix_off == b.pix_off; }

#ifdef CVX_WITH_CHAIN
// ---- eval-mode fusion groups: recognised in the op list, planned as tile-resident chains (conv_chain.hip) ----
bool is_bn_silu_3x3(const cvx_op_desc& o) {
  return o.type == CVX_OP_CONV && o.k == 3 && o.stride == 1 && o.pad == 1 && o.dil == 1 && o.act == CVX_ACT_BN_SILU && !(o.flags & CVX_OPF_CONV_BIAS);
}
// true when no op other than `reader` reads (in / res) buffer `buf`
bool sole_reader(const cvx_engine* e, int buf, int reader) {
  for (size_t i = 0; i < e->ops.size(); ++i) {
    if ((int)i == reader) continue;
    if (e->ops[i].in.buf == buf || e->ops[i].res.buf == buf) return false;
  }
  return true;
}
ChainConvArgs chain_args_bn(const cvx_engine* e, int i) {
  const cvx_op_desc& o = e->ops[i];
  const ConvRt& c = e->conv[i];
  return ChainConvArgs{e->shadow + c.sh_fwd, c.ntaps * c.cin_g, o.out.c, c.scale, c.shift, nullptr, 0, 1};
}
ChainConvArgs chain_args_bias(const cvx_engine* e, int i) {
  const cvx_op_desc& o = e->ops[i];
  const ConvRt& c = e->conv[i];
  return ChainConvArgs{e->shadow + c.sh_fwd, c.ntaps * c.cin_g, o.out.c, nullptr, nullptr, e->params + o.bias_off, 2, 1};
}
// best tile (estimated time) among the divisor pairs of (H, W) for which `make_spec` yields a feasible plan; false if none
template <typename MakeSpec>
bool choose_tile(int H, int W, int max_th, int max_tw, MakeSpec make_spec, int* th_out, int* tw_out, double* cost_out, int pref_th = 0, int pref_tw = 0) {
  double best = -1;
  if (pref_th > 0 && pref_tw > 0 && H % pref_th == 0 && W % pref_tw == 0) {  // the tile measured best on this shape class (tools/chain_probe.py sweep)
    ChainSpec sp;
    ChainPlan plan;
    void* unused = nullptr;
    if (make_spec(&sp, pref_th, pref_tw) == 0 && cvx_chain_plan(sp, &plan, &unused, true) == 0) {
      *th_out = pref_th;
      *tw_out = pref_tw;
      *cost_out = cvx_chain_cost_us(sp, plan);
      return true;
    }
    cvx_set_error("");
  }
  for (int th = 1; th <= std::min(H, max_th); ++th) {
    if (H % th) continue;
    for (int tw = 1; tw <= std::min(W, max_tw); ++tw) {
      if (W % tw || th * tw < 32) continue;
      ChainSpec sp;
      if (make_spec(&sp, th, tw) != 0) continue;
      ChainPlan plan;
      void* unused = nullptr;
      if (cvx_chain_plan(sp, &plan, &unused, true) != 0) continue;
      const double c = cvx_chain_cost_us(sp, plan);
      if (best < 0 || c < best) {
        best = c;
        *th_out = th;
        *tw_out = tw;
      }
    }
  }
  cvx_set_error("");
  if (best < 0) return false;
  *cost_out = best;
  return true;
}

void free_fused(cvx_engine* e) {
  for (auto& g : e->fused)
    if (g.mem) (void)hipFree(g.mem);
  e->fused.clear();
  e->fused_at.assign(e->ops.size(), -1);
  if (e->d_chain_jobs) (void)hipFree(e->d_chain_jobs);
  e->d_chain_jobs = nullptr;
  e->n_chain_jobs = e->chain_max_units = 0;
}

// Bottleneck pairs (core/models/yolov8/modules.py:124-135) with 64+ channels and the Detect levels of at most 40x40 cells
// (modules.py:428-433) run as ONE launch each in eval mode; everything else keeps its per-layer kernels (measured: tools/chain_probe.py
// -- at 32 channels and below, and at 80x80, the per-layer kernels are as fast or faster).
int plan_fused_groups(cvx_engine* e, int B) {
  free_fused(e);
  static const bool off = cvx_tune_set("CVX_NO_CHAIN");
  if (off || !e->chain_fusion || !e->zero_page) return 0;
  const Buf& pb = e->bufs[e->pred_buf];
  const long long A = (long long)pb.d.h * pb.d.w;
  std::vector<ChainPackJob> all_jobs;
  const int nops = (int)e->ops.size();
  for (int i = 0; i < nops; ++i) {
    const cvx_op_desc& o = e->ops[i];
    cvx_engine::FusedGroup g;
    ChainSpec sp;
    bool ok = false;
    // ---- Detect level: 3x3 (box | class) -> 3x3, 3x3 -> 1x1 + bias, 1x1 + bias ----
    static const int kinds = cvx_tune_int("CVX_CHAIN_KINDS", 3);  // tuning build: bit 0 Bottleneck pairs, bit 1 Detect levels
    if ((kinds & 2) && i + 4 < nops && is_bn_silu_3x3(o) && is_bn_silu_3x3(e->ops[i + 1]) && is_bn_silu_3x3(e->ops[i + 2]) && e->ops[i + 3].type == CVX_OP_CONV &&
        e->ops[i + 3].act == CVX_ACT_BIAS && e->ops[i + 3].k == 1 && e->ops[i + 4].type == CVX_OP_CONV && e->ops[i + 4].act == CVX_ACT_BIAS &&
        e->ops[i + 4].k == 1 && o.res.buf < 0) {
      const cvx_op_desc &b1 = e->ops[i + 1], &b2 = e->ops[i + 2], &o1 = e->ops[i + 3], &o2 = e->ops[i + 4];
      const int cb = b1.out.c, cc = b2.out.c;
      const bool shape = o.out.c == cb + cc && b1.in.buf == o.out.buf && b1.in.coff == o.out.coff && b1.in.c == cb && b2.in.buf == o.out.buf &&
                         b2.in.coff == o.out.coff + cb && b2.in.c == cc && same_view(o1.in, b1.out) && same_view(o2.in, b2.out) && o1.out.c == 64 &&
                         o1.out.buf == e->pred_buf && o2.out.buf == e->pred_buf && o1.out.coff == 0 && o2.out.coff == 64 &&
                         o1.out.pix_off == o2.out.pix_off && b1.res.buf < 0 && b2.res.buf < 0 && o.ih * o.iw <= 1600 && cb % 16 == 0 && cc % 16 == 0 &&
                         e->conv[i].cin_g % 16 == 0 && sole_reader(e, b1.out.buf, i + 3) && sole_reader(e, b2.out.buf, i + 4);
      bool readers = shape;
      for (int q = 0; readers && q < nops; ++q)
        if (q != i + 1 && q != i + 2 && (e->ops[q].in.buf == o.out.buf || e->ops[q].res.buf == o.out.buf)) readers = false;
      if (readers) {
        const ViewDesc xin = make_view(e, o.in, false);
        auto mk = [&](ChainSpec* s, int th, int tw) {
          return cvx_chain_spec_detect(s, xin.p, xin.bstride, xin.ld, B, o.ih, o.iw, e->conv[i].cin_g, cb, cc, o2.out.c, chain_args_bn(e, i),
                                       chain_args_bn(e, i + 1), chain_args_bn(e, i + 2), chain_args_bias(e, i + 3), chain_args_bias(e, i + 4), nullptr,
                                       A * pb.d.c, pb.d.c, o1.out.pix_off, th, tw, e->zero_page);
        };
        int th, tw;
        double cost;
        // (the destination is the caller's `pred`, known at forward time: the plan holds offsets, the base comes with the launch)
        // measured: 40x40 -> 8 x 10 tiles, 20x20 -> 5 x 10
        const bool big = o.ih * o.iw > 400;
        if (choose_tile(o.ih, o.iw, 10, 20, mk, &th, &tw, &cost, big ? o.ih / 5 : o.ih / 4, big ? o.iw / 4 : o.iw / 2)) {
          mk(&sp, th, tw);
          g.first = i;
          g.last = i + 4;
          ok = true;
        }
      }
    }
    // ---- Bottleneck pair: 3x3 -> 3x3 (+ shortcut) ----
    if ((kinds & 1) && !ok && i + 1 < nops && is_bn_silu_3x3(o) && is_bn_silu_3x3(e->ops[i + 1]) && o.res.buf < 0) {
      const cvx_op_desc& o2 = e->ops[i + 1];
      const int C = o.out.c;
      const bool shortcut = o2.res.buf >= 0;
      if (same_view(o2.in, o.out) && e->conv[i].cin_g == C && o2.out.c == C && C >= 64 && C % 16 == 0 && (!shortcut || same_view(o2.res, o.in)) &&
          o.out.buf != o.in.buf && sole_reader(e, o.out.buf, i + 1) && e->bufs[o.out.buf].d.c == C) {
        const ViewDesc xin = make_view(e, o.in, false), yout = make_view(e, o2.out, false);
        auto mk = [&](ChainSpec* s, int th, int tw) {
          return cvx_chain_spec_pair(s, xin.p, xin.bstride, xin.ld, B, o.ih, o.iw, C, chain_args_bn(e, i), chain_args_bn(e, i + 1), shortcut, yout.p,
                                     yout.bstride, yout.ld, th, tw, e->zero_page);
        };
        int th, tw;
        double cost;
        if (choose_tile(o.ih, o.iw, 16, 40, mk, &th, &tw, &cost, o.ih / 4, o.iw / 2)) {  // measured: 40x40 -> 10 x 20, 20x20 -> 5 x 10
          mk(&sp, th, tw);
          g.first = i;
          g.last = i + 1;
          ok = true;
        }
      }
    }
    if (!ok) continue;
    if (cvx_chain_plan(sp, &g.plan, &g.mem) != 0) {
      if (g.mem) (void)hipFree(g.mem);
      cvx_set_error("");
      continue;
    }
    for (int q = 0; q < g.plan.njobs; ++q) all_jobs.push_back(g.plan.jobs[q]);
    e->chain_max_units = std::max(e->chain_max_units, g.plan.max_job_units);
    e->fused_at[i] = (int)e->fused.size();
    e->fused.push_back(g);
    i = g.last;
  }
  if (!all_jobs.empty()) {
    CVX_HIP(hipMalloc((void**)&e->d_chain_jobs, all_jobs.size() * sizeof(ChainPackJob)));
    CVX_HIP(hipMemcpy(e->d_chain_jobs, all_jobs.data(), all_jobs.size() * sizeof(ChainPackJob), hipMemcpyHostToDevice));
    e->n_chain_jobs = (int)all_jobs.size();
  }
  return 0;
}
#else
void free_fused(cvx_engine*) {}
int plan_fused_groups(cvx_engine*, int) { return 0; }
#endif

int plan_gemm_packs(cvx_engine* e, int B, bool training);
void free_gemm_packs(cvx_engine* e);
int plan_tile_packs(cvx_engine* e, int B, bool training);
void free_tile_packs(cvx_engine* e);

int plan_batch(cvx_engine* e, int B, bool training) {
  if (e->planned_batch == B && (e->planned_train || !training)) return 0;
  CVX_HIP(hipStreamSynchronize(e->stream));
  if (e->side) CVX_HIP(hipStreamSynchronize(e->side));
  if (e->lane) CVX_HIP(hipStreamSynchronize(e->lane));
  free_pool(e->batch_allocs);
  free_fused(e);
  free_gemm_packs(e);
  free_tile_packs(e);
  e->batch_bytes = 0;
  e->planned_batch = 0;
  e->plan_generation++;  // every per-batch buffer moves: a hipGraph captured against the old plan must be dropped
  e->fwd_train_done = false;
  void* p = nullptr;
  long long ytmp_elems = 0, ytmp_lane_elems = 0;
  for (size_t bi = 0; bi < e->bufs.size(); ++bi) {
    Buf& b = e->bufs[bi];
    b.act = b.grad = nullptr;
    if (b.d.kind != CVX_BUF_ACT_F16 || ((int)bi == e->image_buf && !e->image_nhwc)) continue;
    long long bytes = (long long)B * b.d.h * b.d.w * b.d.c * 2;
    CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, bytes));
    b.act = (half_t*)p;
    if (training) {
      CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, bytes));
      b.grad = (half_t*)p;
    }
  }
  long long stat_floats = 0, slab_total = 0;
  std::vector<SlabDesc> sdescs;
  std::vector<BlockRef> sblocks;
  std::vector<BnFoldDesc> folds;
  e->slab_tail_blocks = 0;
  e->slab_tail_op = -1;
  for (size_t i = 0; i < e->ops.size(); ++i) {
    const cvx_op_desc& o = e->ops[i];
    if (o.type == CVX_OP_L2NORM) {
      if (training) {  // per-workgroup partial sums of the weight gradient
        CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, (long long)cvx_l2norm_bwd_blocks((long long)B * o.ih * o.iw) * o.in.c * 4));
        e->pool[i].idx = (uint8_t*)p;
      }
      continue;
    }
    if (o.type == CVX_OP_DWCONVT) {
      if (training) {  // per-workgroup partial sums of the weight gradient
        CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, cvx_dwconvt_bwd_scratch_floats((long long)B * o.ih * o.iw, o.in.c, o.stride) * 4));
        e->pool[i].idx = (uint8_t*)p;
      }
      continue;
    }
    if (o.type == CVX_OP_MAXPOOL5 || o.type == CVX_OP_MAXPOOL3S2 || o.type == CVX_OP_MAXPOOL3S1) {
      if (training) {  // argmax byte per output element: the operand of the backward gather
        CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, (long long)B * o.oh * o.ow * o.out.c));
        e->pool[i].idx = (uint8_t*)p;
      }
      continue;
    }
    if (o.type != CVX_OP_CONV) continue;
    ConvRt& c = e->conv[i];
    const long long M = (long long)B * o.oh * o.ow;
    const int C = o.out.c;
    CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, 4LL * C * 4));
    c.mean = (float*)p;
    c.invstd = c.mean + C;
    c.scale = c.invstd + C;
    c.shift = c.scale + C;
    if (o.act == CVX_ACT_BN_SILU || o.act == CVX_ACT_BN_RELU || o.act == CVX_ACT_BN_LINEAR)
      folds.push_back(BnFoldDesc{o.gamma_off, o.beta_off, o.rmean_off, o.rvar_off, c.scale, c.shift, C, 0,
                                 (o.flags & CVX_OPF_CONV_BIAS) ? (long long)o.bias_off : -1LL});
    else if (o.act == CVX_ACT_BIAS_RELU || o.act == CVX_ACT_BIAS_LINEAR)
      folds.push_back(BnFoldDesc{0, o.bias_off, 0, 0, c.scale, c.shift, C, 1, -1LL});
    if (training && (o.act == CVX_ACT_BIAS_RELU || o.act == CVX_ACT_BIAS_LINEAR)) {  // no BatchNorm: only the masked gradient is kept
      CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, M * C * 2));
      c.dybuf = (half_t*)p;
    }
    // CVX_OPF_RAW_F16 is honoured for plain Conv + BatchNorm + SiLU layers: no residual, no bias in front of the BatchNorm, not the fp32 stem
    static const bool raw16_off = cvx_tune_set("CVX_NO_RAW_F16");
    c.raw16 = training && !raw16_off && (o.flags & CVX_OPF_RAW_F16) && o.act == CVX_ACT_BN_SILU && o.res.buf < 0 && !(o.flags & CVX_OPF_CONV_BIAS) && !c.stem;
    if (training && (o.act == CVX_ACT_BN_SILU || o.act == CVX_ACT_BN_RELU || o.act == CVX_ACT_BN_LINEAR)) {
      CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, M * C * 2));
      c.ybuf = (half_t*)p;
      if (!c.stem) {  // the stem's dy is never materialised (cvx_stem_backward)
        CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, M * C * 2));
        c.dybuf = (half_t*)p;
      }
      if (!c.stem && !c.raw16) ytmp_elems = std::max(ytmp_elems, M * C);
      if (e->use_lanes && (o.lane >= 2 || (e->head_overlap && o.lane >= 1)) && !c.raw16) ytmp_lane_elems = std::max(ytmp_lane_elems, M * C);
    }
    c.stat_fwd = (long long*)nullptr + stat_floats;  // offset for now, rebased below
    stat_floats += (long long)cvx_stat_replicas(C) * C * CVX_STAT_WORDS + CVX_STAT_GATE_WORDS;  // + the gate counter of the one-launch BN backward
    if (training) {
      int co_b, j_b;
      const int Jtot = c.ntaps * c.cin_pad16;
      cvx_conv_wgrad_tile(C, Jtot, &co_b, &j_b);
      // the GEMM-shaped weight-gradient kernel (conv_wgrad_gemm.hip) takes the BatchNorm / bias layers with 128+ output channels: its tiles
      // and its 512-thread workgroups size the pixel splits
      // the first conv ops are the LAST of the backward pass: their weight gradients are the tail of the step (nothing of the main chain is
      // left to disturb) and may keep the whole-chip workgroup counts -- measured: no difference (6.16-6.23 ms for 0 ... 7 such layers), so off
      static const int tail_convs = cvx_tune_int("CVX_WGRAD_TAIL_OPS", -1);
      const bool tail_layer = (int)i <= tail_convs;
      bool wgg = false;
      int ws_splits = 0;  // > 0: the streaming kernel (conv_wgrad_stream.hip) takes the layer, with its planner's pixel splits
      if (!c.stem) {
        WgradParams q;
        memset(&q, 0, sizeof(q));
        const Buf& xb = e->bufs[o.in.buf];
        q.x_ld = xb.d.c;
        q.x_bstride = (long long)xb.d.h * xb.d.w * xb.d.c;
        q.IH = o.ih;
        q.IW = o.iw;
        q.Cin = c.cin_g;
        q.dy_ld = C;
        q.dy_bstride = (long long)o.oh * o.ow * C;
        if (o.act == CVX_ACT_BIAS) {  // the head's output convs: dy is a slice of dpred
          const Buf& pb = e->bufs[e->pred_buf];
          q.dy_ld = pb.d.c;
          q.dy_bstride = (long long)pb.d.h * pb.d.w * pb.d.c;
        }
        q.Cout = C;
        q.B = B;
        q.OH = o.oh;
        q.OW = o.ow;
        q.stride = o.stride;
        q.ntaps = c.ntaps;
        q.cin_pad16 = c.cin_pad16;
        q.std3x3 = c.std3x3;
        if (cvx_conv_wgrad_k3_supported(q)) {
          ws_splits = cvx_conv_wgrad_k3_nsplit(q, tail_layer);
        } else if (cvx_conv_wgrad_stream_supported(q)) {
          ws_splits = cvx_conv_wgrad_stream_nsplit(q);
        } else if (o.act != CVX_ACT_BIAS && !cvx_conv_wgrad_halo_supported(q) && cvx_conv_wgrad_gemm_supported(q)) {
          wgg = true;
          cvx_conv_wgrad_gemm_tile(C, Jtot, &co_b, &j_b);
        }
      }
      const long long tiles = (long long)cvx_cdiv(C, co_b) * cvx_cdiv(Jtot, j_b);
      // the 128 x 128 tile (ResNet-sized layers) has few, fat tiles: it takes its parallelism from more pixel splits, and its slabs
      // may grow accordingly (measured on DeepLabv3+ R101: 8 -> 64 MB of slabs per layer took the step from 48.6 to 38.5 ms)
      static const long long blk_narrow = cvx_tune_int("CVX_WGRAD_BLOCKS", 2048), blk_wide = cvx_tune_int("CVX_WGRAD_BLOCKS_WIDE", 1024);
      static const long long blk_wgg_big = cvx_tune_int("CVX_WGG_BLOCKS_BIG", 256), blk_wgg = cvx_tune_int("CVX_WGG_BLOCKS", 512);
      // Round 5: these launches run beside the backward pass on the lowest-priority stream, and what they cost the step is the CUs their
      // workgroups hold while a main-chain kernel waits to be placed -- not their own duration.  A small layer therefore gets
      // only as many workgroups as its work needs (YOLOv8-n, same box: 512 -> 96 and 2048 -> 256 took 0.12 ms off the step); the big layers
      // of the other models keep the counts that were tuned on them.
      static const long long wgg_mflop = cvx_tune_int("CVX_WGG_MFLOP", 40), gen_mflop = cvx_tune_int("CVX_WGRAD_MFLOP", 20);
      static const long long wgg_min = cvx_tune_int("CVX_WGG_BLOCKS_MIN", 96), gen_min = cvx_tune_int("CVX_WGRAD_BLOCKS_MIN", 256);
      const double wflops = 2.0 * (double)M * C * c.cin_g * c.ntaps;
      const long long wgg_by_work = std::max(wgg_min, std::min(blk_wgg, (long long)(wflops / (wgg_mflop * 1e6))));
      const long long gen_by_work = std::max(gen_min, std::min(blk_narrow, (long long)(wflops / (gen_mflop * 1e6))));
      const long long blk_target = wgg ? (co_b == 256 ? blk_wgg_big : tail_layer ? blk_wgg : wgg_by_work) : co_b == 128 ? blk_wide : tail_layer ? blk_narrow : gen_by_work;
      long long ns = std::min<long long>(std::max<long long>(1, M / 256), std::max<long long>(1, blk_target / tiles));
      const long long slab_elems = (long long)C * Jtot;
      static const long long cap_narrow = cvx_tune_int("CVX_SLAB_MB", 8), cap_wide = cvx_tune_int("CVX_SLAB_MB_WIDE", 64);
      const long long slab_cap_mb = (wgg || co_b == 128) ? cap_wide : cap_narrow;
      static const long long ns_cap = cvx_tune_int("CVX_NSPLIT_CAP", 512);
      ns = std::min(ns, std::max<long long>(1, (slab_cap_mb << 20) / (slab_elems * 4)));
      ns = std::min<long long>(ns, ns_cap);
      {  // 3x3 stride-1 layers take the register-tile kernel: few, fat workgroups per pixel split
        WgradParams probe;
        memset(&probe, 0, sizeof(probe));
        probe.std3x3 = c.std3x3;
        probe.stride = o.stride;
        probe.ntaps = c.ntaps;
        probe.Cin = c.cin_g;
        probe.Cout = C;
        probe.IH = o.ih;
        probe.IW = o.iw;
        probe.OH = o.oh;
        probe.OW = o.ow;
        if (c.stem) {
          ns = cvx_stem_wgrad_splits(M);
        } else if (ws_splits > 0) {
          ns = ws_splits;
        } else if (cvx_conv_wgrad_halo_supported(probe)) {
          int gx, gy;
          cvx_conv_wgrad_halo_grid(C, c.cin_g, &gx, &gy);
          const long long ptiles = cvx_conv_wgrad_halo_tiles(B, o.oh, o.ow);
          // split count by the layer's work: 128 workgroups for the small layers (YOLOv8-n: 128 beats 64/256/512 -- slab volume vs.
          // parallelism), 512 from 8 GFLOP up (SSD300's 150^2 / 300^2 layers: -9 % on the whole step; tools/sweeps/sweep_wh.sh, sweep_wh2.sh)
          static const long long wh_small = cvx_tune_int("CVX_WH_BLOCKS", 128), wh_big = cvx_tune_int("CVX_WH_BLOCKS_BIG", 512);
          static const long long wh_big_gf = cvx_tune_int("CVX_WH_BIG_GF", 8);
          static const long long wh_slab_mb = cvx_tune_int("CVX_WH_SLAB_MB", 16);
          const double gflop = 2.0 * (double)M * C * c.cin_g * c.ntaps * 1e-9;
          const long long wh_blocks = gflop >= (double)wh_big_gf ? wh_big : wh_small;
          ns = std::max<long long>(1, wh_blocks / ((long long)gx * gy));
          ns = std::min(ns, ptiles);
          ns = std::min(ns, std::max<long long>(1, (wh_slab_mb << 20) / (slab_elems * 4)));
        }
      }
      c.nsplit = (int)ns;
      c.slab_off = slab_total;
      slab_total += ns * slab_elems;
      SlabDesc sd;
      sd.slab_off = c.slab_off;
      sd.dst_off = o.w_off;
      sd.nsplit = c.stem ? 0 : c.nsplit;  // the stem folds its own slabs / partial blocks (cvx_stem_backward_fold): nothing for the table-driven reducer
      sd.rows = C * c.ntaps;
      sd.Cin = o.w_cin;
      sd.Cin_pad = c.cin_pad16;
      sd.lanes = cvx_slab_lanes(sd.nsplit);
      const long long total = (long long)sd.rows * sd.Cin;
      static const int tail_ops = cvx_tune_int("CVX_SLAB_TAIL", 2);
      if ((int)sdescs.size() == tail_ops && tail_ops > 0) {  // this is the first conv op outside the tail
        e->slab_tail_blocks = (int)sblocks.size();
        e->slab_tail_op = (int)i;
      }
      c.slab_blk0 = (int)sblocks.size();
      for (long long s0 = 0; s0 < total; s0 += 256 / sd.lanes) sblocks.push_back(BlockRef{(int)sdescs.size(), (int)s0});
      c.slab_blk1 = (int)sblocks.size();
      sdescs.push_back(sd);
    }
  }
  e->ytmp = nullptr;
  if (training && ytmp_elems > 0) {
    CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, ytmp_elems * 4));
    e->ytmp = (float*)p;
  }
  e->ytmp_lane = nullptr;
  if (training && ytmp_lane_elems > 0) {
    CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, ytmp_lane_elems * 4));
    e->ytmp_lane = (float*)p;
  }
  CVX_TRY(upload(e, e->batch_allocs, e->batch_bytes, &e->d_fold, folds));
  e->n_fold = (int)folds.size();
  CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, 2 * stat_floats * 8));
  e->stat_region = (long long*)p;
  e->stat_half = stat_floats;
  for (size_t i = 0; i < e->ops.size(); ++i) {
    if (e->ops[i].type != CVX_OP_CONV) continue;
    ConvRt& c = e->conv[i];
    const long long off = c.stat_fwd - (long long*)nullptr;
    c.stat_fwd = e->stat_region + off;
    c.stat_bwd = e->stat_region + stat_floats + off;
  }
  e->n_colsum = e->n_colsum_blocks = e->colsum_max_c = 0;
  if (training) {
    const Buf& pb = e->bufs[e->pred_buf];
    const long long A = (long long)pb.d.h * pb.d.w;
    std::vector<ColsumDesc> cds;
    std::vector<ColsumBlock> cbs;
    for (size_t i = 0; i < e->ops.size(); ++i) {
      const cvx_op_desc& o = e->ops[i];
      if (o.type != CVX_OP_CONV || o.act != CVX_ACT_BIAS) continue;
      ColsumDesc d;
      d.off = (long long)o.out.pix_off * pb.d.c + o.out.coff;
      d.bstride = A * pb.d.c;
      d.ld = pb.d.c;
      d.hw = o.oh * o.ow;
      d.M = (long long)B * d.hw;
      d.C = o.out.c;
      d.rows_per_block = cvx_stream_rows_per_block(d.M, d.C, 32);
      d.part = e->conv[i].stat_bwd;
      d.dbias_off = o.bias_off;
      const int nb = (int)((d.M + d.rows_per_block - 1) / d.rows_per_block);
      for (int b = 0; b < nb; ++b) cbs.push_back(ColsumBlock{(int)cds.size(), b});
      if (d.C > e->colsum_max_c) e->colsum_max_c = d.C;
      cds.push_back(d);
    }
    CVX_TRY(upload(e, e->batch_allocs, e->batch_bytes, &e->d_colsum, cds));
    CVX_TRY(upload(e, e->batch_allocs, e->batch_bytes, &e->d_colsum_blocks, cbs));
    e->n_colsum = (int)cds.size();
    e->n_colsum_blocks = (int)cbs.size();
  }
  if (training) {
    CVX_TRY(dev_alloc(e, e->batch_allocs, e->batch_bytes, &p, slab_total * 4));
    e->slabs = (float*)p;
    CVX_TRY(upload(e, e->batch_allocs, e->batch_bytes, &e->d_slab, sdescs));
    CVX_TRY(upload(e, e->batch_allocs, e->batch_bytes, &e->d_slab_blocks, sblocks));
    e->n_slab_blocks = (int)sblocks.size();
  }
  e->planned_batch = B;
  e->planned_train = training;
  // the fused groups hold pointers into this plan's buffers; a training plan serves eval forwards too
  CVX_TRY(plan_fused_groups(e, B));
  CVX_TRY(plan_gemm_packs(e, B, training));
  CVX_TRY(plan_tile_packs(e, B, training));
  return 0;
}

void fill_conv_fwd(const cvx_engine* e, int i, int B, ConvParams* cp) {
  const cvx_op_desc& o = e->ops[i];
  const ConvRt& c = e->conv[i];
  ViewDesc in = make_view(e, o.in, false);
  memset(cp, 0, sizeof(*cp));
  cp->in = in.p;
  cp->in_bstride = in.bstride;
  cp->in_ld = in.ld;
  cp->IH = o.ih;
  cp->IW = o.iw;
  cp->Cin = c.cin_g;
  cp->wt = e->shadow + c.sh_fwd;
  cp->wt_ld = c.ntaps * c.cin_g;
  cp->Cout = o.out.c;
  cp->B = B;
  cp->OH2 = o.oh;
  cp->OW2 = o.ow;
  cp->IS = o.stride;
  cp->OS = 1;
  cp->OWr = o.ow;
  cp->ntaps = c.ntaps;
  cp->taps = c.taps_fwd;
  cp->zeros = e->zero_page;
  cp->halo_taps_ok = c.halo_ok ? 1 : 0;
  cp->pointwise = c.pointwise;
  cp->halo_pos = c.halo_pos;
  cp->halo_wt = c.halo_wt;
  cp->std7x7 = c.std7x7;
  cp->std3x3 = c.std3x3;
  cp->wt_packed = c.gemm_fwd;
  cp->wt_packed_bn = c.gemm_fwd_bn;
  cp->wt_packed_kc = c.gemm_fwd_kc;
  cp->tile_packed = c.tile_fwd;
  cp->tile_packed_bn = c.tile_fwd_bn;
}

void free_gemm_packs(cvx_engine* e) {
  if (e->gemm_arena) (void)hipFree(e->gemm_arena);
  if (e->d_gemm_jobs) (void)hipFree(e->d_gemm_jobs);
  e->gemm_arena = nullptr;
  e->d_gemm_jobs = nullptr;
  e->n_gemm_jobs = e->n_gemm_fwd_jobs = e->gemm_blocks = e->gemm_fwd_blocks = 0;
  for (auto& c : e->conv) {
    c.gemm_fwd = nullptr;
    c.gemm_fwd_bn = c.gemm_fwd_kc = 0;
    c.ps_on = false;
    c.ps_pk = nullptr;
    c.ps_bn = c.ps_kc = 0;
    for (auto& d : c.dg) {
      d.gemm_pk = nullptr;
      d.gemm_bn = d.gemm_kc = 0;
    }
  }
}

// the launch geometry of one phase class of a data gradient, as far as the dispatcher's choice of kernel depends on it
void fill_conv_dgrad_shape(const cvx_engine* e, int i, int q, int B, ConvParams* cp) {
  const cvx_op_desc& o = e->ops[i];
  const ConvRt& c = e->conv[i];
  const DgClass& dc = c.dg[q];
  const int C = o.out.c;
  memset(cp, 0, sizeof(*cp));
  cp->in_ld = C;
  cp->in_bstride = (long long)o.oh * o.ow * C;  // dy of a BN layer is dense; a head's dpred slice is wider (the launch re-checks its own view)
  cp->IH = o.oh;
  cp->IW = o.ow;
  cp->Cin = C;
  cp->wt = e->shadow + c.sh_dg;
  cp->wt_ld = c.ntaps * C;
  cp->Cout = o.in.c;
  cp->B = B;
  cp->OH2 = dc.OH2;
  cp->OW2 = dc.OW2;
  cp->IS = 1;
  cp->OS = o.stride;
  cp->ntaps = dc.ntaps;
  cp->taps = dc.taps;
  cp->zeros = e->zero_page;
  cp->oph = dc.oph;
  cp->opw = dc.opw;
  cp->OWr = o.iw;
  cp->halo_taps_ok = dc.halo_ok ? 1 : 0;
  cp->halo_pos = dc.halo_pos;
  cp->halo_wt = dc.halo_wt;
  cp->pointwise = dc.pointwise;
}

// The data gradient of a 3x3 / stride-2 convolution as ONE launch of the GEMM-shaped kernel.  dx[2i + p, 2j + q] only sees dy[i + a, j + b],
// a, b in {0, 1}: a stride-1 convolution over dy with the 2 x 2 window as its taps and 4 x Cin outputs -- phase (p, q) = channel block
// 2p + q -- stored with a pixel shuffle (ConvParams::ps_cin).  7 of the 16 (phase, window position) weight blocks are zero (16 / 9 of the
// multiplications), but it is one GEMM with K = 4 C and N = 4 Cin where the merged-phase launch of the ring kernel is four with K = C .. 4 C
// and N = Cin: measured in DESIGN 5c.
void fill_conv_ps_shape(const cvx_engine* e, int i, int B, ConvParams* cp) {
  const cvx_op_desc& o = e->ops[i];
  const ConvRt& c = e->conv[i];
  const int C = o.out.c;
  memset(cp, 0, sizeof(*cp));
  cp->in_ld = C;
  cp->in_bstride = (long long)o.oh * o.ow * C;
  cp->IH = o.oh;
  cp->IW = o.ow;
  cp->Cin = C;
  cp->wt = e->shadow + c.sh_ps;
  cp->wt_ld = 4 * C;
  cp->Cout = 4 * o.in.c;
  cp->B = B;
  cp->OH2 = o.oh;
  cp->OW2 = o.ow;
  cp->IS = 1;
  cp->OS = 2;
  cp->OWr = o.iw;
  cp->ntaps = 4;
  cp->taps = c.ps_taps;
  cp->zeros = e->zero_page;
  cp->epi = CVX_EPI_PLAIN;
  cp->ps_cin = o.in.c;
}

// Every conv launch the dispatcher will give to the GEMM-shaped kernel (conv_gemm.hip) gets its weights in ring image order from one
// batched launch per forward instead of a 4..8-us launch of its own in front of every convolution.
constexpr int kPsRef = 1000;  // plan_gemm_packs: the job belongs to an op's pixel-shuffle data gradient
int plan_gemm_packs(cvx_engine* e, int B, bool training) {
  free_gemm_packs(e);
  struct Ref {
    int op, q;  // q < 0: forward
  };
  std::vector<GemmPackJob> jobs;
  std::vector<Ref> refs;
  std::vector<size_t> offs;
  size_t total = 0;
  int blocks = 0;
  auto add = [&](const ConvParams& cp, int op, int q) {
    GemmPackJob j;
    size_t bytes = 0;
    if (!cvx_conv_gemm_plan(cp, &j, &bytes)) return;
    j.blk0 = blocks;
    blocks += j.nblk;
    offs.push_back(total);
    total += (bytes + 255) / 256 * 256;
    jobs.push_back(j);
    refs.push_back(Ref{op, q});
  };
  for (size_t i = 0; i < e->ops.size(); ++i) {
    if (e->ops[i].type != CVX_OP_CONV || e->conv[i].stem) continue;
    ConvParams cp;
    fill_conv_fwd(e, (int)i, B, &cp);
    add(cp, (int)i, -1);
  }
  const int n_fwd = (int)jobs.size(), fwd_blocks = blocks;
  if (training) {
    for (size_t i = 0; i < e->ops.size(); ++i) {
      const cvx_op_desc& o = e->ops[i];
      const ConvRt& c = e->conv[i];
      if (o.type != CVX_OP_CONV || c.stem || !o.needs_dgrad || c.sh_dg < 0 || c.ndg != 1) continue;  // strided data gradients go out as merged phases
      if (c.dg[0].OH2 <= 0 || c.dg[0].OW2 <= 0 || c.dg[0].ntaps <= 0) continue;
      ConvParams cp;
      fill_conv_dgrad_shape(e, (int)i, 0, B, &cp);
      add(cp, (int)i, 0);
    }
    static const bool ps_off = cvx_tune_set("CVX_NO_PS_DGRAD");
    for (size_t i = 0; i < e->ops.size() && !ps_off; ++i) {  // stride-2 data gradients as one pixel-shuffle GEMM where that kernel takes the shape
      const cvx_op_desc& o = e->ops[i];
      ConvRt& c = e->conv[i];
      if (o.type != CVX_OP_CONV || c.stem || !o.needs_dgrad || c.sh_ps < 0 || c.ndg != 4) continue;
      ConvParams cp;
      fill_conv_ps_shape(e, (int)i, B, &cp);
      if (!cvx_conv_gemm_supported(cp)) continue;
      const size_t before = jobs.size();
      add(cp, (int)i, kPsRef);
      c.ps_on = jobs.size() > before;
    }
  }
  if (jobs.empty()) return 0;
  CVX_HIP(hipMalloc((void**)&e->gemm_arena, total));
  for (size_t k = 0; k < jobs.size(); ++k) {
    jobs[k].dst = reinterpret_cast<half_t*>(reinterpret_cast<char*>(e->gemm_arena) + offs[k]);
    ConvRt& c = e->conv[refs[k].op];
    if (refs[k].q < 0) {
      c.gemm_fwd = jobs[k].dst;
      c.gemm_fwd_bn = jobs[k].BN;
      c.gemm_fwd_kc = jobs[k].kc;
    } else if (refs[k].q == kPsRef) {
      c.ps_pk = jobs[k].dst;
      c.ps_bn = jobs[k].BN;
      c.ps_kc = jobs[k].kc;
    } else {
      c.dg[refs[k].q].gemm_pk = jobs[k].dst;
      c.dg[refs[k].q].gemm_bn = jobs[k].BN;
      c.dg[refs[k].q].gemm_kc = jobs[k].kc;
    }
  }
  CVX_HIP(hipMalloc((void**)&e->d_gemm_jobs, jobs.size() * sizeof(GemmPackJob)));
  CVX_HIP(hipMemcpy(e->d_gemm_jobs, jobs.data(), jobs.size() * sizeof(GemmPackJob), hipMemcpyHostToDevice));
  e->n_gemm_jobs = (int)jobs.size();
  e->n_gemm_fwd_jobs = n_fwd;
  e->gemm_blocks = blocks;
  e->gemm_fwd_blocks = fwd_blocks;
  return 0;
}

void free_tile_packs(cvx_engine* e) {
  if (e->tile_arena) (void)hipFree(e->tile_arena);
  if (e->d_tile_jobs) (void)hipFree(e->d_tile_jobs);
  e->tile_arena = nullptr;
  e->d_tile_jobs = nullptr;
  e->n_tile_jobs = e->n_tile_fwd_jobs = e->tile_blocks = e->tile_fwd_blocks = 0;
  for (auto& c : e->conv) {
    c.tile_fwd = nullptr;
    c.tile_fwd_bn = 0;
    for (auto& d : c.dg) {
      d.tile_pk = nullptr;
      d.tile_bn = 0;
    }
  }
}

// Every conv launch the dispatcher will give to the row-band kernel (conv_tile.hip: 3x3 stride 1 on the small maps) gets its weights in that
// kernel's LDS image order from one batched launch per forward (beside cvx_pack_weights and the GEMM-shaped kernel's pack).
int plan_tile_packs(cvx_engine* e, int B, bool training) {
  free_tile_packs(e);
  struct Ref {
    int op, q;  // q < 0: forward
    TilePackPlan tp;
    ConvParams cp;
  };
  std::vector<Ref> refs;
  std::vector<size_t> offs;
  size_t total = 0;
  auto add = [&](const ConvParams& cp, int op, int q) {
    TilePackPlan tp;
    if (!cvx_conv_tile_supported(cp) || !cvx_conv_tile_plan(cp, &tp)) return;
    offs.push_back(total);
    total += (tp.bytes + 255) / 256 * 256;
    refs.push_back(Ref{op, q, tp, cp});
  };
  for (size_t i = 0; i < e->ops.size(); ++i) {
    if (e->ops[i].type != CVX_OP_CONV || e->conv[i].stem) continue;
    ConvParams cp;
    fill_conv_fwd(e, (int)i, B, &cp);
    add(cp, (int)i, -1);
  }
  const int n_fwd = (int)refs.size();
  if (training) {
    for (size_t i = 0; i < e->ops.size(); ++i) {
      const cvx_op_desc& o = e->ops[i];
      const ConvRt& c = e->conv[i];
      if (o.type != CVX_OP_CONV || c.stem || !o.needs_dgrad || c.sh_dg < 0 || c.ndg != 1) continue;
      if (c.dg[0].OH2 <= 0 || c.dg[0].OW2 <= 0 || c.dg[0].ntaps <= 0) continue;
      ConvParams cp;
      fill_conv_dgrad_shape(e, (int)i, 0, B, &cp);
      add(cp, (int)i, 0);
    }
  }
  if (refs.empty()) return 0;
  CVX_HIP(hipMalloc((void**)&e->tile_arena, total));
  const size_t jb = cvx_conv_tile_job_bytes();
  std::vector<unsigned char> jobs(refs.size() * jb);
  int blocks = 0, fwd_blocks = 0;
  for (size_t k = 0; k < refs.size(); ++k) {
    half_t* dst = reinterpret_cast<half_t*>(reinterpret_cast<char*>(e->tile_arena) + offs[k]);
    blocks += cvx_conv_tile_fill_job(refs[k].cp, refs[k].tp, dst, blocks, jobs.data() + k * jb);
    if ((int)k + 1 == n_fwd) fwd_blocks = blocks;
    ConvRt& c = e->conv[refs[k].op];
    if (refs[k].q < 0) {
      c.tile_fwd = dst;
      c.tile_fwd_bn = refs[k].tp.BN;
    } else {
      c.dg[refs[k].q].tile_pk = dst;
      c.dg[refs[k].q].tile_bn = refs[k].tp.BN;
    }
  }
  if (n_fwd == 0) fwd_blocks = 0;
  CVX_HIP(hipMalloc(&e->d_tile_jobs, jobs.size()));
  CVX_HIP(hipMemcpy(e->d_tile_jobs, jobs.data(), jobs.size(), hipMemcpyHostToDevice));
  e->n_tile_jobs = (int)refs.size();
  e->n_tile_fwd_jobs = n_fwd;
  e->tile_blocks = blocks;
  e->tile_fwd_blocks = fwd_blocks;
  return 0;
}

}  // namespace

// =============================================================================================
// The auxiliary streams (weight gradients, slab reduction, lanes) are ONE set per device for all engines of the process.  Measured on the
// ROCm 7 runtime (tools/stream_probe.py): as soon as a process has put more than four hardware queues to work -- the
// default stream plus three per engine was already the limit -- the train step of EVERY engine takes 2.2-2.5x as long (6.5 -> 14.6 ms with a
// second engine of another input size that had merely run before; 16.0 ms with one more stream at work).  Engines of one process run one
// after the other on the caller's stream anyway; sharing the side streams only orders work that was ordered already.  The streams live as
// long as the process.
namespace {
struct AuxStreams {
  hipStream_t side = nullptr, red = nullptr, lane = nullptr, xchg = nullptr;
};
std::mutex g_aux_mu;
std::map<int, AuxStreams> g_aux;
template <typename Make>
hipError_t shared_stream(int device, hipStream_t AuxStreams::*which, hipStream_t* out, Make make) {
  std::lock_guard<std::mutex> lk(g_aux_mu);
  AuxStreams& a = g_aux[device];
  if (!(a.*which)) {
    hipStream_t s = nullptr;
    const hipError_t rc = make(&s);
    if (rc != hipSuccess) return rc;
    a.*which = s;
  }
  *out = a.*which;
  return hipSuccess;
}
}  // namespace

static std::atomic<int> g_live_engines{0};  // engines of this process: the process-global stand-alone pack caches go with the last one

extern "C" int cvx_engine_create(cvx_engine** out, const cvx_buf_desc* bufs, int32_t nbufs, const cvx_op_desc* ops, int32_t nops,
                                 int32_t image_buf, int32_t pred_buf, int32_t device, void* hip_stream) {
  CVX_CHECK(out && bufs && ops && nbufs > 0 && nops > 0, "null arguments");
  CVX_HIP(hipSetDevice(device));
  cvx_engine* e = new cvx_engine();
  e->device = device;
  e->stream = (hipStream_t)hip_stream;
  e->bufs.resize(nbufs);
  for (int i = 0; i < nbufs; ++i) e->bufs[i].d = bufs[i];
  e->ops.assign(ops, ops + nops);
  e->image_buf = image_buf;
  e->pred_buf = pred_buf;
  if (!(image_buf >= 0 && image_buf < nbufs && bufs[image_buf].c == 8 && pred_buf >= 0 && pred_buf < nbufs &&
        bufs[pred_buf].kind == CVX_BUF_PRED_F32)) {
    delete e;
    CVX_FAIL("image_buf must be an 8-channel fp16 buffer and pred_buf a PRED_F32 buffer");
  }
  int rc = build_static(e);
  if (rc == 0) {
    // The weight-gradient stream is background work: lowest priority, so the dispatcher prefers the main chain, and -- the
    // reason it matters -- a priority class of its own keeps it off the main stream's hardware queue.  (With the default
    // round-robin mapping onto 4 hardware queues, a process that had created other streams first, e.g. RCCL's, got main
    // and side on ONE queue: the weight gradients ran serialised, 8.8 instead of 7.4 ms/step.)  CVX_SIDE_PRIO=0: plain stream.
    static const bool side_prio = cvx_tune_int("CVX_SIDE_PRIO", 1) != 0;
    int prio_least = 0, prio_greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    hipError_t side_rc;
    static const int side_cus = cvx_tune_int("CVX_SIDE_CUS", 0);  // tuning build: confine the weight-gradient stream to the first n CUs
    side_rc = shared_stream(e->device, &AuxStreams::side, &e->side, [&](hipStream_t* s) {
      if (side_cus > 0) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int c = 0; c < side_cus && c < 256; ++c) mask[c >> 5] |= 1u << (c & 31);
        return hipExtStreamCreateWithCUMask(s, 8, mask);
      }
      return side_prio ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio_least) : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    });
    // The slab reduction shares the weight-gradient stream: a stream of its own measured no faster (6.50-6.52 vs 6.56-6.57 ms) and the engine
    // then works three hardware queues instead of four -- the fourth is left to the caller (a launch stream of its own: 6.5 ms instead of
    // 16.0) or to RCCL's internal stream (the exchange itself rides this stream too).  CVX_RED_OWN_STREAM=1 (tuning build): the round-2 arrangement.
    static const bool red_own = cvx_tune_int("CVX_RED_OWN_STREAM", 0) != 0;
    if (side_rc == hipSuccess && !red_own) e->red = e->side;
    else if (side_rc == hipSuccess)
      side_rc = shared_stream(e->device, &AuxStreams::red, &e->red, [&](hipStream_t* s) {
        return side_prio ? hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio_least) : hipStreamCreateWithFlags(s, hipStreamNonBlocking);
      });
    if (side_rc != hipSuccess || hipEventCreateWithFlags(&e->ev_red, cvx_event_flags()) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_fork, cvx_event_flags()) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_join, cvx_event_flags()) != hipSuccess ||
        hipEventCreateWithFlags(&e->ev_mid, cvx_event_flags()) != hipSuccess) {
      cvx_set_error("cvx_engine_create: could not create the side stream / events");
      rc = -1;
    }
    for (size_t i = 0; rc == 0 && i < e->ops.size(); ++i)
      if (e->ops[i].type == CVX_OP_CONV && hipEventCreateWithFlags(&e->conv[i].ev_dy, cvx_event_flags()) != hipSuccess) {
        cvx_set_error("cvx_engine_create: could not create events");
        rc = -1;
      }
    // lanes: a HIGH-priority stream, i.e. a priority class (and hardware queue) of its own -- plain extra streams can land on
    // the main or the side stream's hardware queue and serialise with it (that is what made per-level streams slower before)
    static const bool lanes_on = cvx_tune_int("CVX_LANES", 1) != 0;
    bool any_lane = false;
    for (cvx_op_desc& o : e->ops) {
      if (!lanes_on || e->inference_only) o.lane = 0;
      any_lane = any_lane || o.lane >= 2;
    }
    const bool stem_first = !e->ops.empty() && e->ops[0].type == CVX_OP_CONV && e->conv[0].stem;  // the weight packing can run beside it
    if (rc == 0 && (any_lane || (lanes_on && stem_first))) {
      if (shared_stream(e->device, &AuxStreams::lane, &e->lane,
                        [&](hipStream_t* s) { return hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio_greatest); }) == hipSuccess &&
          hipEventCreateWithFlags(&e->ev_lane_fork, cvx_event_flags()) == hipSuccess &&
          hipEventCreateWithFlags(&e->ev_lane_join, cvx_event_flags()) == hipSuccess &&
          hipEventCreateWithFlags(&e->ev_pack, cvx_event_flags()) == hipSuccess) {
        e->use_lanes = any_lane;
        static const int head_overlap = cvx_tune_int("CVX_HEAD_OVERLAP", 1);
        if (any_lane && head_overlap) {
          bool ok = true;
          for (int l = 0; l < 3 && ok; ++l) {
            int first = -1;
            for (size_t i = 0; i < e->ops.size(); ++i)
              if (e->ops[i].lane == l + 1) {
                first = (int)i;
                break;
              }
            if (first < 0) {
              ok = false;
              break;
            }
            for (int j = first - 1; j >= 0; --j)
              if (e->ops[j].out.buf == e->ops[first].in.buf && e->ops[j].lane == 0) {
                e->head_producer[l] = j;
                break;
              }
            for (int j = first - 1; j >= 0; --j) {
              const cvx_op_desc& q = e->ops[j];
              if (q.lane == 0 && (q.out.buf == e->ops[first].in.buf || q.in.buf == e->ops[first].in.buf || q.res.buf == e->ops[first].in.buf)) {
                e->head_join_op[l] = j;
                break;
              }
            }
            ok = e->head_producer[l] >= 0 && e->head_join_op[l] >= 0 && hipEventCreateWithFlags(&e->ev_head[l], cvx_event_flags()) == hipSuccess;
          }
          e->head_overlap = ok;
        }
      } else {
        (void)hipGetLastError();
        e->lane = nullptr;  // (the shared stream stays: other engines may be using it)
        for (cvx_op_desc& o : e->ops) o.lane = 0;
      }
    }
  }
  if (rc != 0) {
    free_pool(e->static_allocs);
    delete e;
    return rc;
  }
  // SPPF (core/models/yolov8/modules.py:304-318): three chained 5x5 max pools whose maps fit in LDS go out as one launch each way
  e->sppf3.assign(e->ops.size(), 0);
  {
    static const bool off = cvx_tune_set("CVX_NO_SPPF3");
    for (size_t k = 0; !off && k + 2 < e->ops.size(); ++k) {
      const cvx_op_desc &a = e->ops[k], &b = e->ops[k + 1], &c = e->ops[k + 2];
      if (a.type != CVX_OP_MAXPOOL5 || b.type != CVX_OP_MAXPOOL5 || c.type != CVX_OP_MAXPOOL5) continue;
      if (!same_view(b.in, a.out) || !same_view(c.in, b.out) || a.lane != b.lane || b.lane != c.lane) continue;
      if (a.ih != b.ih || b.ih != c.ih || a.iw != b.iw || b.iw != c.iw || a.in.c != b.in.c || b.in.c != c.in.c) continue;
      if (!cvx_sppf_pool3_fits(a.ih, a.iw)) continue;
      e->sppf3[k] = 1;
      e->sppf3[k + 1] = e->sppf3[k + 2] = 2;
      k += 2;
    }
  }
  g_live_engines.fetch_add(1);
  *out = e;
  return 0;
}

extern "C" int cvx_engine_destroy(cvx_engine* e) {
  if (!e) return 0;
  (void)hipStreamSynchronize(e->stream);
  if (e->side) (void)hipStreamSynchronize(e->side);  // shared per device (AuxStreams): never destroyed
  for (auto& c : e->conv)
    if (c.ev_dy) (void)hipEventDestroy(c.ev_dy);
  if (e->lane) (void)hipStreamSynchronize(e->lane);
  if (e->ev_lane_fork) (void)hipEventDestroy(e->ev_lane_fork);
  for (int l = 0; l < 3; ++l)
    if (e->ev_head[l]) (void)hipEventDestroy(e->ev_head[l]);
  if (e->ev_lane_join) (void)hipEventDestroy(e->ev_lane_join);
  if (e->ev_pack) (void)hipEventDestroy(e->ev_pack);
  if (e->red) (void)hipStreamSynchronize(e->red);
  if (e->ev_red) (void)hipEventDestroy(e->ev_red);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->ev_mid) (void)hipEventDestroy(e->ev_mid);
  for (hipEvent_t ev : e->ev_seg)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : e->ev_pool) (void)hipEventDestroy(ev);
  free_fused(e);
  {
    if (e->gemm_arena) (void)hipFree(e->gemm_arena);
    if (e->d_gemm_jobs) (void)hipFree(e->d_gemm_jobs);
    if (e->tile_arena) (void)hipFree(e->tile_arena);
    if (e->d_tile_jobs) (void)hipFree(e->d_tile_jobs);
  }
  free_pool(e->batch_allocs);
  free_pool(e->static_allocs);
  // The stand-alone pack caches (unit-op entry points) are process-global and keyed by weight pointers; other live engines and concurrent
  // unit-op callers may be using them, so they go only with the LAST engine of the process (a stale entry is harmless: a stand-alone
  // launch re-packs its weights into the cached buffer every time, the cache only saves the allocation)
  if (g_live_engines.fetch_sub(1) == 1) {
    cvx_conv_gemm_release();
    cvx_conv_tile_release();
  }
  cvx_engine_free_bw(e);
  delete e;
  return 0;
}

extern "C" int cvx_engine_bind(cvx_engine* e, float* params, float* grads, int64_t n_params, float* stats, int64_t n_stats) {
  CVX_CHECK(e && params && stats, "null arguments");
  CVX_CHECK(((uintptr_t)params % 16) == 0 && ((uintptr_t)grads % 16) == 0, "arenas must be 16-byte aligned");
  for (const cvx_op_desc& o : e->ops) {
    if (o.type == CVX_OP_DWCONVT) CVX_CHECK(o.w_off >= 0 && o.w_off + (int64_t)o.out.c * 4 * o.stride * o.stride <= n_params, "dwconvt weight offset");
    if (o.type != CVX_OP_CONV) continue;
    CVX_CHECK(o.w_off >= 0 && o.w_off + (int64_t)o.out.c * o.k * o.k * o.w_cin <= n_params, "weight offset out of range");
    if (o.act == CVX_ACT_BN_SILU || o.act == CVX_ACT_BN_RELU || o.act == CVX_ACT_BN_LINEAR) {
      CVX_CHECK(o.gamma_off >= 0 && o.gamma_off + o.out.c <= n_params && o.beta_off >= 0 && o.beta_off + o.out.c <= n_params, "bn offsets");
      CVX_CHECK(o.rmean_off >= 0 && o.rmean_off + o.out.c <= n_stats && o.rvar_off >= 0 && o.rvar_off + o.out.c <= n_stats, "stat offsets");
      CVX_CHECK(o.gamma_off % 4 == 0 && o.beta_off % 4 == 0 && o.rmean_off % 4 == 0 && o.rvar_off % 4 == 0, "bn offsets must be 4-aligned");
    } else {
      CVX_CHECK(o.bias_off >= 0 && o.bias_off + o.out.c <= n_params && o.bias_off % 4 == 0, "bias offset");
    }
  }
  e->params = params;
  e->grads = grads;
  e->stats = stats;
  e->n_params = n_params;
  e->n_stats = n_stats;
  e->shadows_gen = -1;  // the fp16 images were made from the arena bound before: cvx_engine_keep_shadows must not vouch for them
  return 0;
}

extern "C" int cvx_engine_set_bn(cvx_engine* e, float eps, float momentum) {
  CVX_CHECK(e, "null engine");
  e->bn_eps = eps;
  e->bn_momentum = momentum;
  return 0;
}

extern "C" int cvx_engine_set_fusion(cvx_engine* e, int32_t enable) {
  CVX_CHECK(e, "null engine");
#ifndef CVX_WITH_CHAIN
  CVX_CHECK(!enable, "fusion groups: the tile-resident chain kernel (conv_chain.hip) is not part of the release library -- it measured slower "
                     "than the per-layer launches (DESIGN.md 5b); tools/build_tuning.sh builds a library that carries it");
#else
  if (e->chain_fusion != (enable != 0)) {
    e->chain_fusion = enable != 0;
    if (e->planned_batch > 0) {  // re-derive the groups of the current plan (the per-batch buffers stay)
      CVX_HIP(hipStreamSynchronize(e->stream));
      if (e->lane) CVX_HIP(hipStreamSynchronize(e->lane));
      CVX_TRY(plan_fused_groups(e, e->planned_batch));
      e->plan_generation++;  // captured hipGraphs hold the old launch sequence
    }
  }
#endif
  return 0;
}

extern "C" int32_t cvx_engine_fused_groups(const cvx_engine* e) {
#ifdef CVX_WITH_CHAIN
  return e ? (int32_t)e->fused.size() : -1;
#else
  return e ? 0 : -1;
#endif
}

extern "C" int cvx_engine_keep_shadows(cvx_engine* e) {
  CVX_CHECK(e, "null engine");
  e->keep_shadows_once = true;
  return 0;
}

extern "C" int cvx_engine_set_seed(cvx_engine* e, uint64_t seed) {
  CVX_CHECK(e, "null engine");
  e->seed = seed;
  return 0;
}

extern "C" int cvx_engine_debug_copy(cvx_engine* e, int32_t buf, int32_t which, void* dst, int64_t bytes) {
  CVX_CHECK(e && dst && buf >= 0, "bad arguments");
  if (which == 4) {  // the normalised output as the backward passes use it, fp32: `buf` is an op index
    CVX_CHECK(buf < (int)e->ops.size() && e->ops[buf].type == CVX_OP_CONV, "which 4: `buf` must be the index of a conv op");
    const ConvRt& c = e->conv[buf];
    const cvx_op_desc& o = e->ops[buf];
    CVX_CHECK(c.ybuf && e->planned_batch > 0 && e->planned_train, "no training plan, or the op keeps no such tensor");
    const long long M = (long long)e->planned_batch * o.oh * o.ow;
    const int64_t want = (int64_t)M * o.out.c * 4;
    CVX_CHECK(bytes == want, "size mismatch: the tensor holds " + std::to_string(want) + " bytes");
    float* tmp = nullptr;
    CVX_HIP(hipMalloc((void**)&tmp, (size_t)bytes));
    int rc = cvx_kept_to_xhat_f32(c.ybuf, M, o.out.c, c.raw16 ? c.mean : nullptr, c.invstd, tmp, e->stream);
    if (rc == 0 && hipMemcpyAsync(dst, tmp, (size_t)bytes, hipMemcpyDefault, e->stream) != hipSuccess) rc = -1;
    (void)hipStreamSynchronize(e->stream);
    (void)hipFree(tmp);
    return rc;
  }
  if (which == 2 || which == 3) {  // per-layer operands of the backward pass: `buf` is an op index
    CVX_CHECK(buf < (int)e->ops.size() && e->ops[buf].type == CVX_OP_CONV, "which 2/3: `buf` must be the index of a conv op");
    const ConvRt& c = e->conv[buf];
    const cvx_op_desc& o = e->ops[buf];
    const half_t* src = which == 2 ? c.ybuf : c.dybuf;
    CVX_CHECK(src && e->planned_batch > 0 && e->planned_train, "no training plan, or the op keeps no such tensor");
    const int64_t want = (int64_t)e->planned_batch * o.oh * o.ow * o.out.c * 2;
    CVX_CHECK(bytes == want, "size mismatch: the tensor holds " + std::to_string(want) + " bytes");
    if (which == 2 && c.raw16) {  // the layer keeps its RAW output (CVX_OPF_RAW_F16): hand out what the backward passes make of it
      half_t* tmp = nullptr;
      CVX_HIP(hipMalloc((void**)&tmp, (size_t)bytes));
      int rc = cvx_raw16_to_xhat(c.ybuf, (long long)e->planned_batch * o.oh * o.ow, o.out.c, c.mean, c.invstd, tmp, e->stream);
      if (rc == 0 && hipMemcpyAsync(dst, tmp, (size_t)bytes, hipMemcpyDefault, e->stream) != hipSuccess) rc = -1;
      (void)hipStreamSynchronize(e->stream);
      (void)hipFree(tmp);
      return rc;
    }
    CVX_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, e->stream));
    return 0;
  }
  CVX_CHECK(buf < (int)e->bufs.size(), "bad arguments");
  const Buf& b = e->bufs[buf];
  const half_t* src = which ? b.grad : b.act;
  CVX_CHECK(src && e->planned_batch > 0, "buffer not allocated (no forward yet, or eval-only plan)");
  const int64_t want = (int64_t)e->planned_batch * b.d.h * b.d.w * b.d.c * 2;
  CVX_CHECK(bytes == want, "size mismatch: buffer holds " + std::to_string(want) + " bytes");
  CVX_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDefault, e->stream));
  return 0;
}

extern "C" int64_t cvx_engine_workspace_bytes(const cvx_engine* e) { return e ? e->batch_bytes + e->static_bytes : 0; }
extern "C" int64_t cvx_engine_plan_generation(const cvx_engine* e) { return e ? e->plan_generation : -1; }

extern "C" int cvx_engine_forward(cvx_engine* e, const float* images, int32_t batch, int32_t training, float* pred) {
  CVX_CHECK(e && images && pred && batch > 0, "bad arguments");
  CVX_CHECK(e->params, "cvx_engine_bind was not called");
  CVX_CHECK(!(training && e->inference_only), "this graph holds inference-only ops: training-mode forward is not available");
  CVX_HIP(hipSetDevice(e->device));
  CVX_TRY(plan_batch(e, batch, training != 0));
  hipStream_t st = e->stream;
  const int B = batch;
  // fp32 master -> fp16 shadows (forward layout + transposed layout for the data gradient)
  const Buf& ib = e->bufs[e->image_buf];
  CVX_CHECK(((uintptr_t)images % 8) == 0, "images must be 8-byte aligned");
  if (training) {  // forward AND backward statistic slabs in one fill (the backward's own fill sat at the head of its critical path)
    CVX_HIP(hipMemsetAsync(e->stat_region, 0, (size_t)e->stat_half * 16, st));
    e->bwd_slabs_clean = true;
  }
  // fp16 weight shadows: the fp32 stem does not need them, so they are prepared on the lane stream BESIDE it (33 us off the main chain); the first op after the stem waits for them
  static const int pack_lane = cvx_tune_int("CVX_PACK_LANE", 1);  // bit 0: training forward, bit 1: eval forward (measured: +0.02 ms there)
  bool keep = e->keep_shadows_once && e->shadows_gen == e->plan_generation && (e->shadows_train || !training);
#ifdef CVX_WITH_CHAIN
  if (!training && e->shadows_train && e->n_chain_jobs > 0) keep = false;  // the chain kernel's images are packed by eval forwards only
#endif
  e->keep_shadows_once = false;
  const bool prep_beside_stem = !keep && e->lane && !e->ops.empty() && e->ops[0].type == CVX_OP_CONV && e->conv[0].stem && !e->profile &&
                                (pack_lane & (training ? 1 : 2)) != 0;
  hipStream_t prep = st;
  if (prep_beside_stem) {
    CVX_HIP(hipEventRecord(e->ev_pack, st));  // the caller's parameter update precedes on this stream
    CVX_HIP(hipStreamWaitEvent(e->lane, e->ev_pack, 0));
    prep = e->lane;
  }
  if (!keep) {
    e->shadows_gen = -1;  // (recorded as packed only once every pack launch below has been queued: a failed launch leaves no stale claim)
    ProfScope ps(e, PROF_MISC, 0, 6.0 * e->n_params, prep);
    CVX_TRY(cvx_pack_weights(e->params, e->shadow, e->d_pack, e->d_pack_blocks, e->n_pack_blocks, prep));
#ifdef CVX_WITH_CHAIN
    if (!training && e->n_chain_jobs > 0) CVX_TRY(cvx_chain_pack_jobs(e->d_chain_jobs, e->n_chain_jobs, e->chain_max_units, prep));
#endif
    if (training) {  // pixel-shuffle data-gradient weights: from the transposed shadows just written, before the GEMM kernel's pack reads them
      for (const ConvRt& c : e->conv)
        if (c.ps_on) CVX_TRY(cvx_pack_ps_weights(e->shadow, e->shadow, c.ps_desc, prep));
    }
    if (training)
      CVX_TRY(cvx_conv_gemm_pack_jobs(e->d_gemm_jobs, e->n_gemm_jobs, e->gemm_blocks, prep));
    else
      CVX_TRY(cvx_conv_gemm_pack_jobs(e->d_gemm_jobs, e->n_gemm_fwd_jobs, e->gemm_fwd_blocks, prep));
    CVX_TRY(cvx_conv_tile_pack_jobs(e->d_tile_jobs, training ? e->n_tile_jobs : e->n_tile_fwd_jobs, training ? e->tile_blocks : e->tile_fwd_blocks, prep));
    e->shadows_gen = e->plan_generation;  // every image of this plan is queued
    e->shadows_train = training;
  }
  e->last_images = training ? images : nullptr;
  if (training) e->train_pass++;  // dropout masks: one per (seed, training forward, op)
  if (e->image_nhwc) CVX_TRY(cvx_image_to_nhwc8(images, B, ib.d.h, ib.d.w, ib.act, st));
  const Buf& pb = e->bufs[e->pred_buf];
  const long long A = (long long)pb.d.h * pb.d.w;
  // eval: running stats -> scale/shift (the stem's too); training graphs with bias-only epilogues (conv + bias + ReLU) need their table as well
  if (!training || e->has_bias_act) CVX_TRY(cvx_bn_fold_all(e->d_fold, e->n_fold, e->params, e->stats, e->bn_eps, st));
  bool prep_pending = prep_beside_stem;
  if (prep_beside_stem) CVX_HIP(hipEventRecord(e->ev_pack, e->lane));

  const hipStream_t main_st = st;
  bool lane_forked = false, lane_used = false;
  bool head_rec[3] = {false, false, false}, head_waited[3] = {false, false, false};
  static const int head_lane_mask = cvx_tune_int("CVX_HEAD_LANE_MASK", 6);  // bit l: Detect level l runs on the lane stream (levels 1 and 2: measured best of the eight)
  for (size_t i = 0; i < e->ops.size(); ++i) {
    const cvx_op_desc& o = e->ops[i];
    e->cur_op = (int)i;
    if (prep_pending && i > 0) {
      CVX_HIP(hipStreamWaitEvent(main_st, e->ev_pack, 0));
      prep_pending = false;
    }
    // lanes (Detect levels 1, 2 beside level 0): fork where the first op of ANY lane comes up -- everything the lanes read
    // exists by then --, the lane stream joins back after the last op
    st = main_st;
    if (e->use_lanes && o.lane >= 1 && !lane_forked) {
      CVX_HIP(hipEventRecord(e->ev_lane_fork, main_st));
      lane_forked = true;
    }
    if (e->head_overlap && !e->profile) {
      for (int l = 0; l < 3; ++l)
        if (!head_rec[l] && (int)i > e->head_producer[l]) {  // the level's input is complete on the main stream from here on
          CVX_HIP(hipEventRecord(e->ev_head[l], main_st));
          head_rec[l] = true;
        }
      const int lvl = o.lane - 1;
      if (lvl >= 0 && lvl < 3 && ((head_lane_mask >> lvl) & 1)) {
        if (!head_waited[lvl]) CVX_HIP(hipStreamWaitEvent(e->lane, e->ev_head[lvl], 0));
        head_waited[lvl] = true;
        lane_used = true;
        st = e->lane;
      }
    } else if (e->use_lanes && o.lane >= 2) {
      if (!lane_used) CVX_HIP(hipStreamWaitEvent(e->lane, e->ev_lane_fork, 0));
      lane_used = true;
      st = e->lane;
    }
    float* const ytmp = st == main_st ? e->ytmp : e->ytmp_lane;
#ifdef CVX_WITH_CHAIN
    if (!training && !e->fused_at.empty() && e->fused_at[i] >= 0) {  // a fused group starts here: one tile-resident launch for ops i .. last
      const cvx_engine::FusedGroup& g = e->fused[e->fused_at[i]];
      ProfScope ps(e, PROF_CONV_FWD, g.plan.flops, g.plan.bytes, st);
      CVX_TRY(cvx_chain_launch(g.plan, st, pred));
      i = (size_t)g.last;
      continue;
    }
#endif
    if (o.type == CVX_OP_MAXPOOL5) {
      if (!e->sppf3.empty() && e->sppf3[i] == 1) {  // SPPF: y1 = m(x), y2 = m(y1), y3 = m(y2) as one launch
        ProfScope ps(e, PROF_MISC, 0, 8.0 * B * o.ih * o.iw * o.in.c, st);
        CVX_TRY(cvx_sppf_pool3_fwd(make_view(e, o.in, false), make_view(e, o.out, false), make_view(e, e->ops[i + 1].out, false),
                                   make_view(e, e->ops[i + 2].out, false), B, o.ih, o.iw, o.in.c, training ? e->pool[i].idx : nullptr,
                                   training ? e->pool[i + 1].idx : nullptr, training ? e->pool[i + 2].idx : nullptr, st));
        i += 2;
        continue;
      }
      ProfScope ps(e, PROF_MISC, 0, 4.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_maxpool5_fwd(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih, o.iw, o.in.c,
                               training ? e->pool[i].idx : nullptr, st));
      continue;
    }
    if (o.type == CVX_OP_UPSAMPLE2) {
      ProfScope ps(e, PROF_MISC, 0, 10.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_upsample2_fwd(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih, o.iw, o.in.c, st));
      continue;
    }
    if (o.type == CVX_OP_MAXPOOL2) {
      ProfScope ps(e, PROF_MISC, 0, 2.5 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_maxpool2(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih, o.iw, o.oh, o.ow, o.in.c, st));
      continue;
    }
    if (o.type == CVX_OP_DWCONVT) {
      ProfScope ps(e, PROF_MISC, 0, 2.0 * B * (o.ih * o.iw + o.oh * o.ow) * o.in.c, st);
      CVX_TRY(cvx_dwconvt(make_view(e, o.in, false), make_view(e, o.out, false), e->params + o.w_off, B, o.ih, o.iw, o.in.c, o.stride, st));
      continue;
    }
    if (o.type == CVX_OP_COPY) {
      ProfScope ps(e, PROF_MISC, 0, 4.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_copy_slice(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih * o.iw, o.in.c, st));
      continue;
    }
    if (o.type == CVX_OP_MAXPOOL3S2) {
      ProfScope ps(e, PROF_MISC, 0, 2.0 * B * (o.ih * o.iw + o.oh * o.ow) * o.in.c, st);
      CVX_CHECK(o.oh == (o.ih - 1) / 2 + 1 && o.ow == (o.iw - 1) / 2 + 1, "maxpool3s2: output size must be floor((i - 1) / 2) + 1");
      CVX_TRY(cvx_maxpool3(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih, o.iw, o.in.c, 2, training ? e->pool[i].idx : nullptr, st));
      continue;
    }
    if (o.type == CVX_OP_MAXPOOL3S1) {
      ProfScope ps(e, PROF_MISC, 0, 4.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_CHECK(o.oh == o.ih && o.ow == o.iw, "maxpool3s1: same-size output");
      CVX_TRY(cvx_maxpool3(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih, o.iw, o.in.c, 1, training ? e->pool[i].idx : nullptr, st));
      continue;
    }
    if (o.type == CVX_OP_L2NORM) {
      ProfScope ps(e, PROF_MISC, 0, 4.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_l2norm(make_view(e, o.in, false), make_view(e, o.out, false), e->params + o.gamma_off, B, o.ih * o.iw, o.in.c, st));
      continue;
    }
    if (o.type == CVX_OP_AVGPOOL) {
      ProfScope ps(e, PROF_MISC, 0, 2.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_CHECK(o.oh == 1 && o.ow == 1, "avgpool: global pooling only (1x1 output)");
      CVX_TRY(cvx_avgpool_global(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih * o.iw, o.in.c, st));
      continue;
    }
    if (o.type == CVX_OP_RESIZE) {
      ProfScope ps(e, PROF_MISC, 0, 2.0 * B * (o.ih * o.iw + o.oh * o.ow) * o.in.c, st);
      CVX_TRY(cvx_resize_bilinear(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih, o.iw, o.oh, o.ow, o.in.c, st));
      continue;
    }
    if (o.type == CVX_OP_DROPOUT) {  // eval: identity (p = 0 keeps every element at scale 1)
      ProfScope ps(e, PROF_MISC, 0, 4.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_dropout(make_view(e, o.in, false), make_view(e, o.out, false), B, o.ih * o.iw, o.in.c, training ? dropout_p(o) : 0.f,
                          dropout_seed(e, (int)i), 0, st));
      continue;
    }
    ConvRt& c = e->conv[i];
    const long long M = (long long)B * o.oh * o.ow;
    const int C = o.out.c;
    if (c.stem) {  // fp32, straight from the caller's NCHW images: statistics pass + recompute/normalise pass (stem.hip)
      const StemParams sp{images, B, ib.d.h, ib.d.w, o.oh, o.ow, e->params + o.w_off, C};
      const double img_bytes = 12.0 * B * ib.d.h * ib.d.w;
      ViewDesc outv = make_view(e, o.out, false);
      if (training) {
        {
          ProfScope ps(e, PROF_CONV_FWD, conv_flops(o, B), img_bytes, st);
          CVX_TRY(cvx_stem_stats(sp, c.stat_fwd, st));
        }
        ProfScope ps(e, PROF_CONV_FWD, conv_flops(o, B), img_bytes + 4.0 * M * C, st);
        BnTrainArgs ta{c.stat_fwd,           e->params + o.gamma_off, e->params + o.beta_off, c.mean, c.invstd, e->stats + o.rmean_off,
                       e->stats + o.rvar_off, e->bn_eps,       e->bn_momentum};
        // (xhat is not stored where the backward pass recomputes it from the images: cvx_stem_keeps_xhat, the same test on both sides)
        const bool keep_xhat = cvx_stem_keeps_xhat(sp, make_view(e, o.out, true), c.nsplit);
        CVX_TRY(cvx_stem_apply_train(sp, ta, outv, keep_xhat ? c.ybuf : nullptr, st));
      } else {
        ProfScope ps(e, PROF_CONV_FWD, conv_flops(o, B), img_bytes + 2.0 * M * C, st);
        CVX_TRY(cvx_stem_apply_eval(sp, c.scale, c.shift, outv, st));
      }
      continue;
    }
    ConvParams cp;
    fill_conv_fwd(e, (int)i, B, &cp);
    if (o.act == CVX_ACT_BIAS) {
      cp.epi = CVX_EPI_BIAS_F32;
      cp.bias = e->params + o.bias_off;
      cp.out32 = pred + (long long)o.out.pix_off * pb.d.c + o.out.coff;
      cp.out_ld = pb.d.c;
      cp.out_bstride = A * pb.d.c;
      ProfScope ps(e, PROF_CONV_FWD, conv_flops(o, B), conv_bytes(o, B) + 2.0 * M * C, st);
      CVX_TRY(cvx_conv_igemm_launch(cp, st, nullptr));
      continue;
    }
    ViewDesc outv = make_view(e, o.out, false);
    ViewDesc resv = make_view(e, o.res, false);
    if (training && o.act != CVX_ACT_BIAS_RELU && o.act != CVX_ACT_BIAS_LINEAR) {
      cp.epi = CVX_EPI_RAW_STATS;
      cp.out32 = ytmp;  // raw fp32 output: lives until the normalisation pass right below, then the next layer (of this stream) reuses it
      cp.out_ld = C;
      cp.out_bstride = (long long)o.oh * o.ow * C;
      if (c.raw16) {    // CVX_OPF_RAW_F16: rounded to fp16 straight into the tensor the backward pass keeps
        cp.raw16 = 1;
        cp.out32 = nullptr;
        cp.out16 = c.ybuf;
      }
      cp.stats = c.stat_fwd;
      cp.stats_replicas = cvx_stat_replicas(C);
      int P = 0;
      {
        ProfScope ps(e, PROF_CONV_FWD, conv_flops(o, B), conv_bytes(o, B) + (c.raw16 ? 0.0 : 2.0 * M * C), st);
        CVX_TRY(cvx_conv_igemm_launch(cp, st, &P));
      }
      ProfScope ps(e, PROF_BN_FWD, 0, (c.raw16 ? 4.0 : resv.p ? 10.0 : 8.0) * M * C, st);
      BnTrainArgs ta{c.stat_fwd,           e->params + o.gamma_off, e->params + o.beta_off, c.mean, c.invstd, e->stats + o.rmean_off,
                     e->stats + o.rvar_off, e->bn_eps,       e->bn_momentum};
      if (o.flags & CVX_OPF_CONV_BIAS) ta.cbias = e->params + o.bias_off;  // (its gradient is exactly zero: BatchNorm removes the mean)
      if (c.raw16) CVX_TRY(cvx_bn_silu_apply_raw16(c.ybuf, M, C, o.oh * o.ow, ta, outv, st));
      else CVX_TRY(cvx_bn_act_apply(ytmp, M, C, o.oh * o.ow, ta, outv, resv, act_kind(o), (o.flags & CVX_OPF_RES_PRE_ACT) ? 1 : 0, c.ybuf, st));
    } else {
      // scale / shift were folded for every layer at once before the op loop (cvx_bn_fold_all)
      cp.epi = CVX_EPI_AFFINE_SILU;
      cp.act_kind = act_kind(o);
      cp.res_pre = (o.flags & CVX_OPF_RES_PRE_ACT) ? 1 : 0;
      cp.scale = c.scale;
      cp.shift = c.shift;
      cp.out16 = outv.p;
      cp.out_ld = outv.ld;
      cp.out_bstride = outv.bstride;
      cp.res = resv.p;
      cp.res_ld = resv.ld;
      cp.res_bstride = resv.bstride;
      ProfScope ps(e, PROF_CONV_FWD, conv_flops(o, B), conv_bytes(o, B), st);
      CVX_TRY(cvx_conv_igemm_launch(cp, st, nullptr));
    }
  }
  if (lane_used) {
    CVX_HIP(hipEventRecord(e->ev_lane_join, e->lane));
    CVX_HIP(hipStreamWaitEvent(main_st, e->ev_lane_join, 0));
  }
  e->cur_op = -1;
  e->fwd_train_done = training != 0;
  e->last_batch = B;
  return 0;
}

// ---- backward pass state shared by the whole-pass and the segmented entry points ----
struct PendingWgrad {
  WgradParams wp;
  bool stem;
  StemParams sp;
  ViewDesc gout;             // stem: fused BN-apply + weight gradient (cvx_stem_backward) reads gout / xhat instead of dy
  BnCoef coef;
  const long long* part;
  float inv_scale;
  float *dgamma, *dbeta;
  double flops, bytes;
  int op;
  float* dw = nullptr;       // stem: its slice of the gradient arena (cvx_stem_backward_fold adds the folded weight gradient itself)
};
struct cvx_bw_state {
  half_t* dpred = nullptr;
  float inv_scale = 1.f;
  int B = 0;
  bool active = false;
  int wg_batch = 1;
  int next_op = -1;  // next op (descending) the segmented interface expects
  bool lane_pending = false;       // kernels queued on the lane stream since the last join
  int lane_join_op = -1;           // head overlap: the main chain joins the lane stream at the first op i <= lane_join_op
  hipStream_t pending_stream = nullptr;  // stream that produced the dy tensors of `pending`
  std::vector<PendingWgrad> pending;
};
static cvx_bw_state& bw_of(cvx_engine* e) {
  if (!e->bw) e->bw = new cvx_bw_state();
  return *e->bw;
}

namespace {

// weight gradients go to the side stream in batches: one event record on the producing stream per batch
int flush_wgrads(cvx_engine* e, hipEvent_t ev, hipStream_t producer) {
  cvx_bw_state& w = bw_of(e);
  if (w.pending.empty()) return 0;
  CVX_HIP(hipEventRecord(ev, producer));
  CVX_HIP(hipStreamWaitEvent(e->side, ev, 0));
  for (const PendingWgrad& g : w.pending) {
    e->cur_op = g.op;
    if (g.stem) {
      // the last op of the pass: the main stream has nothing else left, while the side stream still holds the previous
      // layers' weight gradients -- the fused stem backward runs on the producer (main) stream, beside them
      ProfScope ps(e, PROF_CONV_WGRAD, g.flops, g.bytes, producer);
      CVX_TRY(cvx_stem_backward_fold(g.sp, g.wp.dy /* = xhat */, g.gout, g.coef, g.part, g.inv_scale, g.dgamma, g.dbeta, g.dw, g.wp.slabs, g.wp.nsplit, producer));
      continue;
    }
    // tuning build only (cvx_tune_int is a constant in the release library): time the main stream without its competitor
    static const bool skip = cvx_tune_int("CVX_TUNE_SKIP_WGRAD", 0) != 0;
    if (skip) continue;
    ProfScope ps(e, PROF_CONV_WGRAD, g.flops, g.bytes, e->side);
    CVX_TRY(cvx_conv_wgrad_launch(g.wp, e->side));
  }
  w.pending.clear();
  return 0;
}

int join_lane(cvx_engine* e) {
  cvx_bw_state& w = bw_of(e);
  if (!w.lane_pending) return 0;
  if (!w.pending.empty() && w.pending_stream == e->lane) CVX_TRY(flush_wgrads(e, e->ev_lane_fork, e->lane));
  CVX_HIP(hipEventRecord(e->ev_lane_join, e->lane));
  CVX_HIP(hipStreamWaitEvent(e->stream, e->ev_lane_join, 0));
  w.lane_pending = false;
  w.lane_join_op = -1;
  return 0;
}

int backward_begin(cvx_engine* e, const void* dpred_f16, float loss_scale) {
  CVX_CHECK(e && dpred_f16, "bad arguments");
  CVX_CHECK(e->fwd_train_done, "cvx_engine_backward needs a preceding training-mode forward");
  CVX_CHECK(e->grads, "no gradient arena bound");
  CVX_CHECK(loss_scale > 0.f, "loss_scale must be positive");
  CVX_HIP(hipSetDevice(e->device));
  hipStream_t st = e->stream;
  cvx_bw_state& w = bw_of(e);
  w.dpred = (half_t*)dpred_f16;
  w.inv_scale = 1.0f / loss_scale;
  w.B = e->last_batch;
  w.pending.clear();
  w.active = true;
  w.next_op = (int)e->ops.size() - 1;
  static const int wg_batch_env = cvx_tune_int("CVX_WGRAD_BATCH", 2);
  w.wg_batch = wg_batch_env < 1 ? 1 : wg_batch_env;
  if (!e->bwd_slabs_clean) CVX_HIP(hipMemsetAsync(e->stat_region + e->stat_half, 0, (size_t)e->stat_half * 8, st));  // (a second backward on one forward)
  e->bwd_slabs_clean = false;
  if (e->n_colsum > 0) {  // every bias gradient (the head's 1x1 output convs) at once: column sums of dpred
    double by = 0;
    for (size_t i = 0; i < e->ops.size(); ++i)
      if (e->ops[i].type == CVX_OP_CONV && e->ops[i].act == CVX_ACT_BIAS) by += 2.0 * w.B * e->ops[i].oh * e->ops[i].ow * e->ops[i].out.c;
    e->cur_op = -1;
    ProfScope ps(e, PROF_MISC, 0, by, st);
    CVX_TRY(cvx_colsum_multi(w.dpred, e->d_colsum, e->n_colsum, e->colsum_max_c, e->d_colsum_blocks, e->n_colsum_blocks, w.inv_scale, e->grads, st));
  }
  // fork: the side stream (weight gradients) starts after everything already queued on the main stream
  CVX_HIP(hipEventRecord(e->ev_fork, st));
  CVX_HIP(hipStreamWaitEvent(e->side, e->ev_fork, 0));
  if (e->use_lanes) CVX_HIP(hipStreamWaitEvent(e->lane, e->ev_fork, 0));
  w.lane_pending = false;
  w.lane_join_op = -1;
  w.pending_stream = nullptr;
  return 0;
}

// backward of op i: BN/bias gradients and the data gradient on the op's stream, the weight gradient queued for the side stream
int backward_op(cvx_engine* e, int i) {
  cvx_bw_state& w = bw_of(e);
  const int B = w.B;
  const Buf& pb = e->bufs[e->pred_buf];
  const long long A = (long long)pb.d.h * pb.d.w;
  const cvx_op_desc& o = e->ops[i];
  e->cur_op = i;
    hipStream_t st = e->stream;
    static const int head_bwd_mask = cvx_tune_int("CVX_HEAD_BWD_MASK", 6);  // bit l: Detect level l's backward runs on the lane stream
    if (e->head_overlap && !e->profile) {
      // Detect levels by mask on the lane stream; the main chain (the other levels, then the neck) joins it only where it first touches the
      // gradient of a lane level's input (head_join_op): the neck's bottom-up path runs beside a level that is still on the lane stream
      const int lvl = o.lane - 1;
      if (lvl >= 0 && lvl < 3 && ((head_bwd_mask >> lvl) & 1)) {
        st = e->lane;
        w.lane_pending = true;
        w.lane_join_op = std::max(w.lane_join_op, e->head_join_op[lvl]);
      } else if (w.lane_pending && lvl < 0 && i <= w.lane_join_op) {
        CVX_TRY(join_lane(e));
        w.lane_join_op = -1;
      }
    } else if (e->use_lanes && o.lane >= 2) {
      st = e->lane;
      w.lane_pending = true;
    } else if (w.lane_pending && o.lane == 0) {
      CVX_TRY(join_lane(e));  // the first op after the head: it consumes gradients the lanes wrote
    }
    // weight gradients wait for ONE event per batch, recorded on the stream that produced their dy: a batch never mixes streams
    if (!w.pending.empty() && w.pending_stream != st) CVX_TRY(flush_wgrads(e, e->ev_lane_fork, w.pending_stream));
    w.pending_stream = st;
    if (o.type == CVX_OP_MAXPOOL5) {
      if (!e->sppf3.empty() && e->sppf3[i] == 2) {
        if (i >= 2 && e->sppf3[i - 2] == 1) {  // the last pool of an SPPF triple, met first on the way back: the whole chain in one launch
          const cvx_op_desc &p0 = e->ops[i - 2], &p1 = e->ops[i - 1];
          ProfScope ps(e, PROF_MISC, 0, 14.0 * B * o.ih * o.iw * o.in.c, st);
          const int acc_mask = (e->pool[i].in_accum ? 4 : 0) | (e->pool[i - 1].in_accum ? 2 : 0) | (e->pool[i - 2].in_accum ? 1 : 0);
          CVX_TRY(cvx_sppf_pool3_bwd(make_view(e, o.out, true), make_view(e, o.in, true), make_view(e, p1.in, true), make_view(e, p0.in, true), B, o.ih,
                                     o.iw, o.in.c, e->pool[i - 2].idx, e->pool[i - 1].idx, e->pool[i].idx, acc_mask, st));
        }
        return 0;  // (the middle pool: done with the last one)
      }
      if (!e->sppf3.empty() && e->sppf3[i] == 1) return 0;  // the first pool of the triple: done with the last one
      ProfScope ps(e, PROF_MISC, 0, 7.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_maxpool5_bwd(make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih, o.iw, o.in.c, e->pool[i].idx,
                               e->pool[i].in_accum, st));
      return 0;
    }
    if (o.type == CVX_OP_UPSAMPLE2) {
      ProfScope ps(e, PROF_MISC, 0, 12.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_upsample2_bwd(make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih, o.iw, o.in.c, e->pool[i].in_accum, st));
      return 0;
    }
    if (o.type == CVX_OP_L2NORM) {
      ProfScope ps(e, PROF_MISC, 0, 6.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_l2norm_bwd(make_view(e, o.in, false), make_view(e, o.out, true), make_view(e, o.in, true), e->params + o.gamma_off,
                             e->grads + o.gamma_off, w.inv_scale, B, o.ih * o.iw, o.in.c, e->pool[i].in_accum, (float*)e->pool[i].idx, st));
      return 0;
    }
    if (o.type == CVX_OP_COPY) {
      ProfScope ps(e, PROF_MISC, 0, (e->pool[i].in_accum ? 6.0 : 4.0) * B * o.ih * o.iw * o.in.c, st);
      if (e->pool[i].in_accum)
        CVX_TRY(cvx_add_slice(make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih * o.iw, o.in.c, st));
      else
        CVX_TRY(cvx_copy_slice(make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih * o.iw, o.in.c, st));
      return 0;
    }
    if (o.type == CVX_OP_DWCONVT) {
      ProfScope ps(e, PROF_MISC, 0, 4.0 * B * (o.ih * o.iw + o.oh * o.ow) * o.in.c, st);
      CVX_TRY(cvx_dwconvt_bwd(make_view(e, o.in, false), make_view(e, o.out, true), make_view(e, o.in, true), e->params + o.w_off,
                              e->grads + o.w_off, w.inv_scale, B, o.ih, o.iw, o.in.c, o.stride, e->pool[i].in_accum, (float*)e->pool[i].idx, st));
      return 0;
    }
    if (o.type == CVX_OP_MAXPOOL2) {
      ProfScope ps(e, PROF_MISC, 0, (4.0 * o.ih * o.iw + 2.0 * o.oh * o.ow) * B * o.in.c, st);
      CVX_TRY(cvx_maxpool2_bwd(make_view(e, o.in, false), make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih, o.iw, o.oh, o.ow, o.in.c,
                               e->pool[i].in_accum, st));
      return 0;
    }
    if (o.type == CVX_OP_MAXPOOL3S2 || o.type == CVX_OP_MAXPOOL3S1) {
      ProfScope ps(e, PROF_MISC, 0, (2.0 * o.ih * o.iw + 3.0 * o.oh * o.ow) * B * o.in.c, st);
      CVX_TRY(cvx_maxpool3_bwd(make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih, o.iw, o.in.c, o.type == CVX_OP_MAXPOOL3S2 ? 2 : 1,
                               e->pool[i].idx, e->pool[i].in_accum, st));
      return 0;
    }
    if (o.type == CVX_OP_AVGPOOL) {
      ProfScope ps(e, PROF_MISC, 0, (e->pool[i].in_accum ? 4.0 : 2.0) * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_avgpool_global_bwd(make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih * o.iw, o.in.c, e->pool[i].in_accum, st));
      return 0;
    }
    if (o.type == CVX_OP_RESIZE) {
      ProfScope ps(e, PROF_MISC, 0, 2.0 * B * (o.ih * o.iw + o.oh * o.ow) * o.in.c, st);
      CVX_TRY(cvx_resize_bilinear_bwd(make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih, o.iw, o.oh, o.ow, o.in.c, e->pool[i].in_accum,
                                      st));
      return 0;
    }
    if (o.type == CVX_OP_DROPOUT) {  // the forward's mask again, from the same (seed, pass, op)
      ProfScope ps(e, PROF_MISC, 0, 4.0 * B * o.ih * o.iw * o.in.c, st);
      CVX_TRY(cvx_dropout(make_view(e, o.out, true), make_view(e, o.in, true), B, o.ih * o.iw, o.in.c, dropout_p(o), dropout_seed(e, i),
                          e->pool[i].in_accum, st));
      return 0;
    }
    CVX_CHECK(o.type == CVX_OP_CONV, "op without a backward pass");
    ConvRt& c = e->conv[i];
    const long long M = (long long)B * o.oh * o.ow;
    const int C = o.out.c;
    const int hw = o.oh * o.ow;
    ViewDesc dyv;  // gradient w.r.t. the raw conv output
    if (o.act == CVX_ACT_BIAS) {
      dyv.p = w.dpred + (long long)o.out.pix_off * pb.d.c + o.out.coff;
      dyv.ld = pb.d.c;
      dyv.bstride = A * pb.d.c;
      // (the bias gradient = column sums of dy went out with all the others in backward_begin: cvx_colsum_multi)
    } else if (o.act == CVX_ACT_BIAS_RELU || o.act == CVX_ACT_BIAS_LINEAR) {
      ProfScope ps(e, PROF_BN_BWD, 0, 6.0 * M * C, st);
      CVX_TRY(cvx_bias_act_bwd(make_view(e, o.out, true), make_view(e, o.out, false), o.act == CVX_ACT_BIAS_RELU ? 1 : 0, M, C, hw, c.dybuf, c.stat_bwd,
                               w.inv_scale, e->grads + o.bias_off, st));
      dyv.p = c.dybuf;
      dyv.ld = C;
      dyv.bstride = (long long)hw * C;
    } else {
      ViewDesc gout = make_view(e, o.out, true);
      ViewDesc gres = make_view(e, o.res, true);
      BnCoef k{c.invstd, e->params + o.gamma_off, e->params + o.beta_off, c.raw16 ? c.mean : nullptr};
      // SiLU with a pre-activation residual needs the residual's forward value (ReLU's mask rides in xhat's lowest bit)
      const bool pre = (o.flags & CVX_OPF_RES_PRE_ACT) != 0 && o.res.buf >= 0;
      const BnActKind ak{act_kind(o), pre ? 1 : 0, act_kind(o) == 0 && pre ? make_view(e, o.res, false) : ViewDesc{nullptr, 0, 0}};
      ProfScope ps(e, PROF_BN_BWD, 0, (c.stem ? 4.0 : (gres.p ? 14.0 : 10.0)) * M * C, st);
      static const bool tune_skip_reduce = cvx_tune_int("CVX_TUNE_SKIP_BN_REDUCE", 0) != 0;  // tuning build: timing without the pass (wrong results)
      // one launch (reduce, grid gate, apply from registers) where the layer qualifies -- on the main stream only: two gated kernels side
      // by side (the Detect lanes) could keep each other's blocks off the CUs
      int one = 1;
      if (!c.stem && !c.raw16 && st == e->stream && !tune_skip_reduce)
        one = cvx_bn_bwd_fused(c.ybuf, M, C, hw, k, c.stat_bwd, reinterpret_cast<unsigned long long*>(c.stat_bwd + (long long)cvx_stat_replicas(C) * C * CVX_STAT_WORDS),
                               w.inv_scale, e->grads + o.gamma_off, e->grads + o.beta_off, gout, ak, c.dybuf, gres, c.res_accum, st);
      if (one < 0) return one;
      // (the stem's one-pass backward produces these sums itself: cvx_stem_backward_fold, queued below)
      bool stem_onepass = false;
      if (c.stem && e->last_images) {
        const Buf& ib = e->bufs[e->image_buf];
        stem_onepass = cvx_stem_backward_onepass_ok(StemParams{e->last_images, B, ib.d.h, ib.d.w, o.oh, o.ow, e->params + o.w_off, C}, c.ybuf, gout, c.nsplit);
      }
      if (one == 1 && !tune_skip_reduce && !stem_onepass) CVX_TRY(cvx_bn_bwd_reduce(c.ybuf, M, C, hw, k, gout, ak, c.stat_bwd, st));
      // the stem's "apply" half is fused into its weight gradient (cvx_stem_backward, queued below): dy is never materialised
      if (one == 1 && !c.stem)
        CVX_TRY(cvx_bn_bwd_apply(c.ybuf, M, C, hw, k, c.stat_bwd, w.inv_scale, e->grads + o.gamma_off, e->grads + o.beta_off, gout, ak, c.dybuf,
                                 gres, c.res_accum, st));
      dyv.p = c.dybuf;
      dyv.ld = C;
      dyv.bstride = (long long)hw * C;
    }
    // dy of this layer is complete here; the side stream learns it through an event, recorded once per `w.wg_batch`
    // layers (a marker packet between two main-chain kernels costs ~5 us, see flush_wgrads below)
    // ---- weight gradient -> fp32 slabs (queued for the side stream) ----
    auto queue_wgrad = [&]() -> int {
      ViewDesc xin = c.stem ? ViewDesc{nullptr, 0, 0} : make_view(e, o.in, false);
      WgradParams wp;
      memset(&wp, 0, sizeof(wp));
      wp.x = xin.p;
      wp.x_bstride = xin.bstride;
      wp.x_ld = xin.ld;
      wp.IH = o.ih;
      wp.IW = o.iw;
      wp.Cin = c.cin_g;
      wp.dy = dyv.p;
      wp.dy_bstride = dyv.bstride;
      wp.dy_ld = dyv.ld;
      wp.Cout = C;
      wp.B = B;
      wp.OH = o.oh;
      wp.OW = o.ow;
      wp.stride = o.stride;
      wp.ntaps = c.ntaps;
      wp.taps = c.taps_fwd;
      wp.slabs = e->slabs + c.slab_off;
      wp.nsplit = c.nsplit;
      wp.cin_pad16 = c.cin_pad16;
      wp.std3x3 = c.std3x3;
      PendingWgrad pw{wp, c.stem, StemParams{}, ViewDesc{nullptr, 0, 0}, BnCoef{nullptr, nullptr, nullptr}, nullptr, 0.f, nullptr, nullptr,
                      conv_flops(o, B), conv_bytes(o, B) + 4.0 * c.nsplit * C * c.ntaps * c.cin_pad16, i};
      if (c.stem) {
        const Buf& ib = e->bufs[e->image_buf];
        CVX_CHECK(e->last_images, "the stem's weight gradient needs the images of the training forward");
        pw.sp = StemParams{e->last_images, B, ib.d.h, ib.d.w, o.oh, o.ow, e->params + o.w_off, C};
        pw.bytes = 12.0 * B * ib.d.h * ib.d.w + 4.0 * M * C + 4.0 * c.nsplit * C * 144;
        pw.wp.dy = c.ybuf;  // xhat
        pw.gout = make_view(e, o.out, true);
        pw.coef = BnCoef{c.invstd, e->params + o.gamma_off, e->params + o.beta_off, c.mean};  // (mean: for the recomputed xhat)
        pw.part = c.stat_bwd;
        pw.inv_scale = w.inv_scale;
        pw.dgamma = e->grads + o.gamma_off;
        pw.dbeta = e->grads + o.beta_off;
        pw.dw = e->grads + o.w_off;
      }
      w.pending.push_back(pw);
      // batches of wg_batch layers share one event record -- except at the end of the pass: the last layers' weight
      // gradients are the largest and form the tail of the step, they start the moment their dy exists
      if ((int)w.pending.size() >= w.wg_batch || i <= e->slab_tail_op || e->slab_tail_op < 0) CVX_TRY(flush_wgrads(e, c.ev_dy, st));
      return 0;
    };
    // The weight gradient needs dy only: for the layers of the step's tail it is queued BEFORE the layer's data gradient, so that it
    // starts beside it instead of behind it (the last layers' weight gradients are the largest and nothing follows them to hide behind).
    static const int wgrad_early = cvx_tune_int("CVX_WGRAD_EARLY", 1);  // 0: never, 1: the tail layers, 2: every layer
    const bool wgrad_first = !c.stem && (wgrad_early >= 2 || (wgrad_early == 1 && e->slab_tail_op >= 0 && i <= e->slab_tail_op));
    if (wgrad_first) CVX_TRY(queue_wgrad());
    // ---- data gradient: dx = dy (*) W^T, one launch per output phase of the forward stride ----
    if (o.needs_dgrad) {
      ViewDesc gin = make_view(e, o.in, true);
      // the phases of a strided data gradient differ only in tap subset and output phase: one launch (blockIdx.z)
      static const bool merge_off = cvx_tune_set("CVX_NO_PHASE_MERGE");
      bool merged = !merge_off && c.ndg > 1 && c.ndg <= 4;
      for (int q = 0; q < c.ndg && merged; ++q)
        if (c.dg[q].OH2 <= 0 || c.dg[q].OW2 <= 0 || c.dg[q].ntaps <= 0) merged = false;
      // stride > kernel (ResNet's 1x1 stride-2 downsample): input pixels of the tap-less phases receive no gradient -- the first
      // writer of the slice zeroes it, the phases with taps then overwrite their own pixels
      bool empty_phase = false;
      for (int q = 0; q < c.ndg; ++q) empty_phase |= c.dg[q].OH2 > 0 && c.dg[q].OW2 > 0 && c.dg[q].ntaps == 0;
      if (empty_phase && !c.in_accum) CVX_TRY(cvx_zero_slice(gin, B, o.ih * o.iw, o.in.c, st));
      if (c.ps_on) {  // the four phases as one GEMM with a pixel-shuffle store (fill_conv_ps_shape)
        ConvParams cp;
        fill_conv_ps_shape(e, i, B, &cp);
        cp.in = dyv.p;
        cp.in_bstride = dyv.bstride;
        cp.in_ld = dyv.ld;
        cp.accumulate = c.in_accum;
        cp.wt_packed = c.ps_pk;
        cp.wt_packed_bn = c.ps_bn;
        cp.wt_packed_kc = c.ps_kc;
        cp.out16 = gin.p;
        cp.out_ld = gin.ld;
        cp.out_bstride = gin.bstride;
        const double fl = 2.0 * B * o.oh * o.ow * (double)o.in.c * c.ntaps * C;  // the useful multiplications (the zero blocks are not counted)
        ProfScope ps(e, PROF_CONV_DGRAD, fl, conv_bytes(o, B) + (c.in_accum ? 2.0 * B * o.ih * o.iw * o.in.c : 0.0), st);
        CVX_TRY(cvx_conv_igemm_launch(cp, st, nullptr));
      }
      for (int q = 0; q < c.ndg && !c.ps_on; ++q) {
        if (merged && q > 0) break;  // everything went out with phase 0
        const DgClass& dc = c.dg[q];
        if (dc.OH2 <= 0 || dc.OW2 <= 0 || dc.ntaps == 0) continue;
        ConvParams cp;
        memset(&cp, 0, sizeof(cp));
        cp.in = dyv.p;
        cp.in_bstride = dyv.bstride;
        cp.in_ld = dyv.ld;
        cp.IH = o.oh;
        cp.IW = o.ow;
        cp.Cin = C;
        cp.wt = e->shadow + c.sh_dg;
        cp.wt_ld = c.ntaps * C;
        cp.Cout = o.in.c;
        cp.B = B;
        cp.OH2 = dc.OH2;
        cp.OW2 = dc.OW2;
        cp.IS = 1;
        cp.OS = o.stride;
        cp.oph = dc.oph;
        cp.opw = dc.opw;
        cp.OWr = o.iw;
        cp.ntaps = dc.ntaps;
        cp.taps = dc.taps;
        cp.epi = CVX_EPI_PLAIN;
        cp.zeros = e->zero_page;
        cp.halo_taps_ok = dc.halo_ok ? 1 : 0;
        cp.pointwise = dc.pointwise;
        cp.halo_pos = dc.halo_pos;
        cp.halo_wt = dc.halo_wt;
        cp.accumulate = c.in_accum;
        cp.wt_packed = dc.gemm_pk;
        cp.wt_packed_bn = dc.gemm_bn;
        cp.wt_packed_kc = dc.gemm_kc;
        cp.tile_packed = dc.tile_pk;
        cp.tile_packed_bn = dc.tile_bn;
        cp.out16 = gin.p;
        cp.out_ld = gin.ld;
        cp.out_bstride = gin.bstride;
        double fl = 2.0 * B * dc.OH2 * dc.OW2 * (double)o.in.c * dc.ntaps * C;
        double by = (conv_bytes(o, B) + (c.in_accum ? 2.0 * B * o.ih * o.iw * o.in.c : 0.0)) / c.ndg;
        if (merged) {
          cp.nphase = c.ndg;
          fl = 0;
          for (int z = 0; z < c.ndg; ++z) {
            const DgClass& dz = c.dg[z];
            cp.phase[z] = ConvParams::Phase{dz.taps, dz.ntaps, dz.OH2, dz.OW2, dz.oph, dz.opw};
            fl += 2.0 * B * dz.OH2 * dz.OW2 * (double)o.in.c * dz.ntaps * C;
          }
          by *= c.ndg;
          cp.halo_taps_ok = 0;
          cp.pointwise = 0;
        }
        ProfScope ps(e, PROF_CONV_DGRAD, fl, by, st);
        CVX_TRY(cvx_conv_igemm_launch(cp, st, nullptr));
      }
    }
    if (!wgrad_first) CVX_TRY(queue_wgrad());
  return 0;
}

}  // namespace

extern "C" int cvx_engine_backward(cvx_engine* e, const void* dpred_f16, float loss_scale) {
  CVX_TRY(backward_begin(e, dpred_f16, loss_scale));
  cvx_bw_state& w = bw_of(e);
  hipStream_t st = e->stream;
  const float inv_scale = w.inv_scale;
  // The gradient slabs are folded into the arena on a stream of their own, in chunks of a few layers: a chunk is reduced as
  // soon as its weight gradients have been queued on the side stream, i.e. DURING the backward pass, where the main chain is
  // latency-bound and leaves HBM bandwidth unused -- not at its end, where the stem's gradient kernels need all of it (one
  // reduction of everything at the end moved 0.4 GB beside them).  The main stream only waits for the last, tiny chunk.
  hipStream_t rs = e->red;
  static const int chunk_convs = cvx_tune_int("CVX_SLAB_CHUNK", 6);
  double slab_bytes = 0;
  int chunk_hi = -1, in_chunk = 0;  // conv ops [i, chunk_hi] whose slabs are not reduced yet
  auto reduce_chunk = [&](int lo, int hi, bool first, hipStream_t on) -> int {
    const int blk0 = e->conv[lo].slab_blk0, blk1 = e->conv[hi].slab_blk1;
    if (blk1 <= blk0) return 0;
    ProfScope ps(e, PROF_SLAB_REDUCE, 0, first ? slab_bytes + 8.0 * e->n_params : 0.0, on);
    return cvx_reduce_slabs(e->slabs, e->grads, inv_scale, e->d_slab, e->d_slab_blocks + blk0, blk1 - blk0, on);
  };
  for (size_t i = 0; i < e->ops.size(); ++i)
    if (e->ops[i].type == CVX_OP_CONV) slab_bytes += 4.0 * e->conv[i].nsplit * e->ops[i].out.c * e->conv[i].ntaps * e->conv[i].cin_pad16;
  bool first = true;
  int lo_conv = -1;
  for (int i = (int)e->ops.size() - 1; i >= 0; --i) {
    CVX_TRY(backward_op(e, i));
    if (e->ops[i].type != CVX_OP_CONV) continue;
    if (chunk_hi < 0) chunk_hi = i;
    lo_conv = i;
    ++in_chunk;
    if (in_chunk >= chunk_convs && w.pending.empty() && i > 1) {  // every weight gradient of [i, chunk_hi] is on the side stream
      if (rs != e->side) {  // (a reduction stream of its own: it follows the weight gradients through an event)
        CVX_HIP(hipEventRecord(e->ev_mid, e->side));
        CVX_HIP(hipStreamWaitEvent(rs, e->ev_mid, 0));
      }
      CVX_TRY(reduce_chunk(i, chunk_hi, first, rs));
      first = false;
      chunk_hi = -1;
      in_chunk = 0;
    }
  }
  w.active = false;
  CVX_TRY(join_lane(e));
  CVX_TRY(flush_wgrads(e, e->ev_fork, st));  // (non-conv first ops: nothing pending in practice)
  e->cur_op = -1;
  // The last, small chunk: the first layers' weight-gradient slabs (the stem folds its own partial blocks on the main stream, inside
  // cvx_stem_backward_fold).  It is reduced on the reduction stream behind those weight gradients -- i.e. beside the stem's backward kernel
  // -- and the main stream waits for it once, at the very end.  CVX_TAIL_REDUCE_MAIN=1 (tuning): on the main stream, behind the stem.
  static const bool tail_on_main = cvx_tune_int("CVX_TAIL_REDUCE_MAIN", 0) != 0;
  if (tail_on_main) {
    CVX_HIP(hipEventRecord(e->ev_red, rs));        // everything queued on the reduction / weight-gradient streams so far
    if (rs != e->side) {
      CVX_HIP(hipEventRecord(e->ev_join, e->side));
      CVX_HIP(hipStreamWaitEvent(st, e->ev_join, 0));
    }
    CVX_HIP(hipStreamWaitEvent(st, e->ev_red, 0));
    if (chunk_hi >= 0) CVX_TRY(reduce_chunk(lo_conv, chunk_hi, first, st));
    return 0;
  }
  if (rs != e->side) {
    CVX_HIP(hipEventRecord(e->ev_join, e->side));
    CVX_HIP(hipStreamWaitEvent(rs, e->ev_join, 0));
  }
  if (chunk_hi >= 0) CVX_TRY(reduce_chunk(lo_conv, chunk_hi, first, rs));
  CVX_HIP(hipEventRecord(e->ev_red, rs));
  CVX_HIP(hipStreamWaitEvent(st, e->ev_red, 0));
  return 0;
}

void cvx_engine_free_bw(cvx_engine* e) {
  delete e->bw;
  e->bw = nullptr;
}

// ---- segmented backward: lets the caller exchange (all-reduce) the gradients of finished parameter ranges while the
// rest of the pass still runs.  begin -> range(hi, lo) ... -> grads_ready(hi, lo, stream) per range -> end. ----
extern "C" int cvx_engine_backward_begin(cvx_engine* e, const void* dpred_f16, float loss_scale) {
  return backward_begin(e, dpred_f16, loss_scale);
}

extern "C" int cvx_engine_backward_range(cvx_engine* e, int32_t op_hi, int32_t op_lo) {
  CVX_CHECK(e && e->bw && e->bw->active, "cvx_engine_backward_range without cvx_engine_backward_begin");
  cvx_bw_state& w = bw_of(e);
  CVX_CHECK(op_hi == w.next_op && op_lo >= 0 && op_lo <= op_hi, "ranges must tile the op list from the last op down to 0");
  for (int i = op_hi; i >= op_lo; --i) CVX_TRY(backward_op(e, i));
  w.next_op = op_lo - 1;
  CVX_TRY(join_lane(e));                             // (a range that ends inside the head)
  CVX_TRY(flush_wgrads(e, e->ev_fork, e->stream));  // every weight gradient of the range is queued on the side stream
  return 0;
}

extern "C" int cvx_engine_grads_ready(cvx_engine* e, int32_t op_hi, int32_t op_lo, void* hip_stream) {
  CVX_CHECK(e && e->bw && op_lo >= 0 && op_hi < (int)e->ops.size() && op_lo <= op_hi, "bad arguments");
  cvx_bw_state& w = bw_of(e);
  CVX_CHECK(w.next_op < op_lo, "cvx_engine_grads_ready: the range has not been run yet");
  hipStream_t cs = (hipStream_t)hip_stream;
  // the caller's stream waits for everything queued so far on the main and the side stream ...
  hipEvent_t* ev = &e->ev_seg[(e->ev_seg_next++ % 4) * 2];
  for (int k = 0; k < 2; ++k)
    if (!ev[k]) CVX_HIP(hipEventCreateWithFlags(&ev[k], cvx_event_flags()));
  if (cs != e->stream) {
    CVX_HIP(hipEventRecord(ev[0], e->stream));
    CVX_HIP(hipStreamWaitEvent(cs, ev[0], 0));
  }
  if (cs != e->side) {
    CVX_HIP(hipEventRecord(ev[1], e->side));
    CVX_HIP(hipStreamWaitEvent(cs, ev[1], 0));
  }
  // ... then folds the weight-gradient slabs of the range's conv ops into the gradient arena there
  int blk0 = -1, blk1 = -1;
  for (int i = op_lo; i <= op_hi; ++i) {
    if (e->ops[i].type != CVX_OP_CONV) continue;
    const ConvRt& c = e->conv[i];
    if (blk0 < 0) blk0 = c.slab_blk0;
    CVX_CHECK(blk1 < 0 || c.slab_blk0 == blk1, "slab block table is not in op order");
    blk1 = c.slab_blk1;
  }
  if (blk0 >= 0 && blk1 > blk0)
    CVX_TRY(cvx_reduce_slabs(e->slabs, e->grads, w.inv_scale, e->d_slab, e->d_slab_blocks + blk0, blk1 - blk0, cs));
  return 0;
}

int cvx_comm_allreduce_slice(float* base, long long p0, long long p1, void* comm, hipStream_t stream);  // comm.hip

extern "C" int cvx_allreduce_grads(cvx_engine* e, void* nccl_comm, void* hip_stream) {
  CVX_CHECK(e && nccl_comm && e->grads && e->n_params > 0, "bad arguments (engine bound with a gradient arena, RCCL communicator)");
  return cvx_comm_allreduce_slice(e->grads, 0, e->n_params, nccl_comm, (hipStream_t)hip_stream);
}

extern "C" int cvx_engine_backward_exchange(cvx_engine* e, const void* dpred_f16, float loss_scale, void* nccl_comm, const int64_t* buckets,
                                            int32_t n_buckets, void* comm_stream) {
  CVX_CHECK(e && dpred_f16 && nccl_comm && buckets && n_buckets > 0 && comm_stream, "bad arguments");
  CVX_CHECK(e->grads, "cvx_engine_bind was called without a gradient arena");
  hipStream_t cs = (hipStream_t)comm_stream;
  CVX_TRY(cvx_engine_backward_begin(e, dpred_f16, loss_scale));
  for (int b = 0; b < n_buckets; ++b) {
    const int64_t* q = buckets + 4 * b;  // op_hi, op_lo, p_start, p_end
    CVX_CHECK(q[2] >= 0 && q[3] <= e->n_params && q[2] <= q[3], "bucket slice outside the gradient arena");
    CVX_TRY(cvx_engine_backward_range(e, (int32_t)q[0], (int32_t)q[1]));
    CVX_TRY(cvx_engine_grads_ready(e, (int32_t)q[0], (int32_t)q[1], comm_stream));
    CVX_TRY(cvx_comm_allreduce_slice(e->grads, q[2], q[3], nccl_comm, cs));
  }
  CVX_TRY(cvx_engine_backward_end(e));
  // the engine's stream (the optimiser step comes next on it) waits for the last exchange
  hipEvent_t* ev = &e->ev_seg[(e->ev_seg_next++ % 4) * 2];
  if (!ev[0]) CVX_HIP(hipEventCreateWithFlags(&ev[0], cvx_event_flags()));
  CVX_HIP(hipEventRecord(ev[0], cs));
  CVX_HIP(hipStreamWaitEvent(e->stream, ev[0], 0));
  return 0;
}

extern "C" int cvx_engine_backward_end(cvx_engine* e) {
  CVX_CHECK(e && e->bw && e->bw->active, "cvx_engine_backward_end without cvx_engine_backward_begin");
  cvx_bw_state& w = bw_of(e);
  CVX_CHECK(w.next_op < 0, "cvx_engine_backward_end: not every op range has been run");
  w.active = false;
  e->cur_op = -1;
  CVX_HIP(hipEventRecord(e->ev_join, e->side));  // the main stream continues after the last weight gradient
  CVX_HIP(hipStreamWaitEvent(e->stream, e->ev_join, 0));
  return 0;
}

extern "C" int cvx_engine_profile(cvx_engine* e, int32_t enable) {
  CVX_CHECK(e, "null engine");
  CVX_HIP(hipStreamSynchronize(e->stream));
  e->profile = enable != 0;
  e->ev_used = 0;
  e->prof_recs.clear();
  return 0;
}

extern "C" int cvx_engine_profile_read(cvx_engine* e, int32_t n_classes, double* ms, double* flops, double* bytes, int64_t* launches) {
  CVX_CHECK(e && ms && flops && bytes && launches && n_classes >= 7, "bad arguments (7 classes)");
  CVX_HIP(hipStreamSynchronize(e->stream));
  for (int i = 0; i < n_classes; ++i) {
    ms[i] = flops[i] = bytes[i] = 0;
    launches[i] = 0;
  }
  for (size_t r = 0; r < e->prof_recs.size() && 2 * r + 1 < e->ev_used + 1; ++r) {
    float t = 0.f;
    CVX_HIP(hipEventElapsedTime(&t, e->ev_pool[2 * r], e->ev_pool[2 * r + 1]));
    const auto& rec = e->prof_recs[r];
    ms[rec.cls] += t;
    flops[rec.cls] += rec.flops;
    bytes[rec.cls] += rec.bytes;
    launches[rec.cls] += 1;
  }
  e->ev_used = 0;
  e->prof_recs.clear();
  return 0;
}

extern "C" int cvx_debug_clock_buffer(void* buf) {
  g_cvx_clk = (unsigned long long*)buf;
  return 0;
}

extern "C" int cvx_engine_profile_dump(cvx_engine* e, const char* path) {
  CVX_CHECK(e && path, "bad arguments");
  CVX_HIP(hipStreamSynchronize(e->stream));
  CVX_HIP(hipStreamSynchronize(e->side));
  FILE* f = fopen(path, "w");
  CVX_CHECK(f, "cannot open the profile dump file");
  fprintf(f, "class,op,flops,bytes,ms\n");
  for (size_t r = 0; r < e->prof_recs.size() && 2 * r + 1 < e->ev_used + 1; ++r) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, e->ev_pool[2 * r], e->ev_pool[2 * r + 1]) != hipSuccess) t = -1.f;
    const auto& rec = e->prof_recs[r];
    fprintf(f, "%d,%d,%.0f,%.0f,%.6f\n", rec.cls, rec.op, rec.flops, rec.bytes, t);
  }
  fclose(f);
  e->ev_used = 0;
  e->prof_recs.clear();
  return 0;
}

// ---- thin C wrappers over the launchers ------------------------------------------------------
extern "C" int cvx_pred_level_to_nchw(const float* pred, int32_t batch, int32_t anchors, int32_t no, int32_t a_off, int32_t h, int32_t w,
                                      float* out_nchw, void* hip_stream) {
  return cvx_pred_to_nchw(pred, batch, anchors, no, a_off, h, w, out_nchw, (hipStream_t)hip_stream);
}
extern "C" int cvx_nchw_grad_to_dpred(const float* grad_nchw, int32_t batch, int32_t anchors, int32_t no, int32_t a_off, int32_t h, int32_t w,
                                      float scale, void* dpred_f16, void* hip_stream) {
  return cvx_nchw_to_pred_f16(grad_nchw, batch, anchors, no, a_off, h, w, scale, (half_t*)dpred_f16, (hipStream_t)hip_stream);
}
extern "C" int cvx_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                             float eps, int32_t step, const int32_t* found_inf, int32_t zero_grad, void* hip_stream) {
  CVX_CHECK(params && grads && exp_avg && exp_avg_sq && n > 0, "bad arguments");
  CVX_CHECK(((uintptr_t)params % 16) == 0 && ((uintptr_t)grads % 16) == 0 && ((uintptr_t)exp_avg % 16) == 0 && ((uintptr_t)exp_avg_sq % 16) == 0,
            "adam arenas must be 16-byte aligned");
  return cvx_adam(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, found_inf, zero_grad, (hipStream_t)hip_stream);
}
extern "C" int cvx_adam_step_dev(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float beta1, float beta2, float eps,
                                 float* state, const int32_t* found_inf, int32_t zero_grad, float grad_scale, void* hip_stream) {
  CVX_CHECK(params && grads && exp_avg && exp_avg_sq && state && n > 0, "bad arguments");
  return cvx_adam_dev(params, grads, exp_avg, exp_avg_sq, n, beta1, beta2, eps, state, found_inf, zero_grad, grad_scale,
                      (hipStream_t)hip_stream);
}
extern "C" int cvx_engine_set_stream(cvx_engine* e, void* hip_stream) {
  CVX_CHECK(e, "null engine");
  e->stream = (hipStream_t)hip_stream;
  return 0;
}
// The stream of the data-parallel gradient exchange = the engine's weight-gradient stream: a range's slab fold and all-reduce queue up
// behind the weight gradients they depend on (no events needed), the next range's weight gradients behind them.  A dedicated stream
// overlaps a little better (the all-reduces do not hold up the next weight gradients) but is the process's FOURTH hardware queue, and
// RCCL orders every multi-rank launch against an internal stream of its own (a fifth: 2.2-2.5x on every step) --
// which a 1-GPU box cannot measure; three queues leave room for it.  CVX_XCHG_OWN_STREAM=1 (tuning build): the dedicated stream.
extern "C" void* cvx_engine_exchange_stream(cvx_engine* e) {
  if (!e) return nullptr;
  static const bool own = cvx_tune_int("CVX_XCHG_OWN_STREAM", 0) != 0;
  if (!own) return (void*)e->side;
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  hipStream_t s = nullptr;
  if (shared_stream(e->device, &AuxStreams::xchg, &s, [&](hipStream_t* out) { return hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio_least); }) !=
      hipSuccess) {
    (void)hipGetLastError();
    cvx_set_error("cvx_engine_exchange_stream: could not create the stream");
    return nullptr;
  }
  return (void*)s;
}

extern "C" int cvx_check_finite(const float* grads, int64_t n, int32_t* found_inf, void* hip_stream) {
  CVX_CHECK(grads && found_inf, "bad arguments");
  return cvx_check_finite_launch(grads, n, found_inf, (hipStream_t)hip_stream);
}

// ---- single-op entry points --------------------------------------------------------------------
namespace {
int make_taps(std::vector<ConvTap>& host, ConvTap** dev) {
  // the tap table is followed by a 256-byte zero page (DMA padding source of the second-generation kernel)
  const size_t tb = ((host.size() * sizeof(ConvTap) + 255) / 256) * 256;
  CVX_HIP(hipMalloc((void**)dev, tb + 256));
  CVX_HIP(hipMemset(*dev, 0, tb + 256));
  CVX_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(ConvTap), hipMemcpyHostToDevice));
  return 0;
}
const half_t* zeros_after(const ConvTap* dev, size_t ntaps) {
  const size_t tb = ((ntaps * sizeof(ConvTap) + 255) / 256) * 256;
  return reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(dev) + tb);
}
}  // namespace

extern "C" int cvx_conv2d_nhwc(const void* x_f16, int32_t batch, int32_t ih, int32_t iw, int32_t cin, const void* w_f16, int32_t cout,
                               int32_t k, int32_t stride, int32_t pad, int32_t dil, int32_t mode, const float* scale_or_bias,
                               const float* shift, void* out, void* hip_stream) {
  CVX_CHECK(x_f16 && w_f16 && out && k * k <= CVX_MAX_TAPS && stride >= 1 && dil >= 1, "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  const bool force_gemm = (mode & 0x100) != 0;  // unit tests of the GEMM-shaped kernel on shapes the dispatcher gives to another one
  const int gemm_variant = (mode >> 9) & 15;    // ... and of one particular variant of it (0: the cost model's choice)
  const bool force_tile = (mode & 0x2000) != 0;  // likewise the row-band kernel (conv_tile.hip); 0x4000: never the row-band kernel
  const bool no_tile = (mode & 0x4000) != 0;
  mode &= 0xff;
  const int oh = (ih + 2 * pad - dil * (k - 1) - 1) / stride + 1, ow = (iw + 2 * pad - dil * (k - 1) - 1) / stride + 1;
  std::vector<ConvTap> taps;
  for (int r = 0; r < k; ++r)
    for (int s = 0; s < k; ++s) taps.push_back(ConvTap{r * dil - pad, s * dil - pad, r * k + s, 0});
  ConvTap* dt = nullptr;
  CVX_TRY(make_taps(taps, &dt));
  ConvParams cp;
  memset(&cp, 0, sizeof(cp));
  cp.in = (const half_t*)x_f16;
  cp.in_bstride = (long long)ih * iw * cin;
  cp.in_ld = cin;
  cp.IH = ih;
  cp.IW = iw;
  cp.Cin = cin;
  cp.wt = (const half_t*)w_f16;
  cp.wt_ld = k * k * cin;
  cp.Cout = cout;
  cp.B = batch;
  cp.OH2 = oh;
  cp.OW2 = ow;
  cp.IS = stride;
  cp.OS = 1;
  cp.OWr = ow;
  cp.ntaps = k * k;
  cp.taps = dt;
  cp.zeros = zeros_after(dt, taps.size());
  cp.halo_taps_ok = cvx_halo_pack_taps(taps.data(), (int)taps.size(), &cp.halo_pos, &cp.halo_wt) ? 1 : 0;
  cp.pointwise = cvx_taps_pointwise(taps.data(), (int)taps.size());
  cp.std7x7 = dil == 1 ? cvx_taps_std7x7(taps.data(), (int)taps.size()) : 0;
  cp.std3x3 = dil == 1 ? cvx_taps_std3x3(taps.data(), (int)taps.size()) : 0;
  cp.out_ld = cout;
  cp.out_bstride = (long long)oh * ow * cout;
  if (mode == 0) {
    cp.epi = CVX_EPI_PLAIN;
    cp.out16 = (half_t*)out;
  } else if (mode == 1) {
    cp.epi = CVX_EPI_AFFINE_SILU;
    cp.scale = scale_or_bias;
    cp.shift = shift;
    cp.out16 = (half_t*)out;
  } else if (mode == 2) {
    cp.epi = CVX_EPI_BIAS_F32;
    cp.bias = scale_or_bias;
    cp.out32 = (float*)out;
  } else {  // mode 3: the training epilogue -- raw fp32 output + per-channel (sum, sumsq) into `shift` viewed as the replica slabs
    CVX_CHECK(shift, "mode 3 needs the statistics slab (cvx_stat_replicas(cout) * cout * 4 64-bit words, zeroed) in `shift`");
    cp.epi = CVX_EPI_RAW_STATS;
    cp.out32 = (float*)out;
    cp.stats = (long long*)shift;
    cp.stats_replicas = cvx_stat_replicas(cout);
  }
  int rc;
  cp.gemm_variant = gemm_variant;
  if (force_gemm) {
    if (!cvx_conv_gemm_shape_ok(cp)) {
      (void)hipFree(dt);
      CVX_CHECK(false, "shape outside the GEMM-shaped kernel (cin % 8, cout % 4)");
    }
    rc = cvx_conv_gemm_launch(cp, st);
  } else if (force_tile) {
    if (!cvx_conv_tile_shape_ok(cp)) {
      (void)hipFree(dt);
      CVX_CHECK(false, "shape outside the row-band kernel (3x3 stride 1, cin % 8, cin >= 32)");
    }
    cp.clk = g_cvx_clk;
    rc = cvx_conv_tile_launch(cp, st);
  } else {
    cp.no_tile = no_tile ? 1 : 0;
    rc = cvx_conv_igemm_launch(cp, st, nullptr);
  }
  (void)hipStreamSynchronize(st);
  (void)hipFree(dt);
  return rc;
}

extern "C" int cvx_conv2d_dgrad_nhwc(const void* dy_f16, int32_t batch, int32_t ih, int32_t iw, int32_t cin, const void* wt_f16, int32_t cout,
                                     int32_t k, int32_t stride, int32_t pad, int32_t dil, void* dx_f16, void* hip_stream) {
  CVX_CHECK(dy_f16 && wt_f16 && dx_f16 && k * k <= CVX_MAX_TAPS && stride >= 1 && stride <= 4, "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  const int oh = (ih + 2 * pad - dil * (k - 1) - 1) / stride + 1, ow = (iw + 2 * pad - dil * (k - 1) - 1) / stride + 1;
  int rc = 0;
  // 3x3 / stride 2 / pad 1 on an even map: the engine's route -- ONE launch of the GEMM-shaped kernel over the 2 x 2 window of dy, the four
  // phases as channel blocks, pixel-shuffle store (ConvParams::ps_cin) -- where that kernel takes the shape
  if (k == 3 && stride == 2 && pad == 1 && dil == 1 && ih == 2 * oh && iw == 2 * ow && cout % 32 == 0 && cin % 8 == 0) {
    PsPackDesc pd;
    pd.dg_off = pd.ps_off = 0;
    pd.cin_pad = cin;
    pd.T = 9;
    pd.C = cout;
    for (int q = 0; q < 16; ++q) pd.wtap[q] = -1;
    for (int ph = 0; ph < 2; ++ph)
      for (int pw = 0; pw < 2; ++pw)
        for (int r = 0; r < 3; ++r)
          for (int s = 0; s < 3; ++s) {
            const int nh = ph + 1 - r, nw = pw + 1 - s;
            if ((nh & 1) || (nw & 1)) continue;
            pd.wtap[(ph * 2 + pw) * 4 + (nh / 2) * 2 + nw / 2] = r * 3 + s;
          }
    std::vector<ConvTap> pt(4);
    for (int tau = 0; tau < 4; ++tau) pt[tau] = ConvTap{tau >> 1, tau & 1, tau, 0};
    ConvTap* dt = nullptr;
    CVX_TRY(make_taps(pt, &dt));
    ConvParams cp;
    memset(&cp, 0, sizeof(cp));
    cp.in = (const half_t*)dy_f16;
    cp.in_bstride = (long long)oh * ow * cout;
    cp.in_ld = cout;
    cp.IH = oh;
    cp.IW = ow;
    cp.Cin = cout;
    cp.wt_ld = 4 * cout;
    cp.Cout = 4 * cin;
    cp.B = batch;
    cp.OH2 = oh;
    cp.OW2 = ow;
    cp.IS = 1;
    cp.OS = 2;
    cp.OWr = iw;
    cp.ntaps = 4;
    cp.taps = dt;
    cp.zeros = zeros_after(dt, pt.size());
    cp.epi = CVX_EPI_PLAIN;
    cp.ps_cin = cin;
    cp.out16 = (half_t*)dx_f16;
    cp.out_ld = cin;
    cp.out_bstride = (long long)ih * iw * cin;
    if (cvx_conv_gemm_supported(cp)) {
      half_t* psw = nullptr;
      if (hipMalloc((void**)&psw, (size_t)16 * cin * cout * 2) != hipSuccess) {
        (void)hipFree(dt);
        CVX_CHECK(false, "dgrad: out of memory for the pixel-shuffle weights");
      }
      cp.wt = psw;
      rc = cvx_pack_ps_weights((const half_t*)wt_f16, psw, pd, st);
      if (rc == 0) rc = cvx_conv_igemm_launch(cp, st, nullptr);
      (void)hipStreamSynchronize(st);
      (void)hipFree(psw);
      (void)hipFree(dt);
      return rc;
    }
    (void)hipFree(dt);
  }
  for (int ph = 0; ph < stride && rc == 0; ++ph)
    for (int pw = 0; pw < stride && rc == 0; ++pw) {
      std::vector<ConvTap> taps;
      for (int r = 0; r < k; ++r) {
        int nh = ph + pad - r * dil;
        if (((nh % stride) + stride) % stride) continue;
        for (int s = 0; s < k; ++s) {
          int nw = pw + pad - s * dil;
          if (((nw % stride) + stride) % stride) continue;
          taps.push_back(ConvTap{nh / stride, nw / stride, r * k + s, 0});
        }
      }
      const int OH2 = (ih - ph + stride - 1) / stride, OW2 = (iw - pw + stride - 1) / stride;
      if (OH2 <= 0 || OW2 <= 0) continue;
      CVX_CHECK(!taps.empty(), "dgrad phase without taps");
      ConvTap* dt = nullptr;
      CVX_TRY(make_taps(taps, &dt));
      ConvParams cp;
      memset(&cp, 0, sizeof(cp));
      cp.in = (const half_t*)dy_f16;
      cp.in_bstride = (long long)oh * ow * cout;
      cp.in_ld = cout;
      cp.IH = oh;
      cp.IW = ow;
      cp.Cin = cout;
      cp.wt = (const half_t*)wt_f16;
      cp.wt_ld = k * k * cout;
      cp.Cout = cin;
      cp.B = batch;
      cp.OH2 = OH2;
      cp.OW2 = OW2;
      cp.IS = 1;
      cp.OS = stride;
      cp.oph = ph;
      cp.opw = pw;
      cp.OWr = iw;
      cp.ntaps = (int)taps.size();
      cp.taps = dt;
      cp.zeros = zeros_after(dt, taps.size());
      cp.halo_taps_ok = cvx_halo_pack_taps(taps.data(), (int)taps.size(), &cp.halo_pos, &cp.halo_wt) ? 1 : 0;
      cp.pointwise = cvx_taps_pointwise(taps.data(), (int)taps.size());
      cp.epi = CVX_EPI_PLAIN;
      cp.out16 = (half_t*)dx_f16;
      cp.out_ld = cin;
      cp.out_bstride = (long long)ih * iw * cin;
      rc = cvx_conv_igemm_launch(cp, st, nullptr);
      (void)hipStreamSynchronize(st);
      (void)hipFree(dt);
    }
  return rc;
}

extern "C" int64_t cvx_conv2d_wgrad_workspace_bytes(int32_t batch, int32_t oh, int32_t ow, int32_t cin, int32_t cout, int32_t k) {
  const long long M = (long long)batch * oh * ow;
  long long ns = std::min<long long>(std::max<long long>(1, M / 256), 64);
  return ns * cout * k * k * round_up(cin, 16) * 4 + 4096;
}

extern "C" int cvx_conv2d_wgrad_nhwc(const void* x_f16, const void* dy_f16, int32_t batch, int32_t ih, int32_t iw, int32_t cin, int32_t cout,
                                     int32_t k, int32_t stride, int32_t pad, int32_t dil, float* dw, void* workspace, int64_t workspace_bytes,
                                     void* hip_stream) {
  CVX_CHECK(x_f16 && dy_f16 && dw && workspace && k * k <= CVX_MAX_TAPS, "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  const int oh = (ih + 2 * pad - dil * (k - 1) - 1) / stride + 1, ow = (iw + 2 * pad - dil * (k - 1) - 1) / stride + 1;
  CVX_CHECK(workspace_bytes >= cvx_conv2d_wgrad_workspace_bytes(batch, oh, ow, cin, cout, k), "workspace too small");
  const long long M = (long long)batch * oh * ow;
  std::vector<ConvTap> taps;
  for (int r = 0; r < k; ++r)
    for (int s = 0; s < k; ++s) taps.push_back(ConvTap{r * dil - pad, s * dil - pad, r * k + s, 0});
  ConvTap* dt = nullptr;
  CVX_TRY(make_taps(taps, &dt));
  WgradParams wp;
  memset(&wp, 0, sizeof(wp));
  wp.x = (const half_t*)x_f16;
  wp.x_bstride = (long long)ih * iw * cin;
  wp.x_ld = cin;
  wp.IH = ih;
  wp.IW = iw;
  wp.Cin = cin;
  wp.dy = (const half_t*)dy_f16;
  wp.dy_bstride = (long long)oh * ow * cout;
  wp.dy_ld = cout;
  wp.Cout = cout;
  wp.B = batch;
  wp.OH = oh;
  wp.OW = ow;
  wp.stride = stride;
  wp.ntaps = k * k;
  wp.taps = dt;
  wp.slabs = (float*)workspace;
  wp.nsplit = (int)std::min<long long>(std::max<long long>(1, M / 256), 64);
  wp.cin_pad16 = round_up(cin, 16);
  wp.std3x3 = cvx_taps_std3x3(taps.data(), (int)taps.size());
  int rc = cvx_conv_wgrad_launch(wp, st);
  if (rc == 0) {
    // reduce the slabs into dw (overwrite): zero, then the table-driven reducer with one descriptor
    rc = hipMemsetAsync(dw, 0, (size_t)cout * k * k * cin * 4, st) == hipSuccess ? 0 : -1;
    SlabDesc sd{0, 0, wp.nsplit, cout * k * k, cin, wp.cin_pad16, cvx_slab_lanes(wp.nsplit)};
    std::vector<BlockRef> blocks;
    const long long total = (long long)sd.rows * sd.Cin;
    for (long long s0 = 0; s0 < total; s0 += 256 / sd.lanes) blocks.push_back(BlockRef{0, (int)s0});
    SlabDesc* dsd = nullptr;
    BlockRef* dbl = nullptr;
    if (hipMalloc((void**)&dsd, sizeof(sd)) != hipSuccess || hipMalloc((void**)&dbl, blocks.size() * sizeof(BlockRef)) != hipSuccess) rc = -1;
    if (rc == 0) {
      (void)hipMemcpy(dsd, &sd, sizeof(sd), hipMemcpyHostToDevice);
      (void)hipMemcpy(dbl, blocks.data(), blocks.size() * sizeof(BlockRef), hipMemcpyHostToDevice);
      rc = cvx_reduce_slabs((float*)workspace, dw, 1.0f, dsd, dbl, (int)blocks.size(), st);
    }
    (void)hipStreamSynchronize(st);
    (void)hipFree(dsd);
    (void)hipFree(dbl);
  }
  (void)hipStreamSynchronize(st);
  (void)hipFree(dt);
  return rc;
}

#ifdef CVX_TUNING
// Tuning build only (include/cvx_engine_experimental.h): device time of ONE weight-gradient launch (stride 1, pad k / 2) on operands with pixel
// pitches x_ld / dy_ld, mean over `reps` launches between two HIP events after one warm-up launch; nsplit <= 0: the streaming kernel's
// planner (1 when it does not take the shape).  The slabs are left in `workspace` (no reduction).
extern "C" int cvx_wgrad_time_unit(const void* x_f16, const void* dy_f16, int32_t batch, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t k,
                                   int32_t x_ld, int32_t dy_ld, int32_t nsplit, int32_t reps, void* workspace, int64_t workspace_bytes,
                                   float* us_out, int32_t* nsplit_out, void* hip_stream) {
  CVX_CHECK(x_f16 && dy_f16 && workspace && us_out && reps >= 1 && (k == 1 || k == 3), "bad arguments");
  hipStream_t st = (hipStream_t)hip_stream;
  std::vector<ConvTap> taps;
  for (int r = 0; r < k; ++r)
    for (int s = 0; s < k; ++s) taps.push_back(ConvTap{r - k / 2, s - k / 2, r * k + s, 0});
  ConvTap* dt = nullptr;
  CVX_TRY(make_taps(taps, &dt));
  WgradParams wp;
  memset(&wp, 0, sizeof(wp));
  wp.x = (const half_t*)x_f16;
  wp.x_bstride = (long long)h * w * x_ld;
  wp.x_ld = x_ld;
  wp.IH = wp.OH = h;
  wp.IW = wp.OW = w;
  wp.Cin = cin;
  wp.dy = (const half_t*)dy_f16;
  wp.dy_bstride = (long long)h * w * dy_ld;
  wp.dy_ld = dy_ld;
  wp.Cout = cout;
  wp.B = batch;
  wp.stride = 1;
  wp.ntaps = k * k;
  wp.taps = dt;
  wp.slabs = (float*)workspace;
  wp.cin_pad16 = round_up(cin, 16);
  wp.std3x3 = cvx_taps_std3x3(taps.data(), (int)taps.size());
  wp.nsplit = nsplit > 0 ? nsplit : (cvx_conv_wgrad_k3_supported(wp) ? cvx_conv_wgrad_k3_nsplit(wp) : cvx_conv_wgrad_stream_supported(wp) ? cvx_conv_wgrad_stream_nsplit(wp) : 1);
  if (nsplit_out) *nsplit_out = wp.nsplit;
  int rc = 0;
  if ((long long)wp.nsplit * cout * k * k * wp.cin_pad16 * 4 > workspace_bytes) {
    cvx_set_error("cvx_wgrad_time_unit: workspace too small");
    rc = -1;
  }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (rc == 0 && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) rc = -1;
  if (rc == 0) rc = cvx_conv_wgrad_launch(wp, st);
  if (rc == 0) {
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < reps && rc == 0; ++i) rc = cvx_conv_wgrad_launch(wp, st);
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) rc = -1;
    float ms = 0.f;
    if (rc == 0 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) *us_out = ms * 1e3f / reps;
  }
  (void)hipStreamSynchronize(st);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(dt);
  return rc;
}
#endif
