// Visformer meta-tuning step (meta_tuning_sun_m/train_meta.py:161-177): train-mode forward
// (batch-statistics BatchNorm with running-stat update, DropPath) with saved activations, and the full backward
// to every parameter gradient.  GEMM-shaped work (forward convs, data gradients, split-K weight gradients) runs on
// conv_gemm_v2; everything else is train_kernels.hip / attention_bwd.hip / head.hip.  This first version is
// correctness-first: nothing is fused and BatchNorm is never folded (its statistics depend on the batch).
#include "../../include/fsvit.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "train_kernels.h"

using namespace fsvit;

extern "C" const char* fsvit_last_error(void);
int fsvit_set_error(int code, const char* fmt, ...);      // engine.hip

#define T_TRY(expr)                                                                     \
  do {                                                                                  \
    int _rc = (expr);                                                                   \
    if (_rc != 0) return _rc > 0 ? fsvit_set_error(_rc, "%s: %s", #expr, hipGetErrorString((hipError_t)_rc)) : _rc; \
  } while (0)

// launches are skipped in the sizing pass (dry arenas hand out fake addresses)
#define T_RUN(expr)                                                                     \
  do {                                                                                  \
    if (!t->save.dry) {                                                                 \
      int _rc = (expr);                                                                 \
      if (_rc != 0) return _rc > 0 ? fsvit_set_error(_rc, "%s: %s", #expr, hipGetErrorString((hipError_t)_rc)) : _rc; \
    }                                                                                   \
  } while (0)

namespace {

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

struct Arena {
  unsigned char* base = nullptr;
  size_t size = 0, off = 0, peak = 0;
  bool dry = false;                       // dry run: only measure
  void* take(size_t bytes) {
    size_t o = off;
    off += align256(bytes);
    if (off > peak) peak = off;
    if (dry) return (void*)(uintptr_t)(o + 256);
    return off <= size ? base + o : nullptr;
  }
};

struct ConvSpec {                         // one nn.Conv2d of the model (PyTorch layout weight [O][Ig][KH][KW])
  std::string wname;
  int O = 0, Ig = 0, KH = 1, KW = 1, stride = 1, pad = 0, groups = 1;
  int hd_rows = 1, hdp_rows = 1, hd_cols = 1, hdp_cols = 1;      // head-dim padding (qkv rows / proj columns)
  bool via_patches = false;               // stem conv1 / downsample: input is the 32-wide im2col row
  int rows_fwd() const { return (O / groups) / hd_rows * hdp_rows; }                         // padded N per group
  int kreal() const { return KH * KW * Ig; }
  int kpad_cols() const { return via_patches ? 32 : kreal() / hd_cols * hdp_cols; }         // padded K (elements)
};

struct BnSave { void* z = nullptr; float *mean = nullptr, *invstd = nullptr, *sa = nullptr, *sb = nullptr; int M = 0, C = 0; };

}  // namespace

struct fsvit_visformer_trainer {
  fsvit_visformer_cfg cfg;
  int dtype = 0, es = 4;          // storage type of activations / activation gradients (FSVIT_F32 | FSVIT_BF16) and its element size
  int gdt = 0;                    // GEMM kernel type (conv_gemm.h): = dtype, or 2 for FSVIT_BF16X2 (fp32 storage, every GEMM as two-limb 16-bit MFMAs on limb-packed weights)
  int C0, C1, C2, C3, H0, H1, H2, H3, hid1, hid2, hid3, hd2, hdp2, hd3, hdp3, Cg;
  // ---- per-call state
  std::map<std::string, const fsvit_param*> P;
  hipStream_t st = nullptr;
  Arena save, tmp;
  int B = 0;
  float dp_rate = 0.f;
  bool freeze_bn = false;                 // BatchNorm layers in eval mode inside the step (utils.freeze_bn): running statistics, no update
  // batched weight pack: the sizing pass (dry arenas) records every pack of forward + backward in call order; the real forward runs them in
  // one or two launches and the real conv calls pick their packed weights up by position
  std::vector<PackJob> jobs;
  std::vector<size_t> job_off;
  size_t pack_bytes = 0, job_cursor = 0;
  unsigned char* pack_base = nullptr;
  const float* masks = nullptr;           // [n_droppath_calls][B] 0/1
  // saved forward state (pointers into `save`)
  struct Stem { void *patches, *z1, *a1, *zd, *ad, *z2, *a2, *z3, *a3; unsigned char* arg; BnSave b1, bd, b2, b3; void* x1; } stem;
  struct S1 { void *x, *xn, *z1, *h1, *z2, *h2; BnSave bn; const float* scale; void* out; };
  struct SA { void *x, *xn1, *qkv, *ctx, *xa, *xn2, *z1, *h; BnSave bn1, bn2; const float *s1, *s2; void* out; };
  struct PE { void *xin, *z; BnSave bn; void* out; };
  std::vector<S1> s1;
  std::vector<SA> s2, s3;
  PE pe2, pe3;
  BnSave bnf;
  float* scales = nullptr;                // [n_calls][B] = mask / keep
  const float* dtokens = nullptr;         // optional gradient of the post-norm token map for the next backward (distillation head)
  // weight-gradient split slabs of a backward pass stay in the `save` arena and are summed by ONE table-driven launch at the end of the pass
  std::vector<FinJob> fin;
  size_t save_after_forward = 0;
  // Side stream of the backward pass: the direct weight-gradient kernels (wgrad3x3 / wgrad1x1) have no consumer before the end of the pass, so they
  // run on a second HIP stream next to the data-gradient / BatchNorm chain (launch tails and HBM-bound elementwise passes overlap with them).
  // `pend` = the byte ranges pending side launches still READ; a main-stream launch that writes into one of them first waits for the side stream.
  // Two side streams (FSVIT_SIDE_STREAMS = 1 .. 4): side launch i goes to stream i % n (each stream in order), so that the weight gradients of two
  // layers overlap each other as well - with that, half as many row splits per launch (wgrad1x1_splits: 128 workgroups instead of 256) fill the
  // chip, and the split slabs written by the kernels and read by the finalize pass halve (2.2 GB each way per step): 13.69 -> 13.47 ms per
  // 800-image step on one box (1 x 256: 13.69, 2 x 256: 13.80, 2 x 128: 13.47, 3 x 128: 13.59, 3 x 96: 13.53, 4 x 128: 14.10, 4 x 64: 14.01).
  static constexpr int MAX_SIDE = 4;
  hipStream_t sides[MAX_SIDE] = {nullptr, nullptr, nullptr, nullptr};
  int n_side = 1;
  hipStream_t side = nullptr;             // the stream of the side launch being issued (side_begin)
  hipEvent_t ev_a = nullptr;
  std::vector<hipEvent_t> ev_done;        // one per side launch of a pass (recorded behind it; a side stream runs in order)
  int side_seq = 0;                       // side launches of this pass so far
  int side_waited[MAX_SIDE] = {-1, -1, -1, -1};       // per side stream: the youngest launch the main stream has already waited for
  // default: on (ViT / DeiT trainer: 19.2 -> 17.6 ms per 200-image DeiT-S step; Visformer trainer: nothing in round 3 - 16.55 vs 16.67 ms -, 14.26 -> 14.15 ms
  // once round 4 had taken HBM-bound passes out of the main stream); FSVIT_WGRAD_SIDE_STREAM=0 / 1 forces it
  bool side_on = false;
  struct Range { const unsigned char *lo, *hi; int seq; };
  std::vector<Range> pend;
  ~fsvit_visformer_trainer() {
    if (ev_a) (void)hipEventDestroy(ev_a);
    for (hipEvent_t e : ev_done) (void)hipEventDestroy(e);
    for (hipStream_t q : sides) if (q) (void)hipStreamDestroy(q);
  }
};

namespace {

typedef fsvit_visformer_trainer TR;

const fsvit_param* getp(TR* t, const std::string& name) {
  auto it = t->P.find(name);
  if (it == t->P.end()) { fsvit_set_error(FSVIT_ERR_KEY, "missing parameter: %s", name.c_str()); return nullptr; }
  return it->second;
}

ConvGemmParams gemm_params(const void* x, const void* w, void* y, int B, int H, int W, int Cin, int x_cstride, int KH, int KW, int stride, int pad,
                           int N, int y_cstride, int K, int Kw, int groups) {
  ConvGemmParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.w = w; p.y = y;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.x_cstride = x_cstride;
  p.OH = (H + 2 * pad - KH) / stride + 1; p.OW = (W + 2 * pad - KW) / stride + 1;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.N = N; p.y_cstride = y_cstride; p.K = K; p.Kw = Kw; p.M = B * p.OH * p.OW;
  p.groups = groups; p.act = ACT_NONE; p.log2Cin = ilog2(Cin);
  return p;
}

// packed weights of one layer: recorded in the sizing pass, served from the batch in the real pass (fallback: packed here and now)
int packed_weight(TR* t, const PackJob& job, size_t bytes, void** out) {
  if (t->tmp.dry) {
    t->jobs.push_back(job);
    t->job_off.push_back(t->pack_bytes);
    t->pack_bytes += align256(bytes);
    *out = (void*)(uintptr_t)256;
    return 0;
  }
  if (t->pack_base && t->job_cursor < t->jobs.size()) {
    const PackJob& r = t->jobs[t->job_cursor];
    if (r.w == job.w && r.mode == job.mode && r.rows_pad == job.rows_pad && r.Kw == job.Kw) {
      *out = t->pack_base + t->job_off[t->job_cursor++];
      return 0;
    }
  }
  void* pk = t->tmp.take(bytes);
  if (!pk) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (weights)");
  T_RUN(launch_pack_weight(job.w, pk, job.O, job.Ig, job.KH, job.KW, job.groups, job.mode, job.rows_pad, job.Kw, job.hd_rows, job.hdp_rows, job.hd_cols, job.hdp_cols,
                           t->gdt, t->st));
  *out = pk;
  return 0;
}
// runs every recorded pack (call after the sizing pass, with the real arenas in place)
int run_packs(TR* t) {
  t->pack_base = nullptr; t->job_cursor = 0;
  if (t->jobs.empty()) return 0;
  unsigned char* base = (unsigned char*)t->save.take(t->pack_bytes);
  if (!base) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (packed weights)");
  std::vector<PackJob> js = t->jobs;
  for (size_t i = 0; i < js.size(); ++i) js[i].out = base + t->job_off[i];
  T_RUN(launch_pack_weight_multi(js.data(), (int)js.size(), t->gdt, t->st));
  t->pack_base = base;
  return 0;
}

// ---------------------------------------------------------------- conv forward: z = conv(x) (+ bias)
// gp != nullptr: z = GELU(conv(x) + bias) and gp = the GELU's derivative at the pre-activation, both written by the GEMM's epilogue (conv_gemm.h y2)
// the forward pack of a layer: [groups][rows_fwd][Kw], row n = output channel, k = (tap, input channel)
int conv_pack_fwd(TR* t, const ConvSpec& c, void** pk, int* Kw_out = nullptr) {
  const fsvit_param* w = getp(t, c.wname);
  if (!w) return FSVIT_ERR_KEY;
  const int bke = 128 / t->es, Ng = c.rows_fwd(), K = c.kpad_cols(), Kw = round_up(K, bke);
  if (Kw_out) *Kw_out = Kw;
  return packed_weight(t, PackJob{w->data, nullptr, c.O, c.Ig, c.KH, c.KW, c.groups, 0, Ng, Kw, c.hd_rows, c.hdp_rows, c.hd_cols, c.hdp_cols},
                       (size_t)c.groups * Ng * Kw * t->es, pk);
}
// st_rows != nullptr: ask the layer's kernel for the BatchNorm statistics of z (ConvGemmParams::stats); *st_partial / *st_rows receive the partial
// sums (tmp arena) and their row count, 0 rows = this layer's kernel does not produce them (the caller runs the reduce pass)
int conv_fwd(TR* t, const ConvSpec& c, const void* x, int B, int H, int W, void* z, const float* bias, void* gp = nullptr, float** st_partial = nullptr,
             int* st_rows = nullptr) {
  if (st_rows) *st_rows = 0;
  const fsvit_param* w = getp(t, c.wname);
  if (!w) return FSVIT_ERR_KEY;
  const int bke = 128 / t->es, Ng = c.rows_fwd(), K = c.kpad_cols(), Kw = round_up(K, bke);
  void* pk = nullptr;
  T_TRY(packed_weight(t, PackJob{w->data, nullptr, c.O, c.Ig, c.KH, c.KW, c.groups, 0, Ng, Kw, c.hd_rows, c.hdp_rows, c.hd_cols, c.hdp_cols},
                      (size_t)c.groups * Ng * Kw * t->es, &pk));
  if (!bias && !c.via_patches && gconv3x3_supported(t->dtype, c.O, c.Ig, c.groups, c.KH, c.KW, c.stride, c.pad, W)) {
    T_RUN(launch_gconv3x3(x, pk, Kw, z, B, H, W, t->st, gp));      // wave = group, weights in registers (wgrad3x3.hip)
    return 0;
  }
  ConvGemmParams p;
  if (c.via_patches) p = gemm_params(x, pk, z, B, H, W, 32, 32, 1, 1, 1, 0, Ng, Ng, 32, Kw, 1);
  else p = gemm_params(x, pk, z, B, H, W, K / (c.KH * c.KW), c.groups * (K / (c.KH * c.KW)), c.KH, c.KW, c.stride, c.pad, Ng, c.groups * Ng, K, Kw, c.groups);
  p.bias = bias;
  if (gp) { p.act = ACT_GELU; p.y2 = gp; }
  if (st_rows) {
    static const bool off = [] { const char* e = getenv("FSVIT_BN_PRODUCER_STATS"); return e && e[0] == '0'; }();
    *st_rows = (off || t->freeze_bn) ? 0 : conv_stats_rows(p, t->gdt);
    if (*st_rows > 0) {
      *st_partial = (float*)t->tmp.take((size_t)*st_rows * 2 * p.y_cstride * 4);
      if (!*st_partial) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (producer statistics)");
      p.stats = *st_partial;
    }
  }
  T_RUN(launch_conv_gemm(p, t->gdt, t->st));
  return 0;
}

// ---------------------------------------------------------------- data gradient: dx = conv^T(dz)   (stride-1 convs and 1x1)
int side_guard(TR* t, const void* p, size_t bytes);       // (side stream of the backward pass, below)
// mul != nullptr: dx = conv^T(dz) * mul (the saved GELU derivative of the layer in front: its backward rides in this epilogue)
// the transposed ("dgrad") pack of a layer: [groups][Ig_pad rows][Kw], row = input channel, k = (flipped tap, output channel)
int conv_pack_bwd(TR* t, const ConvSpec& c, void** pk, int* Kw_out = nullptr) {
  const fsvit_param* w = getp(t, c.wname);
  if (!w) return FSVIT_ERR_KEY;
  const int bke = 128 / t->es, Ng_pad = c.rows_fwd(), Ig_pad = c.Ig / c.hd_cols * c.hdp_cols;
  const int K = c.KH * c.KW * Ng_pad, Kw = round_up(K, bke);
  if (Kw_out) *Kw_out = Kw;
  return packed_weight(t, PackJob{w->data, nullptr, c.O, c.Ig, c.KH, c.KW, c.groups, 1, Ig_pad, Kw, c.hd_cols, c.hdp_cols, c.hd_rows, c.hdp_rows},
                       (size_t)c.groups * Ig_pad * Kw * t->es, pk);
}
int conv_bwd_data(TR* t, const ConvSpec& c, const void* dz, int B, int OH, int OW, void* dx, const void* mul = nullptr) {
  const fsvit_param* w = getp(t, c.wname);
  if (!w) return FSVIT_ERR_KEY;
  const int bke = 128 / t->es;
  const int Ng_pad = c.rows_fwd();                        // dz channels per group (padded)
  const int Ig_pad = c.Ig / c.hd_cols * c.hdp_cols;       // dx channels per group (padded for proj's ctx input)
  const int K = c.KH * c.KW * Ng_pad, Kw = round_up(K, bke);
  // rows = input channels (padded like the forward K columns), K = output channels (padded like the forward rows)
  void* pk = nullptr;
  T_TRY(packed_weight(t, PackJob{w->data, nullptr, c.O, c.Ig, c.KH, c.KW, c.groups, 1, Ig_pad, Kw, c.hd_cols, c.hdp_cols, c.hd_rows, c.hdp_rows},
                      (size_t)c.groups * Ig_pad * Kw * t->es, &pk));
  T_TRY(side_guard(t, dx, (size_t)B * OH * OW * c.groups * Ig_pad * t->es));
  if (gconv3x3_supported(t->dtype, c.O, c.Ig, c.groups, c.KH, c.KW, c.stride, c.pad, OW) && Ng_pad == 32 && Ig_pad == 32) {
    T_RUN(launch_gconv3x3(dz, pk, Kw, dx, B, OH, OW, t->st, nullptr, mul));      // the same kernel on the transposed, tap-flipped pack
    return 0;
  }
  ConvGemmParams p = gemm_params(dz, pk, dx, B, OH, OW, Ng_pad, c.groups * Ng_pad, c.KH, c.KW, 1, c.pad, Ig_pad, c.groups * Ig_pad, K, Kw, c.groups);
  if (mul) { p.act = ACT_MUL; p.res = mul; }
  T_RUN(launch_conv_gemm(p, t->gdt, t->st));
  return 0;
}

// ---------------------------------------------------------------- side stream of the backward pass
int side_begin(TR* t) {                       // the side stream may start once everything queued on the main stream so far is done
  if (!t->sides[0]) {
    static const int n = [] { const char* e = getenv("FSVIT_SIDE_STREAMS"); const int v = e ? atoi(e) : 2; return v < 1 ? 1 : (v > TR::MAX_SIDE ? TR::MAX_SIDE : v); }();
    t->n_side = n;
    for (int i = 0; i < n; ++i) T_TRY((int)hipStreamCreateWithFlags(&t->sides[i], hipStreamNonBlocking));
    T_TRY((int)hipEventCreateWithFlags(&t->ev_a, hipEventDisableTiming));
  }
  t->side = t->sides[t->side_seq % t->n_side];
  T_TRY((int)hipEventRecord(t->ev_a, t->st));
  T_TRY((int)hipStreamWaitEvent(t->side, t->ev_a, 0));
  return 0;
}
// behind a side launch: its completion event, and the byte ranges it reads
int side_end(TR* t, const void* p0, size_t b0, const void* p1, size_t b1) {
  if ((int)t->ev_done.size() <= t->side_seq) {
    hipEvent_t e = nullptr;
    T_TRY((int)hipEventCreateWithFlags(&e, hipEventDisableTiming));
    t->ev_done.push_back(e);
  }
  T_TRY((int)hipEventRecord(t->ev_done[t->side_seq], t->side));
  t->pend.push_back(TR::Range{(const unsigned char*)p0, (const unsigned char*)p0 + b0, t->side_seq});
  t->pend.push_back(TR::Range{(const unsigned char*)p1, (const unsigned char*)p1 + b1, t->side_seq});
  ++t->side_seq;
  return 0;
}
// the main stream waits for the side launches up to `seq` (the side stream is in order) and forgets their ranges
int side_wait(TR* t, int seq) {
  for (int q = 0; q < t->n_side; ++q) {        // per side stream: its youngest launch <= seq (the stream runs in order), unless already waited for
    const int i = seq - ((seq - q) % t->n_side + t->n_side) % t->n_side;
    if (i >= 0 && i > t->side_waited[q]) {
      T_TRY((int)hipStreamWaitEvent(t->st, t->ev_done[i], 0));
      t->side_waited[q] = i;
    }
  }
  size_t k = 0;
  for (size_t i = 0; i < t->pend.size(); ++i)
    if (t->pend[i].seq > seq) t->pend[k++] = t->pend[i];
  t->pend.resize(k);
  return 0;
}
int side_sync(TR* t) {                        // ... for every side launch so far
  if (t->save.dry || t->side_seq == 0) { t->pend.clear(); return 0; }
  T_TRY(side_wait(t, t->side_seq - 1));
  t->pend.clear();
  return 0;
}
// error exit of a backward pass: nothing may still be running on a side stream when the caller sees the error (it may free or reuse the workspace the
// pending weight-gradient launches read)
template <typename T_>
int side_drain(T_* t, int rc) {
  if (rc != 0)
    for (hipStream_t q : t->sides)
      if (q) (void)hipStreamSynchronize(q);
  return rc;
}
// call before a main-stream launch that WRITES [p, p + bytes): waits for the youngest pending side launch that still reads any of it
int side_guard(TR* t, const void* p, size_t bytes) {
  const unsigned char *lo = (const unsigned char*)p, *hi = lo + bytes;
  int seq = -1;
  for (const TR::Range& r : t->pend)
    if (lo < r.hi && r.lo < hi && r.seq > seq) seq = r.seq;
  return seq >= 0 ? side_wait(t, seq) : 0;
}

// ---------------------------------------------------------------- weight gradient: dW = sum_m dz[m] (x) xcol[m]   (split-K GEMM over transposed operands)
// A grouped conv is computed as ONE dense GEMM over all channels (the cross-group blocks are discarded by the finalize
// pass): 8 x the useful FLOPs of the stage-1 3x3 conv, but one full-width launch instead of 8 quarter-empty ones.
int conv_bwd_weight(TR* t, const ConvSpec& c, const void* x, int B, int H, int W, const void* dz) {
  const fsvit_param* w = getp(t, c.wname);
  if (!w) return FSVIT_ERR_KEY;
  if (!w->grad && !t->save.dry) return 0;      // (the sizing pass always counts the weight-gradient scratch)
  const int OH = c.via_patches ? H : (H + 2 * c.pad - c.KH) / c.stride + 1, OW = c.via_patches ? W : (W + 2 * c.pad - c.KW) / c.stride + 1;
  const int M = B * OH * OW;
  const int rows = c.groups * c.rows_fwd();                                     // all output channels (padded)
  const int Cin_tot = c.via_patches ? 32 : c.groups * (c.kpad_cols() / (c.KH * c.KW));
  const int Kc_pad = c.via_patches ? 32 : round_up(c.KH * c.KW * Cin_tot, 4);
  const int wdt = t->gdt == 2 ? 2 : t->dtype;          // the direct weight-gradient kernels: 16-bit rows, or fp32 rows with two-limb arithmetic
  if (!c.via_patches && c.KH == 3 && c.KW == 3 && c.stride == 1 && c.pad == 1 && wgrad3x3_supported(wdt, c.O, c.Ig, c.groups, W)) {
    // direct kernel: no transposed copies of dz / im2col(x) (wgrad3x3.hip)
    float* scratch = (float*)t->save.take(wgrad3x3_scratch_bytes(c.O, c.Ig, c.groups, M, wdt));
    if (!scratch) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (wgrad3x3)");
    int d[2] = {0, 0};
    const bool on_side = t->side_on && !t->save.dry;
    if (on_side) T_TRY(side_begin(t));
    T_RUN(launch_wgrad3x3(x, Cin_tot, dz, rows, w->grad, scratch, B, H, W, c.O, c.Ig, c.groups, on_side ? t->side : t->st, d, wdt));
    if (on_side) T_TRY(side_end(t, x, (size_t)M * Cin_tot * t->es, dz, (size_t)M * rows * t->es));
    if (!t->save.dry) t->fin.push_back(FinJob{scratch, w->grad, 3, 0, c.Ig, 3, 3, c.groups == 8 ? 1 : 0, d[1], d[0], 1, 1, 1, 1});
    return 0;
  }
  if ((c.via_patches || (c.KH == 1 && c.KW == 1 && c.stride == 1 && c.groups == 1)) && wgrad1x1_supported(wdt, rows, Cin_tot)) {
    // direct kernel on the row-major operands (x [M][Cin_tot] - or the 32-wide patch rows - and dz [M][rows]); the finalize pass is the round-1 one
    const int splits = wgrad1x1_splits(rows, Cin_tot, M, wdt);
    float* ysp = (float*)t->save.take((size_t)round_up(rows, 4) * splits * Kc_pad * 4);
    if (!ysp) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (wgrad1x1)");
    const bool on_side = t->side_on && !t->save.dry;
    if (on_side) T_TRY(side_begin(t));
    T_RUN(launch_wgrad1x1(x, Cin_tot, Cin_tot, dz, rows, rows, ysp, M, Kc_pad, on_side ? t->side : t->st, wdt));
    if (on_side) T_TRY(side_end(t, x, (size_t)M * Cin_tot * t->es, dz, (size_t)M * rows * t->es));
    const int KHf = c.via_patches ? 3 : c.KH, KWf = c.via_patches ? 3 : c.KW;
    const bool fast = KHf == 1 && KWf == 1 && c.hd_rows == c.hdp_rows && c.hd_cols == c.hdp_cols && (c.Ig & 3) == 0 && (Kc_pad & 3) == 0;
    if (!t->save.dry) t->fin.push_back(FinJob{ysp, w->grad, fast ? 1 : 0, c.O, c.Ig, KHf, KWf, 0, splits, Kc_pad, c.hd_rows, c.hdp_rows, c.hd_cols, c.hdp_cols});
    return 0;
  }
  const int tiles = ((rows + 127) / 128) * ((Kc_pad + 127) / 128);
  int splits = (1024 + tiles - 1) / tiles;
  if (splits > 64) splits = 64;
  if (splits > M / 512) splits = M / 512;
  if (splits < 1) splits = 1;
  const int Ks = round_up((M + splits - 1) / splits, 64), Mpad = splits * Ks;
  float* ysp = (float*)t->save.take((size_t)round_up(rows, 4) * splits * Kc_pad * 4);
  const size_t tmp_mark = t->tmp.off;
  void* dzt = t->tmp.take((size_t)rows * Mpad * t->es);
  void* xct = t->tmp.take((size_t)Kc_pad * Mpad * t->es);
  if (!dzt || !xct || !ysp) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (wgrad)");
  T_TRY(side_guard(t, dzt, (size_t)rows * Mpad * t->es));
  T_TRY(side_guard(t, xct, (size_t)Kc_pad * Mpad * t->es));
  T_RUN(launch_transpose_cols(dz, dzt, M, rows, 0, rows, Mpad, t->dtype, t->st));
  if (c.via_patches) T_RUN(launch_transpose_cols(x, xct, M, 32, 0, 32, Mpad, t->gdt, t->st));           // (the GEMM's weight-side operand: limb words under bf16x2)
  else T_RUN(launch_im2col_t(x, xct, B, H, W, Cin_tot, 0, Cin_tot, c.KH, c.KW, c.stride, c.pad, OH, OW, Mpad, t->gdt, t->st));
  // Y[n][s*Kc_pad + k] = sum_{m in split s} dzt[n][m] * xct[k][m]
  ConvGemmParams p = gemm_params(dzt, xct, ysp, 1, rows, 1, Ks, Mpad, 1, 1, 1, 0, Kc_pad, splits * Kc_pad, Ks, Ks, splits);
  p.w_gstride = Ks; p.w_rstride = Mpad; p.out_f32 = 1;
  T_RUN(launch_conv_gemm(p, t->gdt, t->st));
  if (!t->save.dry) {
    if (c.groups > 1) t->fin.push_back(FinJob{ysp, w->grad, 2, c.O / c.groups, c.Ig, c.KH, c.KW, c.groups, splits, Kc_pad, 1, 1, 1, 1});
    else {
      const int KHf = c.via_patches ? 3 : c.KH, KWf = c.via_patches ? 3 : c.KW;
      const bool fast = KHf == 1 && KWf == 1 && c.hd_rows == c.hdp_rows && c.hd_cols == c.hdp_cols && (c.Ig & 3) == 0 && (Kc_pad & 3) == 0;
      t->fin.push_back(FinJob{ysp, w->grad, fast ? 1 : 0, c.O, c.Ig, KHf, KWf, 0, splits, Kc_pad, c.hd_rows, c.hdp_rows, c.hd_cols, c.hdp_cols});
    }
  }
  t->tmp.off = tmp_mark;
  return 0;
}

// every deferred split-slab finalize of this backward pass (one or two launches)
int run_finalizes(TR* t) {
  T_TRY(side_sync(t));
  if (t->save.dry || t->fin.empty()) return 0;
  T_RUN(launch_wgrad_finalize_multi(t->fin.data(), (int)t->fin.size(), t->st));
  t->fin.clear();
  return 0;
}

// ---------------------------------------------------------------- BatchNorm (train)
// A residual add queued by the block that produced z (z = add.a + add.scale[image] * add.b, not yet computed) rides in the reduce pass.
struct PendingAdd { const void* a = nullptr; const void* b = nullptr; const float* scale = nullptr; void* out = nullptr; size_t n = 0, per_img = 0; };
// pre / pre_rows: partial sums of z that the producing kernel already wrote (conv_fwd's st_partial / st_rows): no reduce pass
int bn_fwd(TR* t, const std::string& name, const void* z, int M, int C, int act, const void* res, void* y, BnSave* sv, PendingAdd* add = nullptr,
           const float* pre = nullptr, int pre_rows = 0) {
  const fsvit_param *g = getp(t, name + ".weight"), *b = getp(t, name + ".bias"), *rm = getp(t, name + ".running_mean"), *rv = getp(t, name + ".running_var");
  if (!g || !b || !rm || !rv) return FSVIT_ERR_KEY;
  float* stats = (float*)t->save.take((size_t)4 * C * 4);
  float* partial = (float*)t->tmp.take((size_t)bn_reduce_blocks(M) * 2 * C * 4);
  if (!stats || !partial) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (bn)");
  sv->z = const_cast<void*>(z); sv->M = M; sv->C = C;
  sv->mean = stats; sv->invstd = stats + C; sv->sa = stats + 2 * C; sv->sb = stats + 3 * C;
  const bool fuse_add = add && add->out && add->out == z && !t->freeze_bn;
  if (add && add->out && !fuse_add) {      // a queued add this BatchNorm cannot take (frozen statistics: no reduce pass)
    T_RUN(launch_add_scaled(add->a, add->b, add->scale, add->out, add->n, add->per_img, t->dtype, t->st));
  }
  if (t->freeze_bn) {
    T_RUN(launch_bn_frozen_coeffs(C, t->cfg.bn_eps, g->data, b->data, rm->data, rv->data, sv->mean, sv->invstd, sv->sa, sv->sb, t->st));
  } else {
    if (pre && pre_rows > 0 && !fuse_add) {
      T_RUN(launch_bn_fwd_finalize_nblk(pre, pre_rows, M, C, t->cfg.bn_eps, 0.1f, g->data, b->data, rm->data, rv->data, sv->mean, sv->invstd, sv->sa, sv->sb, t->st));
    } else {
      if (fuse_add) T_RUN(launch_bn_reduce(z, nullptr, nullptr, nullptr, partial, M, C, 0, t->dtype, t->st, add->a, add->b, add->scale, (int)(add->per_img / C)));
      else T_RUN(launch_bn_reduce(z, nullptr, nullptr, nullptr, partial, M, C, 0, t->dtype, t->st));
      T_RUN(launch_bn_fwd_finalize(partial, M, C, t->cfg.bn_eps, 0.1f, g->data, b->data, rm->data, rv->data, sv->mean, sv->invstd, sv->sa, sv->sb, t->st));
    }
  }
  if (add) *add = PendingAdd{};
  if (y) T_RUN(launch_bn_apply(z, sv->sa, sv->sb, res, y, (size_t)M, C, act, t->dtype, t->st));     // (y == nullptr: the caller applies sa / sb in a fused pass)
  return 0;
}

// dy: gradient w.r.t. the BN output (after undoing the activation) -> dz; writes dgamma / dbeta
// acc / scale2 / out2 / rows_per_img: the fused tail of launch_bn_bwd_apply (dz = acc + ..., out2 = scale2[image] * dz)
// act: dy is the gradient BEHIND the LeakyReLU that followed this BatchNorm (no residual); the slope is applied inside the reduce and apply passes
int bn_bwd(TR* t, const std::string& name, const BnSave& sv, const void* dy, void* dz, const void* acc = nullptr, const float* scale2 = nullptr,
           void* out2 = nullptr, size_t rows_per_img = 0, bool act = false) {
  const fsvit_param *g = getp(t, name + ".weight"), *b = getp(t, name + ".bias");
  if (!g || !b) return FSVIT_ERR_KEY;
  const size_t mark = t->tmp.off;
  float* partial = (float*)t->tmp.take((size_t)bn_reduce_blocks(sv.M) * 2 * sv.C * 4);
  float* coef = (float*)t->tmp.take((size_t)5 * sv.C * 4);
  if (!partial || !coef) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (bn bwd)");
  float* dgamma = g->grad ? g->grad : coef + 3 * sv.C;
  float* dbeta = b->grad ? b->grad : coef + 4 * sv.C;
  T_RUN(launch_bn_reduce(dy, sv.z, sv.mean, sv.invstd, partial, sv.M, sv.C, 1, t->dtype, t->st, nullptr, nullptr, nullptr, 0, act ? sv.sa : nullptr, act ? sv.sb : nullptr));
  T_RUN(launch_bn_bwd_finalize(partial, sv.M, sv.C, g->data, sv.invstd, dgamma, dbeta, coef, coef + sv.C, coef + 2 * sv.C, t->freeze_bn ? 1 : 0, t->st));
  T_TRY(side_guard(t, dz, (size_t)sv.M * sv.C * t->es));
  if (out2) T_TRY(side_guard(t, out2, (size_t)sv.M * sv.C * t->es));
  T_RUN(launch_bn_bwd_apply(dy, sv.z, sv.mean, sv.invstd, coef, coef + sv.C, coef + 2 * sv.C, dz, (size_t)sv.M, sv.C, t->dtype, t->st, acc, scale2, out2, rows_per_img,
                            act ? sv.sa : nullptr, act ? sv.sb : nullptr));
  t->tmp.off = mark;
  return 0;
}

void* take_act(TR* t, size_t elems) { return t->save.take(elems * t->es); }
void* take_tmp(TR* t, size_t elems) { return t->tmp.take(elems * t->es); }
#define NEED(ptr) do { if (!(ptr)) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small"); } while (0)

const float* dp_scale(TR* t, int call, int block_index, int nblocks) {
  // per-block rates linspace(0, rate, depth) (visformer.py:312); rate 0 -> plain residual
  const float r = nblocks > 1 ? t->dp_rate * (float)block_index / (float)(nblocks - 1) : 0.f;
  if (r == 0.f || !t->scales) return nullptr;
  return t->scales + (size_t)call * t->B;
}

// ================================================================ specs
struct Specs {
  ConvSpec conv1, down, conv2, conv3, pe2, pe3;
  std::vector<ConvSpec> s1c1, s1c2, s1c3;
  struct A { ConvSpec qkv, proj, fc1, fc2; };
  std::vector<A> s2, s3;
};

Specs make_specs(const TR* t) {
  Specs s;
  auto mk = [](std::string n, int O, int Ig, int k, int stride, int pad, int groups) { ConvSpec c; c.wname = n; c.O = O; c.Ig = Ig; c.KH = c.KW = k; c.stride = stride; c.pad = pad; c.groups = groups; return c; };
  s.conv1 = mk("stem.conv1.weight", t->C0, 3, 3, 2, 1, 1); s.conv1.via_patches = true;
  s.down = mk("stem.downsample.0.weight", t->C1, 3, 3, 2, 1, 1); s.down.via_patches = true;
  s.conv2 = mk("stem.conv2.weight", t->C1, t->C0, 3, 1, 1, 1);
  s.conv3 = mk("stem.conv3.weight", t->C1, t->C1, 3, 1, 1, 1);
  for (int i = 0; i < t->cfg.depth[0]; ++i) {
    const std::string p = "stage1." + std::to_string(i) + ".mlp.";
    s.s1c1.push_back(mk(p + "conv1.weight", t->hid1, t->C1, 1, 1, 0, 1));
    s.s1c2.push_back(mk(p + "conv2.weight", t->hid1, t->Cg, 3, 1, 1, t->cfg.group));
    s.s1c3.push_back(mk(p + "conv3.weight", t->C1, t->hid1, 1, 1, 0, 1));
  }
  s.pe2 = mk("patch_embed2.proj.weight", t->C2, t->C1, 2, 2, 0, 1);
  s.pe3 = mk("patch_embed3.proj.weight", t->C3, t->C2, 2, 2, 0, 1);
  for (int st = 2; st <= 3; ++st) {
    const int C = st == 2 ? t->C2 : t->C3, hid = st == 2 ? t->hid2 : t->hid3, hd = st == 2 ? t->hd2 : t->hd3, hdp = st == 2 ? t->hdp2 : t->hdp3;
    const int heads = t->cfg.num_heads;
    for (int i = 0; i < t->cfg.depth[st - 1]; ++i) {
      const std::string p = "stage" + std::to_string(st) + "." + std::to_string(i) + ".";
      Specs::A a;
      a.qkv = mk(p + "attn.qkv.weight", 3 * heads * hd, C, 1, 1, 0, 1); a.qkv.hd_rows = hd; a.qkv.hdp_rows = hdp;
      a.proj = mk(p + "attn.proj.weight", C, heads * hd, 1, 1, 0, 1); a.proj.hd_cols = hd; a.proj.hdp_cols = hdp;
      a.fc1 = mk(p + "mlp.conv1.weight", hid, C, 1, 1, 0, 1);
      a.fc2 = mk(p + "mlp.conv3.weight", C, hid, 1, 1, 0, 1);
      (st == 2 ? s.s2 : s.s3).push_back(a);
    }
  }
  return s;
}

// ================================================================ forward
int train_forward_impl(TR* t, const float* x, float* feat) {
  const Specs sp = make_specs(t);
  const int B = t->B, dt = t->dtype;
  hipStream_t st = t->st;
  const int img = t->cfg.img_size, H0 = t->H0, H1 = t->H1;
  const size_t M0 = (size_t)B * H0 * H0, M1 = (size_t)B * H1 * H1;
  const int nblk = t->cfg.depth[0] + t->cfg.depth[1] + t->cfg.depth[2];
  int dp_call = 0, blk = 0;

  // ---- stem (visformer.py:219-239)
  auto& S = t->stem;
  NEED(S.patches = take_act(t, M0 * 32)); NEED(S.z1 = take_act(t, M0 * t->C0)); NEED(S.a1 = take_act(t, M0 * t->C0));
  NEED(S.zd = take_act(t, M0 * t->C1)); S.ad = nullptr; NEED(S.z2 = take_act(t, M0 * t->C1)); NEED(S.a2 = take_act(t, M0 * t->C1));
  NEED(S.z3 = take_act(t, M0 * t->C1)); S.a3 = nullptr; NEED(S.arg = (unsigned char*)t->save.take(M1 * t->C1)); NEED(S.x1 = take_act(t, M1 * t->C1));
  T_RUN(launch_im2col27(x, S.patches, B, img, img, H0, H0, dt, st));
  {
    // (the producing GEMMs hand the BatchNorm statistics of the maps they store to the finalize: no reduce pass, ConvGemmParams::stats)
    const size_t mark = t->tmp.off;
    float* stp = nullptr;
    int str = 0;
    T_TRY(conv_fwd(t, sp.conv1, S.patches, B, H0, H0, S.z1, nullptr, nullptr, &stp, &str));
    T_TRY(bn_fwd(t, "stem.bn1", S.z1, (int)M0, t->C0, ACT_LRELU, nullptr, S.a1, &S.b1, nullptr, stp, str));
    t->tmp.off = mark;
    T_TRY(conv_fwd(t, sp.down, S.patches, B, H0, H0, S.zd, nullptr, nullptr, &stp, &str));
    T_TRY(bn_fwd(t, "stem.downsample.1", S.zd, (int)M0, t->C1, ACT_NONE, nullptr, nullptr, &S.bd, nullptr, stp, str));      // statistics only: applied inside the pooling pass
    t->tmp.off = mark;
  }
  {
    // conv2 / conv3 (conv3x3_halo) hand their BatchNorm the statistics of the map they store: no reduce pass over 328 MB
    const size_t mark = t->tmp.off;
    float* stp = nullptr;
    int str = 0;
    T_TRY(conv_fwd(t, sp.conv2, S.a1, B, H0, H0, S.z2, nullptr, nullptr, &stp, &str));
    T_TRY(bn_fwd(t, "stem.bn2", S.z2, (int)M0, t->C1, ACT_LRELU, nullptr, S.a2, &S.b2, nullptr, stp, str));
    t->tmp.off = mark;
    T_TRY(conv_fwd(t, sp.conv3, S.a2, B, H0, H0, S.z3, nullptr, nullptr, &stp, &str));
    T_TRY(bn_fwd(t, "stem.bn3", S.z3, (int)M0, t->C1, ACT_LRELU, nullptr, nullptr, &S.b3, nullptr, stp, str));      // statistics only: applied inside the pooling pass below
    t->tmp.off = mark;
  }
  {
    const fsvit_param* pos = getp(t, "pos_embed1");
    if (!pos) return FSVIT_ERR_KEY;
    // pos_embed is [1,C,H,W] in the checkpoint; transpose to [HW][C] in tmp
    float* pt = (float*)t->tmp.take((size_t)H1 * H1 * t->C1 * 4);
    NEED(pt);
    T_RUN(launch_transpose_cols(pos->data, pt, t->C1, H1 * H1, 0, H1 * H1, t->C1, 0, st));   // in [C][HW] -> out [HW][C]
    T_RUN(launch_bn_pool_fwd(S.z3, S.b3.sa, S.b3.sb, S.zd, pt, S.x1, S.arg, B, H1, H1, t->C1, dt, st, S.bd.sa, S.bd.sb));      // LeakyReLU(bn3(z3) + bn_d(zd)) -> MaxPool -> + pos
  }
  void* xcur = S.x1;
  // residual adds in front of a BatchNorm are queued here and computed by that BatchNorm's reduce pass (bn_fwd); flush_add() launches a queued
  // add on its own where the consumer is not a BatchNorm
  PendingAdd pend;
  auto flush_add = [&]() -> int {
    if (pend.out) T_RUN(launch_add_scaled(pend.a, pend.b, pend.scale, pend.out, pend.n, pend.per_img, dt, st));
    pend = PendingAdd{};
    return 0;
  };

  // ---- stage 1 (Block with attn_disabled: x + DropPath(mlp(norm2(x))), visformer.py:259-263)
  t->s1.resize(t->cfg.depth[0]);
  // Round 4: a block is ONE launch behind its BatchNorm's statistics (stage1_ring.hip MODE 3): the batch statistics are folded into conv1 as the eval
  // engine folds the running statistics (fold_prenorm), the residual with the DropPath scale is added in the same kernel - no apply pass, and the
  // reduce pass reads one finished map instead of adding two and storing a third.  (The statistics of the block's output from the same kernel were
  // built and measured: 16 more VGPRs than the kernel has, 25 spills, +95 us per launch.)  (FSVIT_STAGE1_BLOCK_FUSED / _TRAIN_FUSED: retired in round 6, dispatch_switches.r06.patch.)
  constexpr bool block_off = false;
  constexpr bool fused_off = false;
  const bool block_fused = !block_off && !fused_off && stage1_ring_supported(t->gdt, t->C1, t->hid1, t->cfg.group, H1);
  const size_t s1_mark = t->tmp.off;
  for (int i = 0; i < t->cfg.depth[0]; ++i, ++blk) {
    auto& b = t->s1[i];
    const std::string p = "stage1." + std::to_string(i) + ".";
    b.x = xcur;
    NEED(b.xn = take_act(t, M1 * t->C1)); NEED(b.z1 = take_act(t, M1 * t->hid1)); NEED(b.h1 = take_act(t, M1 * t->hid1));
    NEED(b.z2 = take_act(t, M1 * t->hid1)); NEED(b.h2 = take_act(t, M1 * t->hid1)); NEED(b.out = take_act(t, M1 * t->C1));
    const size_t mark = t->tmp.off;
    if (block_fused) {
      T_TRY(bn_fwd(t, p + "norm2.bn", b.x, (int)M1, t->C1, ACT_NONE, nullptr, nullptr, &b.bn));   // statistics only
      const fsvit_param* w1 = getp(t, sp.s1c1[i].wname);
      if (!w1) return FSVIT_ERR_KEY;
      void* w1f = t->tmp.take((size_t)t->hid1 * t->C1 * t->es); NEED(w1f);
      float* b1f = (float*)t->tmp.take((size_t)t->hid1 * 4); NEED(b1f);
      T_RUN(launch_fold_prenorm(w1->data, b.bn.sa, b.bn.sb, w1f, b1f, t->hid1, t->C1, t->C1, dt, st));
      void *pk2 = nullptr, *pk3 = nullptr;
      int kw2 = 0;
      T_TRY(conv_pack_fwd(t, sp.s1c2[i], &pk2, &kw2));
      T_TRY(conv_pack_fwd(t, sp.s1c3[i], &pk3));
      if (kw2 != 320) return fsvit_set_error(FSVIT_ERR_ARG, "stage-1 grouped conv pack: Kw %d", kw2);
      b.scale = dp_scale(t, dp_call, blk, nblk);
      if (t->dp_rate * blk > 0.f) ++dp_call;
      T_RUN(launch_stage1_ring_block_train(b.x, b.out, w1f, b1f, pk2, pk3, b.h1, b.z1, b.h2, b.z2, b.xn, b.bn.sa, b.bn.sb, b.scale, nullptr, B, H1, H1, st));
      t->tmp.off = mark;
      xcur = b.out;
      continue;
    }
    void* z3 = take_tmp(t, M1 * t->C1); NEED(z3);
    T_TRY(bn_fwd(t, p + "norm2.bn", b.x, (int)M1, t->C1, ACT_NONE, nullptr, b.xn, &b.bn, &pend));
    // (z1 / z2 hold the GELU DERIVATIVES at the pre-activations, written next to h1 / h2 by the conv epilogues)
    if (!fused_off && stage1_ring_supported(t->gdt, t->C1, t->hid1, t->cfg.group, H1)) {
      // the three GEMM launches of the Mlp and the round trips of its hidden maps in ONE kernel (stage1_ring.hip, training variant)
      void *pk1 = nullptr, *pk2 = nullptr, *pk3 = nullptr;
      int kw2 = 0;
      T_TRY(conv_pack_fwd(t, sp.s1c1[i], &pk1));
      T_TRY(conv_pack_fwd(t, sp.s1c2[i], &pk2, &kw2));
      T_TRY(conv_pack_fwd(t, sp.s1c3[i], &pk3));
      if (kw2 != 320) return fsvit_set_error(FSVIT_ERR_ARG, "stage-1 grouped conv pack: Kw %d", kw2);
      T_RUN(launch_stage1_ring_train(b.xn, z3, pk1, pk2, pk3, b.h1, b.z1, b.h2, b.z2, B, H1, H1, st));
    } else {
      T_TRY(conv_fwd(t, sp.s1c1[i], b.xn, B, H1, H1, b.h1, nullptr, b.z1));
      T_TRY(conv_fwd(t, sp.s1c2[i], b.h1, B, H1, H1, b.h2, nullptr, b.z2));
      T_TRY(conv_fwd(t, sp.s1c3[i], b.h2, B, H1, H1, z3, nullptr));
    }
    b.scale = dp_scale(t, dp_call, blk, nblk);
    if (t->dp_rate * blk > 0.f) ++dp_call;
    // b.out = b.x + scale * z3: queued for the next block's BatchNorm reduce pass (z3 must outlive this block's tmp scope: it is the first
    // allocation after `mark` in every block, and the next block's BatchNorm runs before anything else is written there)
    pend = PendingAdd{b.x, z3, b.scale, b.out, M1 * t->C1, (size_t)H1 * H1 * t->C1};
    if (i + 1 == t->cfg.depth[0]) T_TRY(flush_add());
    t->tmp.off = mark;
    xcur = b.out;
  }
  t->tmp.off = s1_mark;

  // ---- stages 2, 3
  for (int sg = 2; sg <= 3; ++sg) {
    const int C = sg == 2 ? t->C2 : t->C3, Hi = sg == 2 ? t->H1 : t->H2, Ho = sg == 2 ? t->H2 : t->H3;
    const int hid = sg == 2 ? t->hid2 : t->hid3, hd = sg == 2 ? t->hd2 : t->hd3, hdp = sg == 2 ? t->hdp2 : t->hdp3, heads = t->cfg.num_heads;
    const size_t M = (size_t)B * Ho * Ho;
    auto& pe = sg == 2 ? t->pe2 : t->pe3;
    const std::string pn = "patch_embed" + std::to_string(sg) + ".";
    pe.xin = xcur;
    NEED(pe.z = take_act(t, M * C)); NEED(pe.out = take_act(t, M * C));
    {
      const fsvit_param *bias = getp(t, pn + "proj.bias"), *pos = getp(t, "pos_embed" + std::to_string(sg));
      if (!bias || !pos) return FSVIT_ERR_KEY;
      const size_t mark = t->tmp.off;
      T_TRY(conv_fwd(t, sg == 2 ? sp.pe2 : sp.pe3, pe.xin, B, Hi, Hi, pe.z, bias->data));
      void* y = take_tmp(t, M * C); NEED(y);
      T_TRY(bn_fwd(t, pn + "norm.bn", pe.z, (int)M, C, ACT_NONE, nullptr, y, &pe.bn));
      float* pt = (float*)t->tmp.take((size_t)Ho * Ho * C * 4); NEED(pt);
      T_RUN(launch_transpose_cols(pos->data, pt, C, Ho * Ho, 0, Ho * Ho, C, 0, st));
      T_RUN(launch_bcast_add(y, pt, pe.out, B, (size_t)Ho * Ho * C, dt, st));
      t->tmp.off = mark;
    }
    xcur = pe.out;
    auto& blocks = sg == 2 ? t->s2 : t->s3;
    blocks.resize(t->cfg.depth[sg - 1]);
    const float scale = 1.0f / std::sqrt((float)hd);
    for (int i = 0; i < t->cfg.depth[sg - 1]; ++i, ++blk) {
      auto& b = blocks[i];
      const auto& cs = (sg == 2 ? sp.s2 : sp.s3)[i];
      const std::string p = "stage" + std::to_string(sg) + "." + std::to_string(i) + ".";
      b.x = xcur;
      NEED(b.xn1 = take_act(t, M * C)); NEED(b.qkv = take_act(t, M * 3 * heads * hdp)); NEED(b.ctx = take_act(t, M * heads * hdp));
      NEED(b.xa = take_act(t, M * C)); NEED(b.xn2 = take_act(t, M * C)); NEED(b.z1 = take_act(t, M * hid)); NEED(b.h = take_act(t, M * hid));
      NEED(b.out = take_act(t, M * C));
      const size_t mark = t->tmp.off;
      void* zp = take_tmp(t, M * C); NEED(zp);
      T_TRY(bn_fwd(t, p + "norm1.bn", b.x, (int)M, C, ACT_NONE, nullptr, b.xn1, &b.bn1, &pend));
      T_TRY(conv_fwd(t, cs.qkv, b.xn1, B, Ho, Ho, b.qkv, nullptr));
      T_RUN(launch_attention(b.qkv, b.ctx, B, Ho * Ho, heads, hdp, scale, t->gdt == 2 ? 2 : dt, st));
      T_TRY(conv_fwd(t, cs.proj, b.ctx, B, Ho, Ho, zp, nullptr));
      b.s1 = dp_scale(t, dp_call, blk, nblk);
      if (t->dp_rate * blk > 0.f) ++dp_call;
      pend = PendingAdd{b.x, zp, b.s1, b.xa, M * C, (size_t)Ho * Ho * C};                       // b.xa = b.x + s1 * zp inside norm2's reduce pass
      T_TRY(bn_fwd(t, p + "norm2.bn", b.xa, (int)M, C, ACT_NONE, nullptr, b.xn2, &b.bn2, &pend));
      T_TRY(conv_fwd(t, cs.fc1, b.xn2, B, Ho, Ho, b.h, nullptr, b.z1));                        // b.z1 = GELU'(fc1(xn2))
      T_TRY(conv_fwd(t, cs.fc2, b.h, B, Ho, Ho, zp, nullptr));
      b.s2 = dp_scale(t, dp_call, blk, nblk);
      if (t->dp_rate * blk > 0.f) ++dp_call;
      // b.out = b.xa + s2 * zp: queued for the next block's norm1 (zp is the first tmp allocation of every block of the stage and is
      // rewritten only by that block's proj conv, after norm1); the stage's last block adds now
      pend = PendingAdd{b.xa, zp, b.s2, b.out, M * C, (size_t)Ho * Ho * C};
      if (i + 1 == t->cfg.depth[sg - 1]) T_TRY(flush_add());
      t->tmp.off = mark;
      xcur = b.out;
    }
  }
  // ---- final norm + pool (visformer.py:455-462)
  const size_t M3 = (size_t)B * t->H3 * t->H3;
  // the normalised map is only ever averaged: feat = sa * mean_hw(x) + sb in the pooling pass (no apply pass, no stored map)
  T_TRY(bn_fwd(t, "norm.bn", xcur, (int)M3, t->C3, ACT_NONE, nullptr, nullptr, &t->bnf));
  T_RUN(launch_pool_affine(xcur, t->bnf.sa, t->bnf.sb, feat, B, t->H3 * t->H3, t->C3, dt, st));
  return 0;
}

// ================================================================ backward
int train_backward_impl(TR* t, const float* dfeat) {
  const Specs sp = make_specs(t);
  const int B = t->B, dt = t->dtype;
  hipStream_t st = t->st;
  const int H0 = t->H0, H1 = t->H1;
  const size_t M0 = (size_t)B * H0 * H0, M1 = (size_t)B * H1 * H1, M3 = (size_t)B * t->H3 * t->H3;
  // gradient of the residual stream, ping-pong in tmp
  void* dx = take_tmp(t, M3 * t->C3); NEED(dx);
  {
    void* dxn = take_tmp(t, M3 * t->C3); NEED(dxn);
    T_RUN(launch_avgpool_bwd(dfeat, dxn, B, t->H3 * t->H3, t->C3, dt, st));
    if (t->dtokens) {                      // the token map also feeds the per-token classifier (token_label.py:49-53)
      T_RUN(launch_add_f32_into(dxn, t->dtokens, M3 * t->C3, dt, st));
      t->dtokens = nullptr;
    }
    T_TRY(bn_bwd(t, "norm.bn", t->bnf, dxn, dx));
  }
  for (int sg = 3; sg >= 2; --sg) {
    const int Ci = sg == 2 ? t->C1 : t->C2, C = sg == 2 ? t->C2 : t->C3, Hi = sg == 2 ? t->H1 : t->H2, Ho = sg == 2 ? t->H2 : t->H3;
    const int hid = sg == 2 ? t->hid2 : t->hid3, hd = sg == 2 ? t->hd2 : t->hd3, hdp = sg == 2 ? t->hdp2 : t->hdp3, heads = t->cfg.num_heads;
    const size_t M = (size_t)B * Ho * Ho;
    auto& blocks = sg == 2 ? t->s2 : t->s3;
    const float scale = 1.0f / std::sqrt((float)hd);
    for (int i = (int)blocks.size() - 1; i >= 0; --i) {
      auto& b = blocks[i];
      const auto& cs = (sg == 2 ? sp.s2 : sp.s3)[i];
      const std::string p = "stage" + std::to_string(sg) + "." + std::to_string(i) + ".";
      const size_t mark = t->tmp.off;
      // mlp branch:  out = xa + s2 * fc2(gelu(fc1(bn2(xa))))
      // (dz2 sits at the same tmp offset in every block of the stage: the drop-path-scaled copy of dx each branch starts from is written by
      // the fused tail of the preceding BatchNorm backward, only the stage's first block launches the copy itself)
      void* dz2 = take_tmp(t, M * C); NEED(dz2);
      if (i == (int)blocks.size() - 1) { T_TRY(side_guard(t, dz2, M * C * t->es)); T_RUN(launch_add_scaled(nullptr, dx, b.s2, dz2, M * C, (size_t)Ho * Ho * C, dt, st)); }
      T_TRY(conv_bwd_weight(t, cs.fc2, b.h, B, Ho, Ho, dz2));
      // (round 5's row-wise Mlp kernel - forward and data gradient in one launch each - was a draw against these gemm256 launches for two rounds and is
      // retired: tools/probes/variants/mlp_train/)
      void* dh = take_tmp(t, M * hid); NEED(dh);
      T_TRY(conv_bwd_data(t, cs.fc2, dz2, B, Ho, Ho, dh, b.z1));                               // dh := dz1 (x GELU' in the epilogue)
      T_TRY(conv_bwd_weight(t, cs.fc1, b.xn2, B, Ho, Ho, dh));
      void* dxn2 = take_tmp(t, M * C); NEED(dxn2);
      T_TRY(conv_bwd_data(t, cs.fc1, dh, B, Ho, Ho, dxn2));
      // dx := d(xa) total = dx + norm2 backward;  attention branch:  xa = x + s1 * proj(attn(qkv(bn1(x)))):  dz2 := dzp = s1 * dx
      T_TRY(bn_bwd(t, p + "norm2.bn", b.bn2, dxn2, dx, dx, b.s1, dz2, (size_t)Ho * Ho));
      T_TRY(conv_bwd_weight(t, cs.proj, b.ctx, B, Ho, Ho, dz2));
      void* dctx = take_tmp(t, M * heads * hdp); NEED(dctx);
      T_TRY(conv_bwd_data(t, cs.proj, dz2, B, Ho, Ho, dctx));
      void* dqkv = take_tmp(t, M * 3 * heads * hdp); NEED(dqkv);
      T_TRY(side_guard(t, dqkv, M * 3 * heads * hdp * t->es));
      T_RUN(launch_attention_bwd(b.qkv, dctx, dqkv, B, Ho * Ho, heads, hd, hdp, scale, dt, st));
      T_TRY(conv_bwd_weight(t, cs.qkv, b.xn1, B, Ho, Ho, dqkv));
      T_TRY(conv_bwd_data(t, cs.qkv, dqkv, B, Ho, Ho, dxn2));                                 // dxn2 := d(xn1)
      // dx += norm1 backward; the next block (i - 1) of the stage starts from dz2 = s2' * dx
      if (i > 0) T_TRY(bn_bwd(t, p + "norm1.bn", b.bn1, dxn2, dx, dx, blocks[i - 1].s2, dz2, (size_t)Ho * Ho));
      else T_TRY(bn_bwd(t, p + "norm1.bn", b.bn1, dxn2, dx, dx));
      t->tmp.off = mark;
    }
    // patch embed:  out = bn(conv_k2s2(xin) + bias) + pos
    auto& pe = sg == 2 ? t->pe2 : t->pe3;
    const std::string pn = "patch_embed" + std::to_string(sg) + ".";
    const ConvSpec& pc = sg == 2 ? sp.pe2 : sp.pe3;
    const size_t Mi = (size_t)B * Hi * Hi;
    {
      const fsvit_param *bias = getp(t, pn + "proj.bias"), *pos = getp(t, "pos_embed" + std::to_string(sg));
      if (!bias || !pos) return FSVIT_ERR_KEY;
      if (pos->grad || t->save.dry) {
        float* ps = (float*)t->tmp.take((size_t)Ho * Ho * C * 4); NEED(ps);
        T_TRY(side_guard(t, ps, (size_t)Ho * Ho * C * 4));
        T_RUN(launch_batch_sum(dx, ps, B, (size_t)Ho * Ho * C, dt, st));
        T_RUN(launch_transpose_cols(ps, pos->grad, Ho * Ho, C, 0, C, Ho * Ho, 0, st));         // [HW][C] -> [C][HW]
      }
      void* dz = take_tmp(t, M * C); NEED(dz);
      T_TRY(bn_bwd(t, pn + "norm.bn", pe.bn, dx, dz));
      if (bias->grad || t->save.dry) {
        float* partial = (float*)t->tmp.take((size_t)bn_reduce_blocks((int)M) * 2 * C * 4); NEED(partial);
        T_TRY(side_guard(t, partial, (size_t)bn_reduce_blocks((int)M) * 2 * C * 4));
        T_RUN(launch_colsum(dz, partial, bias->grad, (int)M, C, dt, st));
      }
      T_TRY(conv_bwd_weight(t, pc, pe.xin, B, Hi, Hi, dz));
      // data gradient of the non-overlapping k2s2 conv: G[m][(ky,kx,c)] = dz[m] . W[:, c, ky, kx], scattered to the input grid
      const fsvit_param* w = getp(t, pc.wname);
      const int bke = 128 / t->es, Kw = round_up(C, bke);
      void* pk = nullptr;
      T_TRY(packed_weight(t, PackJob{w->data, nullptr, C, Ci, 2, 2, 1, 2, 4 * Ci, Kw, 1, 1, 1, 1}, (size_t)4 * Ci * Kw * t->es, &pk));
      void* G = take_tmp(t, M * 4 * Ci); NEED(G);
      ConvGemmParams p = gemm_params(dz, pk, G, B, Ho, Ho, C, C, 1, 1, 1, 0, 4 * Ci, 4 * Ci, C, Kw, 1);
      T_TRY(side_guard(t, G, M * 4 * Ci * t->es));
      T_RUN(launch_conv_gemm(p, t->gdt, st));
      void* dxi = take_tmp(t, Mi * Ci); NEED(dxi);
      T_TRY(side_guard(t, dxi, Mi * Ci * t->es));
      T_RUN(launch_unpatch2(G, dxi, B, Ho, Ho, Ci, dt, st));
      // the new residual-stream gradient lives at the start of tmp: move it there
      t->tmp.off = 0;
      dx = take_tmp(t, Mi * Ci);
      T_TRY(side_sync(t));                    // (the start of tmp held this stage's gradient buffers)
      T_RUN((int)hipMemcpyAsync(dx, dxi, Mi * Ci * t->es, hipMemcpyDeviceToDevice, st));
    }
  }
  // ---- stage 1
  for (int i = (int)t->s1.size() - 1; i >= 0; --i) {
    auto& b = t->s1[i];
    const std::string p = "stage1." + std::to_string(i) + ".";
    const size_t mark = t->tmp.off;
    void* dz3 = take_tmp(t, M1 * t->C1); NEED(dz3);
    if (i == (int)t->s1.size() - 1) { T_TRY(side_guard(t, dz3, M1 * t->C1 * t->es)); T_RUN(launch_add_scaled(nullptr, dx, b.scale, dz3, M1 * t->C1, (size_t)H1 * H1 * t->C1, dt, st)); }
    constexpr bool fused_off = false;
    void* dxn = dz3;                                                                           // d(xn): in place over dz3 on the three-launch route
    if (!fused_off && stage1_ring_supported(t->gdt, t->C1, t->hid1, t->cfg.group, H1)) {
      // the block's data-gradient chain dz3 -> dz2 -> dz1 -> d(xn) in ONE kernel (stage1_ring.hip MODE 2), then the three weight gradients
      void *pk3 = nullptr, *pk2 = nullptr, *pk1 = nullptr;
      int kw2 = 0;
      T_TRY(conv_pack_bwd(t, sp.s1c3[i], &pk3));
      T_TRY(conv_pack_bwd(t, sp.s1c2[i], &pk2, &kw2));
      T_TRY(conv_pack_bwd(t, sp.s1c1[i], &pk1));
      if (kw2 != 320) return fsvit_set_error(FSVIT_ERR_ARG, "stage-1 grouped conv dgrad pack: Kw %d", kw2);
      void* dh2 = take_tmp(t, M1 * t->hid1); NEED(dh2);
      void* dh1 = take_tmp(t, M1 * t->hid1); NEED(dh1);
      dxn = take_tmp(t, M1 * t->C1); NEED(dxn);
      T_TRY(side_guard(t, dh2, M1 * t->hid1 * t->es));
      T_TRY(side_guard(t, dh1, M1 * t->hid1 * t->es));
      T_TRY(side_guard(t, dxn, M1 * t->C1 * t->es));
      T_RUN(launch_stage1_ring_dgrad(dz3, dxn, pk3, pk2, pk1, b.z2, b.z1, dh2, dh1, B, H1, H1, st));
      T_TRY(conv_bwd_weight(t, sp.s1c3[i], b.h2, B, H1, H1, dz3));
      T_TRY(conv_bwd_weight(t, sp.s1c2[i], b.h1, B, H1, H1, dh2));
      T_TRY(conv_bwd_weight(t, sp.s1c1[i], b.xn, B, H1, H1, dh1));
    } else {
      T_TRY(conv_bwd_weight(t, sp.s1c3[i], b.h2, B, H1, H1, dz3));
      void* dh2 = take_tmp(t, M1 * t->hid1); NEED(dh2);
      T_TRY(conv_bwd_data(t, sp.s1c3[i], dz3, B, H1, H1, dh2, b.z2));                           // x GELU'(z2) in the epilogue
      T_TRY(conv_bwd_weight(t, sp.s1c2[i], b.h1, B, H1, H1, dh2));
      void* dh1 = take_tmp(t, M1 * t->hid1); NEED(dh1);
      T_TRY(conv_bwd_data(t, sp.s1c2[i], dh2, B, H1, H1, dh1, b.z1));                           // x GELU'(z1) in the epilogue
      T_TRY(conv_bwd_weight(t, sp.s1c1[i], b.xn, B, H1, H1, dh1));
      T_TRY(conv_bwd_data(t, sp.s1c1[i], dh1, B, H1, H1, dz3));                                  // dz3 := d(xn)
    }
    // dx += norm2 backward (read from d(xn)); the next block's dz3 = scale' * dx goes to dz3's buffer
    if (i > 0) T_TRY(bn_bwd(t, p + "norm2.bn", b.bn, dxn, dx, dx, t->s1[i - 1].scale, dz3, (size_t)H1 * H1));
    else T_TRY(bn_bwd(t, p + "norm2.bn", b.bn, dxn, dx, dx));
    t->tmp.off = mark;
  }
  // ---- stem
  {
    auto& S = t->stem;
    const fsvit_param* pos = getp(t, "pos_embed1");
    if (!pos) return FSVIT_ERR_KEY;
    if (pos->grad || t->save.dry) {
      float* ps = (float*)t->tmp.take((size_t)H1 * H1 * t->C1 * 4); NEED(ps);
      T_TRY(side_guard(t, ps, (size_t)H1 * H1 * t->C1 * 4));
      T_RUN(launch_batch_sum(dx, ps, B, (size_t)H1 * H1 * t->C1, dt, st));
      T_RUN(launch_transpose_cols(ps, pos->grad, H1 * H1, t->C1, 0, t->C1, H1 * H1, 0, st));
    }
    void* da3 = take_tmp(t, M0 * t->C1); NEED(da3);
    void* g3 = take_tmp(t, M0 * t->C1); NEED(g3);
    T_TRY(side_guard(t, da3, M0 * t->C1 * t->es));
    T_TRY(side_guard(t, g3, M0 * t->C1 * t->es));
    void* da2 = nullptr;
    if (pool_bn_bwd_supported(t->C1, dt)) {
      // bn3 and the identity path's BatchNorm see the same gradient - the pooled gradient routed to each window's arg-max with the LeakyReLU slope;
      // their reductions and apply passes are formed from the pooled gradient itself (train_kernels.hip pool_bn_bwd_*): da3 := dz3, g3 := dzd
      const fsvit_param *g3w = getp(t, "stem.bn3.weight"), *b3w = getp(t, "stem.bn3.bias"), *gdw = getp(t, "stem.downsample.1.weight"), *bdw = getp(t, "stem.downsample.1.bias");
      if (!g3w || !b3w || !gdw || !bdw) return FSVIT_ERR_KEY;
      const int C = t->C1, nb = pool_bn_bwd_blocks(B, H1, H1, C, dt);
      float* part = (float*)t->tmp.take((size_t)nb * 4 * C * 4); NEED(part);
      float* coef = (float*)t->tmp.take((size_t)10 * C * 4); NEED(coef);
      float *coef3 = coef, *coefd = coef + 3 * C, *scr = coef + 6 * C;
      T_RUN(launch_pool_bn_bwd_reduce(dx, S.arg, S.z3, S.zd, S.b3.mean, S.b3.invstd, S.bd.mean, S.bd.invstd, part, part + (size_t)nb * 2 * C, B, H1, H1, C, dt, st));
      T_RUN(launch_bn_bwd_finalize_nblk(part, nb, (int)M0, C, g3w->data, S.b3.invstd, g3w->grad ? g3w->grad : scr, b3w->grad ? b3w->grad : scr + C, coef3, coef3 + C,
                                        coef3 + 2 * C, t->freeze_bn ? 1 : 0, st));
      T_RUN(launch_bn_bwd_finalize_nblk(part + (size_t)nb * 2 * C, nb, (int)M0, C, gdw->data, S.bd.invstd, gdw->grad ? gdw->grad : scr + 2 * C, bdw->grad ? bdw->grad : scr + 3 * C,
                                        coefd, coefd + C, coefd + 2 * C, t->freeze_bn ? 1 : 0, st));
      T_RUN(launch_pool_bn_bwd_apply(dx, S.arg, S.z3, S.zd, S.b3.mean, S.b3.invstd, S.bd.mean, S.bd.invstd, coef3, coefd, da3, g3, B, H1, H1, C, dt, st));
      T_TRY(conv_bwd_weight(t, sp.conv3, S.a2, B, H0, H0, da3));
      da2 = take_tmp(t, M0 * t->C1); NEED(da2);
      T_TRY(conv_bwd_data(t, sp.conv3, da3, B, H0, H0, da2));
      T_TRY(conv_bwd_weight(t, sp.down, S.patches, B, H0, H0, g3));
    } else {
      T_RUN(launch_pool_act_bwd(dx, S.arg, g3, B, H1, H1, t->C1, dt, st));                      // gradient at (bn3(z3) + identity): max-pool routing x LeakyReLU slope
      T_TRY(bn_bwd(t, "stem.bn3", S.b3, g3, da3));                                              // da3 := dz3
      T_TRY(conv_bwd_weight(t, sp.conv3, S.a2, B, H0, H0, da3));
      da2 = take_tmp(t, M0 * t->C1); NEED(da2);
      T_TRY(conv_bwd_data(t, sp.conv3, da3, B, H0, H0, da2));
      // identity path: ad = bn_d(zd)
      T_TRY(bn_bwd(t, "stem.downsample.1", S.bd, g3, da3));                                     // da3 := dzd
      T_TRY(conv_bwd_weight(t, sp.down, S.patches, B, H0, H0, da3));
    }
    // bn2 / bn1 are followed by a LeakyReLU: its slope rides in the BatchNorm backward's two passes (no bn_act_bwd pass, no 328 MB map)
    T_TRY(bn_bwd(t, "stem.bn2", S.b2, da2, da2, nullptr, nullptr, nullptr, 0, true));         // da2 := dz2 (in place)
    T_TRY(conv_bwd_weight(t, sp.conv2, S.a1, B, H0, H0, da2));
    void* da1 = take_tmp(t, M0 * t->C0); NEED(da1);
    T_TRY(conv_bwd_data(t, sp.conv2, da2, B, H0, H0, da1));
    T_TRY(bn_bwd(t, "stem.bn1", S.b1, da1, da1, nullptr, nullptr, nullptr, 0, true));         // da1 := dz1 (in place)
    T_TRY(conv_bwd_weight(t, sp.conv1, S.patches, B, H0, H0, da1));
  }
  return run_finalizes(t);
}

}  // namespace

// ================================================================ ViT / DeiT trainer (deit.py:61-78 Block, :139-218 VisionTransformer)
// Same machinery as the Visformer trainer (parameter table, two arenas, conv_fwd / conv_bwd_data / conv_bwd_weight on 1x1 "convs" = nn.Linear,
// attention forward / backward, GELU, DropPath scales); new: LayerNorm with kept row statistics, token assembly, the final norm on the cls row.
struct fsvit_vit_trainer : fsvit_visformer_trainer {
  fsvit_vit_cfg vcfg;
  int D = 0, S = 0, np = 0, npw = 0, K = 0, Kp = 0, hidv = 0, heads = 0, hd = 0, hdp = 0;
  struct Blk { void *x, *xn1, *qkv, *ctx, *x1, *xn2, *z1, *h; float *m1, *r1, *m2, *r2; const float *s1, *s2; };
  std::vector<Blk> blk;
  void* patches = nullptr;
  void* xlast = nullptr;
  float *mf = nullptr, *rf = nullptr;
};

namespace {

typedef fsvit_vit_trainer VT;

ConvSpec lin(const std::string& w, int O, int I) { ConvSpec c; c.wname = w; c.O = O; c.Ig = I; return c; }

struct VSpecs { ConvSpec pe; std::vector<ConvSpec> qkv, proj, fc1, fc2; };
VSpecs vit_specs(const VT* t) {
  VSpecs s;
  s.pe = lin("patch_embed.proj.weight", t->D, t->K);
  s.pe.hd_cols = t->K; s.pe.hdp_cols = t->Kp;                      // the patch rows are padded to Kp columns (one "head" of K real ones)
  for (int i = 0; i < t->vcfg.depth; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    ConvSpec q = lin(p + "attn.qkv.weight", 3 * t->heads * t->hd, t->D);
    q.hd_rows = t->hd; q.hdp_rows = t->hdp;
    ConvSpec pr = lin(p + "attn.proj.weight", t->D, t->heads * t->hd);
    pr.hd_cols = t->hd; pr.hdp_cols = t->hdp;
    s.qkv.push_back(q); s.proj.push_back(pr);
    s.fc1.push_back(lin(p + "mlp.fc1.weight", t->hidv, t->D));
    s.fc2.push_back(lin(p + "mlp.fc2.weight", t->D, t->hidv));
  }
  return s;
}

// bias of a Linear whose output rows are head-padded (qkv): the padded bias vector lives in tmp
int padded_bias(VT* t, const fsvit_param* b, int n_real, int hd, int hdp, const float** out) {
  if (hd == hdp) { *out = b->data; return 0; }
  const int groups = n_real / hd;
  float* pb = (float*)t->tmp.take((size_t)groups * hdp * 4);
  if (!pb) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (bias)");
  if (!t->tmp.dry) {
    hipError_t e = hipMemsetAsync(pb, 0, (size_t)groups * hdp * 4, t->st);
    if (e == hipSuccess) e = hipMemcpy2DAsync(pb, (size_t)hdp * 4, b->data, (size_t)hd * 4, (size_t)hd * 4, groups, hipMemcpyDeviceToDevice, t->st);
    if (e != hipSuccess) return fsvit_set_error((int)e, "padded bias");
  }
  *out = pb;
  return 0;
}
// bias gradient = column sums of dz (real columns only when head-padded)
int bias_grad(VT* t, const fsvit_param* b, const void* dz, int M, int n_real, int hd, int hdp) {
  if (!b->grad && !t->tmp.dry) return 0;
  const int cols = n_real / hd * hdp;
  const size_t mark = t->tmp.off;
  float* partial = (float*)t->tmp.take((size_t)bn_reduce_blocks(M) * 2 * cols * 4);
  float* full = hd == hdp ? b->grad : (float*)t->tmp.take((size_t)cols * 4);
  if (!partial || (!full && !t->tmp.dry)) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (bias grad)");
  T_RUN(launch_colsum(dz, partial, full, M, cols, t->dtype, t->st));
  if (hd != hdp && !t->tmp.dry) {
    hipError_t e = hipMemcpy2DAsync(b->grad, (size_t)hd * 4, full, (size_t)hdp * 4, (size_t)hd * 4, n_real / hd, hipMemcpyDeviceToDevice, t->st);
    if (e != hipSuccess) return fsvit_set_error((int)e, "bias grad");
  }
  t->tmp.off = mark;
  return 0;
}

int vit_ln_fwd(VT* t, const std::string& name, const void* x, void* y, float** mean, float** rstd, int M) {
  const fsvit_param *g = getp(t, name + ".weight"), *b = getp(t, name + ".bias");
  if (!g || !b) return FSVIT_ERR_KEY;
  *mean = (float*)t->save.take((size_t)M * 4);
  *rstd = (float*)t->save.take((size_t)M * 4);
  if (!*mean || !*rstd) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (ln)");
  T_RUN(launch_ln_train_fwd(x, g->data, b->data, y, *mean, *rstd, M, t->D, t->vcfg.ln_eps, t->dtype, t->st));
  return 0;
}
// dx = add + LN^T(dy)
int vit_ln_bwd(VT* t, const std::string& name, const void* dy, const void* x, const float* mean, const float* rstd, const void* add, void* dx, int M) {
  const fsvit_param *g = getp(t, name + ".weight"), *b = getp(t, name + ".bias");
  if (!g || !b) return FSVIT_ERR_KEY;
  const size_t mark = t->tmp.off;
  float* partial = (float*)t->tmp.take((size_t)ln_bwd_blocks(M) * 2 * t->D * 4);
  if (!partial) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small (ln bwd)");
  T_RUN(launch_ln_bwd(dy, x, mean, rstd, g->data, add, dx, partial, g->grad, b->grad, M, t->D, t->dtype, t->st));
  t->tmp.off = mark;
  return 0;
}

const float* vit_dp_scale(VT* t, int call, int i) {
  const int depth = t->vcfg.depth;
  const float r = depth > 1 ? t->dp_rate * (float)i / (float)(depth - 1) : 0.f;          // deit.py:161 linspace(0, rate, depth)
  if (r == 0.f || !t->scales) return nullptr;
  return t->scales + (size_t)call * t->B;
}

int vit_forward_impl(VT* t, const float* x, float* feat) {
  const VSpecs sp = vit_specs(t);
  const int B = t->B, S = t->S, D = t->D, dt = t->dtype, heads = t->heads, hdp = t->hdp, hid = t->hidv;
  const size_t M = (size_t)B * S, Mp = (size_t)B * t->np;
  hipStream_t st = t->st;
  const fsvit_param *cls = getp(t, "cls_token"), *pos = getp(t, "pos_embed"), *peb = getp(t, "patch_embed.proj.bias");
  if (!cls || !pos || !peb) return FSVIT_ERR_KEY;
  NEED(t->patches = take_act(t, Mp * t->Kp));
  T_RUN(launch_patchify(x, t->patches, B, t->vcfg.img_size, t->vcfg.patch_size, t->Kp, dt, st));
  void* xcur = take_act(t, M * D); NEED(xcur);
  {
    const size_t mark = t->tmp.off;
    void* zpe = take_tmp(t, Mp * D); NEED(zpe);
    T_TRY(conv_fwd(t, sp.pe, t->patches, (int)Mp, 1, 1, zpe, peb->data));
    T_RUN(launch_vit_assemble(zpe, cls->data, pos->data, xcur, B, S, D, dt, st));
    t->tmp.off = mark;
  }
  t->blk.resize(t->vcfg.depth);
  const float scale = 1.0f / std::sqrt((float)t->hd);
  int dp_call = 0;
  for (int i = 0; i < t->vcfg.depth; ++i) {
    auto& b = t->blk[i];
    const std::string p = "blocks." + std::to_string(i) + ".";
    const fsvit_param *bq = getp(t, p + "attn.qkv.bias"), *bp = getp(t, p + "attn.proj.bias"), *b1 = getp(t, p + "mlp.fc1.bias"), *b2 = getp(t, p + "mlp.fc2.bias");
    if (!bq || !bp || !b1 || !b2) return FSVIT_ERR_KEY;
    b.x = xcur;
    NEED(b.xn1 = take_act(t, M * D)); NEED(b.qkv = take_act(t, M * 3 * heads * hdp)); NEED(b.ctx = take_act(t, M * heads * hdp));
    NEED(b.x1 = take_act(t, M * D)); NEED(b.xn2 = take_act(t, M * D)); NEED(b.z1 = take_act(t, M * hid)); NEED(b.h = take_act(t, M * hid));
    void* xout = take_act(t, M * D); NEED(xout);
    const size_t mark = t->tmp.off;
    void* zp = take_tmp(t, M * D); NEED(zp);
    T_TRY(vit_ln_fwd(t, p + "norm1", b.x, b.xn1, &b.m1, &b.r1, (int)M));
    const float* bqp = nullptr;
    T_TRY(padded_bias(t, bq, 3 * heads * t->hd, t->hd, hdp, &bqp));
    T_TRY(conv_fwd(t, sp.qkv[i], b.xn1, B, S, 1, b.qkv, bqp));
    T_RUN(launch_attention(b.qkv, b.ctx, B, S, heads, hdp, scale, t->gdt == 2 ? 2 : dt, st));
    T_TRY(conv_fwd(t, sp.proj[i], b.ctx, B, S, 1, zp, bp->data));
    b.s1 = vit_dp_scale(t, dp_call, i);
    if (t->dp_rate * i > 0.f) ++dp_call;
    T_RUN(launch_add_scaled(b.x, zp, b.s1, b.x1, M * D, (size_t)S * D, dt, st));
    T_TRY(vit_ln_fwd(t, p + "norm2", b.x1, b.xn2, &b.m2, &b.r2, (int)M));
    T_TRY(conv_fwd(t, sp.fc1[i], b.xn2, B, S, 1, b.h, b1->data, b.z1));                       // b.z1 = GELU'(fc1(xn2) + b1)
    T_TRY(conv_fwd(t, sp.fc2[i], b.h, B, S, 1, zp, b2->data));
    b.s2 = vit_dp_scale(t, dp_call, i);
    if (t->dp_rate * i > 0.f) ++dp_call;
    T_RUN(launch_add_scaled(b.x1, zp, b.s2, xout, M * D, (size_t)S * D, dt, st));
    t->tmp.off = mark;
    xcur = xout;
  }
  t->xlast = xcur;
  const fsvit_param *ng = getp(t, "norm.weight"), *nb = getp(t, "norm.bias");
  if (!ng || !nb) return FSVIT_ERR_KEY;
  t->mf = (float*)t->save.take((size_t)B * 4); t->rf = (float*)t->save.take((size_t)B * 4);
  NEED(t->mf); NEED(t->rf);
  T_RUN(launch_vit_cls_ln_fwd(t->xlast, ng->data, nb->data, feat, t->mf, t->rf, B, S, D, t->vcfg.ln_eps, dt, st));
  return 0;
}

int vit_backward_impl(VT* t, const float* dfeat) {
  const VSpecs sp = vit_specs(t);
  const int B = t->B, S = t->S, D = t->D, dt = t->dtype, heads = t->heads, hdp = t->hdp, hid = t->hidv;
  const size_t M = (size_t)B * S, Mp = (size_t)B * t->np;
  hipStream_t st = t->st;
  const fsvit_param *ng = getp(t, "norm.weight"), *nb = getp(t, "norm.bias");
  if (!ng || !nb) return FSVIT_ERR_KEY;
  void* dx = take_tmp(t, M * D); NEED(dx);
  {
    if (!t->tmp.dry) { hipError_t e = hipMemsetAsync(dx, 0, M * D * t->es, st); if (e != hipSuccess) return fsvit_set_error((int)e, "memset"); }
    const size_t mark = t->tmp.off;
    float* partial = (float*)t->tmp.take((size_t)B * 2 * D * 4); NEED(partial);
    T_RUN(launch_vit_cls_ln_bwd(dfeat, t->xlast, t->mf, t->rf, ng->data, dx, partial, ng->grad, nb->grad, B, S, D, dt, st));
    t->tmp.off = mark;
  }
  const float scale = 1.0f / std::sqrt((float)t->hd);
  for (int i = t->vcfg.depth - 1; i >= 0; --i) {
    auto& b = t->blk[i];
    const std::string p = "blocks." + std::to_string(i) + ".";
    const fsvit_param *bq = getp(t, p + "attn.qkv.bias"), *bp = getp(t, p + "attn.proj.bias"), *b1 = getp(t, p + "mlp.fc1.bias"), *b2 = getp(t, p + "mlp.fc2.bias");
    if (!bq || !bp || !b1 || !b2) return FSVIT_ERR_KEY;
    const size_t mark = t->tmp.off;
    // mlp branch: out = x1 + s2 * fc2(gelu(fc1(norm2(x1))))
    void* dz2 = take_tmp(t, M * D); NEED(dz2);
    T_TRY(side_guard(t, dz2, M * D * t->es));
    T_RUN(launch_add_scaled(nullptr, dx, b.s2, dz2, M * D, (size_t)S * D, dt, st));
    T_TRY(conv_bwd_weight(t, sp.fc2[i], b.h, B, S, 1, dz2));
    T_TRY(bias_grad(t, b2, dz2, (int)M, D, 1, 1));
    void* dh = take_tmp(t, M * hid); NEED(dh);
    T_TRY(conv_bwd_data(t, sp.fc2[i], dz2, B, S, 1, dh, b.z1));
    T_TRY(conv_bwd_weight(t, sp.fc1[i], b.xn2, B, S, 1, dh));
    T_TRY(bias_grad(t, b1, dh, (int)M, hid, 1, 1));
    void* dxn = take_tmp(t, M * D); NEED(dxn);
    T_TRY(conv_bwd_data(t, sp.fc1[i], dh, B, S, 1, dxn));
    T_TRY(vit_ln_bwd(t, p + "norm2", dxn, b.x1, b.m2, b.r2, dx, dx, (int)M));                // dx := d(x1) total
    // attention branch: x1 = x + s1 * proj(attn(qkv(norm1(x))))
    T_TRY(side_guard(t, dz2, M * D * t->es));
    T_RUN(launch_add_scaled(nullptr, dx, b.s1, dz2, M * D, (size_t)S * D, dt, st));
    T_TRY(conv_bwd_weight(t, sp.proj[i], b.ctx, B, S, 1, dz2));
    T_TRY(bias_grad(t, bp, dz2, (int)M, D, 1, 1));
    void* dctx = take_tmp(t, M * heads * hdp); NEED(dctx);
    T_TRY(conv_bwd_data(t, sp.proj[i], dz2, B, S, 1, dctx));
    void* dqkv = take_tmp(t, M * 3 * heads * hdp); NEED(dqkv);
    T_TRY(side_guard(t, dqkv, M * 3 * heads * hdp * t->es));
    T_RUN(launch_attention_bwd(b.qkv, dctx, dqkv, B, S, heads, t->hd, hdp, scale, dt, st));
    T_TRY(conv_bwd_weight(t, sp.qkv[i], b.xn1, B, S, 1, dqkv));
    T_TRY(bias_grad(t, bq, dqkv, (int)M, 3 * heads * t->hd, t->hd, hdp));
    T_TRY(conv_bwd_data(t, sp.qkv[i], dqkv, B, S, 1, dxn));
    T_TRY(vit_ln_bwd(t, p + "norm1", dxn, b.x, b.m1, b.r1, dx, dx, (int)M));
    t->tmp.off = mark;
  }
  // embedding: tokens = [cls + pos0 | conv(patches) + bias + pos]
  const fsvit_param *cls = getp(t, "cls_token"), *pos = getp(t, "pos_embed"), *peb = getp(t, "patch_embed.proj.bias");
  if (!cls || !pos || !peb) return FSVIT_ERR_KEY;
  {
    float* ps = (float*)t->tmp.take((size_t)S * D * 4); NEED(ps);
    T_RUN(launch_batch_sum(dx, ps, B, (size_t)S * D, dt, st));
    if (!t->tmp.dry) {
      hipError_t e = hipSuccess;
      if (pos->grad) e = hipMemcpyAsync(pos->grad, ps, (size_t)S * D * 4, hipMemcpyDeviceToDevice, st);
      if (e == hipSuccess && cls->grad) e = hipMemcpyAsync(cls->grad, ps, (size_t)D * 4, hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return fsvit_set_error((int)e, "pos / cls gradient");
    }
    void* dzpe = take_tmp(t, Mp * D); NEED(dzpe);
    T_TRY(side_sync(t));
    T_RUN(launch_vit_patch_rows(dx, dzpe, B, S, D, dt, st));
    T_TRY(conv_bwd_weight(t, sp.pe, t->patches, (int)Mp, 1, 1, dzpe));
    T_TRY(bias_grad(t, peb, dzpe, (int)Mp, D, 1, 1));
  }
  return run_finalizes(t);
}

int vit_droppath_calls(const VT* t, float rate, std::vector<float>* keep) {
  int n = 0;
  for (int i = 0; i < t->vcfg.depth; ++i) {
    const float r = t->vcfg.depth > 1 ? rate * (float)i / (float)(t->vcfg.depth - 1) : 0.f;
    if (r == 0.f) continue;
    for (int k = 0; k < 2; ++k) if (keep) keep->push_back(1.0f - r);
    n += 2;
  }
  return n;
}

int vit_size_workspace(VT* t, int n_img, float rate, size_t* save_bytes, size_t* tmp_bytes) {
  t->B = n_img; t->dp_rate = rate; t->scales = nullptr;
  t->save = Arena(); t->tmp = Arena();
  t->save.dry = t->tmp.dry = true;
  t->jobs.clear(); t->job_off.clear(); t->pack_bytes = 0; t->pack_base = nullptr; t->job_cursor = 0;
  int rc = vit_forward_impl(t, nullptr, nullptr);
  if (rc) return rc;
  size_t tp = t->tmp.peak;
  t->tmp.off = 0;
  rc = vit_backward_impl(t, nullptr);
  if (rc) return rc;
  if (t->tmp.peak > tp) tp = t->tmp.peak;
  *save_bytes = align256(t->save.peak + t->pack_bytes + 256 + (size_t)vit_droppath_calls(t, rate, nullptr) * n_img * 4 + 256);
  *tmp_bytes = align256(tp);
  return 0;
}

}  // namespace

static bool side_stream_default(bool dflt);
extern "C" int fsvit_vit_trainer_create(const fsvit_vit_cfg* cfg, int dtype, fsvit_vit_trainer** out) {
  if (!cfg || !out) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  if (dtype != FSVIT_F32 && dtype != FSVIT_BF16 && dtype != FSVIT_BF16X2) return fsvit_set_error(FSVIT_ERR_ARG, "trainer dtype %d (FSVIT_F32 | FSVIT_BF16 | FSVIT_BF16X2)", dtype);
  if (cfg->embed_dim % cfg->num_heads || cfg->img_size % cfg->patch_size || cfg->embed_dim % 8) return fsvit_set_error(FSVIT_ERR_ARG, "bad ViT configuration");
  VT* t = new VT();
  t->vcfg = *cfg; t->gdt = dtype == FSVIT_BF16X2 ? 2 : dtype; t->dtype = dtype == FSVIT_BF16X2 ? FSVIT_F32 : dtype; t->es = t->dtype == FSVIT_F32 ? 4 : 2;
  const int kch = 64 / t->es;
  t->D = cfg->embed_dim; t->npw = cfg->img_size / cfg->patch_size; t->np = t->npw * t->npw; t->S = t->np + 1;
  t->K = 3 * cfg->patch_size * cfg->patch_size; t->Kp = round_up(t->K, 128 / t->es);
  t->hidv = (int)(cfg->embed_dim * cfg->mlp_ratio); t->heads = cfg->num_heads; t->hd = cfg->embed_dim / cfg->num_heads; t->hdp = round_up(t->hd, kch);
  t->side_on = side_stream_default(true);
  *out = t;
  return 0;
}
extern "C" void fsvit_vit_trainer_destroy(fsvit_vit_trainer* t) { delete t; }
extern "C" int fsvit_vit_trainer_droppath_calls(const fsvit_vit_trainer* t, float drop_path_rate) { return t ? vit_droppath_calls(t, drop_path_rate, nullptr) : 0; }

extern "C" size_t fsvit_vit_trainer_workspace_bytes(fsvit_vit_trainer* t, const fsvit_param* params, int n_params, int n_img, float drop_path_rate) {
  if (!t || !params || n_img <= 0) return 0;
  t->P.clear();
  for (int i = 0; i < n_params; ++i) t->P[params[i].name] = &params[i];
  size_t sb = 0, tb = 0;
  if (vit_size_workspace(t, n_img, drop_path_rate, &sb, &tb)) return 0;
  return sb + tb;
}

extern "C" int fsvit_vit_train_forward(fsvit_vit_trainer* t, const fsvit_param* params, int n_params, const float* x_nchw_dev, int n_img, int img_h, int img_w,
                                       float drop_path_rate, const float* masks_dev, float* feat_dev, void* ws_dev, size_t ws_bytes, void* stream) {
  if (!t || !params || !x_nchw_dev || !feat_dev || !ws_dev || n_img <= 0) return fsvit_set_error(FSVIT_ERR_ARG, "bad argument");
  if (img_h != t->vcfg.img_size || img_w != t->vcfg.img_size)
    return fsvit_set_error(FSVIT_ERR_IMG_SIZE, "Input image size (%d*%d) doesn't match model (%d*%d).", img_h, img_w, t->vcfg.img_size, t->vcfg.img_size);
  if (drop_path_rate > 0.f && !masks_dev) return fsvit_set_error(FSVIT_ERR_ARG, "DropPath masks required when drop_path_rate > 0");
  t->P.clear();
  for (int i = 0; i < n_params; ++i) t->P[params[i].name] = &params[i];
  size_t sb = 0, tb = 0;
  T_TRY(vit_size_workspace(t, n_img, drop_path_rate, &sb, &tb));
  if (ws_bytes < sb + tb) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace of %zu bytes is too small (need %zu)", ws_bytes, sb + tb);
  t->st = (hipStream_t)stream;
  t->save = Arena(); t->save.base = (unsigned char*)ws_dev; t->save.size = sb;
  t->tmp = Arena(); t->tmp.base = (unsigned char*)ws_dev + sb; t->tmp.size = ws_bytes - sb;
  t->scales = nullptr;
  if (drop_path_rate > 0.f) {                       // DropPath scale = mask / keep_prob per call (timm DropPath, deit.py:70,76-77)
    std::vector<float> keep;
    const int ncalls = vit_droppath_calls(t, drop_path_rate, &keep);
    t->scales = (float*)t->save.take((size_t)ncalls * n_img * 4);
    if (!t->scales) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small");
    T_RUN(launch_droppath_scales(masks_dev, t->scales, ncalls, n_img, keep.data(), t->st));
  }
  T_TRY(run_packs(t));
  const int rc_f = vit_forward_impl(t, x_nchw_dev, feat_dev);
  t->save_after_forward = t->save.off;
  return rc_f;
}

extern "C" int fsvit_vit_train_backward(fsvit_vit_trainer* t, const fsvit_param* params, int n_params, const float* dfeat_dev, void* stream) {
  if (!t || !params || !dfeat_dev) return fsvit_set_error(FSVIT_ERR_ARG, "bad argument");
  if (!t->save.base || t->save.dry) return fsvit_set_error(FSVIT_ERR_ARG, "train_backward called without a preceding train_forward");
  t->P.clear();
  for (int i = 0; i < n_params; ++i) t->P[params[i].name] = &params[i];
  t->st = (hipStream_t)stream;
  t->tmp.off = 0;
  t->save.off = t->save_after_forward;            // the pass's weight-gradient slabs are appended to the saved activations
  t->fin.clear();
  t->side_seq = 0; t->pend.clear();
  for (int& v : t->side_waited) v = -1;
  return side_drain(t, vit_backward_impl(t, dfeat_dev));
}

// ================================================================ C ABI
extern "C" int fsvit_visformer_trainer_create(const fsvit_visformer_cfg* cfg, int dtype, fsvit_visformer_trainer** out) {
  if (!cfg || !out) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  if (dtype != FSVIT_F32 && dtype != FSVIT_BF16 && dtype != FSVIT_BF16X2) return fsvit_set_error(FSVIT_ERR_ARG, "trainer dtype %d (FSVIT_F32 | FSVIT_BF16 | FSVIT_BF16X2)", dtype);
  TR* t = new TR();
  t->cfg = *cfg; t->gdt = dtype == FSVIT_BF16X2 ? 2 : dtype; t->dtype = dtype == FSVIT_BF16X2 ? FSVIT_F32 : dtype; t->es = t->dtype == FSVIT_F32 ? 4 : 2;
  const int D = cfg->embed_dim, kch = 64 / t->es;
  t->C0 = cfg->init_channels; t->C1 = D / 2; t->C2 = D; t->C3 = D * 2;
  t->H0 = cfg->img_size / 2; t->H1 = cfg->img_size / 4; t->H2 = cfg->img_size / 8; t->H3 = cfg->img_size / 16;
  t->hid1 = t->C1 * 2; t->hid2 = (int)(t->C2 * cfg->mlp_ratio); t->hid3 = (int)(t->C3 * cfg->mlp_ratio);
  t->hd2 = t->C2 / cfg->num_heads; t->hd3 = t->C3 / cfg->num_heads;
  t->hdp2 = round_up(t->hd2, kch); t->hdp3 = round_up(t->hd3, kch);
  t->Cg = t->hid1 / cfg->group;
  t->side_on = side_stream_default(true);      // round 4: the direct weight-gradient launches beside the data-gradient chain: 14.26 -> 14.15 ms per 800-image step (same box)
  *out = t;
  return 0;
}

static bool side_stream_default(bool dflt) {
  const char* e = getenv("FSVIT_WGRAD_SIDE_STREAM");
  return e ? e[0] != '0' : dflt;
}
extern "C" void fsvit_visformer_trainer_destroy(fsvit_visformer_trainer* t) { delete t; }

static void bind_params(TR* t, const fsvit_param* params, int n) {
  t->P.clear();
  for (int i = 0; i < n; ++i) t->P[params[i].name] = &params[i];
}

static int droppath_calls(const TR* t, float rate, std::vector<float>* keep) {
  const int depth = t->cfg.depth[0] + t->cfg.depth[1] + t->cfg.depth[2];
  int ncalls = 0;
  for (int b = 0; b < depth; ++b) {
    const float r = depth > 1 ? rate * (float)b / (float)(depth - 1) : 0.f;
    if (r == 0.f) continue;
    const int calls = b < t->cfg.depth[0] ? 1 : 2;
    for (int k = 0; k < calls; ++k) if (keep) keep->push_back(1.0f - r);
    ncalls += calls;
  }
  return ncalls;
}

// Sizing pass: walk forward + backward with dry arenas (no launches) and record both peaks.
static int size_workspace(TR* t, int n_img, float rate, size_t* save_bytes, size_t* tmp_bytes) {
  t->B = n_img; t->dp_rate = rate; t->scales = nullptr;
  t->save = Arena(); t->tmp = Arena();
  t->save.dry = t->tmp.dry = true;
  t->jobs.clear(); t->job_off.clear(); t->pack_bytes = 0; t->pack_base = nullptr; t->job_cursor = 0;
  int rc = train_forward_impl(t, nullptr, nullptr);
  if (rc) return rc;
  size_t tp = t->tmp.peak;
  t->tmp.off = 0;
  rc = train_backward_impl(t, nullptr);
  if (rc) return rc;
  if (t->tmp.peak > tp) tp = t->tmp.peak;
  *save_bytes = align256(t->save.peak + t->pack_bytes + 256 + (size_t)droppath_calls(t, rate, nullptr) * n_img * 4 + 256);
  *tmp_bytes = align256(tp);
  return 0;
}

extern "C" size_t fsvit_visformer_trainer_workspace_bytes(fsvit_visformer_trainer* t, const fsvit_param* params, int n_params, int n_img, float drop_path_rate) {
  if (!t || !params || n_img <= 0) return 0;
  bind_params(t, params, n_params);
  size_t sb = 0, tb = 0;
  if (size_workspace(t, n_img, drop_path_rate, &sb, &tb)) return 0;
  return sb + tb;
}

extern "C" int fsvit_visformer_train_forward(fsvit_visformer_trainer* t, const fsvit_param* params, int n_params, const float* x_nchw_dev, int n_img,
                                             int img_h, int img_w, float drop_path_rate, const float* masks_dev, float* feat_dev, void* ws_dev,
                                             size_t ws_bytes, void* stream) {
  if (!t || !params || !x_nchw_dev || !feat_dev || !ws_dev || n_img <= 0) return fsvit_set_error(FSVIT_ERR_ARG, "bad argument");
  if (img_h != t->cfg.img_size || img_w != t->cfg.img_size)
    return fsvit_set_error(FSVIT_ERR_IMG_SIZE, "Input image size (%d*%d) does not match model (%d*%d).", img_h, img_w, t->cfg.img_size, t->cfg.img_size);
  {   // nn.BatchNorm2d in train mode needs more than one value per channel; the smallest map is the final H3 x H3 one (B = 1 at 80x80 is valid)
    const int h3 = t->cfg.img_size / 16;
    if ((long)n_img * h3 * h3 < 2) return fsvit_set_error(FSVIT_ERR_ARG, "Expected more than 1 value per channel when training (BatchNorm)");
  }
  if (drop_path_rate > 0.f && !masks_dev) return fsvit_set_error(FSVIT_ERR_ARG, "DropPath masks required when drop_path_rate > 0");
  bind_params(t, params, n_params);
  size_t sb = 0, tb = 0;
  T_TRY(size_workspace(t, n_img, drop_path_rate, &sb, &tb));
  if (ws_bytes < sb + tb) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace of %zu bytes is too small (need %zu)", ws_bytes, sb + tb);
  t->st = (hipStream_t)stream;
  t->save = Arena(); t->save.base = (unsigned char*)ws_dev; t->save.size = sb;
  t->tmp = Arena(); t->tmp.base = (unsigned char*)ws_dev + sb; t->tmp.size = ws_bytes - sb;
  t->scales = nullptr;
  if (drop_path_rate > 0.f) {                       // DropPath scale = mask / keep_prob per call (visformer.py:92-96)
    std::vector<float> keep;
    const int ncalls = droppath_calls(t, drop_path_rate, &keep);
    t->scales = (float*)t->save.take((size_t)ncalls * n_img * 4);
    if (!t->scales) return fsvit_set_error(FSVIT_ERR_WORKSPACE, "training workspace too small");
    T_RUN(launch_droppath_scales(masks_dev, t->scales, ncalls, n_img, keep.data(), t->st));
  }
  T_TRY(run_packs(t));                              // every weight pack of this step (forward + data-gradient layouts) in one or two launches
  const int rc_f = train_forward_impl(t, x_nchw_dev, feat_dev);
  t->save_after_forward = t->save.off;
  return rc_f;
}

extern "C" int fsvit_visformer_train_backward(fsvit_visformer_trainer* t, const fsvit_param* params, int n_params, const float* dfeat_dev, void* stream) {
  if (!t || !params || !dfeat_dev) return fsvit_set_error(FSVIT_ERR_ARG, "bad argument");
  if (!t->save.base || t->save.dry) return fsvit_set_error(FSVIT_ERR_ARG, "train_backward called without a preceding train_forward");
  bind_params(t, params, n_params);
  t->st = (hipStream_t)stream;
  t->tmp.off = 0;
  t->save.off = t->save_after_forward;            // the pass's weight-gradient slabs are appended to the saved activations
  t->fin.clear();
  t->side_seq = 0; t->pend.clear();
  for (int& v : t->side_waited) v = -1;
  return side_drain(t, train_backward_impl(t, dfeat_dev));
}

extern "C" int fsvit_proto_head_backward(const float* feat_shot, const float* feat_query, const float* dlogits, int E, int way, int shot, int Q, int D,
                                         float temp, float* dfeat_shot, float* dfeat_query, float* dtemp_per_episode, void* stream) {
  if (!feat_shot || !feat_query || !dlogits || !dfeat_shot || !dfeat_query) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  int rc = launch_proto_head_bwd(feat_shot, feat_query, dlogits, E, way, shot, Q, D, temp, dfeat_shot, dfeat_query, dtemp_per_episode, (hipStream_t)stream);
  return rc ? fsvit_set_error(rc, "proto_head_bwd") : 0;
}

extern "C" int fsvit_proto_head_backward_sqr(const float* feat_shot, const float* feat_query, const float* dlogits, int E, int way, int shot, int Q, int D,
                                             float temp, float* dfeat_shot, float* dfeat_query, float* dtemp_per_episode, void* stream) {
  if (!feat_shot || !feat_query || !dlogits || !dfeat_shot || !dfeat_query) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  int rc = launch_proto_head_sqr_bwd(feat_shot, feat_query, dlogits, E, way, shot, Q, D, temp, dfeat_shot, dfeat_query, dtemp_per_episode, (hipStream_t)stream);
  return rc ? fsvit_set_error(rc, "proto_head_sqr_bwd") : 0;
}

extern "C" int fsvit_proto_head_backward_devtemp(const float* feat_shot, const float* feat_query, const float* dlogits, int E, int way, int shot, int Q, int D,
                                                 const float* temp_dev, int method, float* dfeat_shot, float* dfeat_query, float* dtemp_per_episode, void* stream) {
  if (!feat_shot || !feat_query || !dlogits || !dfeat_shot || !dfeat_query || !temp_dev) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  if (method != FSVIT_HEAD_COS && method != FSVIT_HEAD_SQR) return fsvit_set_error(FSVIT_ERR_ARG, "head backward: method 'cos' or 'sqr'");
  int rc = method == FSVIT_HEAD_COS
               ? launch_proto_head_bwd(feat_shot, feat_query, dlogits, E, way, shot, Q, D, 0.f, dfeat_shot, dfeat_query, dtemp_per_episode, (hipStream_t)stream, temp_dev)
               : launch_proto_head_sqr_bwd(feat_shot, feat_query, dlogits, E, way, shot, Q, D, 0.f, dfeat_shot, dfeat_query, dtemp_per_episode, (hipStream_t)stream, temp_dev);
  return rc ? fsvit_set_error(rc, "proto_head_bwd") : 0;
}

extern "C" int fsvit_proto_head_ce(const float* feat_shot, const float* feat_query, const long long* labels, int E, int way, int shot, int Q, int D, float temp,
                                   const float* temp_dev, int method, float* logits, float* dlogits, float* acc_per_episode, float* loss_per_episode,
                                   float* loss_acc_mean, unsigned* ticket, void* stream) {
  if (!feat_shot || !feat_query || !logits || !acc_per_episode || !loss_per_episode) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  if (method != FSVIT_HEAD_COS && method != FSVIT_HEAD_SQR && method != FSVIT_HEAD_DOT) return fsvit_set_error(FSVIT_ERR_ARG, "unknown head method %d", method);
  if ((loss_acc_mean != nullptr) != (ticket != nullptr)) return fsvit_set_error(FSVIT_ERR_ARG, "fsvit_proto_head_ce: loss_acc_mean needs the ticket word (and the reverse)");
  int rc = launch_proto_head_ce(feat_shot, feat_query, labels, E, way, shot, Q, D, temp_dev ? 0.f : temp, method, logits, acc_per_episode, loss_per_episode, dlogits,
                                loss_acc_mean, ticket, (hipStream_t)stream, temp_dev);
  return rc ? fsvit_set_error(rc, "proto_head_ce") : 0;
}

extern "C" int fsvit_proto_head_ce_backward(const float* feat_shot, const float* feat_query, const float* dlogits, const float* dloss_dev, int E, int way, int shot,
                                            int Q, int D, float temp, const float* temp_dev, int method, float* dfeat_shot, float* dfeat_query, float* dtemp,
                                            unsigned* ticket, void* stream) {
  if (!feat_shot || !feat_query || !dlogits || !dfeat_shot || !dfeat_query) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  if (method != FSVIT_HEAD_COS && method != FSVIT_HEAD_SQR) return fsvit_set_error(FSVIT_ERR_ARG, "head backward: method 'cos' or 'sqr'");
  if (ticket && !dtemp) return fsvit_set_error(FSVIT_ERR_ARG, "fsvit_proto_head_ce_backward: the ticket sums dtemp[0 .. E) into dtemp[E]");
  int rc = method == FSVIT_HEAD_COS ? launch_proto_head_bwd(feat_shot, feat_query, dlogits, E, way, shot, Q, D, temp_dev ? 0.f : temp, dfeat_shot, dfeat_query, dtemp,
                                                            (hipStream_t)stream, temp_dev, dloss_dev, ticket)
                                    : launch_proto_head_sqr_bwd(feat_shot, feat_query, dlogits, E, way, shot, Q, D, temp_dev ? 0.f : temp, dfeat_shot, dfeat_query,
                                                                dtemp, (hipStream_t)stream, temp_dev, dloss_dev, ticket);
  return rc ? fsvit_set_error(rc, "proto_head_ce_backward") : 0;
}

extern "C" int fsvit_attention_backward(const void* qkv, const void* dctx, void* dqkv, int B, int S, int heads, int hd, int hdp, float scale, int dtype,
                                        void* stream) {
  if (!qkv || !dctx || !dqkv) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  int rc = launch_attention_bwd(qkv, dctx, dqkv, B, S, heads, hd, hdp, scale, dtype, (hipStream_t)stream);
  return rc ? fsvit_set_error(rc, "attention_bwd") : 0;
}

extern "C" int fsvit_visformer_trainer_set_freeze_bn(fsvit_visformer_trainer* t, int on) {
  if (!t) return fsvit_set_error(FSVIT_ERR_ARG, "null trainer");
  t->freeze_bn = on != 0;
  return 0;
}

extern "C" int fsvit_sgd_step_multi(const void* items_dev, int n_items, size_t max_numel, float lr, float momentum, float weight_decay, int first_step,
                                    void* stream) {
  if (!items_dev && n_items > 0) return fsvit_set_error(FSVIT_ERR_ARG, "fsvit_sgd_step_multi: null table");
  int rc = launch_sgd_multi(items_dev, n_items, max_numel, lr, momentum, weight_decay, first_step, (hipStream_t)stream);
  return rc ? fsvit_set_error(rc, "sgd_multi") : 0;
}

extern "C" int fsvit_sgd_step(float* param, const float* grad, float* momentum_buf, size_t n, float lr, float momentum, float weight_decay, int first_step,
                              void* stream) {
  if (!param || !grad || !momentum_buf) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  int rc = launch_sgd(param, grad, momentum_buf, n, lr, momentum, weight_decay, first_step, (hipStream_t)stream);
  return rc ? fsvit_set_error(rc, "sgd") : 0;
}

// ---- post-norm token map of the last train_forward (the `x` of `return x, pooled`, sun_meta_training/models/visformer.py:464) and its gradient
extern "C" int fsvit_visformer_train_tokens(fsvit_visformer_trainer* t, float* tokens_dev, void* stream) {
  if (!t || !tokens_dev) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  if (!t->bnf.z || !t->save.base || t->save.dry) return fsvit_set_error(FSVIT_ERR_ARG, "train_tokens called without a preceding train_forward");
  const size_t n = (size_t)t->B * t->H3 * t->H3 * t->C3;
  // (the final BatchNorm's scale / shift are applied here: the training forward keeps no normalised copy of the stage-3 map)
  int rc = fsvit::launch_tokens_to_f32(t->bnf.z, t->bnf.sa, t->bnf.sb, tokens_dev, n, t->C3, t->dtype, (hipStream_t)stream);
  if (rc != 0) return fsvit_set_error(FSVIT_ERR_ARG, "launch failed: %d", rc);
  return 0;
}
extern "C" int fsvit_visformer_train_set_token_grad(fsvit_visformer_trainer* t, const float* dtokens_dev) {
  if (!t) return fsvit_set_error(FSVIT_ERR_ARG, "null argument");
  t->dtokens = dtokens_dev;               // consumed (and cleared) by the next fsvit_visformer_train_backward; NULL: none
  return 0;
}
