// fsvit engine: state-dict packing (eval BatchNorm folding, K-major weight layout, head-dim
// padding), the Visformer eval forward as a fixed chain of kernel launches on one HIP stream, and
// the extern "C" boundary declared in include/fsvit.h.
#include "../../include/fsvit.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <utility>
#include <cstdint>
#include <initializer_list>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "train_kernels.h"

using namespace fsvit;

// Kernel-namespace dispatch: every kernel exists for the two 16-bit storage types (namespace fsvit = bf16, fsvit_f16 = fp16; see
// fsvit_common.h).  K(fn) picks the build that matches `kdt` (a FSVIT_* dtype in scope); kd() maps the public dtype to the kernels'
// own convention (0 = f32, 1 = the namespace's 16-bit type).
// The two-limb modes (FSVIT_BF16X2 / FSVIT_F16X2) are fp32 STORAGE (kd() = 0: every kernel but the GEMM is the fp32 path's) with the GEMM
// arithmetic selected by kg() = 2 (conv_gemm_v2.hip: x2_split) and the weights uploaded pre-split.
#define K(fn) ((kdt == FSVIT_F16 || kdt == FSVIT_F16X2) ? fsvit_f16::fn : fsvit::fn)
static inline bool is_x2(int dtype) { return dtype == FSVIT_BF16X2 || dtype == FSVIT_F16X2; }
static inline bool known_dtype(int dtype) { return dtype >= FSVIT_F32 && dtype <= FSVIT_F16X2; }
static inline int kd(int dtype) { return (dtype == FSVIT_F32 || is_x2(dtype)) ? 0 : 1; }
static inline int kg(int dtype) { return is_x2(dtype) ? 2 : kd(dtype); }
static inline int storage_bytes(int dtype) { return kd(dtype) == 0 ? 4 : 2; }

// ------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int fsvit_set_error(int code, const char* fmt, ...) {      // shared with train_engine.hip
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
static int hipfail(hipError_t e, const char* what) {
  return fail((int)e, "%s: %s", what, hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t _e = (expr);                             \
    if (_e != hipSuccess) return hipfail(_e, #expr);    \
  } while (0)
#define RC_TRY(expr)                                                              \
  do {                                                                            \
    int _rc = (expr);                                                             \
    if (_rc != 0) return _rc > 0 ? hipfail((hipError_t)_rc, #expr) : _rc;         \
  } while (0)

extern "C" const char* fsvit_last_error(void) { return g_err; }
extern "C" const char* fsvit_version(void) { return "fsvit 0.1.0 gfx950"; }

// ------------------------------------------------------------------------------------ helpers
static inline uint16_t f32_to_bf16(float f) {   // round to nearest even
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline uint16_t f32_to_f16(float f) {    // IEEE binary16, round to nearest even, gradual underflow, overflow -> inf
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                    // NaN
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                   // >= 65520 rounds to inf
  if (a < 0x33000001u) return sign;                                          // < 2^-25 (or exactly 2^-25: ties to even = 0)
  const int e = (int)(a >> 23) - 127;                                        // unbiased exponent
  uint32_t m = (a & 0x7fffffu) | 0x800000u;                                  // 24-bit significand
  int shift = e >= -14 ? 13 : 13 + (-14 - e);                                // bits dropped (subnormal halves drop more)
  const uint32_t half = 1u << (shift - 1), rest = m & ((1u << shift) - 1u);
  uint32_t q = m >> shift;
  if (rest > half || (rest == half && (q & 1u))) ++q;
  if (e >= -14) return (uint16_t)(sign | (uint16_t)(((uint32_t)(e + 15 - 1) << 10) + q));   // the hidden bit carries into the exponent
  return (uint16_t)(sign | (uint16_t)q);                                     // subnormal (q may carry into the smallest normal)
}
static inline float f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  float f;
  if (e == 0) { f = std::ldexp((float)m, -24); uint32_t u; memcpy(&u, &f, 4); u |= sign; memcpy(&f, &u, 4); return f; }
  const uint32_t u = sign | (e == 31 ? 0x7f800000u | (m << 13) : ((e + 112) << 23) | (m << 13));
  memcpy(&f, &u, 4);
  return f;
}
// One fp32 weight as the two-limb dword of the x2 GEMM (conv_gemm_v2.hip): upper half = hi = w rounded to the 16-bit type, lower half =
// lo = (w - hi) rounded to it.
static inline uint32_t x2_limbs(float w, bool f16) {
  uint16_t hi, lo;
  if (f16) { hi = f32_to_f16(w); lo = f32_to_f16(w - f16_to_f32(hi)); }
  else {
    hi = f32_to_bf16(w);
    uint32_t hu = (uint32_t)hi << 16; float hf; memcpy(&hf, &hu, 4);
    lo = f32_to_bf16(w - hf);
  }
  return ((uint32_t)hi << 16) | lo;
}
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

struct Layer {
  void* w = nullptr;      // [groups][N][Kw] storage dtype
  float* bias = nullptr;  // [groups*N] fp32 or null
  int N = 0, K = 0, Kw = 0, groups = 1;
};
struct Block1 {
  Layer c1, c2, c3;
  void* c3s = nullptr;      // 8 x the packed conv3 weights (stage1_w4.hip: the residual and the output then need no scaling)
};
struct BlockA {
  Layer qkv, proj, fc1, fc2;
  void* mlp_img = nullptr;     // mlp_rows.hip: fragment-major image of [proj |] fc1 + fc2 (null: the GEMM launches are used)
  float* mlp_b1 = nullptr;
  int mlp_kc = 0;              // > 0: the image starts with the proj fragments and mlp_rows also computes x += proj(ctx)
  void* qa_img = nullptr;      // qkv_attn.hip: fragment-major image of the qkv conv (null: qkv GEMM + attention launches)
  void* qr_img = nullptr;      // mlp_rows.hip gemm_rows: fragment-major image of the qkv conv (row-wise GEMM instead of the 256-tile one)
  void* qar_img = nullptr;     // mlp_rows.hip qkv_attn_rows: head-major image of the qkv conv (qkv + attention in one launch, maps of <= 32 tokens)
};

struct Tap { void* dst; size_t bytes; };

// One timed launch: events bracket the kernel on the stream it was launched on.
struct ProfEntry { std::string layer; int kernel; double flops; hipEvent_t e0, e1; };

// State shared by every encoder handle (the C-ABI functions that take "any encoder" rely on it being
// the first base sub-object).
enum { KIND_VISFORMER = 1, KIND_VIT = 2 };
struct EngineBase {
  int kind = 0;
  int dtype = 0, es = 4;
  std::map<std::string, Tap> taps;
  std::vector<void*> allocs;
  bool profiling = false;
  std::vector<ProfEntry> prof;
  hipEvent_t prof_last = nullptr;      // end event of the previous launch = start of the next
};

struct fsvit_visformer : EngineBase {
  fsvit_visformer_cfg cfg;
  int C0 = 0, C1 = 0, C2 = 0, C3 = 0;
  int H0 = 0, H1 = 0, H2 = 0, H3 = 0;
  int hid1 = 0, hid2 = 0, hid3 = 0;
  int hd2 = 0, hdp2 = 0, hd3 = 0, hdp3 = 0;
  Layer conv1, down, conv2, conv3, conv3f, pe2, pe3;   // conv3f: conv3 + downsample folded in as a tail K slice
  void* pe_img[2] = {nullptr, nullptr};                // mlp_rows.hip: fragment-major image of patch_embed2 / 3 for the row-wise kernel (null: conv_gemm)
  float *pos1 = nullptr, *pos2 = nullptr, *pos3 = nullptr;
  std::vector<Block1> s1;
  std::vector<BlockA> s2, s3;
  float *fscale = nullptr, *fshift = nullptr;
};

namespace {

struct SD {
  const fsvit_tensor* t;
  int n;
  const fsvit_tensor* find(const std::string& name) const {
    for (int i = 0; i < n; ++i)
      if (name == t[i].name) return &t[i];
    return nullptr;
  }
  // returns nullptr + sets error when missing / mis-shaped
  const float* get(const std::string& name, std::initializer_list<int64_t> shape) const {
    const fsvit_tensor* x = find(name);
    if (!x) { fail(FSVIT_ERR_KEY, "missing key in state_dict: %s", name.c_str()); return nullptr; }
    bool ok = x->ndim == (int)shape.size() && x->data != nullptr;
    int i = 0;
    for (int64_t s : shape) { if (ok && x->shape[i] != s) ok = false; ++i; }
    if (!ok) { fail(FSVIT_ERR_KEY, "size mismatch for %s", name.c_str()); return nullptr; }
    return x->data;
  }
};

struct Affine { std::vector<double> s, t; bool ok = false; };

Affine bn_affine(const SD& sd, const std::string& p, int C, double eps) {
  Affine a;
  const float* w = sd.get(p + ".weight", {C});
  const float* b = sd.get(p + ".bias", {C});
  const float* m = sd.get(p + ".running_mean", {C});
  const float* v = sd.get(p + ".running_var", {C});
  if (!w || !b || !m || !v) return a;
  a.s.resize(C); a.t.resize(C);
  for (int c = 0; c < C; ++c) {
    a.s[c] = (double)w[c] / std::sqrt((double)v[c] + eps);
    a.t[c] = (double)b[c] - (double)m[c] * a.s[c];
  }
  a.ok = true;
  return a;
}

int upload(EngineBase* h, const std::vector<float>& src, bool as_storage, void** out) {
  void* d = nullptr;
  if (as_storage && is_x2(h->dtype)) {          // GEMM weights of the two-limb modes: 4 bytes per value, (hi, lo) limbs
    std::vector<uint32_t> tmp(src.size());
    for (size_t i = 0; i < src.size(); ++i) tmp[i] = x2_limbs(src[i], h->dtype == FSVIT_F16X2);
    HIP_TRY(hipMalloc(&d, tmp.size() * 4));
    h->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
  } else if (as_storage && h->dtype != FSVIT_F32) {
    std::vector<uint16_t> tmp(src.size());
    for (size_t i = 0; i < src.size(); ++i) tmp[i] = h->dtype == FSVIT_F16 ? f32_to_f16(src[i]) : f32_to_bf16(src[i]);
    HIP_TRY(hipMalloc(&d, tmp.size() * 2));
    h->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
  } else {
    HIP_TRY(hipMalloc(&d, src.size() * 4));
    h->allocs.push_back(d);
    HIP_TRY(hipMemcpy(d, src.data(), src.size() * 4, hipMemcpyHostToDevice));
  }
  *out = d;
  return 0;
}

// ---- weight-rounding bias correction of the 16-bit modes (restated by oracle/visformer_emul.py operand_means / _corr)
// Rounding a folded weight W'[n][k] to the 16-bit type is the same perturbation dW for every token of every image, so its mean effect
// sum_k dW[n][k] E[a_k] survives the average pooling.  E[a_k] is in the checkpoint: the running mean of the BatchNorm that sees the operand
// (pre-norm 1x1 convs: exact), 0 for the normalised image, and elsewhere the mean of the activation of a Gaussian pre-activation whose
// moments follow from the preceding BatchNorm's (beta, gamma) under channel independence.  The correction is folded into the fp32 bias at
// pack time: no run-time cost.  Bias-free convs that feed the residual stream (conv3 / proj of a block) hand their correction on as a
// per-channel constant of the stream (`cst`), which is folded into the bias of every later consumer (W' cst) and into the pooled feature.
struct WRound {
  bool on = false, f16 = false;
  std::vector<double> cst;                       // deferred constant of the residual stream
  double rounded(double v) const {               // the value the 16-bit weight image holds for v
    const float f = (float)v;
    if (f16) return (double)f16_to_f32(f32_to_f16(f));
    uint32_t u = (uint32_t)f32_to_bf16(f) << 16;
    float r; memcpy(&r, &u, 4);
    return (double)r;
  }
};

// fraction of the output positions of a k x k conv (stride, zero padding pad) on an H x H map whose tap (ky, kx) lies inside the map
std::vector<double> tap_fraction(int H, int stride, int k, int pad) {
  const int Ho = (H + 2 * pad - k) / stride + 1;
  std::vector<double> f1(k), f((size_t)k * k);
  for (int t = 0; t < k; ++t) {
    int in = 0;
    for (int o = 0; o < Ho; ++o) { const int pos = o * stride - pad + t; in += (pos >= 0 && pos < H); }
    f1[t] = (double)in / Ho;
  }
  for (int y = 0; y < k; ++y) for (int x = 0; x < k; ++x) f[(size_t)y * k + x] = f1[y] * f1[x];
  return f;
}

inline double gelu_exact(double z) { return 0.5 * z * (1.0 + std::erf(z * 0.7071067811865476)); }

// E[GELU(y)] and Var[GELU(y)] for y ~ N(mu, sig^2): trapezoid rule on z in [-8, 8], 257 nodes
void gauss_moments_gelu(double mu, double sig, double* mean, double* var) {
  double sw = 0, s1 = 0, s2 = 0;
  for (int j = 0; j <= 256; ++j) {
    const double z = -8.0 + j / 16.0, w = std::exp(-0.5 * z * z), v = gelu_exact(mu + sig * z);
    sw += w; s1 += w * v; s2 += w * v * v;
  }
  *mean = s1 / sw;
  *var = std::max(0.0, s2 / sw - (s1 / sw) * (s1 / sw));
}

// E[LeakyReLU_0.1(N(mu, sig^2))] = mu (0.1 + 0.9 Phi(mu / sig)) + 0.9 sig phi(mu / sig)
double lrelu_mean(double mu, double sig) {
  if (sig <= 0) return mu > 0 ? mu : 0.1 * mu;
  const double t = mu / sig;
  return mu * (0.1 + 0.9 * 0.5 * (1.0 + std::erf(t * 0.7071067811865476))) + 0.9 * sig * std::exp(-0.5 * t * t) * 0.3989422804014327;
}

// Pack a conv weight W[O][Ig][KH][KW] (O = groups*N) into [groups][N][Kw], k = (ky*KW+kx)*Ig + c,
// with optional per-output-channel scale (BN after the conv) and per-input-channel scale (BN before
// a 1x1 conv).  `rowmap`/`colmap` (optional) scatter rows / K columns (head-dim padding).
// Weight-rounding correction (wr != null and on): `in_mean` [groups*Ig] = E[operand channel], `tapf` [KH*KW] = tap_fraction or null;
// the correction -sum_k dW f_tap E[a] and (`use_cst`) the stream constant's image sum_k W' cst go into the bias; a bias-free layer
// returns its correction in `corr_out` [O] instead.
int pack_layer(EngineBase* h, Layer* L, const float* W, int O, int Ig, int KH, int KW, int groups,
               const std::vector<double>* out_scale, const std::vector<double>* in_scale,
               const std::vector<double>& bias, bool has_bias,
               const std::vector<int>* rowmap, int Npad, const std::vector<int>* colmap, int Kpad,
               const WRound* wr = nullptr, const std::vector<double>* in_mean = nullptr, const double* tapf = nullptr,
               bool use_cst = false, std::vector<double>* corr_out = nullptr) {
  const int bke = 128 / h->es;
  const int N = (rowmap ? Npad : O) / groups;
  const int K = colmap ? Kpad : KH * KW * Ig;
  const int Kw = round_up(K, bke);
  // (an operand-mean / stream-constant vector that does not cover this layer's input channels - e.g. one handed on across a stage without blocks - is
  // not applied: no correction instead of another tensor's statistics)
  const bool fix = wr && wr->on && in_mean && in_mean->size() >= (size_t)groups * Ig;
  const bool cst = wr && wr->on && use_cst && wr->cst.size() >= (size_t)groups * Ig;
  if (corr_out) corr_out->assign(O, 0.0);
  std::vector<float> pk((size_t)groups * N * Kw, 0.0f);
  std::vector<float> pb((size_t)groups * N, 0.0f);
  for (int o = 0; o < O; ++o) {
    const int row = rowmap ? (*rowmap)[o] : o;
    const double so = out_scale ? (*out_scale)[o] : 1.0;
    const int g0 = (o / (O / groups)) * Ig;
    double corr = 0.0, wc = 0.0;
    for (int c = 0; c < Ig; ++c) {
      const double si = in_scale ? (*in_scale)[c] : 1.0;
      for (int ky = 0; ky < KH; ++ky)
        for (int kx = 0; kx < KW; ++kx) {
          int k = (ky * KW + kx) * Ig + c;
          if (colmap) k = (*colmap)[k];
          const double v = (double)W[(((size_t)o * Ig + c) * KH + ky) * KW + kx] * so * si;
          pk[(size_t)row * Kw + k] = (float)v;
          if (fix) corr -= (wr->rounded(v) - v) * (tapf ? tapf[ky * KW + kx] : 1.0) * (*in_mean)[g0 + c];
          if (cst) wc += v * wr->cst[g0 + c];
        }
    }
    if (has_bias) pb[row] = (float)(bias[o] + corr + wc);
    else if (corr_out) (*corr_out)[o] = corr;
  }
  L->N = N; L->K = K; L->Kw = Kw; L->groups = groups;
  RC_TRY(upload(h, pk, true, &L->w));
  if (has_bias) { void* b; RC_TRY(upload(h, pb, false, &b)); L->bias = (float*)b; }
  return 0;
}

// bias[n] = sum_c W[n][c] * t[c]  for a 1x1 conv that follows a BatchNorm
std::vector<double> prenorm_bias(const float* W, int O, int C, const std::vector<double>& t) {
  std::vector<double> b(O, 0.0);
  for (int o = 0; o < O; ++o) {
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += (double)W[(size_t)o * C + c] * t[c];
    b[o] = s;
  }
  return b;
}

// pos_embed [1,C,H,W] -> [H*W][C] fp32 (+ optional per-channel constant)
int pack_pos(EngineBase* h, const float* pos, int C, int HW, float** out) {
  std::vector<float> p((size_t)HW * C);
  for (int c = 0; c < C; ++c)
    for (int i = 0; i < HW; ++i) p[(size_t)i * C + c] = pos[(size_t)c * HW + i];
  void* d; RC_TRY(upload(h, p, false, &d));
  *out = (float*)d;
  return 0;
}

int build(fsvit_visformer* h, const SD& sd) {
  const int kdt = h->dtype;
  const fsvit_visformer_cfg& cf = h->cfg;
  const double eps = cf.bn_eps;
  const int epc = 16 / h->es;
  const int D = cf.embed_dim;
  h->C0 = cf.init_channels; h->C1 = D / 2; h->C2 = D; h->C3 = D * 2;
  if (cf.img_size % 16 != 0) return fail(FSVIT_ERR_ARG, "img_size %d must be a multiple of 16", cf.img_size);
  h->H0 = cf.img_size / 2; h->H1 = cf.img_size / 4; h->H2 = cf.img_size / 8; h->H3 = cf.img_size / 16;
  if (cf.group < 2) return fail(FSVIT_ERR_ARG, "group < 2 ('net' setting, visformer.py:137-138) is not supported");
  h->hid1 = h->C1 * 2;
  h->hid2 = (int)(h->C2 * cf.mlp_ratio);
  h->hid3 = (int)(h->C3 * cf.mlp_ratio);
  const int kch = 64 / h->es;                                  // head dim padded to the 64-byte MFMA K chunk
  h->hd2 = (int)std::lround((double)(h->C2 / cf.num_heads));  // round(dim // heads * 1.0), visformer.py:172
  h->hd3 = (int)std::lround((double)(h->C3 / cf.num_heads));
  h->hdp2 = K(attention_padded_head_dim)(h->hd2, h->H2 * h->H2, kd(kdt));
  h->hdp3 = K(attention_padded_head_dim)(h->hd3, h->H3 * h->H3, kd(kdt));
  (void)kch;
  const int Cg = h->hid1 / cf.group;
  if (h->C0 % epc || h->C1 % epc || h->hid1 % cf.group || Cg % epc)
    return fail(FSVIT_ERR_ARG, "channel counts must be multiples of %d for 16-byte K chunks", epc);
  if (!is_pow2(h->C0) || !is_pow2(h->C1) || !is_pow2(h->C2) || !is_pow2(Cg))
    return fail(FSVIT_ERR_ARG, "multi-tap conv input channels must be powers of two");
  if (h->H2 * h->H2 > 128) return fail(FSVIT_ERR_ARG, "attention supports at most 128 tokens (img_size <= 88)");

  std::vector<double> nob;
  WRound wr;                                    // weight-rounding bias correction: the 16-bit modes only (the two-limb modes carry 16+ bits)
  // FSVIT_WROUND=0 (read once per build): the plain 16-bit images without the correction - the ablation switch of the one eval route whose effect
  // rests on statistical assumptions (data that matches the running statistics); tests/test_gpu_unfused_paths.py
  const char* wr_env = getenv("FSVIT_WROUND");
  wr.on = (kdt == FSVIT_BF16 || kdt == FSVIT_F16) && !(wr_env && wr_env[0] == '0');
  wr.f16 = kdt == FSVIT_F16;
  if (wr.on) wr.cst.assign(h->C1, 0.0);
  auto vec = [&](const std::string& name, int C, bool absolute) {      // a [C] state-dict entry as doubles (checked by bn_affine before)
    const float* p = sd.get(name, {C});
    std::vector<double> v(C, 0.0);
    for (int c = 0; p && c < C; ++c) v[c] = absolute ? std::fabs((double)p[c]) : (double)p[c];
    return v;
  };
  // ---- stem (visformer.py:202-239): conv -> BN folded as per-output-channel scale + shift
  {
    const float* w1 = sd.get("stem.conv1.weight", {h->C0, 3, 3, 3});
    const float* wd = sd.get("stem.downsample.0.weight", {h->C1, 3, 3, 3});
    const float* w2 = sd.get("stem.conv2.weight", {h->C1, h->C0, 3, 3});
    const float* w3 = sd.get("stem.conv3.weight", {h->C1, h->C1, 3, 3});
    Affine b1 = bn_affine(sd, "stem.bn1", h->C0, eps), b2 = bn_affine(sd, "stem.bn2", h->C1, eps);
    Affine b3 = bn_affine(sd, "stem.bn3", h->C1, eps), bd = bn_affine(sd, "stem.downsample.1", h->C1, eps);
    if (!w1 || !wd || !w2 || !w3 || !b1.ok || !b2.ok || !b3.ok || !bd.ok) return FSVIT_ERR_KEY;
    // conv1 / downsample consume the 32-wide im2col rows: K columns 0..26 = (ky,kx,c), 27..31 = 0
    std::vector<int> cm(27);
    for (int k = 0; k < 27; ++k) cm[k] = k;
    // operand means: the normalised image 0 (no correction for conv1 / downsample); conv2 / conv3 read LeakyReLU(BN(z)), z_c ~ N(beta_c, gamma_c^2)
    std::vector<double> m2(h->C0), m3(h->C1);
    {
      const std::vector<double> be1 = vec("stem.bn1.bias", h->C0, false), ga1 = vec("stem.bn1.weight", h->C0, true);
      const std::vector<double> be2 = vec("stem.bn2.bias", h->C1, false), ga2 = vec("stem.bn2.weight", h->C1, true);
      for (int c = 0; c < h->C0; ++c) m2[c] = lrelu_mean(be1[c], ga1[c]);
      for (int c = 0; c < h->C1; ++c) m3[c] = lrelu_mean(be2[c], ga2[c]);
    }
    const std::vector<double> tf0 = tap_fraction(h->H0, 1, 3, 1);
    RC_TRY(pack_layer(h, &h->conv1, w1, h->C0, 3, 3, 3, 1, &b1.s, nullptr, b1.t, true, nullptr, 0, &cm, 32));
    RC_TRY(pack_layer(h, &h->down, wd, h->C1, 3, 3, 3, 1, &bd.s, nullptr, bd.t, true, nullptr, 0, &cm, 32));
    RC_TRY(pack_layer(h, &h->conv2, w2, h->C1, h->C0, 3, 3, 1, &b2.s, nullptr, b2.t, true, nullptr, 0, nullptr, 0, &wr, &m2, tf0.data()));
    RC_TRY(pack_layer(h, &h->conv3, w3, h->C1, h->C1, 3, 3, 1, &b3.s, nullptr, b3.t, true, nullptr, 0, nullptr, 0, &wr, &m3, tf0.data()));
    {  // conv3f rows = [ bn3-scaled conv3 (K = 9*C1, padded to the K slice) | one tail slice: bn_d-scaled downsample taps ]
      const int bke = 128 / h->es, K = 9 * h->C1, Kmain = round_up(K, bke), Kw = Kmain + bke;
      std::vector<float> pk((size_t)h->C1 * Kw, 0.0f), pb(h->C1);
      for (int o = 0; o < h->C1; ++o) {
        double corr = 0.0;
        for (int c = 0; c < h->C1; ++c)
          for (int t9 = 0; t9 < 9; ++t9) {
            const double v = (double)w3[((size_t)o * h->C1 + c) * 9 + t9] * b3.s[o];
            pk[(size_t)o * Kw + t9 * h->C1 + c] = (float)v;
            if (wr.on) corr -= (wr.rounded(v) - v) * tf0[t9] * m3[c];
          }
        for (int c = 0; c < 3; ++c)
          for (int t9 = 0; t9 < 9; ++t9)
            pk[(size_t)o * Kw + Kmain + t9 * 3 + c] = (float)((double)wd[((size_t)o * 3 + c) * 9 + t9] * bd.s[o]);
        pb[o] = (float)(b3.t[o] + bd.t[o] + corr);
      }
      h->conv3f.N = h->C1; h->conv3f.K = K; h->conv3f.Kw = Kw; h->conv3f.groups = 1;
      RC_TRY(upload(h, pk, true, &h->conv3f.w));
      void* bdev; RC_TRY(upload(h, pb, false, &bdev)); h->conv3f.bias = (float*)bdev;
    }
  }
  {
    const float* p1 = sd.get("pos_embed1", {1, h->C1, h->H1, h->H1});
    const float* p2 = sd.get("pos_embed2", {1, h->C2, h->H2, h->H2});
    const float* p3 = sd.get("pos_embed3", {1, h->C3, h->H3, h->H3});
    if (!p1 || !p2 || !p3) return FSVIT_ERR_KEY;
    RC_TRY(pack_pos(h, p1, h->C1, h->H1 * h->H1, &h->pos1));
    RC_TRY(pack_pos(h, p2, h->C2, h->H2 * h->H2, &h->pos2));
    RC_TRY(pack_pos(h, p3, h->C3, h->H3 * h->H3, &h->pos3));
  }
  // ---- stage 1 (Block with attn_disabled, spatial_conv; visformer.py:241-263, Mlp :127-163)
  // WRound: estimated mean of the residual stream where the next PatchEmbed reads it.  Seeded with zeros (a stage without blocks hands on "no
  // estimate" of the right size: ADVICE r04 - depth (0, 1, 1) indexed an empty vector) and re-seeded behind every PatchEmbed.
  std::vector<double> x_mean(h->C1, 0.0);
  h->s1.resize(cf.depth[0]);
  for (int i = 0; i < cf.depth[0]; ++i) {
    const std::string p = "stage1." + std::to_string(i) + ".";
    Affine n2 = bn_affine(sd, p + "norm2.bn", h->C1, eps);
    const float* w1 = sd.get(p + "mlp.conv1.weight", {h->hid1, h->C1, 1, 1});
    const float* w2 = sd.get(p + "mlp.conv2.weight", {h->hid1, Cg, 3, 3});
    const float* w3 = sd.get(p + "mlp.conv3.weight", {h->C1, h->hid1, 1, 1});
    if (!n2.ok || !w1 || !w2 || !w3) return FSVIT_ERR_KEY;
    // operand means: conv1 reads the stream the BatchNorm saw (running mean); conv3 reads h2 = GELU(conv2(h1)), h1 = GELU(conv1(BN(x))) with
    // Gaussian pre-activations; the grouped conv2 has no bias and a GELU behind it: not corrected
    const std::vector<double> rm = vec(p + "norm2.bn.running_mean", h->C1, false);
    std::vector<double> mh2(h->hid1, 0.0), corr3;
    if (wr.on) {
      if (wr.cst.empty()) wr.cst.assign(h->C1, 0.0);
      const std::vector<double> be = vec(p + "norm2.bn.bias", h->C1, false), ga = vec(p + "norm2.bn.weight", h->C1, true);
      const std::vector<double> tf1 = tap_fraction(h->H1, 1, 3, 1);
      std::vector<double> mh1(h->hid1), vh1(h->hid1);
      for (int n = 0; n < h->hid1; ++n) {
        double mu = 0, var = 0;
        for (int c = 0; c < h->C1; ++c) { const double w = w1[(size_t)n * h->C1 + c]; mu += w * be[c]; var += w * w * ga[c] * ga[c]; }
        gauss_moments_gelu(mu, std::sqrt(var), &mh1[n], &vh1[n]);
      }
      for (int n = 0; n < h->hid1; ++n) {
        const int g0 = (n / (h->hid1 / cf.group)) * Cg;
        double mu = 0, var = 0, dummy;
        for (int c = 0; c < Cg; ++c)
          for (int t9 = 0; t9 < 9; ++t9) {
            const double w = w2[((size_t)n * Cg + c) * 9 + t9];
            mu += w * tf1[t9] * mh1[g0 + c]; var += w * w * tf1[t9] * vh1[g0 + c];
          }
        gauss_moments_gelu(mu, std::sqrt(var), &mh2[n], &dummy);
      }
    }
    RC_TRY(pack_layer(h, &h->s1[i].c1, w1, h->hid1, h->C1, 1, 1, 1, nullptr, &n2.s, prenorm_bias(w1, h->hid1, h->C1, n2.t), true, nullptr, 0, nullptr, 0,
                      &wr, &rm, nullptr, true));
    RC_TRY(pack_layer(h, &h->s1[i].c2, w2, h->hid1, Cg, 3, 3, cf.group, nullptr, nullptr, nob, false, nullptr, 0, nullptr, 0));
    RC_TRY(pack_layer(h, &h->s1[i].c3, w3, h->C1, h->hid1, 1, 1, 1, nullptr, nullptr, nob, false, nullptr, 0, nullptr, 0, &wr, &mh2, nullptr, false, &corr3));
    if (kd(kdt) == 1 && h->s1[i].c3.Kw == h->hid1) {
      const size_t n3 = (size_t)h->C1 * h->s1[i].c3.Kw;
      HIP_TRY(hipMalloc(&h->s1[i].c3s, n3 * 2));
      h->allocs.push_back(h->s1[i].c3s);
      RC_TRY(K(launch_stage1_w4_prescale)(h->s1[i].c3.w, h->s1[i].c3s, (int)n3, nullptr));
      HIP_TRY(hipDeviceSynchronize());
    }
    if (wr.on) {
      x_mean.assign(h->C1, 0.0);                  // E[x] behind this block = running mean + conv3 E[h2]: the next PatchEmbed's operand mean
      for (int o = 0; o < h->C1; ++o) {
        double a = rm[o];
        for (int c = 0; c < h->hid1; ++c) a += (double)w3[(size_t)o * h->hid1 + c] * mh2[c];
        x_mean[o] = a;
        wr.cst[o] += corr3[o];
      }
    }
  }
  // ---- per stage s = 2, 3: PatchEmbed (visformer.py:266-288: conv k2 s2 + bias -> BN ; pos_embed added in the epilogue), then its
  // attention + MLP blocks (Attention :166-194, Block :259-263)
  for (int s = 2; s <= 3; ++s) {
   {
    const int Ci = s == 2 ? h->C1 : h->C2, Co = s == 2 ? h->C2 : h->C3;
    const std::string p = "patch_embed" + std::to_string(s) + ".";
    const float* w = sd.get(p + "proj.weight", {Co, Ci, 2, 2});
    const float* b = sd.get(p + "proj.bias", {Co});
    Affine bn = bn_affine(sd, p + "norm.bn", Co, eps);
    if (!w || !b || !bn.ok) return FSVIT_ERR_KEY;
    std::vector<double> bias(Co);
    for (int o = 0; o < Co; ++o) bias[o] = bn.s[o] * (double)b[o] + bn.t[o];
    // WRound: the operand is the residual stream leaving the previous stage (estimated mean x_mean); its deferred constant ends here
    RC_TRY(pack_layer(h, s == 2 ? &h->pe2 : &h->pe3, w, Co, Ci, 2, 2, 1, &bn.s, nullptr, bias, true, nullptr, 0, nullptr, 0, &wr, &x_mean, nullptr, true));
    if (wr.on) {
      wr.cst.assign(Co, 0.0);
      // the stream behind this PatchEmbed: E = BatchNorm beta + mean_hw(pos_embed) until a block of the stage refines it
      const std::vector<double> be = vec(p + "norm.bn.bias", Co, false);
      const int HW = (s == 2 ? h->H2 : h->H3) * (s == 2 ? h->H2 : h->H3);
      const float* pos = sd.get("pos_embed" + std::to_string(s), {1, Co, s == 2 ? h->H2 : h->H3, s == 2 ? h->H2 : h->H3});
      x_mean.assign(Co, 0.0);
      for (int o = 0; o < Co; ++o) {
        double a = 0.0;
        for (int q = 0; pos && q < HW; ++q) a += (double)pos[(size_t)o * HW + q];
        x_mean[o] = be[o] + a / HW;
      }
    }
    const Layer& pe = s == 2 ? h->pe2 : h->pe3;
    if (K(patch_embed_rows_supported)(kd(kdt), Ci, s == 2 ? h->H1 : h->H2, Co) && pe.Kw == 4 * Ci) {
      void* img = nullptr;
      HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(4 * Ci, Co)));
      h->allocs.push_back(img);
      RC_TRY(K(launch_ln_gemm_pack)(pe.w, pe.Kw, img, 4 * Ci, Co, nullptr));
      HIP_TRY(hipDeviceSynchronize());
      h->pe_img[s - 2] = img;
    }
   }
    const int C = s == 2 ? h->C2 : h->C3, hid = s == 2 ? h->hid2 : h->hid3;
    const int hd = s == 2 ? h->hd2 : h->hd3, hdp = s == 2 ? h->hdp2 : h->hdp3;
    const int heads = cf.num_heads;
    if (C % epc || hid % epc) return fail(FSVIT_ERR_ARG, "channel counts must be multiples of %d", epc);
    std::vector<BlockA>& blocks = s == 2 ? h->s2 : h->s3;
    blocks.resize(cf.depth[s - 1]);
    // head-dim padding: qkv rows (x, y, z) -> x*heads*hdp + y*hdp + z ; proj K columns (y, z) -> y*hdp + z
    std::vector<int> rowmap(3 * heads * hd), colmap(heads * hd);
    for (int x = 0; x < 3; ++x)
      for (int y = 0; y < heads; ++y)
        for (int z = 0; z < hd; ++z) rowmap[(x * heads + y) * hd + z] = (x * heads + y) * hdp + z;
    for (int y = 0; y < heads; ++y)
      for (int z = 0; z < hd; ++z) colmap[y * hd + z] = y * hdp + z;
    for (int i = 0; i < cf.depth[s - 1]; ++i) {
      const std::string p = "stage" + std::to_string(s) + "." + std::to_string(i) + ".";
      Affine n1 = bn_affine(sd, p + "norm1.bn", C, eps), n2 = bn_affine(sd, p + "norm2.bn", C, eps);
      const float* wq = sd.get(p + "attn.qkv.weight", {3 * heads * hd, C, 1, 1});
      const float* wp = sd.get(p + "attn.proj.weight", {C, heads * hd, 1, 1});
      const float* w1 = sd.get(p + "mlp.conv1.weight", {hid, C, 1, 1});
      const float* w3 = sd.get(p + "mlp.conv3.weight", {C, hid, 1, 1});
      if (!n1.ok || !n2.ok || !wq || !wp || !w1 || !w3) return FSVIT_ERR_KEY;
      // WRound operand means: qkv / conv1 read the stream their BatchNorm saw (running means); proj reads ctx, E[ctx] ~ E[v] = W_v beta_1
      // (softmax rows sum to 1); conv3 reads GELU of a Gaussian pre-activation N(W1 beta_2, W1^2 gamma_2^2)
      const std::vector<double> rm1 = vec(p + "norm1.bn.running_mean", C, false), rm2 = vec(p + "norm2.bn.running_mean", C, false);
      std::vector<double> mctx(heads * hd, 0.0), mh(hid, 0.0), corr_p, corr_3;
      if (wr.on) {
        const std::vector<double> be1 = vec(p + "norm1.bn.bias", C, false);
        const std::vector<double> be2 = vec(p + "norm2.bn.bias", C, false), ga2 = vec(p + "norm2.bn.weight", C, true);
        for (int n = 0; n < heads * hd; ++n) {
          double a = 0;
          for (int c = 0; c < C; ++c) a += (double)wq[((size_t)2 * heads * hd + n) * C + c] * be1[c];
          mctx[n] = a;
        }
        for (int n = 0; n < hid; ++n) {
          double mu = 0, var = 0, dummy;
          for (int c = 0; c < C; ++c) { const double w = w1[(size_t)n * C + c]; mu += w * be2[c]; var += w * w * ga2[c] * ga2[c]; }
          gauss_moments_gelu(mu, std::sqrt(var), &mh[n], &dummy);
        }
      }
      RC_TRY(pack_layer(h, &blocks[i].qkv, wq, 3 * heads * hd, C, 1, 1, 1, nullptr, &n1.s, prenorm_bias(wq, 3 * heads * hd, C, n1.t), true, &rowmap, 3 * heads * hdp, nullptr, 0,
                        &wr, &rm1, nullptr, true));
      RC_TRY(pack_layer(h, &blocks[i].proj, wp, C, heads * hd, 1, 1, 1, nullptr, nullptr, nob, false, nullptr, 0, &colmap, heads * hdp, &wr, &mctx, nullptr, false, &corr_p));
      if (wr.on) for (int o = 0; o < C; ++o) wr.cst[o] += corr_p[o];
      RC_TRY(pack_layer(h, &blocks[i].fc1, w1, hid, C, 1, 1, 1, nullptr, &n2.s, prenorm_bias(w1, hid, C, n2.t), true, nullptr, 0, nullptr, 0, &wr, &rm2, nullptr, true));
      RC_TRY(pack_layer(h, &blocks[i].fc2, w3, C, hid, 1, 1, 1, nullptr, nullptr, nob, false, nullptr, 0, nullptr, 0, &wr, &mh, nullptr, false, &corr_3));
      if (wr.on) {
        x_mean.assign(C, 0.0);
        for (int o = 0; o < C; ++o) {
          double a = rm2[o];
          for (int c = 0; c < hid; ++c) a += (double)w3[(size_t)o * hid + c] * mh[c];
          x_mean[o] = a;
          wr.cst[o] += corr_3[o];
        }
      }
      if (K(mlp_rows_supported)(kd(kdt), C, hid)) {                // fused row-wise Mlp: re-pack [proj,] fc1, fc2 as the MFMA fragment stream
        const int kc = K(mlp_rows_proj_supported)(C, hid, heads * hdp) ? heads * hdp : 0;
        void *img = nullptr, *b1i = nullptr;
        HIP_TRY(hipMalloc(&img, K(mlp_rows_image_bytes)(C, hid, kc)));
        h->allocs.push_back(img);
        HIP_TRY(hipMalloc(&b1i, (size_t)hid * 4));
        h->allocs.push_back(b1i);
        RC_TRY(K(launch_mlp_pack)(blocks[i].fc1.w, blocks[i].fc1.Kw, blocks[i].fc1.bias, blocks[i].fc2.w, blocks[i].fc2.Kw, blocks[i].proj.w, blocks[i].proj.Kw,
                               kc, img, (float*)b1i, C, hid, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        blocks[i].mlp_img = img;
        blocks[i].mlp_b1 = (float*)b1i;
        blocks[i].mlp_kc = kc;
      }
      if (K(qkv_attn_supported)(kd(kdt), C, heads, hdp, (s == 2 ? h->H2 : h->H3) * (s == 2 ? h->H2 : h->H3))) {   // fused qkv conv + attention: re-pack qkv as the MFMA fragment stream
        void* img = nullptr;
        HIP_TRY(hipMalloc(&img, K(qkv_attn_image_bytes)()));
        h->allocs.push_back(img);
        RC_TRY(K(launch_qkv_attn_pack)(blocks[i].qkv.w, blocks[i].qkv.Kw, img, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        blocks[i].qa_img = img;
      } else if (K(qkv_attn_rows_supported)(kd(kdt), C, heads, hdp, (s == 2 ? h->H2 : h->H3) * (s == 2 ? h->H2 : h->H3))) {   // stage 3, <= 32 tokens
        void* img = nullptr;
        HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(C, 3 * heads * hdp)));
        h->allocs.push_back(img);
        RC_TRY(K(launch_qkv_attn_rows_pack)(blocks[i].qkv.w, blocks[i].qkv.Kw, img, C, heads, hdp, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        blocks[i].qar_img = img;
      } else if (K(gemm_rows_supported)(kd(kdt), C, 3 * heads * hdp)) {      // stage 3: the qkv conv as a row-wise GEMM
        void* img = nullptr;
        HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(C, 3 * heads * hdp)));
        h->allocs.push_back(img);
        RC_TRY(K(launch_ln_gemm_pack)(blocks[i].qkv.w, blocks[i].qkv.Kw, img, C, 3 * heads * hdp, nullptr));
        HIP_TRY(hipDeviceSynchronize());
        blocks[i].qr_img = img;
      }
    }
  }
  // ---- final BN -> pooled feature affine (visformer.py:455-462)
  {
    Affine bn = bn_affine(sd, "norm.bn", h->C3, eps);
    if (!bn.ok) return FSVIT_ERR_KEY;
    std::vector<float> s(h->C3), t(h->C3);
    for (int c = 0; c < h->C3; ++c) { s[c] = (float)bn.s[c]; t[c] = (float)(bn.t[c] + (wr.on ? bn.s[c] * wr.cst[c] : 0.0)); }
    void* d; RC_TRY(upload(h, s, false, &d)); h->fscale = (float*)d;
    RC_TRY(upload(h, t, false, &d)); h->fshift = (float*)d;
  }
  return 0;
}

// ---- workspace plan for a chunk of Bc images
struct Plan {
  size_t patches, c1, ident, c2, c3;     // stem scratch
  size_t ha, hb;                         // stage-1 hidden
  size_t qkv, ctx, hid;                  // stage-2/3 scratch
  size_t x1, x1b, x2, x3;                // residual streams (x1b: ping-pong partner for the fused stage-1 block)
  size_t total;
};

size_t align256(size_t v) { return (v + 255) / 256 * 256; }

Plan make_plan(const fsvit_visformer* h, size_t Bc) {
  Plan p;
  const size_t es = h->es, P0 = (size_t)h->H0 * h->H0, P1 = (size_t)h->H1 * h->H1, P2 = (size_t)h->H2 * h->H2, P3 = (size_t)h->H3 * h->H3;
  const int heads = h->cfg.num_heads;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
  p.x1 = take(Bc * P1 * h->C1 * es);
  p.x1b = take(Bc * P1 * h->C1 * es);
  p.x2 = take(Bc * P2 * h->C2 * es);
  p.x3 = take(Bc * P3 * h->C3 * es);
  const size_t scratch0 = off;
  // stem
  p.patches = take(Bc * P0 * 32 * es);
  p.c1 = take(Bc * P0 * h->C0 * es);
  p.ident = take(Bc * P0 * h->C1 * es);
  p.c2 = take(Bc * P0 * h->C1 * es);
  p.c3 = take(Bc * P0 * h->C1 * es);
  size_t hi = off;
  // stage 1 (reuses the stem scratch)
  off = scratch0;
  p.ha = take(Bc * P1 * h->hid1 * es);
  p.hb = take(Bc * P1 * h->hid1 * es);
  if (off > hi) hi = off;
  // stage 2 / 3 (same offsets, sized for the larger)
  off = scratch0;
  size_t q2 = Bc * P2 * 3 * heads * h->hdp2 * es, q3 = Bc * P3 * 3 * heads * h->hdp3 * es;
  size_t c2 = Bc * P2 * heads * h->hdp2 * es, c3 = Bc * P3 * heads * h->hdp3 * es;
  size_t h2 = Bc * P2 * h->hid2 * es, h3 = Bc * P3 * h->hid3 * es;
  p.qkv = take(q2 > q3 ? q2 : q3);
  p.ctx = take(c2 > c3 ? c2 : c3);
  p.hid = take(h2 > h3 ? h2 : h3);
  if (off > hi) hi = off;
  p.total = hi;
  return p;
}

ConvGemmParams conv_params(const Layer& L, const void* x, void* y, int B, int H, int W, int Cin, int x_cstride,
                           int KH, int KW, int stride, int pad, int y_cstride, int act,
                           const void* res, int res_first, const float* pos) {
  ConvGemmParams p;
  p.x = x; p.w = L.w; p.bias = L.bias; p.res = res; p.pos = pos; p.y = y;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.x_cstride = x_cstride;
  p.OH = (H + 2 * pad - KH) / stride + 1; p.OW = (W + 2 * pad - KW) / stride + 1;
  p.KH = KH; p.KW = KW; p.stride = stride; p.pad = pad;
  p.N = L.N; p.y_cstride = y_cstride; p.K = L.K; p.Kw = L.Kw; p.M = B * p.OH * p.OW;
  p.groups = L.groups; p.act = act; p.res_first = res_first; p.log2Cin = ilog2(Cin);
  p.x2 = nullptr; p.x2_cstride = 0; p.K2 = 0; p.pool2 = 0; p.y_rpi = 0; p.y_row0 = 0;
  p.w_gstride = 0; p.w_rstride = 0; p.out_f32 = 0; p.y2 = nullptr; p.stats = nullptr;
  return p;
}

int tap(EngineBase* h, const std::string& name, const void* src, size_t bytes, bool first, hipStream_t st) {
  if (!first || h->taps.empty()) return 0;
  auto it = h->taps.find(name);
  if (it == h->taps.end() || it->second.dst == nullptr) return 0;
  size_t n = bytes < it->second.bytes ? bytes : it->second.bytes;
  HIP_TRY(hipMemcpyAsync(it->second.dst, src, n, hipMemcpyDeviceToDevice, st));
  return 0;
}

// kernel ids reported by the profiler (names in fsvit_kernel_name)
enum { KID_GEMM256 = 0, KID_GEMM64 = 1, KID_GEMM32 = 2, KID_IM2COL = 3, KID_MAXPOOL = 4, KID_ATTN = 5, KID_POOL = 6, KID_HEAD = 7, KID_STAGE1 = 8, KID_GEMM128 = 9, KID_HALO = 12, KID_MLPROWS = 13, KID_QKVATTN = 15, KID_STEMCONV1 = 16, KID_GCONV_X2 = 14, KID_STAGE1RING = 17, KID_LNGEMM = 18, KID_QKVATTNROWS = 19, KID_VITATTNROWS = 20 };


// Runs one launch; in profiling mode brackets it with HIP events on the same stream.
template <typename F>
int timed(EngineBase* h, hipStream_t st, const char* layer, int kernel, double flops, F&& launch) {
  if (!h->profiling) return launch();
  hipEvent_t e0 = h->prof_last, e1;
  if (!e0) { HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventRecord(e0, st)); }
  int rc = launch();
  if (rc != 0) return rc;
  HIP_TRY(hipEventCreate(&e1));
  HIP_TRY(hipEventRecord(e1, st));
  h->prof.push_back(ProfEntry{layer, kernel, flops, e0, e1});
  h->prof_last = e1;
  return 0;
}

// images per slice of the dedicated stem (forward_chunk): swept 1600 / 3200 / 6400 / 12 800 on one box - 29.14 / 29.06 / 29.24 / 29.26 ms per 12 800-image step
static constexpr int stem_slice_images() { return 3200; }

int run_gemm(EngineBase* h, hipStream_t st, const char* layer, const Layer& L, const ConvGemmParams& p, double n_true, double k_true) {
  const int kdt = h->dtype;
  const double flops = 2.0 * (double)p.M * n_true * k_true * (double)p.groups;     // algorithmic: unpadded N and K
  static const int kid_of_cfg[4] = {KID_GEMM256, KID_GEMM64, KID_GEMM32, KID_GEMM128};
  const int route = K(conv_gemm_route)(p, kg(kdt));
  const int kid = route == 0 ? KID_HALO : route == 1 ? KID_GEMM256 : route == 3 ? KID_GCONV_X2 : kid_of_cfg[K(conv_gemm_v2_config)(p)];
  return timed(h, st, layer, kid, flops, [&]() { return K(launch_conv_gemm)(p, kg(kdt), st); });
}

// xsrc2 != nullptr: images [0, B1) come from x, images [B1, Bc) from xsrc2 (the shot and query tensors of one MetaBaseline call: one
// pass over cat([shot, query]) as the reference does, meta_baseline.py:29-32, without materialising the concatenation)
int forward_chunk(fsvit_visformer* h, const float* x, int Bc, float* feat, unsigned char* ws, bool first, hipStream_t st,
                  const float* xsrc2 = nullptr, int B1 = 0) {
  const Plan pl = make_plan(h, Bc);
  const int kdt = h->dtype, dt = kd(kdt);
  const size_t es = h->es;
  const int heads = h->cfg.num_heads;
  void *patches = ws + pl.patches, *c1 = ws + pl.c1, *ident = ws + pl.ident, *c2 = ws + pl.c2, *c3 = ws + pl.c3;
  void *x1 = ws + pl.x1, *x1b = ws + pl.x1b, *x2 = ws + pl.x2, *x3 = ws + pl.x3, *ha = ws + pl.ha, *hb = ws + pl.hb;
  void *qkv = ws + pl.qkv, *ctx = ws + pl.ctx, *hid = ws + pl.hid;
  const int img = h->cfg.img_size;

  // stem: conv1 / downsample as K=32 GEMMs over the im2col rows, conv2, conv3 (+identity, LeakyReLU), max-pool + pos1
  h->prof_last = nullptr;
  if (!xsrc2) B1 = Bc;
  constexpr bool stem_fused_on = true;
  // Round 6: when the whole stem runs on its dedicated kernels (stem_conv1 -> conv3x3_halo -> conv3x3_halo + tail) the two intermediate maps c1 / c2 are
  // stored row-chunk-planar (conv_gemm.h x_planar): the halo fetch of the consumer then reads row segments instead of 16 bytes of every 128-byte line
  ConvGemmParams p2 = conv_params(h->conv2, c1, c2, Bc, h->H0, h->H0, h->C0, h->C0, 3, 3, 1, 1, h->C1, ACT_LRELU, nullptr, 0, nullptr);
  ConvGemmParams p3 = conv_params(h->conv3f, c2, x1, Bc, h->H0, h->H0, h->C1, h->C1, 3, 3, 1, 1, h->C1, ACT_LRELU, nullptr, 0, h->pos1);
  p3.x2 = patches; p3.x2_cstride = 32; p3.K2 = 32; p3.pool2 = 1;
  const bool planar = stem_fused_on && K(stem_conv1_supported)(dt, img, h->C0) && K(conv_gemm_route)(p2, kg(kdt)) == 0 && K(conv_gemm_route)(p3, kg(kdt)) == 0;
  p2.x_planar = p2.y_planar = p3.x_planar = planar ? 1 : 0;
  if (planar) {
    // The dedicated stem runs in SLICES of the chunk (stem_conv1 -> conv2 -> conv3 + tail per slice of STEM_SLICE images): the two halo convs take
    // ~10 % less per image at 1600 ... 3200-image launches than at 12 800 (measured per launch size, profiles/r06_*; their halo fetch is bound by HBM
    // latency, and a slice's maps are produced and consumed within a few ms), while every later kernel wants the whole chunk.
    const int S = stem_slice_images();
    const size_t s_pat = (size_t)h->H0 * h->H0 * 32 * es, s_c1 = (size_t)h->H0 * h->H0 * h->C0 * es, s_c2 = (size_t)h->H0 * h->H0 * h->C1 * es,
                 s_x1 = (size_t)h->H1 * h->H1 * h->C1 * es, s_img = (size_t)3 * img * img * sizeof(float);
    const double fl1 = 2.0 * 27.0 * h->C0 * h->H0 * h->H0;
    for (int g0 = 0; g0 < Bc; g0 += S) {
      const int g1 = g0 + S < Bc ? g0 + S : Bc;
      for (int part = 0; part < 2; ++part) {               // images [g0, g1) of cat([shot, query]): [0, B1) from x, [B1, Bc) from xsrc2
        const int a = part == 0 ? g0 : (g0 > B1 ? g0 : B1), b = part == 0 ? (g1 < B1 ? g1 : B1) : g1;
        if (a >= b) continue;
        const float* src = part == 0 ? x + (size_t)a * 3 * img * img : xsrc2 + (size_t)(a - B1) * 3 * img * img;
        (void)s_img;
        RC_TRY(timed(h, st, "stem.im2col+conv1", KID_STEMCONV1, fl1 * (b - a), [&]() {
          return K(launch_stem_conv1)(src, (unsigned char*)patches + a * s_pat, (unsigned char*)c1 + a * s_c1, h->conv1.w, h->conv1.Kw, h->conv1.bias, b - a, st, 1); }));
      }
      ConvGemmParams q2 = p2, q3 = p3;
      q2.x = (const unsigned char*)c1 + g0 * s_c1; q2.y = (unsigned char*)c2 + g0 * s_c2; q2.B = g1 - g0; q2.M = (g1 - g0) * q2.OH * q2.OW;
      q3.x = (const unsigned char*)c2 + g0 * s_c2; q3.x2 = (const unsigned char*)patches + g0 * s_pat; q3.y = (unsigned char*)x1 + g0 * s_x1;
      q3.B = g1 - g0; q3.M = (g1 - g0) * q3.OH * q3.OW;
      RC_TRY(run_gemm(h, st, "stem.conv2", h->conv2, q2, h->C1, 9.0 * h->C0));
      RC_TRY(run_gemm(h, st, "stem.conv3+down+pool", h->conv3f, q3, h->C1, 9.0 * h->C1 + 27.0));
    }
  } else {
  if (stem_fused_on && K(stem_conv1_supported)(dt, img, h->C0)) {      // im2col + conv1 + bn1 + LeakyReLU in one pass over the image
      const double fl1 = 2.0 * 27.0 * h->C0 * h->H0 * h->H0;
      RC_TRY(timed(h, st, "stem.im2col+conv1", KID_STEMCONV1, fl1 * B1, [&]() {
        return K(launch_stem_conv1)(x, patches, c1, h->conv1.w, h->conv1.Kw, h->conv1.bias, B1, st, planar); }));
      if (Bc > B1)
        RC_TRY(timed(h, st, "stem.im2col+conv1", KID_STEMCONV1, fl1 * (Bc - B1), [&]() {
          return K(launch_stem_conv1)(xsrc2, (unsigned char*)patches + (size_t)B1 * h->H0 * h->H0 * 32 * es, (unsigned char*)c1 + (size_t)B1 * h->H0 * h->H0 * h->C0 * es,
                                   h->conv1.w, h->conv1.Kw, h->conv1.bias, Bc - B1, st, planar); }));
    } else {
      RC_TRY(timed(h, st, "stem.im2col", KID_IM2COL, 0.0, [&]() { return K(launch_im2col27)(x, patches, B1, img, img, h->H0, h->H0, dt, st); }));
      if (Bc > B1)
        RC_TRY(timed(h, st, "stem.im2col", KID_IM2COL, 0.0, [&]() {
          return K(launch_im2col27)(xsrc2, (unsigned char*)patches + (size_t)B1 * h->H0 * h->H0 * 32 * es, Bc - B1, img, img, h->H0, h->H0, dt, st); }));
      RC_TRY(run_gemm(h, st, "stem.conv1", h->conv1, conv_params(h->conv1, patches, c1, Bc, h->H0, h->H0, 32, 32, 1, 1, 1, 0, h->C0, ACT_LRELU, nullptr, 0, nullptr), h->C0, 27));
    }
    RC_TRY(run_gemm(h, st, "stem.conv2", h->conv2, p2, h->C1, 9.0 * h->C0));
    constexpr bool split_stem = false;
    if (!split_stem) {
      // conv3 + bn3 + (downsample conv + bn_d as a tail K slice over the im2col rows) + LeakyReLU + MaxPool2d(2) + pos_embed1
      RC_TRY(run_gemm(h, st, "stem.conv3+down+pool", h->conv3f, p3, h->C1, 9.0 * h->C1 + 27.0));
    } else {
      RC_TRY(run_gemm(h, st, "stem.downsample", h->down, conv_params(h->down, patches, ident, Bc, h->H0, h->H0, 32, 32, 1, 1, 1, 0, h->C1, ACT_NONE, nullptr, 0, nullptr), h->C1, 27));
      RC_TRY(run_gemm(h, st, "stem.conv3", h->conv3, conv_params(h->conv3, c2, c3, Bc, h->H0, h->H0, h->C1, h->C1, 3, 3, 1, 1, h->C1, ACT_LRELU, ident, 1, nullptr), h->C1, 9.0 * h->C1));
      RC_TRY(timed(h, st, "stem.maxpool", KID_MAXPOOL, 0.0, [&]() { return K(launch_maxpool2_pos)(c3, h->pos1, x1, Bc, h->H1, h->H1, h->C1, dt, st); }));
    }
  }
  RC_TRY(tap(h, "stem", x1, (size_t)Bc * h->H1 * h->H1 * h->C1 * es, first, st));

  // stage 1: x += conv3(GELU(conv2_g(GELU(conv1(BN(x))))))
  const int Cg = h->hid1 / h->cfg.group;
  constexpr bool no_fuse = false;
  // one LDS-resident kernel per block (stage1_w4.hip / stage1_ring.hip, any map up to 20 wide); other geometries / numerics modes and FSVIT_NO_FUSE=1 take the three-launch route
  const bool ring_ok = K(stage1_ring_supported)(dt, h->C1, h->hid1, h->cfg.group, h->H1) && h->s1.size() && h->s1[0].c2.Kw == 320;
  const bool fuse1 = !no_fuse && ring_ok && K(stage1_ring_preferred)();
  for (size_t i = 0; i < h->s1.size(); ++i) {
    const Block1& b = h->s1[i];
    if (fuse1) {   // one LDS-resident kernel per block, ping-pong between x1 and x1b
      const double fl = 2.0 * Bc * h->H1 * h->H1 * ((double)h->hid1 * h->C1 + (double)h->hid1 * 9 * Cg + (double)h->C1 * h->hid1);
      RC_TRY(timed(h, st, "stage1.block", KID_STAGE1RING, fl,
                   [&]() { return K(launch_stage1_ring)(x1, x1b, b.c1.w, b.c1.bias, b.c2.w, b.c3.w, Bc, h->H1, h->H1, st, b.c3s); }));
      std::swap(x1, x1b);
    } else {
      RC_TRY(run_gemm(h, st, "stage1.mlp.conv1", b.c1, conv_params(b.c1, x1, ha, Bc, h->H1, h->H1, h->C1, h->C1, 1, 1, 1, 0, h->hid1, ACT_GELU, nullptr, 0, nullptr), h->hid1, h->C1));
      RC_TRY(run_gemm(h, st, "stage1.mlp.conv2", b.c2, conv_params(b.c2, ha, hb, Bc, h->H1, h->H1, Cg, h->hid1, 3, 3, 1, 1, h->hid1, ACT_GELU, nullptr, 0, nullptr), Cg, 9.0 * Cg));
      RC_TRY(run_gemm(h, st, "stage1.mlp.conv3", b.c3, conv_params(b.c3, hb, x1, Bc, h->H1, h->H1, h->hid1, h->hid1, 1, 1, 1, 0, h->C1, ACT_NONE, x1, 0, nullptr), h->C1, h->hid1));
    }
    RC_TRY(tap(h, "stage1." + std::to_string(i), x1, (size_t)Bc * h->H1 * h->H1 * h->C1 * es, first, st));
  }

  // stages 2 and 3
  for (int s = 2; s <= 3; ++s) {
    const int Ci = s == 2 ? h->C1 : h->C2, C = s == 2 ? h->C2 : h->C3;
    const int Hi = s == 2 ? h->H1 : h->H2, Ho = s == 2 ? h->H2 : h->H3;
    const int hidc = s == 2 ? h->hid2 : h->hid3, hd = s == 2 ? h->hd2 : h->hd3, hdp = s == 2 ? h->hdp2 : h->hdp3;
    void* xin = s == 2 ? x1 : x2;
    void* xs = s == 2 ? x2 : x3;
    const Layer& pe = s == 2 ? h->pe2 : h->pe3;
    const float* pos = s == 2 ? h->pos2 : h->pos3;
    const size_t xbytes = (size_t)Bc * Ho * Ho * C * es;
    const std::string sp = "stage" + std::to_string(s);
    if (h->pe_img[s - 2]) {
      RC_TRY(timed(h, st, s == 2 ? "patch_embed2" : "patch_embed3", KID_LNGEMM, 2.0 * Bc * Ho * Ho * 4.0 * Ci * C,
                   [&]() { return K(launch_patch_embed_rows)(xin, xs, h->pe_img[s - 2], pe.bias, pos, Bc, Hi, Ci, C, st); }));
    } else
    RC_TRY(run_gemm(h, st, s == 2 ? "patch_embed2" : "patch_embed3", pe, conv_params(pe, xin, xs, Bc, Hi, Hi, Ci, Ci, 2, 2, 2, 0, C, ACT_NONE, nullptr, 0, pos), C, 4.0 * Ci));
    RC_TRY(tap(h, "patch_embed" + std::to_string(s), xs, xbytes, first, st));
    const std::vector<BlockA>& blocks = s == 2 ? h->s2 : h->s3;
    const float scale = 1.0f / std::sqrt((float)hd);                       // head_dim ** -0.5 (visformer.py:174)
    const int S = Ho * Ho;
    for (size_t i = 0; i < blocks.size(); ++i) {
      const BlockA& b = blocks[i];
      if (b.qa_img) {     // qkv conv + attention core in one launch: q / k / v never leave the chip
        RC_TRY(timed(h, st, (sp + ".attn.qkv+core").c_str(), KID_QKVATTN, 2.0 * Bc * S * (3.0 * heads * hd) * C + 4.0 * Bc * heads * (double)S * S * hd,
                     [&]() { return K(launch_qkv_attn)(xs, ctx, b.qa_img, b.qkv.bias, Bc, S, scale, st); }));
      } else if (b.qar_img) {
        RC_TRY(timed(h, st, (sp + ".attn.qkv+core").c_str(), KID_QKVATTNROWS, 2.0 * Bc * S * (3.0 * heads * hd) * C + 4.0 * Bc * heads * (double)S * S * hd,
                     [&]() { return K(launch_qkv_attn_rows)(xs, ctx, b.qar_img, b.qkv.bias, Bc, S, C, heads, hdp, scale, st); }));
      } else {
      if (b.qr_img) {
        RC_TRY(timed(h, st, (sp + ".attn.qkv").c_str(), KID_LNGEMM, 2.0 * Bc * S * (3.0 * heads * hd) * C,
                     [&]() { return K(launch_gemm_rows)(xs, qkv, b.qr_img, b.qkv.bias, Bc * S, C, 3 * heads * hdp, st); }));
      } else
      RC_TRY(run_gemm(h, st, (sp + ".attn.qkv").c_str(), b.qkv, conv_params(b.qkv, xs, qkv, Bc, Ho, Ho, C, C, 1, 1, 1, 0, 3 * heads * hdp, ACT_NONE, nullptr, 0, nullptr), 3.0 * heads * hd, C));
      RC_TRY(timed(h, st, (sp + ".attn.core").c_str(), KID_ATTN, 4.0 * Bc * heads * (double)S * S * hd,
                   [&]() { return K(launch_attention)(qkv, ctx, Bc, S, heads, hdp, scale, is_x2(h->dtype) ? 2 : dt, st); }));
      }
      if (b.mlp_img && b.mlp_kc) {   // proj + residual + conv1 + GELU + conv3 + residual in one launch
        RC_TRY(timed(h, st, (sp + ".proj+mlp").c_str(), KID_MLPROWS, 2.0 * Bc * S * ((double)C * heads * hd + 2.0 * hidc * C),
                     [&]() { return K(launch_mlp_rows)(xs, xs, b.mlp_img, b.mlp_b1, b.fc2.bias, ctx, b.mlp_kc, Bc * S, C, hidc, st); }));
        RC_TRY(tap(h, sp + "." + std::to_string(i), xs, xbytes, first, st));
        continue;
      }
      RC_TRY(run_gemm(h, st, (sp + ".attn.proj").c_str(), b.proj, conv_params(b.proj, ctx, xs, Bc, Ho, Ho, heads * hdp, heads * hdp, 1, 1, 1, 0, C, ACT_NONE, xs, 0, nullptr), C, (double)heads * hd));
      if (b.mlp_img) {     // conv1 + GELU + conv3 + residual in one launch, the hidden map stays in registers
        RC_TRY(timed(h, st, (sp + ".mlp").c_str(), KID_MLPROWS, 4.0 * Bc * S * (double)hidc * C,
                     [&]() { return K(launch_mlp_rows)(xs, xs, b.mlp_img, b.mlp_b1, b.fc2.bias, nullptr, 0, Bc * S, C, hidc, st); }));
      } else {
        RC_TRY(run_gemm(h, st, (sp + ".mlp.conv1").c_str(), b.fc1, conv_params(b.fc1, xs, hid, Bc, Ho, Ho, C, C, 1, 1, 1, 0, hidc, ACT_GELU, nullptr, 0, nullptr), hidc, C));
        RC_TRY(run_gemm(h, st, (sp + ".mlp.conv3").c_str(), b.fc2, conv_params(b.fc2, hid, xs, Bc, Ho, Ho, hidc, hidc, 1, 1, 1, 0, C, ACT_NONE, xs, 0, nullptr), C, hidc));
      }
      RC_TRY(tap(h, sp + "." + std::to_string(i), xs, xbytes, first, st));
    }
  }
  RC_TRY(timed(h, st, "norm.pool", KID_POOL, 0.0, [&]() { return K(launch_pool_affine)(x3, h->fscale, h->fshift, feat, Bc, h->H3 * h->H3, h->C3, dt, st); }));
  return 0;
}

}  // namespace

// ------------------------------------------------------------------------------------ C ABI
extern "C" int fsvit_visformer_create(const fsvit_visformer_cfg* cfg, const fsvit_tensor* state_dict, int n_tensors,
                                      int dtype, fsvit_visformer** out) {
  if (!cfg || !state_dict || !out || n_tensors <= 0) return fail(FSVIT_ERR_ARG, "null argument");
  if (!known_dtype(dtype)) return fail(FSVIT_ERR_ARG, "unknown dtype %d", dtype);
  if (cfg->num_heads < 1 || cfg->embed_dim < 2 || cfg->init_channels < 1 || cfg->depth[0] < 0 || cfg->depth[1] < 0 || cfg->depth[2] < 0)
    return fail(FSVIT_ERR_ARG, "bad Visformer configuration");
  fsvit_visformer* h = new fsvit_visformer();
  h->kind = KIND_VISFORMER;
  h->cfg = *cfg;
  h->dtype = dtype;
  h->es = storage_bytes(dtype);
  SD sd{state_dict, n_tensors};
  int rc = build(h, sd);
  if (rc != 0) { fsvit_visformer_destroy(h); return rc; }
  *out = h;
  return 0;
}

extern "C" void fsvit_visformer_destroy(fsvit_visformer* h) {
  if (!h) return;
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
}

extern "C" int fsvit_visformer_out_dim(const fsvit_visformer* h) { return h ? h->C3 : 0; }
extern "C" int fsvit_visformer_dtype(const fsvit_visformer* h) { return h ? h->dtype : -1; }

extern "C" size_t fsvit_visformer_workspace_bytes(const fsvit_visformer* h, int chunk_images) {
  if (!h || chunk_images <= 0) return 0;
  return make_plan(h, (size_t)chunk_images).total;
}

extern "C" int fsvit_encoder_set_tap(void* hv, const char* name, void* dst_dev, size_t bytes) {
  EngineBase* h = static_cast<EngineBase*>(hv);
  if (!h || !name) return fail(FSVIT_ERR_ARG, "null argument");
  if (!dst_dev) h->taps.erase(name);
  else h->taps[name] = Tap{dst_dev, bytes};
  return 0;
}

extern "C" int fsvit_visformer_forward(fsvit_visformer* h, const float* x, int n_img, int img_h, int img_w,
                                       float* feat, void* ws, size_t ws_bytes, void* stream) {
  if (h && n_img <= 0) return 0;
  if (!h || !x || !feat || !ws) return fail(FSVIT_ERR_ARG, "null argument");
  if (img_h != h->cfg.img_size || img_w != h->cfg.img_size)   // PatchEmbed assert / pos_embed mismatch (visformer.py:283-284,431)
    return fail(FSVIT_ERR_IMG_SIZE, "Input image size (%d*%d) does not match model (%d*%d).", img_h, img_w, h->cfg.img_size, h->cfg.img_size);
  if (n_img <= 0) return 0;
  if (((uintptr_t)ws & 255) != 0) return fail(FSVIT_ERR_ARG, "workspace must be 256-byte aligned");
  // largest chunk that fits the workspace (plan size is monotone in the chunk)
  int lo = 0, hi = n_img;
  while (lo < hi) {
    int mid = (lo + hi + 1) / 2;
    if (make_plan(h, (size_t)mid).total <= ws_bytes) lo = mid; else hi = mid - 1;
  }
  if (lo < 1) return fail(FSVIT_ERR_WORKSPACE, "workspace of %zu bytes cannot hold one image (need %zu)", ws_bytes, make_plan(h, 1).total);
  const int chunk = lo;
  hipStream_t st = (hipStream_t)stream;
  const size_t img_elems = (size_t)3 * img_h * img_w;
  for (int off = 0; off < n_img; off += chunk) {
    const int bc = n_img - off < chunk ? n_img - off : chunk;
    int rc = forward_chunk(h, x + (size_t)off * img_elems, bc, feat + (size_t)off * h->C3, (unsigned char*)ws, off == 0, st);
    if (rc != 0) return rc;
  }
  return 0;
}

/* Post-norm token map [n_img][H3*H3][C3] fp32 of the images of the LAST fsvit_visformer_forward call on this handle (same workspace,
 * untouched since; the call must have fitted one chunk): the `x` of `return x, pooled` in the distillation phase's encoder
 * (sun_meta_training/models/visformer.py:464) for the eval-mode teacher. */
extern "C" int fsvit_visformer_last_tokens(fsvit_visformer* h, const void* ws, size_t ws_bytes, int n_img, float* tokens, void* stream) {
  const int kdt = h ? h->dtype : FSVIT_BF16;
  if (!h || !ws || !tokens || n_img <= 0) return fail(FSVIT_ERR_ARG, "bad argument");
  const Plan pl = make_plan(h, (size_t)n_img);
  if (pl.total > ws_bytes) return fail(FSVIT_ERR_WORKSPACE, "last_tokens: the forward of %d images did not fit one chunk of this workspace", n_img);
  RC_TRY(K(launch_tokens_to_f32)((const unsigned char*)ws + pl.x3, h->fscale, h->fshift, tokens, (size_t)n_img * h->H3 * h->H3 * h->C3, h->C3, kd(kdt),
                              (hipStream_t)stream));
  return 0;
}

extern "C" int fsvit_proto_head(const float* fs, const float* fq, int E, int way, int shot, int Q, int D, float temp,
                                int method, float* logits, float* acc, float* loss, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!fs || !fq || !logits) return fail(FSVIT_ERR_ARG, "null argument");
  if (method != FSVIT_HEAD_COS && method != FSVIT_HEAD_SQR && method != FSVIT_HEAD_DOT) return fail(FSVIT_ERR_ARG, "unknown head method %d", method);
  RC_TRY(K(launch_proto_head)(fs, fq, E, way, shot, Q, D, temp, method, logits, acc, loss, (hipStream_t)stream, nullptr));
  return 0;
}

extern "C" int fsvit_proto_head_devtemp(const float* fs, const float* fq, int E, int way, int shot, int Q, int D, const float* temp_dev,
                                        int method, float* logits, float* acc, float* loss, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!fs || !fq || !logits || !temp_dev) return fail(FSVIT_ERR_ARG, "null argument");
  if (method != FSVIT_HEAD_COS && method != FSVIT_HEAD_SQR && method != FSVIT_HEAD_DOT) return fail(FSVIT_ERR_ARG, "unknown head method %d", method);
  RC_TRY(K(launch_proto_head)(fs, fq, E, way, shot, Q, D, 0.f, method, logits, acc, loss, (hipStream_t)stream, temp_dev));
  return 0;
}

static int encoder_forward_any(void* hv, const float* x, int n, int img_h, int img_w, float* feat, void* ws, size_t ws_bytes, void* stream);
static int encoder_out_dim_any(void* hv);

extern "C" int fsvit_meta_baseline_forward(void* hv, const float* x_shot, const float* x_query, int E, int way,
                                           int shot, int Q, int img_h, int img_w, float temp, int method, float* logits,
                                           float* acc, float* loss, float* feat, void* ws, size_t ws_bytes, void* stream) {
  if (!hv || !feat) return fail(FSVIT_ERR_ARG, "null argument");
  const int ns = E * way * shot, nq = E * Q, D = encoder_out_dim_any(hv);
  // the reference encodes cat([shots, queries]) in one call (meta_baseline.py:29-32); eval mode is
  // per-image independent, so two passes into one feature buffer are equivalent
  {   // Visformer: ONE pass over shots + queries when the workspace holds them all (larger launches, fewer tile tails)
    EngineBase* eb = static_cast<EngineBase*>(hv);
    if (eb->kind == KIND_VISFORMER && x_shot && x_query && ns > 0 && nq > 0) {
      fsvit_visformer* h = static_cast<fsvit_visformer*>(eb);
      if (img_h != h->cfg.img_size || img_w != h->cfg.img_size)
        return fail(FSVIT_ERR_IMG_SIZE, "Input image size (%d*%d) does not match model (%d*%d).", img_h, img_w, h->cfg.img_size, h->cfg.img_size);
      if (ws && ((uintptr_t)ws & 255) == 0 && make_plan(h, (size_t)ns + nq).total <= ws_bytes) {
        int rc1 = forward_chunk(h, x_shot, ns + nq, feat, (unsigned char*)ws, true, (hipStream_t)stream, x_query, ns);
        if (rc1 != 0) return rc1;
        return fsvit_proto_head(feat, feat + (size_t)ns * D, E, way, shot, Q, D, temp, method, logits, acc, loss, stream);
      }
    }
  }
  int rc = encoder_forward_any(hv, x_shot, ns, img_h, img_w, feat, ws, ws_bytes, stream);
  if (rc != 0) return rc;
  rc = encoder_forward_any(hv, x_query, nq, img_h, img_w, feat + (size_t)ns * D, ws, ws_bytes, stream);
  if (rc != 0) return rc;
  return fsvit_proto_head(feat, feat + (size_t)ns * D, E, way, shot, Q, D, temp, method, logits, acc, loss, stream);
}

extern "C" int fsvit_conv_gemm(const void* x, const void* w, const float* bias, const void* res, const float* pos, void* y,
                               int B, int H, int W, int Cin, int x_cstride, int KH, int KW, int stride, int pad, int N,
                               int y_cstride, int Kw, int groups, int act, int res_first, int dtype, void* stream) {
  const int kdt = dtype;
  if (!known_dtype(dtype)) return fail(FSVIT_ERR_ARG, "unknown dtype %d", dtype);
  if (!x || !w || !y) return fail(FSVIT_ERR_ARG, "null argument");
  const int es = storage_bytes(dtype), epc = 16 / es, bke = 128 / es;
  if (Cin % epc || x_cstride % epc || N % 4 || y_cstride % 4 || Kw % bke || Kw < KH * KW * Cin)
    return fail(FSVIT_ERR_ARG, "conv_gemm alignment: Cin/x_cstride %% %d, N/y_cstride %% 4, Kw %% %d", epc, bke);
  if (KH * KW > 1 && !is_pow2(Cin)) return fail(FSVIT_ERR_ARG, "multi-tap conv needs power-of-two Cin");
  Layer L; L.w = const_cast<void*>(w); L.bias = const_cast<float*>(bias); L.N = N; L.K = KH * KW * Cin; L.Kw = Kw; L.groups = groups;
  ConvGemmParams p = conv_params(L, x, y, B, H, W, Cin, x_cstride, KH, KW, stride, pad, y_cstride, act, res, res_first, pos);
  RC_TRY(K(launch_conv_gemm)(p, kg(dtype), (hipStream_t)stream));
  return 0;
}

/* The stem's fused tail (visformer.py:213,224-237 + :431): out = MaxPool2d(2)(LeakyReLU(conv3x3(x, w[:, :9*Cin]) + x2 . w[:, Kmain:] + bias)) + pos
 * - conv3 + bn3 with the downsample conv + bn_d riding as ONE extra K slice over the im2col rows x2, pooled and position-embedded in the
 * epilogue.  Routed exactly as the engine routes it (conv3x3_halo_kernel<128,true> for the Visformer-S geometry, conv_gemm_v2 otherwise). */
extern "C" int fsvit_conv_stem_tail(const void* x, const void* w, const float* bias, const float* pos, const void* x2, int x2_cstride, int K2,
                                    void* y, int B, int H, int W, int Cin, int N, int Kw, int dtype, void* stream) {
  const int kdt = dtype;
  if (!known_dtype(dtype)) return fail(FSVIT_ERR_ARG, "unknown dtype %d", dtype);
  if (!x || !w || !y || !x2 || !pos) return fail(FSVIT_ERR_ARG, "null argument");
  const int es = storage_bytes(dtype), epc = 16 / es, bke = 128 / es;
  if (Cin % epc || !is_pow2(Cin) || N % 4 || Kw % bke || Kw < 9 * Cin + bke || K2 > bke || x2_cstride < K2 || (H & 1) || (W & 1))
    return fail(FSVIT_ERR_ARG, "fsvit_conv_stem_tail: bad geometry");
  Layer L; L.w = const_cast<void*>(w); L.bias = const_cast<float*>(bias); L.N = N; L.K = 9 * Cin; L.Kw = Kw; L.groups = 1;
  ConvGemmParams p = conv_params(L, x, y, B, H, W, Cin, Cin, 3, 3, 1, 1, N, ACT_LRELU, nullptr, 0, pos);
  p.x2 = x2; p.x2_cstride = x2_cstride; p.K2 = K2; p.pool2 = 1;
  RC_TRY(K(launch_conv_gemm)(p, kg(dtype), (hipStream_t)stream));
  return 0;
}

extern "C" int fsvit_stage1_block(const void* x, void* y, const void* w1, const float* b1, const void* w2, const void* w3, int B, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !y || !w1 || !b1 || !w2 || !w3 || x == y || B <= 0) return fail(FSVIT_ERR_ARG, "bad argument");
  RC_TRY(K(launch_stage1_ring16)(x, y, w1, b1, w2, w3, B, 20, 20, (hipStream_t)stream));
  return 0;
}

extern "C" int fsvit_stage1_block_hw(const void* x, void* y, const void* w1, const float* b1, const void* w2, const void* w3, int B, int H, int W, int dtype,
                                     void* stream) {
  const int kdt = dtype;
  if (!x || !y || !w1 || !b1 || !w2 || !w3 || x == y || B <= 0) return fail(FSVIT_ERR_ARG, "bad argument");
  if (dtype != FSVIT_BF16 && dtype != FSVIT_F16) return fail(FSVIT_ERR_ARG, "fsvit_stage1_block_hw: 16-bit storage only (bf16 / f16)");
  if (H != W || !K(stage1_ring_supported)(1, 128, 256, 8, W)) return fail(FSVIT_ERR_ARG, "fsvit_stage1_block_hw: square maps of 4 .. 20 tokens a side");
  RC_TRY(K(launch_stage1_ring)(x, y, w1, b1, w2, w3, B, H, W, (hipStream_t)stream, nullptr));
  return 0;
}

extern "C" int fsvit_conv3x3_wgrad(const void* x, const void* dz, float* dw, int B, int H, int W, int O, int Ig, int groups, int dtype, void* stream) {
  const int kdt = dtype;
  if (!x || !dz || !dw) return fail(FSVIT_ERR_ARG, "null argument");
  if (dtype != FSVIT_BF16 && dtype != FSVIT_F16 && !is_x2(dtype))
    return fail(FSVIT_ERR_ARG, "fsvit_conv3x3_wgrad: bf16 / f16 activations, or fp32 activations with two-limb arithmetic (FSVIT_BF16X2 / FSVIT_F16X2)");
  if (!K(wgrad3x3_supported)(kg(dtype), O, Ig, groups, W)) return fail(FSVIT_ERR_ARG, "fsvit_conv3x3_wgrad: built for 8 groups of 32 -> 32 channels (W <= 20) and dense 64 / 128 -> 128 (W <= 40)");
  hipStream_t st = (hipStream_t)stream;
  void* scratch = nullptr;
  HIP_TRY(hipMalloc(&scratch, K(wgrad3x3_scratch_bytes)(O, Ig, groups, B * H * W, kg(dtype))));
  int rc = K(launch_wgrad3x3)(x, groups * Ig, dz, O, dw, (float*)scratch, B, H, W, O, Ig, groups, st, nullptr, kg(dtype));
  (void)hipStreamSynchronize(st);
  (void)hipFree(scratch);
  if (rc != 0) return hipfail((hipError_t)rc, "fsvit_conv3x3_wgrad");
  return 0;
}

extern "C" int fsvit_gconv3x3(const void* x, const void* w_packed, int Kw, void* y, int B, int H, int W, int dtype, void* stream) {
  const int kdt = dtype;
  if (!x || !w_packed || !y || B <= 0) return fail(FSVIT_ERR_ARG, "bad argument");
  if (dtype != FSVIT_BF16 && dtype != FSVIT_F16) return fail(FSVIT_ERR_ARG, "fsvit_gconv3x3: 16-bit storage only (bf16 / f16)");
  if (!K(gconv3x3_supported)(1, 256, 32, 8, 3, 3, 1, 1, W) || Kw < 288 || (Kw & 7)) return fail(FSVIT_ERR_ARG, "fsvit_gconv3x3: 8 groups of 32 -> 32 channels, W <= 20, Kw >= 288");
  RC_TRY(K(launch_gconv3x3)(x, w_packed, Kw, y, B, H, W, (hipStream_t)stream, nullptr, nullptr));
  return 0;
}

extern "C" int fsvit_conv1x1_wgrad(const void* x, const void* dz, float* dw, int M, int N, int C, int dtype, void* stream) {
  const int kdt = dtype;
  if (!x || !dz || !dw || M <= 0) return fail(FSVIT_ERR_ARG, "bad argument");
  if (dtype != FSVIT_BF16 && dtype != FSVIT_F16 && !is_x2(dtype))
    return fail(FSVIT_ERR_ARG, "fsvit_conv1x1_wgrad: bf16 / f16 rows, or fp32 rows with two-limb arithmetic (FSVIT_BF16X2 / FSVIT_F16X2)");
  if (!K(wgrad1x1_supported)(kg(dtype), N, C)) return fail(FSVIT_ERR_ARG, "fsvit_conv1x1_wgrad: N and C must be multiples of 8");
  hipStream_t st = (hipStream_t)stream;
  const int splits = K(wgrad1x1_splits)(N, C, M, kg(dtype)), Kc_pad = (C + 3) / 4 * 4;
  void* ysp = nullptr;
  HIP_TRY(hipMalloc(&ysp, (size_t)((N + 3) / 4 * 4) * splits * Kc_pad * 4));
  int rc = K(launch_wgrad1x1)(x, C, C, dz, N, N, (float*)ysp, M, Kc_pad, st, kg(dtype));
  if (rc == 0) rc = fsvit::launch_wgrad_finalize((const float*)ysp, dw, N, C, 1, 1, 0, splits, Kc_pad, 1, 1, 1, 1, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(ysp);
  if (rc != 0) return hipfail((hipError_t)rc, "fsvit_conv1x1_wgrad");
  return 0;
}

extern "C" int fsvit_mlp_rows(const void* x, void* y, const void* w1, int k1w, const float* b1, const void* w2, int k2w, const float* b2,
                              int M, int C, int hid, void* stream) {
  return fsvit_proj_mlp_rows(x, y, nullptr, nullptr, 0, 0, w1, k1w, b1, w2, k2w, b2, M, C, hid, stream);
}

extern "C" int fsvit_proj_mlp_rows(const void* x, void* y, const void* ctx, const void* wp, int kpw, int KC, const void* w1, int k1w, const float* b1,
                                   const void* w2, int k2w, const float* b2, int M, int C, int hid, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !y || !w1 || !w2) return fail(FSVIT_ERR_ARG, "null argument");
  if (!K(mlp_rows_supported)(1, C, hid)) return fail(FSVIT_ERR_ARG, "fsvit_mlp_rows: only C = 256 / hidden = 1024 and C = 512 / hidden = 2048 (bf16) are built");
  if (k1w < C || k2w < hid) return fail(FSVIT_ERR_ARG, "weight rows shorter than K");
  if (ctx && (!wp || kpw < KC || !K(mlp_rows_proj_supported)(C, hid, KC))) return fail(FSVIT_ERR_ARG, "proj fusion: (C, KC) must be (256, 288) or (512, 576)");
  if (!ctx) KC = 0;
  hipStream_t st = (hipStream_t)stream;
  void *img = nullptr, *b1i = nullptr;
  HIP_TRY(hipMalloc(&img, K(mlp_rows_image_bytes)(C, hid, KC)));
  hipError_t e = hipMalloc(&b1i, (size_t)hid * 4);
  if (e != hipSuccess) { (void)hipFree(img); return hipfail(e, "hipMalloc"); }
  int rc = K(launch_mlp_pack)(w1, k1w, b1, w2, k2w, wp, kpw, KC, img, (float*)b1i, C, hid, st);
  if (rc == 0) rc = K(launch_mlp_rows)(x, y, img, (const float*)b1i, b2, ctx, KC, M, C, hid, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(img);
  (void)hipFree(b1i);
  if (rc != 0) return hipfail((hipError_t)rc, "fsvit_mlp_rows");
  return 0;
}

// The ViT / DeiT block tail as one operator (deit.py:69-72): x1 = x + bp + wp ctx; y = x1 + b2 + W2 GELU(W1 LN(x1) + b1), LN without affine
// (fold gamma / beta into w1 / b1: W1 diag(gamma), b1 + W1 beta - what the engine's packer does).  Packs the weights on every call.
extern "C" int fsvit_vit_block_tail(const void* x, void* y, const void* ctx, const void* wp, int kpw, int KC, const float* bp, const void* w1, int k1w,
                                    const float* b1, const void* w2, int k2w, const float* b2, int M, int C, int hid, float eps, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !y || !ctx || !wp || !bp || !w1 || !b1 || !w2 || !b2) return fail(FSVIT_ERR_ARG, "null argument");
  if (!K(mlp_rows_ln_supported)(1, C, hid, KC)) return fail(FSVIT_ERR_ARG, "fsvit_vit_block_tail: only C = 384 / hidden = 1536 / KC = 384 (bf16) is built");
  if (k1w < C || k2w < hid || kpw < KC) return fail(FSVIT_ERR_ARG, "weight rows shorter than K");
  hipStream_t st = (hipStream_t)stream;
  void *img = nullptr, *b1i = nullptr;
  HIP_TRY(hipMalloc(&img, K(mlp_rows_image_bytes)(C, hid, KC)));
  hipError_t e = hipMalloc(&b1i, (size_t)hid * 4);
  if (e != hipSuccess) { (void)hipFree(img); return hipfail(e, "hipMalloc"); }
  int rc = K(launch_mlp_pack)(w1, k1w, b1, w2, k2w, wp, kpw, KC, img, (float*)b1i, C, hid, st);
  if (rc == 0) rc = K(launch_mlp_rows_ln)(x, y, img, (const float*)b1i, bp, b2, ctx, KC, M, C, hid, eps, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(img);
  (void)hipFree(b1i);
  if (rc != 0) return hipfail((hipError_t)rc, "fsvit_vit_block_tail");
  return 0;
}

// LayerNorm + Linear on token rows as one operator (the DeiT block's norm1 + qkv, deit.py:40-47,:69): y [M][N] = b + W LN(x), LN without affine
// (gamma / beta folded into w / b by the caller).  bf16, C = 384, N a multiple of 32.  Packs the weights on every call.
extern "C" int fsvit_ln_linear_rows(const void* x, void* y, const void* w, int kw, const float* b, int M, int C, int N, float eps, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !y || !w || (!b && C != 512)) return fail(FSVIT_ERR_ARG, "null argument");
  const bool plain = C == 512;      // the Visformer stage-3 geometry runs the same kernel without the LayerNorm (eps ignored)
  if (!(plain ? K(gemm_rows_supported)(1, C, N) : K(ln_gemm_rows_supported)(1, C, N)) || kw < C)
    return fail(FSVIT_ERR_ARG, "fsvit_ln_linear_rows: C = 384 (LayerNorm + Linear) or 512 (Linear only), N a multiple of 32, rows of at least C weights (bf16)");
  hipStream_t st = (hipStream_t)stream;
  void* img = nullptr;
  HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(C, N)));
  int rc = K(launch_ln_gemm_pack)(w, kw, img, C, N, st);
  if (rc == 0) rc = plain ? K(launch_gemm_rows)(x, y, img, b, M, C, N, st) : K(launch_ln_gemm_rows)(x, y, img, b, M, C, N, eps, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(img);
  if (rc != 0) return hipfail((hipError_t)rc, "fsvit_ln_linear_rows");
  return 0;
}

// PatchEmbed of the Visformer stages as one operator (visformer.py:266-288: conv k2 s2 -> BN, + pos_embed): x NHWC bf16 [B][H][H][Ci] with
// 4 Ci = 512 (patch_embed2 of Visformer-S), w [N][kw >= 4 Ci] K-major in (ky, kx, c) order with the eval BatchNorm folded, bias [N] or NULL,
// pos fp32 [(H/2)^2][N]; y [B (H/2)^2][N].  Packs the weights on every call.
extern "C" int fsvit_patch_embed2x2(const void* x, void* y, const void* w, int kw, const float* bias, const float* pos, int B, int H, int Ci, int N,
                                    void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !y || !w || !pos) return fail(FSVIT_ERR_ARG, "null argument");
  if (!K(patch_embed_rows_supported)(1, Ci, H, N) || kw < 4 * Ci) return fail(FSVIT_ERR_ARG, "fsvit_patch_embed2x2: 4 Ci = 512, even H, N a multiple of 32 (bf16)");
  hipStream_t st = (hipStream_t)stream;
  void* img = nullptr;
  HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(4 * Ci, N)));
  int rc = K(launch_ln_gemm_pack)(w, kw, img, 4 * Ci, N, st);
  if (rc == 0) rc = K(launch_patch_embed_rows)(x, y, img, bias, pos, B, H, Ci, N, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(img);
  if (rc != 0) return hipfail((hipError_t)rc, "fsvit_patch_embed2x2");
  return 0;
}

// ---- distillation head (sun_meta_training/offline.py, models/token_label.py, models/classifier.py)
extern "C" int fsvit_linear_forward(const float* x, const float* w, const float* b, float* y, int M, int N, int K, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !w || !y || (K & 3) || N <= 0 || K <= 0) return fail(FSVIT_ERR_ARG, "fsvit_linear_forward: null argument or K %% 4 != 0");
  RC_TRY(K(launch_linear_fwd)(x, w, b, y, M, N, K, (hipStream_t)stream));
  return 0;
}
extern "C" int fsvit_linear_backward(const float* dy, const float* x, const float* w, float* dx, int accumulate_dx, float* dw, float* db,
                                     int M, int N, int K, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!dy || (dx && !w) || (dw && !x) || N <= 0 || K <= 0) return fail(FSVIT_ERR_ARG, "fsvit_linear_backward: bad argument");
  RC_TRY(K(launch_linear_bwd)(dy, x, w, dx, accumulate_dx, dw, db, M, N, K, (hipStream_t)stream));
  return 0;
}
extern "C" int fsvit_token_softlabel(const float* teacher_logits, float* soft, int B, int T, int C, int k, int bp, double smoothing, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!teacher_logits || !soft || T > 64 || T <= 0 || C < 2 || k < 1 || k > C || bp < 0 || bp > T)
    return fail(FSVIT_ERR_ARG, "fsvit_token_softlabel: bad argument (T <= 64, 1 <= k <= C, 0 <= bp <= T)");
  RC_TRY(K(launch_token_softlabel)(teacher_logits, soft, B, T, C, k, bp, smoothing, (hipStream_t)stream));
  return 0;
}
extern "C" int fsvit_soft_target_ce(const float* logits, const float* target, float* row_loss, float* dlogits, int R, int C, float grad_scale,
                                    void* stream) {
  const int kdt = FSVIT_BF16;
  if (!logits || !target || !row_loss || C <= 0) return fail(FSVIT_ERR_ARG, "fsvit_soft_target_ce: bad argument");
  RC_TRY(K(launch_soft_target_ce)(logits, target, row_loss, dlogits, R, C, grad_scale, (hipStream_t)stream));
  return 0;
}
extern "C" int fsvit_row_normalize(const float* x, float* y, float* inv_norm, int R, int D, void* stream) {
  if (!x || !y || !inv_norm || D <= 0) return fail(FSVIT_ERR_ARG, "fsvit_row_normalize: bad argument");
  RC_TRY(fsvit::launch_row_normalize(x, y, inv_norm, R, D, (hipStream_t)stream));
  return 0;
}
extern "C" int fsvit_row_normalize_backward(const float* y, const float* inv_norm, const float* dy, float* dx, int R, int D, void* stream) {
  if (!y || !inv_norm || !dy || !dx || D <= 0) return fail(FSVIT_ERR_ARG, "fsvit_row_normalize_backward: bad argument");
  RC_TRY(fsvit::launch_row_normalize_bwd(y, inv_norm, dy, dx, R, D, (hipStream_t)stream));
  return 0;
}
extern "C" int fsvit_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                                float weight_decay, int step, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!p || !g || !m || !v || step < 1) return fail(FSVIT_ERR_ARG, "fsvit_adamw_step: bad argument");
  RC_TRY(K(launch_adamw)(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream));
  return 0;
}

extern "C" int fsvit_adamw_step_multi(const void* items_dev, int n_items, size_t max_numel, float lr, float beta1, float beta2, float eps, float weight_decay,
                                      int step, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!items_dev || n_items < 0 || step < 1) return fail(FSVIT_ERR_ARG, "fsvit_adamw_step_multi: bad argument");
  RC_TRY(K(launch_adamw_multi)(items_dev, n_items, max_numel, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream));
  return 0;
}

extern "C" int fsvit_qkv_attention(const void* x, const void* wqkv, int kw, const float* bias, void* ctx, int B, int S, int C, int heads, int hdp,
                                   float scale, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !wqkv || !ctx) return fail(FSVIT_ERR_ARG, "null argument");
  hipStream_t st = (hipStream_t)stream;
  void* img = nullptr;
  if (C == 512) {       // the rows kernel (stage-3 geometry: head dim padded to 96, maps of at most 32 tokens)
    if (!K(qkv_attn_rows_supported)(1, C, heads, hdp, S) || kw < C)
      return fail(FSVIT_ERR_ARG, "fsvit_qkv_attention: C = 512 needs head dim 96 (padded) and S <= 32 (bf16)");
    HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(C, 3 * heads * hdp)));
    int rc = K(launch_qkv_attn_rows_pack)(wqkv, kw, img, C, heads, hdp, st);
    if (rc == 0) rc = K(launch_qkv_attn_rows)(x, ctx, img, bias, B, S, C, heads, hdp, scale, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(img);
    if (rc != 0) return hipfail((hipError_t)rc, "fsvit_qkv_attention");
    return 0;
  }
  if (!K(qkv_attn_supported)(1, C, heads, hdp, S) || kw < C)
    return fail(FSVIT_ERR_ARG, "fsvit_qkv_attention: only C = 256, 6 heads x 48 (padded), S <= 112 (bf16) is built");
  HIP_TRY(hipMalloc(&img, K(qkv_attn_image_bytes)()));
  int rc = K(launch_qkv_attn_pack)(wqkv, kw, img, st);
  if (rc == 0) rc = K(launch_qkv_attn)(x, ctx, img, bias, B, S, scale, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(img);
  if (rc != 0) return hipfail((hipError_t)rc, "fsvit_qkv_attention");
  return 0;
}

extern "C" int fsvit_vit_ln_qkv_attention(const void* x, const void* wqkv, int kw, const float* bias, void* ctx, int B, int S, int C, int heads, int hdp,
                                          float eps, float scale, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !wqkv || !ctx) return fail(FSVIT_ERR_ARG, "null argument");
  if (!K(vit_attn_rows_supported)(1, C, heads, hdp, S) || kw < C)
    return fail(FSVIT_ERR_ARG, "fsvit_vit_ln_qkv_attention: built for C = 384, head dim 64, S <= 256 (bf16)");
  hipStream_t st = (hipStream_t)stream;
  void* img = nullptr;
  HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(C, 3 * heads * hdp)));
  int rc = K(launch_qkv_attn_rows_pack)(wqkv, kw, img, C, heads, hdp, st);
  if (rc == 0) rc = K(launch_vit_attn_rows)(x, ctx, img, bias, B, S, C, heads, hdp, eps, scale, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(img);
  if (rc != 0) return hipfail((hipError_t)rc, "fsvit_vit_ln_qkv_attention");
  return 0;
}

extern "C" int fsvit_attention(const void* qkv, void* ctx, int B, int S, int heads, int hdp, float scale, int dtype, void* stream) {
  const int kdt = dtype;
  if (!known_dtype(dtype)) return fail(FSVIT_ERR_ARG, "unknown dtype %d", dtype);
  if (!qkv || !ctx) return fail(FSVIT_ERR_ARG, "null argument");
  int rc = K(launch_attention)(qkv, ctx, B, S, heads, hdp, scale, kd(dtype), (hipStream_t)stream);
  if (rc != 0) return hipfail((hipError_t)rc, "attention");
  return 0;
}

extern "C" int fsvit_im2col27(const float* x, void* out, int B, int H, int W, int dtype, void* stream) {
  const int kdt = dtype;
  if (!known_dtype(dtype)) return fail(FSVIT_ERR_ARG, "unknown dtype %d", dtype);
  if (!x || !out || (H & 1) || (W & 1)) return fail(FSVIT_ERR_ARG, "bad argument");
  RC_TRY(K(launch_im2col27)(x, out, B, H, W, H / 2, W / 2, kd(dtype), (hipStream_t)stream));
  return 0;
}

extern "C" int fsvit_stem_conv1(const float* x, const void* w, int kw, const float* bias, void* patches, void* c1, int B, int H, int W, void* stream) {
  const int kdt = FSVIT_BF16;
  if (!x || !w || !patches || !c1 || kw < 32) return fail(FSVIT_ERR_ARG, "bad argument");
  if (H != W || !K(stem_conv1_supported)(1, H, 64)) return fail(FSVIT_ERR_ARG, "fsvit_stem_conv1: only 80x80 images, 64 output channels (bf16) are built");
  RC_TRY(K(launch_stem_conv1)(x, patches, c1, w, kw, bias, B, (hipStream_t)stream, 0));
  return 0;
}

extern "C" int fsvit_maxpool2_pos(const void* in, const float* pos, void* out, int B, int OH, int OW, int C, int dtype, void* stream) {
  const int kdt = dtype;
  if (!known_dtype(dtype)) return fail(FSVIT_ERR_ARG, "unknown dtype %d", dtype);
  if (!in || !out || C % 4) return fail(FSVIT_ERR_ARG, "bad argument");
  RC_TRY(K(launch_maxpool2_pos)(in, pos, out, B, OH, OW, C, kd(dtype), (hipStream_t)stream));
  return 0;
}

extern "C" int fsvit_pool_affine(const void* x, const float* scale, const float* shift, float* feat, int B, int HW, int C,
                                 int dtype, void* stream) {
  const int kdt = dtype;
  if (!known_dtype(dtype)) return fail(FSVIT_ERR_ARG, "unknown dtype %d", dtype);
  if (!x || !scale || !shift || !feat || C % 4) return fail(FSVIT_ERR_ARG, "bad argument");
  RC_TRY(K(launch_pool_affine)(x, scale, shift, feat, B, HW, C, kd(dtype), (hipStream_t)stream));
  return 0;
}

// ------------------------------------------------------------------------------------ profiling
static bool K_w4_enabled(int kdt) { return K(stage1_w4_enabled)(); }
extern "C" const char* fsvit_kernel_name(int kernel_id, int dtype) {
  static const char* f32n[] = {"gemm256_kernel", "conv_gemm_v2_kernel<float,128,64,2,2,3>", "conv_gemm_v2_kernel<float,128,32,4,1,3>",
                               "im2col27_kernel<float>", "maxpool2_pos_kernel<float>", "attention_v2_kernel<float,...>", "pool_affine_kernel<float>", "proto_head_kernel", "stage1_block_kernel", "conv_gemm_v2_kernel<float,128,128,2,2,2>",
                               "patchify_kernel<float>", "layernorm_kernel<float>", "conv3x3_halo_kernel", "mlp_rows_kernel", "-", "qkv_attn_kernel", "stem_conv1_kernel", "stage1_ring_kernel", "ln_gemm_rows_kernel", "qkv_attn_rows_kernel", "vit_attn_rows_kernel"};
  static const char* bf16n[] = {"gemm256_kernel", "conv_gemm_v2_kernel<__bf16,128,64,2,2,3>", "conv_gemm_v2_kernel<__bf16,128,32,4,1,3>",
                                "im2col27_kernel<__bf16>", "maxpool2_pos_kernel<__bf16>", "attention_v2_kernel<__bf16,...>", "pool_affine_kernel<__bf16>", "proto_head_kernel", "stage1_block_kernel", "conv_gemm_v2_kernel<__bf16,128,128,2,2,2>",
                                "patchify_kernel<__bf16>", "layernorm_kernel<__bf16>", "conv3x3_halo_kernel", "mlp_rows_kernel", "-", "qkv_attn_kernel", "stem_conv1_kernel", "stage1_ring_kernel", "ln_gemm_rows_kernel", "qkv_attn_rows_kernel", "vit_attn_rows_kernel"};
  static const char* f16n[] = {"gemm256_kernel", "conv_gemm_v2_kernel<_Float16,128,64,2,2,3>", "conv_gemm_v2_kernel<_Float16,128,32,4,1,3>",
                               "im2col27_kernel<_Float16>", "maxpool2_pos_kernel<_Float16>", "attention_v2_kernel<_Float16,...>", "pool_affine_kernel<_Float16>", "proto_head_kernel", "stage1_block_kernel", "conv_gemm_v2_kernel<_Float16,128,128,2,2,2>",
                               "patchify_kernel<_Float16>", "layernorm_kernel<_Float16>", "conv3x3_halo_kernel", "mlp_rows_kernel", "-", "qkv_attn_kernel", "stem_conv1_kernel", "stage1_ring_kernel", "ln_gemm_rows_kernel", "qkv_attn_rows_kernel", "vit_attn_rows_kernel"};
  static const char* x2n[] = {"gemm256_x2_kernel", "conv_gemm_v2_kernel<f32x2l,128,64,2,2,3>", "conv_gemm_v2_kernel<f32x2l,128,32,4,1,3>", "", "", "", "", "", "", "conv_gemm_v2_kernel<f32x2l,128,128,2,2,2>"};
  if (kernel_id < 0 || kernel_id > 20) return "?";
  if (kernel_id == KID_STAGE1RING && kd(dtype) == 1 && K_w4_enabled(dtype)) return "stage1_w4_kernel";      // (launch_stage1_ring dispatches to stage1_w4.hip)
  if (is_x2(dtype)) return kernel_id == 14 ? "gconv3x3_x2_kernel" : (kernel_id == 0 || kernel_id == 1 || kernel_id == 2 || kernel_id == 9) ? x2n[kernel_id] : f32n[kernel_id];
  return dtype == FSVIT_F32 ? f32n[kernel_id] : dtype == FSVIT_F16 ? f16n[kernel_id] : bf16n[kernel_id];
}

extern "C" int fsvit_encoder_profile_begin(void* hv) {
  EngineBase* h = static_cast<EngineBase*>(hv);
  if (!h) return fail(FSVIT_ERR_ARG, "null argument");
  h->prof.clear();
  h->prof_last = nullptr;
  h->profiling = true;
  return 0;
}

extern "C" int fsvit_encoder_profile_end(void* hv, fsvit_prof_rec* out, int max_recs, int* n_out) {
  EngineBase* h = static_cast<EngineBase*>(hv);
  if (!h || !out || !n_out) return fail(FSVIT_ERR_ARG, "null argument");
  h->profiling = false;
  std::vector<fsvit_prof_rec> agg;
  hipEvent_t freed = nullptr;
  for (ProfEntry& e : h->prof) {
    HIP_TRY(hipEventSynchronize(e.e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e.e0, e.e1));
    size_t k = 0;
    for (; k < agg.size(); ++k)
      if (e.layer == agg[k].layer && e.kernel == agg[k].kernel_id) break;
    if (k == agg.size()) {
      fsvit_prof_rec r;
      memset(&r, 0, sizeof(r));
      snprintf(r.layer, sizeof(r.layer), "%s", e.layer.c_str());
      r.kernel_id = e.kernel;
      agg.push_back(r);
    }
    agg[k].flops += e.flops;
    agg[k].ms += ms;
    agg[k].launches += 1;
  }
  // events are shared between neighbours (e1 of launch i is e0 of launch i+1 inside a chunk)
  for (ProfEntry& e : h->prof) {
    if (e.e0 != freed) (void)hipEventDestroy(e.e0);
    (void)hipEventDestroy(e.e1);
    freed = e.e1;
  }
  h->prof.clear();
  h->prof_last = nullptr;
  *n_out = (int)agg.size();
  if ((int)agg.size() > max_recs) return fail(FSVIT_ERR_ARG, "profile has %zu records, buffer holds %d", agg.size(), max_recs);
  for (size_t i = 0; i < agg.size(); ++i) out[i] = agg[i];
  return 0;
}

// ==================================================================================== ViT / DeiT encoder
// test_phase/models/deit.py:139-218.  LayerNorm gains/shifts are folded into the following Linear
// (W' = W diag(gamma), b' = b + W beta); the normalisation itself is a one-wave-per-token kernel.
struct VitBlock {
  Layer qkv, proj, fc1, fc2;
  void* mlp_img = nullptr;      // mlp_rows.hip: fragment-major image of proj | fc1 | fc2 (null: GEMM + LayerNorm launches)
  float* mlp_b1 = nullptr;
  void* qkv_img = nullptr;      // mlp_rows.hip ln_gemm_rows: fragment-major image of the qkv Linear (null: LayerNorm launch + GEMM)
  void* attn_img = nullptr;     // mlp_rows.hip vit_attn_rows: head-major image of the qkv Linear (norm1 + qkv + attention in one launch)
};

struct fsvit_vit : EngineBase {
  fsvit_vit_cfg cfg;
  int D = 0, S = 0, np = 0, npw = 0, Kp = 0, hd = 0, hdp = 0, hid = 0;
  Layer pe;
  float *pos_patch = nullptr;      // pos_embed[1:]  [np][D]
  float *cls_pos0 = nullptr;       // cls_token + pos_embed[0]  [D]
  float *ng = nullptr, *nb = nullptr;
  std::vector<VitBlock> blocks;
};

namespace {

std::vector<double> add_vec(const std::vector<double>& a, const float* b, int n) {
  std::vector<double> r(n);
  for (int i = 0; i < n; ++i) r[i] = a[i] + (double)b[i];
  return r;
}

int build_vit(fsvit_vit* h, const SD& sd) {
  const fsvit_vit_cfg& cf = h->cfg;
  const int epc = 16 / h->es, kch = 64 / h->es;
  const int D = cf.embed_dim, p = cf.patch_size, heads = cf.num_heads;
  if (cf.img_size % p || D % heads || D % 8) return fail(FSVIT_ERR_ARG, "unsupported ViT geometry");
  h->D = D; h->npw = cf.img_size / p; h->np = h->npw * h->npw; h->S = h->np + 1;
  h->hd = D / heads; h->hdp = round_up(h->hd, kch); h->hid = (int)(D * cf.mlp_ratio);
  if (h->hid % 8) return fail(FSVIT_ERR_ARG, "hidden width must be a multiple of 8");
  if (h->S > (h->es == 4 ? 208 : 224)) return fail(FSVIT_ERR_ARG, "attention supports at most %d tokens", h->es == 4 ? 208 : 224);
  const int K = 3 * p * p;
  h->Kp = round_up(K, 8);
  (void)epc;
  {
    const float* w = sd.get("patch_embed.proj.weight", {D, 3, p, p});
    const float* b = sd.get("patch_embed.proj.bias", {D});
    const float* cls = sd.get("cls_token", {1, 1, D});
    const float* pos = sd.get("pos_embed", {1, h->S, D});
    const float* g = sd.get("norm.weight", {D});
    const float* bt = sd.get("norm.bias", {D});
    if (!w || !b || !cls || !pos || !g || !bt) return FSVIT_ERR_KEY;
    std::vector<int> cm(K);
    for (int k = 0; k < K; ++k) cm[k] = k;                       // conv weight [D][3][p][p] is already (c, py, px)-major
    std::vector<double> bias(D);
    for (int i = 0; i < D; ++i) bias[i] = b[i];
    RC_TRY(pack_layer(h, &h->pe, w, D, K, 1, 1, 1, nullptr, nullptr, bias, true, nullptr, 0, &cm, h->Kp));
    std::vector<float> pp((size_t)h->np * D), c0(D), gg(g, g + D), bb(bt, bt + D);
    for (int i = 0; i < h->np; ++i)
      for (int d = 0; d < D; ++d) pp[(size_t)i * D + d] = pos[(size_t)(i + 1) * D + d];
    for (int d = 0; d < D; ++d) c0[d] = cls[d] + pos[d];
    void* dv;
    RC_TRY(upload(h, pp, false, &dv)); h->pos_patch = (float*)dv;
    RC_TRY(upload(h, c0, false, &dv)); h->cls_pos0 = (float*)dv;
    RC_TRY(upload(h, gg, false, &dv)); h->ng = (float*)dv;
    RC_TRY(upload(h, bb, false, &dv)); h->nb = (float*)dv;
  }
  const int hd = h->hd, hdp = h->hdp;
  std::vector<int> rowmap(3 * heads * hd), colmap(heads * hd);
  for (int x = 0; x < 3; ++x)
    for (int y = 0; y < heads; ++y)
      for (int z = 0; z < hd; ++z) rowmap[(x * heads + y) * hd + z] = (x * heads + y) * hdp + z;
  for (int y = 0; y < heads; ++y)
    for (int z = 0; z < hd; ++z) colmap[y * hd + z] = y * hdp + z;
  std::vector<double> nob;
  h->blocks.resize(cf.depth);
  for (int i = 0; i < cf.depth; ++i) {
    const std::string bp = "blocks." + std::to_string(i) + ".";
    const float* g1 = sd.get(bp + "norm1.weight", {D});
    const float* b1 = sd.get(bp + "norm1.bias", {D});
    const float* g2 = sd.get(bp + "norm2.weight", {D});
    const float* b2 = sd.get(bp + "norm2.bias", {D});
    const float* wq = sd.get(bp + "attn.qkv.weight", {3 * D, D});
    const float* bq = sd.get(bp + "attn.qkv.bias", {3 * D});
    const float* wp = sd.get(bp + "attn.proj.weight", {D, D});
    const float* bpj = sd.get(bp + "attn.proj.bias", {D});
    const float* w1 = sd.get(bp + "mlp.fc1.weight", {h->hid, D});
    const float* bf1 = sd.get(bp + "mlp.fc1.bias", {h->hid});
    const float* w2 = sd.get(bp + "mlp.fc2.weight", {D, h->hid});
    const float* bf2 = sd.get(bp + "mlp.fc2.bias", {D});
    if (!g1 || !b1 || !g2 || !b2 || !wq || !bq || !wp || !bpj || !w1 || !bf1 || !w2 || !bf2) return FSVIT_ERR_KEY;
    std::vector<double> s1(g1, g1 + D), t1(b1, b1 + D), s2(g2, g2 + D), t2(b2, b2 + D);
    std::vector<double> vbp(bpj, bpj + D), vb2(bf2, bf2 + D);
    RC_TRY(pack_layer(h, &h->blocks[i].qkv, wq, 3 * D, D, 1, 1, 1, nullptr, &s1, add_vec(prenorm_bias(wq, 3 * D, D, t1), bq, 3 * D), true, &rowmap, 3 * heads * hdp, nullptr, 0));
    RC_TRY(pack_layer(h, &h->blocks[i].proj, wp, D, heads * hd, 1, 1, 1, nullptr, nullptr, vbp, true, nullptr, 0, &colmap, heads * hdp));
    RC_TRY(pack_layer(h, &h->blocks[i].fc1, w1, h->hid, D, 1, 1, 1, nullptr, &s2, add_vec(prenorm_bias(w1, h->hid, D, t2), bf1, h->hid), true, nullptr, 0, nullptr, 0));
    RC_TRY(pack_layer(h, &h->blocks[i].fc2, w2, D, h->hid, 1, 1, 1, nullptr, nullptr, vb2, true, nullptr, 0, nullptr, 0));
    const int kdt = h->dtype;
    if (K(mlp_rows_ln_supported)(kd(kdt), D, h->hid, heads * hdp)) {      // proj + residual + norm2 + Mlp in one row-wise kernel
      VitBlock& b = h->blocks[i];
      void *img = nullptr, *b1i = nullptr;
      HIP_TRY(hipMalloc(&img, K(mlp_rows_image_bytes)(D, h->hid, heads * hdp)));
      h->allocs.push_back(img);
      HIP_TRY(hipMalloc(&b1i, (size_t)h->hid * 4));
      h->allocs.push_back(b1i);
      RC_TRY(K(launch_mlp_pack)(b.fc1.w, b.fc1.Kw, b.fc1.bias, b.fc2.w, b.fc2.Kw, b.proj.w, b.proj.Kw, heads * hdp, img, (float*)b1i, D, h->hid, nullptr));
      HIP_TRY(hipDeviceSynchronize());
      b.mlp_img = img;
      b.mlp_b1 = (float*)b1i;
    }
    if (K(ln_gemm_rows_supported)(kd(kdt), D, 3 * heads * hdp)) {         // norm1 + qkv in one row-wise kernel
      VitBlock& b = h->blocks[i];
      void* img = nullptr;
      HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(D, 3 * heads * hdp)));
      h->allocs.push_back(img);
      RC_TRY(K(launch_ln_gemm_pack)(b.qkv.w, b.qkv.Kw, img, D, 3 * heads * hdp, nullptr));
      HIP_TRY(hipDeviceSynchronize());
      b.qkv_img = img;
    }
    if (K(vit_attn_rows_supported)(kd(kdt), D, heads, hdp, h->S)) {       // norm1 + qkv + attention core in one launch: the qkv tensor never reaches HBM
      VitBlock& b = h->blocks[i];
      void* img = nullptr;
      HIP_TRY(hipMalloc(&img, K(ln_gemm_rows_image_bytes)(D, 3 * heads * hdp)));
      h->allocs.push_back(img);
      RC_TRY(K(launch_qkv_attn_rows_pack)(b.qkv.w, b.qkv.Kw, img, D, heads, hdp, nullptr));
      HIP_TRY(hipDeviceSynchronize());
      b.attn_img = img;
    }
  }
  return 0;
}

struct VitPlan { size_t patches, tokens, xn, qkv, ctx, hid, total; };

VitPlan make_vit_plan(const fsvit_vit* h, size_t Bc) {
  VitPlan p;
  const size_t es = h->es;
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += align256(bytes); return o; };
  p.tokens = take(Bc * h->S * h->D * es);
  p.xn = take(Bc * h->S * h->D * es);
  p.qkv = take(Bc * h->S * 3 * h->cfg.num_heads * h->hdp * es);
  p.ctx = take(Bc * h->S * h->cfg.num_heads * h->hdp * es);
  const size_t pb = Bc * h->np * h->Kp * es, hb = Bc * h->S * h->hid * es;      // patches are dead once the tokens exist
  p.patches = p.hid = take(pb > hb ? pb : hb);
  p.total = off;
  return p;
}

enum { KID_PATCHIFY = 10, KID_LN = 11 };

int vit_forward_chunk(fsvit_vit* h, const float* x, int Bc, float* feat, unsigned char* ws, bool first, hipStream_t st) {
  const VitPlan pl = make_vit_plan(h, Bc);
  const int kdt = h->dtype, dt = kd(kdt), D = h->D, S = h->S, heads = h->cfg.num_heads, hdp = h->hdp;
  const size_t es = h->es;
  void *patches = ws + pl.patches, *tokens = ws + pl.tokens, *xn = ws + pl.xn, *qkv = ws + pl.qkv, *ctx = ws + pl.ctx, *hid = ws + pl.hid;
  const int M = Bc * S;
  h->prof_last = nullptr;
  RC_TRY(timed(h, st, "patch_embed.patchify", KID_PATCHIFY, 0.0, [&]() { return K(launch_patchify)(x, patches, Bc, h->cfg.img_size, h->cfg.patch_size, h->Kp, dt, st); }));
  {
    ConvGemmParams p = conv_params(h->pe, patches, tokens, Bc, h->npw, h->npw, h->Kp, h->Kp, 1, 1, 1, 0, D, ACT_NONE, nullptr, 0, h->pos_patch);
    p.y_rpi = S; p.y_row0 = 1;                                   // patch token i of image b -> row b*S + 1 + i
    RC_TRY(run_gemm(h, st, "patch_embed.proj", h->pe, p, D, 3.0 * h->cfg.patch_size * h->cfg.patch_size));
  }
  RC_TRY(timed(h, st, "cls_token", KID_PATCHIFY, 0.0, [&]() { return K(launch_cls_pos)(h->cls_pos0, tokens, Bc, S, D, dt, st); }));
  RC_TRY(tap(h, "embed", tokens, (size_t)M * D * es, first, st));
  const float scale = 1.0f / std::sqrt((float)h->hd);
  for (size_t i = 0; i < h->blocks.size(); ++i) {
    const VitBlock& b = h->blocks[i];
    if (b.attn_img) {
      RC_TRY(timed(h, st, "blocks.norm1+qkv+attn", KID_VITATTNROWS, 2.0 * M * 3.0 * D * D + 4.0 * Bc * heads * (double)S * S * h->hd, [&]() {
        return K(launch_vit_attn_rows)(tokens, ctx, b.attn_img, b.qkv.bias, Bc, S, D, heads, hdp, h->cfg.ln_eps, scale, st);
      }));
    } else {
    if (b.qkv_img) {
      RC_TRY(timed(h, st, "blocks.norm1+qkv", KID_LNGEMM, 2.0 * M * 3.0 * D * D, [&]() {
        return K(launch_ln_gemm_rows)(tokens, qkv, b.qkv_img, b.qkv.bias, M, D, 3 * heads * hdp, h->cfg.ln_eps, st);
      }));
    } else {
      RC_TRY(timed(h, st, "blocks.norm1", KID_LN, 0.0, [&]() { return K(launch_layernorm)(tokens, xn, M, D, h->cfg.ln_eps, dt, st); }));
      RC_TRY(run_gemm(h, st, "blocks.attn.qkv", b.qkv, conv_params(b.qkv, xn, qkv, Bc, S, 1, D, D, 1, 1, 1, 0, 3 * heads * hdp, ACT_NONE, nullptr, 0, nullptr), 3.0 * D, D));
    }
    RC_TRY(timed(h, st, "blocks.attn.core", KID_ATTN, 4.0 * Bc * heads * (double)S * S * h->hd,
                 [&]() { return K(launch_attention)(qkv, ctx, Bc, S, heads, hdp, scale, is_x2(h->dtype) ? 2 : dt, st); }));
    }
    if (b.mlp_img) {
      RC_TRY(timed(h, st, "blocks.proj+norm2+mlp", KID_MLPROWS, 2.0 * M * ((double)heads * h->hd * D + 2.0 * D * h->hid), [&]() {
        return K(launch_mlp_rows_ln)(tokens, tokens, b.mlp_img, b.mlp_b1, b.proj.bias, b.fc2.bias, ctx, heads * hdp, M, D, h->hid, h->cfg.ln_eps, st);
      }));
      RC_TRY(tap(h, "blocks." + std::to_string(i), tokens, (size_t)M * D * es, first, st));
      continue;
    }
    RC_TRY(run_gemm(h, st, "blocks.attn.proj", b.proj, conv_params(b.proj, ctx, tokens, Bc, S, 1, heads * hdp, heads * hdp, 1, 1, 1, 0, D, ACT_NONE, tokens, 0, nullptr), D, D));
    RC_TRY(timed(h, st, "blocks.norm2", KID_LN, 0.0, [&]() { return K(launch_layernorm)(tokens, xn, M, D, h->cfg.ln_eps, dt, st); }));
    RC_TRY(run_gemm(h, st, "blocks.mlp.fc1", b.fc1, conv_params(b.fc1, xn, hid, Bc, S, 1, D, D, 1, 1, 1, 0, h->hid, ACT_GELU, nullptr, 0, nullptr), h->hid, D));
    RC_TRY(run_gemm(h, st, "blocks.mlp.fc2", b.fc2, conv_params(b.fc2, hid, tokens, Bc, S, 1, h->hid, h->hid, 1, 1, 1, 0, D, ACT_NONE, tokens, 0, nullptr), D, h->hid));
    RC_TRY(tap(h, "blocks." + std::to_string(i), tokens, (size_t)M * D * es, first, st));
  }
  RC_TRY(timed(h, st, "norm.cls", KID_LN, 0.0, [&]() { return K(launch_final_ln_cls)(tokens, h->ng, h->nb, feat, Bc, S, D, h->cfg.ln_eps, dt, st); }));
  return 0;
}

}  // namespace

extern "C" int fsvit_vit_create(const fsvit_vit_cfg* cfg, const fsvit_tensor* state_dict, int n_tensors, int dtype, fsvit_vit** out) {
  if (!cfg || !state_dict || !out || n_tensors <= 0) return fail(FSVIT_ERR_ARG, "null argument");
  if (!known_dtype(dtype)) return fail(FSVIT_ERR_ARG, "unknown dtype %d", dtype);
  if (cfg->num_heads < 1 || cfg->embed_dim < 8 || cfg->depth < 0 || cfg->patch_size < 1) return fail(FSVIT_ERR_ARG, "bad ViT configuration");
  fsvit_vit* h = new fsvit_vit();
  h->kind = KIND_VIT;
  h->cfg = *cfg;
  h->dtype = dtype;
  h->es = storage_bytes(dtype);
  SD sd{state_dict, n_tensors};
  int rc = build_vit(h, sd);
  if (rc != 0) { fsvit_vit_destroy(h); return rc; }
  *out = h;
  return 0;
}

extern "C" void fsvit_vit_destroy(fsvit_vit* h) {
  if (!h) return;
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
}

extern "C" int fsvit_vit_out_dim(const fsvit_vit* h) { return h ? h->D : 0; }

extern "C" size_t fsvit_vit_workspace_bytes(const fsvit_vit* h, int chunk_images) {
  if (!h || chunk_images <= 0) return 0;
  return make_vit_plan(h, (size_t)chunk_images).total;
}

extern "C" int fsvit_vit_forward(fsvit_vit* h, const float* x, int n_img, int img_h, int img_w, float* feat, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (h && n_img <= 0) return 0;
  if (!h || !x || !feat || !ws) return fail(FSVIT_ERR_ARG, "null argument");
  if (img_h != h->cfg.img_size || img_w != h->cfg.img_size)      // PatchEmbed assert (deit.py:96-97)
    return fail(FSVIT_ERR_IMG_SIZE, "Input image size (%d*%d) doesn't match model (%d*%d).", img_h, img_w, h->cfg.img_size, h->cfg.img_size);
  if (((uintptr_t)ws & 255) != 0) return fail(FSVIT_ERR_ARG, "workspace must be 256-byte aligned");
  int lo = 0, hi = n_img;
  while (lo < hi) {
    int mid = (lo + hi + 1) / 2;
    if (make_vit_plan(h, (size_t)mid).total <= ws_bytes) lo = mid; else hi = mid - 1;
  }
  if (lo < 1) return fail(FSVIT_ERR_WORKSPACE, "workspace of %zu bytes cannot hold one image (need %zu)", ws_bytes, make_vit_plan(h, 1).total);
  const size_t img_elems = (size_t)3 * img_h * img_w;
  for (int off = 0; off < n_img; off += lo) {
    const int bc = n_img - off < lo ? n_img - off : lo;
    int rc = vit_forward_chunk(h, x + (size_t)off * img_elems, bc, feat + (size_t)off * h->D, (unsigned char*)ws, off == 0, (hipStream_t)stream);
    if (rc != 0) return rc;
  }
  return 0;
}

static int encoder_forward_any(void* hv, const float* x, int n, int img_h, int img_w, float* feat, void* ws, size_t ws_bytes, void* stream) {
  EngineBase* b = static_cast<EngineBase*>(hv);
  if (b->kind == KIND_VISFORMER) return fsvit_visformer_forward(static_cast<fsvit_visformer*>(b), x, n, img_h, img_w, feat, ws, ws_bytes, stream);
  if (b->kind == KIND_VIT) return fsvit_vit_forward(static_cast<fsvit_vit*>(b), x, n, img_h, img_w, feat, ws, ws_bytes, stream);
  return fail(FSVIT_ERR_ARG, "not an fsvit encoder handle");
}
static int encoder_out_dim_any(void* hv) {
  EngineBase* b = static_cast<EngineBase*>(hv);
  return b->kind == KIND_VISFORMER ? static_cast<fsvit_visformer*>(b)->C3 : (b->kind == KIND_VIT ? static_cast<fsvit_vit*>(b)->D : 0);
}
