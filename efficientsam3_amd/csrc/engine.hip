// Host-side engine: weight store (BN folding, GEMM packing), stack arena, and the
// kernel graphs of the hot path (image encoder, prompt encoder + two-way mask decoder,
// mask post-processing) behind the C ABI of include/esam3.h.
//
// Graph structure follows the reference module tree (file:line cited per function);
// nothing here calls into PyTorch or any CPU fallback -- every tensor op is a HIP kernel
// from gemm_conv.hip / kernels_backbone.hip / kernels_decoder.hip.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>

#include "../../include/esam3.h"
#include "kernels.h"

// --------------------------------------------------------------------------------------
// error channel
// --------------------------------------------------------------------------------------
static thread_local std::string g_err;
void esam3_set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
extern "C" const char* esam3_last_error(void) { return g_err.c_str(); }

#define CK(expr)            \
  do {                      \
    if ((expr) != 0) return -1; \
  } while (0)

namespace {

constexpr int IMG = 1008;   // network resolution (sam3_image_processor.py:17)
constexpr int EMB = 72;     // embedding grid (model_builder.py:913-919)
constexpr int DM = 256;     // d_model
constexpr int TRUNK_C = 1024;
constexpr float BN_EPS = 1e-5f;

const std::string TRUNK = "backbone.vision_backbone.trunk.model.";
const std::string EVBB = TRUNK + "backbone.model.";
const std::string NECK = "backbone.vision_backbone.";
const std::string TEXTP = "backbone.language_backbone.";
const std::string SAM = "inst_interactive_predictor.model.";
const std::string MD = SAM + "sam_mask_decoder.";
const std::string PE = SAM + "sam_prompt_encoder.";

}  // namespace

// ---- profiler scopes (esam3_common.h) ------------------------------------------------------------------------------------------
namespace {
void (*g_scope_push)(const char*) = nullptr;
void (*g_scope_pop)(void) = nullptr;
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    if (void* h = dlopen("libroctx64.so", RTLD_LAZY | RTLD_LOCAL)) {
      push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
      pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
      if (!push || !pop) push = nullptr, pop = nullptr;
    }
  }
};
Roctx& roctx() {
  static Roctx r;
  return r;
}
}  // namespace
void esam3_scope_push(const char* name) {
  if (g_scope_push) g_scope_push(name);
  if (roctx().push) roctx().push(name);
}
void esam3_scope_pop() {
  if (roctx().pop) roctx().pop();
  if (g_scope_pop) g_scope_pop();
}
extern "C" void esam3_set_scope_hooks(void (*push)(const char*), void (*pop)(void)) {
  g_scope_push = push;
  g_scope_pop = pop;
}

// The arithmetic of Engine::compose_upconv (see there) on plain host arrays, shared with the single-operator entry esam3_op_upconv:
// wt [cin][cm][2][2] + bt [cm] = the ConvTranspose2d (already composed with the 1x1 that follows it), w3 [co][cm][3][3] + b3 [co] = the
// 3x3 conv -> w [class][o][tap = kh*2 + kw][ci], bias [co], corr [class][3: row edge, column edge, both][co].  fp64 accumulation.
void esam3_compose_upconv_host(const float* wt, const float* bt, const float* w3, const float* b3, int cin, int cm, int co,
                               std::vector<float>& w, std::vector<float>& bias, std::vector<float>& corr) {
  w.assign((size_t)4 * co * 4 * cin, 0.f);
  // Wt as [t][m][ci] (rows contiguous in ci)
  std::vector<double> wtt((size_t)4 * cm * cin);
  for (int ci = 0; ci < cin; ++ci)
    for (int m = 0; m < cm; ++m)
      for (int t = 0; t < 4; ++t) wtt[((size_t)t * cm + m) * cin + ci] = wt[((size_t)ci * cm + m) * 4 + t];
  std::vector<double> acc((size_t)4 * cin);  // [kh*2+kw][ci] of one (class, o)
  for (int cls = 0; cls < 4; ++cls) {
    const int dy = cls >> 1, dx = cls & 1;
    for (int o = 0; o < co; ++o) {
      std::fill(acc.begin(), acc.end(), 0.0);
      for (int ty = 0; ty < 3; ++ty)
        for (int tx = 0; tx < 3; ++tx) {
          const int ay = dy + ty - 1, ax = dx + tx - 1;  // offset of the ConvT output pixel from (2y, 2x): -1 .. 2
          const int sy = ay < 0 ? -1 : ay >> 1, sx = ax < 0 ? -1 : ax >> 1;  // floor(a / 2)
          const int kh = sy + 1 - dy, kw = sx + 1 - dx, par = (ay & 1) * 2 + (ax & 1);
          double* a = &acc[(size_t)(kh * 2 + kw) * cin];
          for (int m = 0; m < cm; ++m) {
            const double wv = w3[(((size_t)o * cm + m) * 3 + ty) * 3 + tx];
            const double* r = &wtt[((size_t)par * cm + m) * cin];
            for (int ci = 0; ci < cin; ++ci) a[ci] += wv * r[ci];
          }
        }
      float* dst = &w[((size_t)cls * co + o) * 4 * cin];
      for (size_t i = 0; i < (size_t)4 * cin; ++i) dst[i] = (float)acc[i];
    }
  }
  // bias shares S[ty][tx][o] and the composed bias / ring corrections
  std::vector<double> S((size_t)9 * co, 0.0);
  for (int o = 0; o < co; ++o)
    for (int m = 0; m < cm; ++m)
      for (int t = 0; t < 9; ++t) S[(size_t)t * co + o] += (double)w3[((size_t)o * cm + m) * 9 + t] * bt[m];
  bias.resize(co);
  corr.resize((size_t)4 * 3 * co);
  for (int o = 0; o < co; ++o) {
    double a = b3[o];
    for (int t = 0; t < 9; ++t) a += S[(size_t)t * co + o];
    bias[o] = (float)a;
  }
  for (int cls = 0; cls < 4; ++cls) {
    const int iy = (cls >> 1) ? 2 : 0, ix = (cls & 1) ? 2 : 0;  // the 3x3 tap row / column that falls outside at that edge
    for (int o = 0; o < co; ++o) {
      double r = 0.0, c = 0.0;
      for (int k = 0; k < 3; ++k) {
        r += S[(size_t)(iy * 3 + k) * co + o];
        c += S[(size_t)(k * 3 + ix) * co + o];
      }
      const double both = r + c - S[(size_t)(iy * 3 + ix) * co + o];
      corr[((size_t)cls * 3 + 0) * co + o] = (float)-r;
      corr[((size_t)cls * 3 + 1) * co + o] = (float)-c;
      corr[((size_t)cls * 3 + 2) * co + o] = (float)-both;
    }
  }
}

namespace {

struct HostTensor {
  std::vector<float> d;
  std::vector<int64_t> shape;
  bool packed = false;        // consumed by one of the cache-first packers (pk_conv / pk_convT / pk_linear / pk_dw)
  bool released = false;      // esam3_release_host_weights: the name and shape stay, the data is gone
  mutable uint32_t touched = 0;  // epoch of the last find() / need()
};

struct PackedGemm {
  void* w = nullptr;      // [Np][Kp] activation dtype
  void* wn = nullptr;     // bf16 3x3 convs with 32 / 64 output channels: the same weights in conv3x3_narrow's fragment order
  float* bias = nullptr;  // [N] (or [Cout] for convT) fp32
  float* border_corr = nullptr;  // up-conv (ksize 2) only: [4 classes][3 edge cases][Cout] fp32 (GemmParams::border_corr)
  int N = 0, K = 0, Kp = 0, Np = 0, ksize = 1, cin = 0, convt_cout = 0;
  std::string tag;
};
struct PackedDw {
  float* w = nullptr;  // [k*k][C]
  float* bias = nullptr;
  int C = 0, ks = 3;
};

struct T4 {  // NHWC activation view
  void* p = nullptr;
  int B = 0, H = 0, W = 0, C = 0;
  int ld = 0;  // row stride in elements
  int pad = 0; // 1: memory is [B][H+2][W+2][ld] with a zero border (input of a 3x3 conv)
  int64_t rows() const { return (int64_t)B * H * W; }
};

struct Arena {
  char* base = nullptr;
  size_t cap = 0, top = 0, peak = 0;
  bool dry = false;
  static size_t align(size_t x) { return (x + 255) & ~(size_t)255; }
  void* alloc(size_t bytes) {
    const size_t a = align(top);
    top = a + bytes;
    if (top > peak) peak = top;
    if (dry) return reinterpret_cast<void*>(a + 4096);  // never dereferenced
    if (top > cap) return nullptr;
    return base + a;
  }
  size_t mark() const { return top; }
  void release(size_t m) { top = m; }
};

}  // namespace

struct esam3_engine {
  esam3_config cfg{};
  int dtype = 1;
  size_t esz = 2;
  std::vector<int> widths, depths;
  int dim = 16;
  struct RvBlock { int c, se, stride; };
  std::vector<RvBlock> rv_cfg;  // RepViT block table (repvit.py:291-506)
  std::vector<int> tv_dims, tv_depths, tv_heads, tv_windows;  // TinyViT (tiny_vit.py:656-690)
  std::unordered_map<std::string, HostTensor> raw;
  std::unordered_map<std::string, PackedGemm> gemms;
  std::unordered_map<std::string, PackedDw> dws;
  std::unordered_map<std::string, float*> fbufs;
  std::unordered_map<std::string, void*> tbufs;
  std::vector<void*> owned;
  Arena arena;
  bool finalized = false;
  bool dry = false;  // allocate + pack only, launch nothing
  hipStream_t st = nullptr;

  struct ProfRec { std::string tag; hipEvent_t a, b; double flops, bytes; const char* kernel; };
  std::vector<ProfRec> recs;
  // bench.py's live roofline leg: the launches whose tag equals `watch_tag` (esam3_profile_tag)
  // are bracketed by HIP events on the launch stream while everything else runs un-instrumented.
  std::string watch_tag;  // empty: nothing watched
  int timed_gemm(const std::string& tag, double flops, double bytes, const GemmParams& p, hipStream_t s) {
    ProfRec r{tag, nullptr, nullptr, flops, bytes, nullptr};
    esam3_take_last_gemm_kernel();
    HIP_CHECK_RET(hipEventCreate(&r.a));
    HIP_CHECK_RET(hipEventCreate(&r.b));
    HIP_CHECK_RET(hipEventRecord(r.a, s));
    const int rc = esam3_launch_gemm(dtype, p, s);
    HIP_CHECK_RET(hipEventRecord(r.b, s));
    r.kernel = esam3_take_last_gemm_kernel();
    recs.push_back(r);
    return rc;
  }

  // ---------------- optional per-launch profiler (HIP events on the launch stream) --------
  bool prof = false;
  int prof_launch(const std::string& tag, double flops, double bytes, const std::function<int()>& fn) {
    if (!prof && (watch_tag.empty() || tag != watch_tag)) return fn();  // watched tag: timed like a watched GEMM
    ProfRec r{tag, nullptr, nullptr, flops, bytes, nullptr};
    esam3_take_last_gemm_kernel();
    HIP_CHECK_RET(hipEventCreate(&r.a));
    HIP_CHECK_RET(hipEventCreate(&r.b));
    HIP_CHECK_RET(hipEventRecord(r.a, st));
    const int rc = fn();
    HIP_CHECK_RET(hipEventRecord(r.b, st));
    r.kernel = esam3_take_last_gemm_kernel();
    recs.push_back(r);
    return rc;
  }

  // ---------------- raw weight access ----------------
  uint32_t epoch = 1;
  const HostTensor* find(const std::string& n) const {
    auto it = raw.find(n);
    if (it == raw.end()) return nullptr;
    it->second.touched = epoch;
    return &it->second;
  }
  const HostTensor* need(const std::string& n) const {
    const HostTensor* t = find(n);
    if (!t) { esam3_set_error("missing weight '%s'", n.c_str()); return nullptr; }
    if (t->released) {
      esam3_set_error("weight '%s' was freed by esam3_release_host_weights() and is requested again", n.c_str());
      return nullptr;
    }
    return t;
  }
  static void mark_packed(const HostTensor* t) { if (t) const_cast<HostTensor*>(t)->packed = true; }

  void* dev_upload(const void* src, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) {
      esam3_set_error("hipMalloc(%zu) failed", bytes);
      return nullptr;
    }
    if (bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) {
      esam3_set_error("hipMemcpy H2D failed");
      return nullptr;
    }
    owned.push_back(p);
    return p;
  }
  void* upload_T(const std::vector<float>& v) {  // host fp32 -> device activation dtype
    if (dtype == 0) return dev_upload(v.data(), v.size() * 4);
    std::vector<bf16_t> h(v.size());
    for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_bf16(v[i]);
    return dev_upload(h.data(), h.size() * 2);
  }
  float* fvec(const std::string& name) {  // upload a raw tensor as fp32 (cached)
    auto it = fbufs.find(name);
    if (it != fbufs.end()) return it->second;
    const HostTensor* t = need(name);
    if (!t) return nullptr;
    float* p = (float*)dev_upload(t->d.data(), t->d.size() * 4);
    fbufs[name] = p;
    return p;
  }
  float* fvec_raw(const std::string& key, const std::vector<float>& v) {
    auto it = fbufs.find(key);
    if (it != fbufs.end()) return it->second;
    float* p = (float*)dev_upload(v.data(), v.size() * 4);
    fbufs[key] = p;
    return p;
  }

  // BatchNorm (eval) folded into per-channel scale/shift: y = x*scale + shift
  bool bn_fold(const std::string& bn, int C, std::vector<float>& scale, std::vector<float>& shift) {
    scale.assign(C, 1.f);
    shift.assign(C, 0.f);
    if (bn.empty()) return true;
    const HostTensor *g = need(bn + ".weight"), *b = need(bn + ".bias"),
                     *m = need(bn + ".running_mean"), *v = need(bn + ".running_var");
    if (!g || !b || !m || !v) return false;
    for (int c = 0; c < C; ++c) {
      const float s = g->d[c] / std::sqrt(v->d[c] + BN_EPS);
      scale[c] = s;
      shift[c] = b->d[c] - m->d[c] * s;
    }
    return true;
  }

  // dense conv (k = 1 or 3) [Cout][Cin][k][k] (+bias) (+BN) -> packed GEMM
  PackedGemm* pk_conv(const std::string& wname, const std::string& bname, const std::string& bn) {
    auto it = gemms.find(wname);
    if (it != gemms.end()) return &it->second;
    const HostTensor* w = need(wname);
    if (!w) return nullptr;
    const int N = (int)w->shape[0], cin = (int)w->shape[1], ks = (int)w->shape[2];
    std::vector<float> scale, shift;
    if (!bn_fold(bn, N, scale, shift)) return nullptr;
    std::vector<float> bias(N, 0.f);
    bool has_bias = !bn.empty();
    if (!bname.empty()) {
      const HostTensor* b = need(bname);
      if (!b) return nullptr;
      for (int n = 0; n < N; ++n) bias[n] = b->d[n] * scale[n];
      has_bias = true;
    }
    for (int n = 0; n < N; ++n) bias[n] += shift[n];
    PackedGemm g;
    g.N = N; g.cin = cin; g.ksize = ks; g.K = cin * ks * ks;
    g.Kp = esam3_gemm_pad_k(g.K, (int)esz);
    g.Np = esam3_gemm_pad_n(N);
    std::vector<float> pk((size_t)g.Np * g.Kp, 0.f);
    for (int n = 0; n < N; ++n)
      for (int c = 0; c < cin; ++c)
        for (int t = 0; t < ks * ks; ++t)
          pk[(size_t)n * g.Kp + esam3_conv_k_index(cin, ks, (int)esz, t, c)] = w->d[((size_t)n * cin + c) * ks * ks + t] * scale[n];
    g.w = upload_T(pk);
    g.bias = has_bias ? (float*)dev_upload(bias.data(), bias.size() * 4) : nullptr;
    if (!g.w) return nullptr;
    if (ks == 3 && esam3_conv3x3_narrow_ok(dtype, N, cin, 16, 16, 1, 0, 1, false)) {  // weight-side eligibility
      std::vector<float> pn((size_t)N * g.K);
      for (int n = 0; n < N; ++n)
        for (int c = 0; c < cin; ++c)
          for (int t = 0; t < 9; ++t)
            pn[(size_t)esam3_conv3x3_narrow_windex(N, n, t, c)] = w->d[((size_t)n * cin + c) * 9 + t] * scale[n];
      g.wn = upload_T(pn);
      if (!g.wn) return nullptr;
    }
    g.tag = wname;
    mark_packed(w);
    if (!bname.empty()) mark_packed(find(bname));
    return &(gemms[wname] = g);
  }
  // ConvTranspose2d k2 s2: [Cin][Cout][2][2] -> GEMM with N = 4*Cout, n = tap*Cout + co
  PackedGemm* pk_convT(const std::string& wname, const std::string& bname) {
    auto it = gemms.find(wname);
    if (it != gemms.end()) return &it->second;
    const HostTensor* w = need(wname);
    const HostTensor* b = need(bname);
    if (!w || !b) return nullptr;
    const int cin = (int)w->shape[0], cout = (int)w->shape[1];
    PackedGemm g;
    g.N = 4 * cout; g.cin = cin; g.ksize = 1; g.K = cin; g.convt_cout = cout;
    g.Kp = esam3_gemm_pad_k(g.K, (int)esz);
    g.Np = esam3_gemm_pad_n(g.N);
    std::vector<float> pk((size_t)g.Np * g.Kp, 0.f);
    for (int ci = 0; ci < cin; ++ci)
      for (int co = 0; co < cout; ++co)
        for (int t = 0; t < 4; ++t)
          pk[((size_t)t * cout + co) * g.Kp + ci] = w->d[((size_t)ci * cout + co) * 4 + t];
    g.w = upload_T(pk);
    g.bias = (float*)dev_upload(b->d.data(), b->d.size() * 4);
    if (!g.w) return nullptr;
    g.tag = wname;
    mark_packed(w);
    mark_packed(b);
    return &(gemms[wname] = g);
  }
  // ---- exact algebraic composition of linear chains (no activation in between) ----------
  // ConvTranspose2d(k2,s2) followed by a 1x1 conv == one ConvTranspose2d with
  //   W'[ci][co][t] = sum_m W[ci][m][t] * w1[co][m],  b'[co] = sum_m w1[co][m] * b[m] + b1[co]
  // (necks.py:42-92: dconv_2x2(_1) -> conv_1x1).  fp64 accumulation on the host.
  bool compose_convT_1x1(const std::string& tprefix, const std::string& cprefix, const std::string& key) {
    if (find(key + ".weight")) return true;
    const HostTensor *w = need(tprefix + ".weight"), *b = need(tprefix + ".bias"),
                     *w1 = need(cprefix + ".weight"), *b1 = need(cprefix + ".bias");
    if (!w || !b || !w1 || !b1) return false;
    const int cin = (int)w->shape[0], cm = (int)w->shape[1], co = (int)w1->shape[0];
    std::vector<double> w1t((size_t)cm * co);
    for (int o = 0; o < co; ++o)
      for (int m = 0; m < cm; ++m) w1t[(size_t)m * co + o] = w1->d[(size_t)o * cm + m];
    HostTensor ow, ob;
    ow.shape = {cin, co, 2, 2};
    ow.d.assign((size_t)cin * co * 4, 0.f);
    std::vector<double> acc(co);
    for (int ci = 0; ci < cin; ++ci)
      for (int t = 0; t < 4; ++t) {
        std::fill(acc.begin(), acc.end(), 0.0);
        for (int m = 0; m < cm; ++m) {
          const double a = w->d[((size_t)ci * cm + m) * 4 + t];
          const double* r = &w1t[(size_t)m * co];
          for (int o = 0; o < co; ++o) acc[o] += a * r[o];
        }
        for (int o = 0; o < co; ++o) ow.d[((size_t)ci * co + o) * 4 + t] = (float)acc[o];
      }
    ob.shape = {co};
    ob.d.resize(co);
    for (int o = 0; o < co; ++o) {
      double a = b1->d[o];
      for (int m = 0; m < cm; ++m) a += (double)w1->d[(size_t)o * cm + m] * b->d[m];
      ob.d[o] = (float)a;
    }
    raw[key + ".weight"] = std::move(ow);
    raw[key + ".bias"] = std::move(ob);
    return true;
  }
  // k x k conv followed by a 1x1 conv == one k x k conv with
  //   W'[o][c][t] = sum_m w1[o][m] * W[m][c][t],  b'[o] = sum_m w1[o][m] * b[m] + b1[o]
  // (conv_3x3 -> conv_s0/conv_s1, sam3_image_processor.py:62-75); exact also at the zero-padded
  // border because the 1x1 acts on the 3x3's output pixel.
  bool compose_conv_1x1(const std::string& kprefix, const std::string& cprefix, const std::string& key) {
    if (find(key + ".weight")) return true;
    const HostTensor *w = need(kprefix + ".weight"), *b = need(kprefix + ".bias"),
                     *w1 = need(cprefix + ".weight"), *b1 = need(cprefix + ".bias");
    if (!w || !b || !w1 || !b1) return false;
    const int cm = (int)w->shape[0], cin = (int)w->shape[1], ks = (int)w->shape[2], co = (int)w1->shape[0];
    const size_t inner = (size_t)cin * ks * ks;
    HostTensor ow, ob;
    ow.shape = {co, cin, ks, ks};
    ow.d.resize((size_t)co * inner);
    std::vector<double> acc(inner);
    for (int o = 0; o < co; ++o) {
      std::fill(acc.begin(), acc.end(), 0.0);
      for (int m = 0; m < cm; ++m) {
        const double a = w1->d[(size_t)o * cm + m];
        const float* r = &w->d[(size_t)m * inner];
        for (size_t j = 0; j < inner; ++j) acc[j] += a * r[j];
      }
      for (size_t j = 0; j < inner; ++j) ow.d[(size_t)o * inner + j] = (float)acc[j];
    }
    ob.shape = {co};
    ob.d.resize(co);
    for (int o = 0; o < co; ++o) {
      double a = b1->d[o];
      for (int m = 0; m < cm; ++m) a += (double)w1->d[(size_t)o * cm + m] * b->d[m];
      ob.d[o] = (float)a;
    }
    raw[key + ".weight"] = std::move(ow);
    raw[key + ".bias"] = std::move(ob);
    return true;
  }

  // ConvTranspose2d(k2, s2) followed by a 3x3 conv (pad 1) == four "2 x 2 convs" on the ConvT's INPUT, one per parity class
  // (dy, dx) of the output pixel (2y + dy, 2x + dx): tap (ty, tx) of the 3x3 reads the ConvT output at row 2y + dy + ty - 1,
  // which the ConvT produced from input row y + floor((dy + ty - 1) / 2) with its own tap parity (dy + ty - 1) mod 2, so
  //   Wup[class][o][kh][kw][ci] = sum over the (ty, tx) that land on input offset (kh, kw) of
  //                               sum_m W3[o][m][ty][tx] * Wt[ci][m][parity],   kh = floor((dy + ty - 1) / 2) + 1 - dy
  // (necks.py:42-92, level 0: dconv_2x2_1 -> conv_1x1 -> conv_3x3; the first two are composed already).  Per INPUT pixel
  // this is 16 Cin Cout MACs instead of 4 Cin Cmid + 36 Cmid Cout, a saving for Cin < 3 Cmid (512 < 768), and the
  // 4x-upsampled intermediate tensor never exists.  The bias: every 3x3 tap contributes S[ty][tx][o] = sum_m W3 b_t[m],
  // EXCEPT the taps that fall outside the output image (zero padding of the 3x3, not the ConvT's bias): those shares are
  // removed again for the ring pixels through `border_corr` (gemm256p adds it to the accumulators of those pixels).
  // Packed as N = class * Cout + o, K = (ci / 64) * 256 + (kh * 2 + kw) * 64 + ci % 64.  fp64 accumulation on the host.
  // -> w [class][o][tap = kh*2 + kw][ci], bias [co], corr [class][3][co]
  bool compose_upconv(const std::string& tprefix, const std::string& c3prefix, std::vector<float>& w, std::vector<float>& bias,
                      std::vector<float>& corr, int& cin_out, int& co_out) {
    const HostTensor *wt = need(tprefix + ".weight"), *bt = need(tprefix + ".bias"),
                     *w3 = need(c3prefix + ".weight"), *b3 = need(c3prefix + ".bias");
    if (!wt || !bt || !w3 || !b3) return false;
    const int cin = (int)wt->shape[0], cm = (int)wt->shape[1], co = (int)w3->shape[0];
    if ((int)w3->shape[1] != cm || (int)w3->shape[2] != 3) { esam3_set_error("compose_upconv %s: unexpected shape", c3prefix.c_str()); return false; }
    cin_out = cin; co_out = co;
    esam3_compose_upconv_host(wt->d.data(), bt->d.data(), w3->d.data(), b3->d.data(), cin, cm, co, w, bias, corr);
    return true;
  }
  // Packed for gemm256p's up-conv gather: N = class * Cout + o, K = (ci / 64) * 256 + (kh * 2 + kw) * 64 + ci % 64.
  PackedGemm* pk_upconv(const std::string& tprefix, const std::string& c3prefix, const std::string& key) {
    auto it = gemms.find(key);
    if (it != gemms.end()) return &it->second;
    std::vector<float> w, bias, corr;
    int cin = 0, co = 0;
    if (!compose_upconv(tprefix, c3prefix, w, bias, corr, cin, co)) return nullptr;
    if (cin % 64 != 0 || co % 256 != 0 || esz != 2) { esam3_set_error("pk_upconv %s: unsupported shape", key.c_str()); return nullptr; }
    PackedGemm g;
    g.N = 4 * co; g.cin = cin; g.ksize = 2; g.K = 4 * cin; g.convt_cout = co;
    g.Kp = g.K;
    g.Np = esam3_gemm_pad_n(g.N);
    std::vector<float> pk((size_t)g.Np * g.Kp, 0.f);
    for (int n = 0; n < 4 * co; ++n) {
      float* row = &pk[(size_t)n * g.Kp];
      const float* src = &w[(size_t)n * 4 * cin];
      for (int tap = 0; tap < 4; ++tap)
        for (int ci = 0; ci < cin; ++ci) row[esam3_upconv_kindex(tap, ci)] = src[(size_t)tap * cin + ci];
    }
    g.w = upload_T(pk);
    g.bias = (float*)dev_upload(bias.data(), bias.size() * 4);
    g.border_corr = (float*)dev_upload(corr.data(), corr.size() * 4);
    if (!g.w || !g.bias || !g.border_corr) return nullptr;
    g.tag = key;
    return &(gemms[key] = g);
  }
  // Packed for upconv_narrow_kernel (32 output channels per class): esam3_upconv_narrow_windex order; the weights live in `wn`.
  PackedGemm* pk_upconv_narrow(const std::string& tprefix, const std::string& c3prefix, const std::string& key) {
    auto it = gemms.find(key);
    if (it != gemms.end()) return &it->second;
    std::vector<float> w, bias, corr;
    int cin = 0, co = 0;
    if (!compose_upconv(tprefix, c3prefix, w, bias, corr, cin, co)) return nullptr;
    if (!esam3_upconv_narrow_ok(dtype, co, cin, 16, 16)) { esam3_set_error("pk_upconv_narrow %s: unsupported shape", key.c_str()); return nullptr; }
    PackedGemm g;
    g.N = 4 * co; g.cin = cin; g.ksize = 2; g.K = 4 * cin; g.Kp = g.K; g.Np = g.N; g.convt_cout = co;
    std::vector<float> pn((size_t)4 * co * 4 * cin);
    for (int cls = 0; cls < 4; ++cls)
      for (int o = 0; o < co; ++o)
        for (int tap = 0; tap < 4; ++tap)
          for (int ci = 0; ci < cin; ++ci)
            pn[(size_t)esam3_upconv_narrow_windex(o, cls, tap, ci)] = w[(((size_t)cls * co + o) * 4 + tap) * cin + ci];
    g.wn = upload_T(pn);
    g.bias = (float*)dev_upload(bias.data(), bias.size() * 4);
    g.border_corr = (float*)dev_upload(corr.data(), corr.size() * 4);
    if (!g.wn || !g.bias || !g.border_corr) return nullptr;
    g.tag = key;
    return &(gemms[key] = g);
  }

  // Second layer of a fused pointwise MLP (fused_mlp.hip): a 1x1 conv (+BN) or Linear packed [N][K] with the hidden (K) order
  // permuted inside each 32-block as esam3_fused_mlp_kperm prescribes.  Cached under wname + "#kperm".
  PackedGemm* pk_mlp2(const std::string& wname, const std::string& bname, const std::string& bn) {
    const std::string key = wname + "#kperm";
    auto it = gemms.find(key);
    if (it != gemms.end()) return &it->second;
    const HostTensor* w = need(wname);
    if (!w) return nullptr;
    const int N = (int)w->shape[0], K = (int)w->shape[1];
    if (K % 32 != 0 || esz != 2) { esam3_set_error("pk_mlp2 %s: K=%d", wname.c_str(), K); return nullptr; }
    std::vector<float> scale, shift;
    if (!bn_fold(bn, N, scale, shift)) return nullptr;
    std::vector<float> bias(N, 0.f);
    if (!bname.empty()) {
      const HostTensor* b = need(bname);
      if (!b) return nullptr;
      for (int n = 0; n < N; ++n) bias[n] = b->d[n] * scale[n];
    }
    for (int n = 0; n < N; ++n) bias[n] += shift[n];
    PackedGemm g;
    g.N = N; g.cin = K; g.ksize = 1; g.K = K; g.Kp = K; g.Np = N;
    std::vector<float> pk((size_t)N * K);
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) pk[(size_t)n * K + k] = w->d[(size_t)n * K + (k & ~31) + esam3_fused_mlp_kperm(k & 31)] * scale[n];
    g.w = upload_T(pk);
    g.bias = (float*)dev_upload(bias.data(), bias.size() * 4);
    if (!g.w || !g.bias) return nullptr;
    g.tag = key;
    return &(gemms[key] = g);
  }
  // x -> act(W1 x + b1) -> W2 . + b2 (+ res) in one launch where fused_mlp.hip is instantiated for the shape; false = not taken
  int fused_mlp(PackedGemm* g1, PackedGemm* g2p, const void* x, int ldx, int64_t rows, const void* res, int ldr, void* out, int ldo,
                int act, const std::string& tag) {
    if (dry) return 0;
    const double fl = 4.0 * (double)rows * g1->N * g1->K;
    const double by = ((double)rows * (g1->K + g2p->N + (res ? g2p->N : 0)) + 2.0 * (double)g1->N * g1->K) * (double)esz;
    return prof_launch("fused_mlp:" + tag, fl, by, [&]() {
      return esam3_launch_fused_mlp(x, ldx, g1->w, g1->bias, g2p->w, g2p->bias, res, ldr, out, ldo, rows, g1->K, g1->N, g2p->N, act, st);
    });
  }

  PackedGemm* pk_linear(const std::string& prefix, bool bias = true) {
    return pk_conv_like_linear(prefix + ".weight", bias ? prefix + ".bias" : "");
  }
  PackedGemm* pk_conv_like_linear(const std::string& wname, const std::string& bname) {
    auto it = gemms.find(wname);
    if (it != gemms.end()) return &it->second;
    const HostTensor* w = need(wname);
    if (!w) return nullptr;
    const int N = (int)w->shape[0], K = (int)w->shape[1];
    PackedGemm g;
    g.N = N; g.cin = K; g.ksize = 1; g.K = K;
    g.Kp = esam3_gemm_pad_k(K, (int)esz);
    g.Np = esam3_gemm_pad_n(N);
    std::vector<float> pk((size_t)g.Np * g.Kp, 0.f);
    for (int n = 0; n < N; ++n)
      for (int k = 0; k < K; ++k) pk[(size_t)n * g.Kp + k] = w->d[(size_t)n * K + k];
    g.w = upload_T(pk);
    if (!bname.empty()) {
      const HostTensor* b = need(bname);
      if (!b) return nullptr;
      g.bias = (float*)dev_upload(b->d.data(), b->d.size() * 4);
    }
    if (!g.w) return nullptr;
    g.tag = wname;
    mark_packed(w);
    if (!bname.empty()) mark_packed(find(bname));
    return &(gemms[wname] = g);
  }
  PackedDw* pk_dw(const std::string& wname, const std::string& bname, const std::string& bn) {
    auto it = dws.find(wname);
    if (it != dws.end()) return &it->second;
    const HostTensor* w = need(wname);
    if (!w) return nullptr;
    const int C = (int)w->shape[0], ks = (int)w->shape[2];
    std::vector<float> scale, shift;
    if (!bn_fold(bn, C, scale, shift)) return nullptr;
    std::vector<float> pk((size_t)ks * ks * C), bias(C, 0.f);
    for (int c = 0; c < C; ++c)
      for (int t = 0; t < ks * ks; ++t) pk[(size_t)t * C + c] = w->d[(size_t)c * ks * ks + t] * scale[c];
    bool has_bias = !bn.empty();
    if (!bname.empty()) {
      const HostTensor* b = need(bname);
      if (!b) return nullptr;
      for (int c = 0; c < C; ++c) bias[c] = b->d[c] * scale[c];
      has_bias = true;
    }
    for (int c = 0; c < C; ++c) bias[c] += shift[c];
    PackedDw d;
    d.C = C; d.ks = ks;
    d.w = (float*)dev_upload(pk.data(), pk.size() * 4);
    d.bias = has_bias ? (float*)dev_upload(bias.data(), bias.size() * 4) : nullptr;
    if (!d.w) return nullptr;
    mark_packed(w);
    if (!bname.empty()) mark_packed(find(bname));
    return &(dws[wname] = d);
  }

  // ---------------- arena helpers ----------------
  T4 alloc4(int B, int H, int W, int C) {
    T4 t;
    t.B = B; t.H = H; t.W = W; t.C = C; t.ld = C;
    t.p = arena.alloc((size_t)B * H * W * C * esz);
    return t;
  }
  // tensor with a 1-pixel zero border; the border is (re)zeroed here, producers fill the interior
  int alloc4_padded(int B, int H, int W, int C, T4* t) {
    t->B = B; t->H = H; t->W = W; t->C = C; t->ld = C; t->pad = 1;
    t->p = arena.alloc((size_t)B * (H + 2) * (W + 2) * C * esz);
    if (!ok(t->p)) return -1;
    if (dry) return 0;
    return prof_launch("zero_border", 0.0, 0.0, [&]() { return esam3_launch_zero_border(dtype, t->p, B, H + 2, W + 2, C, st); });
  }
  void* allocb(size_t bytes) { return arena.alloc(bytes); }
  bool ok(const void* p) {
    if (!p) esam3_set_error("workspace arena exhausted (cap %zu, need %zu)", arena.cap, arena.top);
    return p != nullptr;
  }

  // ---------------- launch helpers ----------------
  int gemm(const PackedGemm* g, const void* A, int lda, int64_t M, int H, int W, void* out, int ldc,
           int act, const void* res = nullptr, int ldr = 0, int res_after_act = 1, int res_mod = 0,
           const int* res_bidx = nullptr, int in_pad = 0, int out_pad = 0, int stride = 1, int out_f32 = 0) {
    if (!g) return -1;
    if (stride != 1 && (g->ksize != 3 || in_pad || out_pad)) { esam3_set_error("strided conv: only plain 3x3"); return -1; }
    if (in_pad && g->ksize != 3 && g->ksize != 2) { esam3_set_error("padded input given to a %dx%d conv", g->ksize, g->ksize); return -1; }
    if (g->ksize == 2 && !in_pad) { esam3_set_error("the up-conv gather needs a zero-bordered input"); return -1; }
    if (dry) return 0;
    GemmParams p{};
    p.A = A; p.Wt = g->w; p.bias = g->bias; p.res = res; p.out = out;
    p.M = M; p.N = g->N; p.K = g->K; p.Kp = g->Kp;
    p.H = H; p.W = W; p.Cin = g->cin; p.ksize = g->ksize;
    p.korder = g->convt_cout ? 0 : esam3_conv_korder(g->cin, g->ksize, (int)esz);
    p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.act = act; p.res_mod = res_mod;
    p.out_mode = g->convt_cout ? OUT_CONVT2X2 : OUT_PLAIN;
    p.convt_cout = g->convt_cout;
    p.res_after_act = res_after_act;
    p.res_bidx = res_bidx;
    p.in_pad = in_pad;
    p.out_pad = out_pad;
    p.stride = stride;
    p.out_f32 = out_f32;
    p.border_corr = g->border_corr;
    const double uniq_in = (double)M * g->cin * (g->ksize == 3 ? 1 : 1);
    const double bytes = (uniq_in + (double)g->N * g->K + (double)M * g->N + (res ? (double)M * g->N : 0.0)) * (double)esz;
    const double flops = 2.0 * (double)M * g->N * g->K;
    if (g->wn && esam3_conv3x3_narrow_ok(dtype, g->N, g->cin, H, W, in_pad, out_pad, stride, res != nullptr)) {
      p.Wt = g->wn;
      return prof_launch(g->tag, flops, bytes, [&]() { return esam3_launch_conv3x3_narrow(p, st); });
    }
    if (!prof && !watch_tag.empty() && g->tag == watch_tag) return timed_gemm(g->tag, flops, bytes, p, st);
    return prof_launch(g->tag, flops, bytes, [&]() { return esam3_launch_gemm(dtype, p, st); });
  }
  // 1x1 / 3x3 conv on an NHWC view -> new tensor (or into `dst` if given)
  int conv(const std::string& prefix, bool convlayer, const T4& x, int act, T4* y, const T4* res = nullptr,
           const T4* dst = nullptr, bool out_pad = false) {
    // convlayer: EfficientViT ConvLayer naming (.conv.weight/.conv.bias/.norm.*); else plain nn.Conv2d
    PackedGemm* g;
    if (convlayer) {
      const bool has_b = find(prefix + ".conv.bias") != nullptr;
      const bool has_bn = find(prefix + ".norm.weight") != nullptr;
      g = pk_conv(prefix + ".conv.weight", has_b ? prefix + ".conv.bias" : "", has_bn ? prefix + ".norm" : "");
    } else {
      const bool has_b = find(prefix + ".bias") != nullptr;
      g = pk_conv(prefix + ".weight", has_b ? prefix + ".bias" : "", "");
    }
    if (!g) return -1;
    if (g->cin != x.C) { esam3_set_error("conv %s: Cin %d != %d", prefix.c_str(), g->cin, x.C); return -1; }
    if (dst) *y = *dst;
    else if (out_pad) { if (alloc4_padded(x.B, x.H, x.W, g->N, y)) return -1; }
    else *y = alloc4(x.B, x.H, x.W, g->N);
    if (!ok(y->p)) return -1;
    return gemm(g, x.p, x.ld, x.rows(), x.H, x.W, y->p, y->ld, act, res ? res->p : nullptr, res ? res->ld : 0,
                1, 0, nullptr, x.pad, y->pad);
  }
  int convT(const std::string& prefix, const T4& x, int act, T4* y, const void* res = nullptr, int ldr = 0,
            int res_after_act = 1, const int* res_bidx = nullptr, bool out_pad = false) {
    PackedGemm* g = pk_convT(prefix + ".weight", prefix + ".bias");
    if (!g) return -1;
    if (out_pad) { if (alloc4_padded(x.B, 2 * x.H, 2 * x.W, g->convt_cout, y)) return -1; }
    else *y = alloc4(x.B, 2 * x.H, 2 * x.W, g->convt_cout);
    if (!ok(y->p)) return -1;
    return gemm(g, x.p, x.ld, x.rows(), x.H, x.W, y->p, y->ld, act, res, ldr, res_after_act, 0, res_bidx, 0, y->pad);
  }
  int dwconv(const std::string& prefix, bool convlayer, const T4& x, int stride, int act, T4* y) {
    PackedDw* d;
    if (convlayer) {
      const bool has_b = find(prefix + ".conv.bias") != nullptr;
      const bool has_bn = find(prefix + ".norm.weight") != nullptr;
      d = pk_dw(prefix + ".conv.weight", has_b ? prefix + ".conv.bias" : "", has_bn ? prefix + ".norm" : "");
    } else {
      d = pk_dw(prefix + ".weight", find(prefix + ".bias") ? prefix + ".bias" : "", "");
    }
    if (!d) return -1;
    *y = alloc4(x.B, (x.H + stride - 1) / stride, (x.W + stride - 1) / stride, x.C);
    if (!ok(y->p)) return -1;
    if (dry) return 0;
    const double px_in = (double)x.rows() * x.C, px_out = (double)y->rows() * y->C;
    return prof_launch("dwconv" + std::to_string(d->ks) + "s" + std::to_string(stride), 2.0 * px_out * d->ks * d->ks,
                       (px_in + px_out) * (double)esz, [&]() {
                         return esam3_launch_dwconv(dtype, x.p, x.ld, d->w, d->bias, y->p, y->ld, x.B, x.H, x.W,
                                                    x.C, d->ks, stride, act, st);
                       });
  }
  int linear(const std::string& prefix, const void* A, int lda, int64_t M, void* out, int ldc, int act,
             const void* res = nullptr, int ldr = 0, int res_mod = 0, int out_f32 = 0) {
    PackedGemm* g = pk_linear(prefix);
    if (!g) return -1;
    return gemm(g, A, lda, M, 1, 1, out, ldc, act, res, ldr, 1, res_mod, nullptr, 0, 0, 1, out_f32);
  }
  // LayerNorm with separate row dtypes (0 f32, 1 bf16): the fp32 residual stream of the bf16 engine
  int layernorm_io(int in_dtype, int out_dtype, const std::string& prefix, const void* x, void* out, int64_t rows, int C, float eps) {
    float* g = fvec(prefix + ".weight");
    float* b = fvec(prefix + ".bias");
    if (!g || !b) return -1;
    if (dry) return 0;
    return prof_launch("layernorm", 8.0 * (double)rows * C, (double)rows * C * (double)((in_dtype ? 2 : 4) + (out_dtype ? 2 : 4)),
                       [&]() { return esam3_launch_layernorm_io(in_dtype, out_dtype, x, nullptr, g, b, out, rows, C, eps, ACT_NONE, st); });
  }
  int layernorm(const std::string& prefix, const void* x, void* out, int64_t rows, int C, float eps,
                int act = ACT_NONE) {
    float* g = fvec(prefix + ".weight");
    float* b = fvec(prefix + ".bias");
    if (!g || !b) return -1;
    if (dry) return 0;
    return prof_launch("layernorm", 8.0 * (double)rows * C, 2.0 * (double)rows * C * (double)esz,
                       [&]() { return esam3_launch_layernorm(dtype, x, nullptr, g, b, out, rows, C, eps, act, st); });
  }

  // ---------------- graphs ----------------
  int mbconv(const std::string& p, const T4& x, int stride, bool residual, T4* y, const T4* dst = nullptr);
  int evit_block(const std::string& p, const T4& x, T4* y);
  int backbone(const float* img, int B, const esam3_image_features* out, T4* feat);
  int stem_weights(const std::string& wname, const std::string& bn, int cout, float** sw_out, float** sb_out);
  int stem(const std::string& wname, const std::string& bn, int cout, int act, const float* img, int B, T4* y);
  int stem_dsconv_fused(const std::string& stem_w, const std::string& stem_bn, const std::string& ds, const float* img, int B, T4* y);
  int conv_bn(const std::string& p, const T4& x, int stride, int act, T4* y, const T4* res = nullptr, const T4* dst = nullptr,
              int res_after_act = 1);
  int squeeze_excite(const std::string& p, T4& x);
  PackedDw* pk_repvggdw(const std::string& q);
  int repvit_block(const std::string& p, const T4& x, bool use_se, int stride, T4* y);
  int backbone_repvit(const float* img, int B, const esam3_image_features* out, T4* feat);
  int tap(const esam3_image_features* out, int i, const T4& t);
  int dw_launch(PackedDw* d, const T4& in, int stride, int act, T4* o);
  int dw_bn(const std::string& p, const T4& x, int stride, int act, T4* y);
  int tv_mbconv(const std::string& p, const T4& x, T4* y);
  int tv_patch_merging(const std::string& p, const T4& x, T4* y);
  int tv_block(const std::string& p, const T4& x, int heads, int ws, T4* y);
  int backbone_tinyvit(const float* img, int B, const esam3_image_features* out, T4* feat);
  float* vit_rope_table(int end, float scale);
  int backbone_vit(const float* img, int B, const esam3_image_features* out, T4* feat);
  int neck(const std::string& which, const T4& trunk, void* const outs[3], bool sam2, const T4* pre = nullptr);
  int encode(const float* img, int B, const esam3_image_features* out);
  int decode(const esam3_prompts* pr, const esam3_decode_out* out);
  int precompute_pe();
  PackedGemm* pk_kv_cat(const std::string& ap);
  // ---- PCS text-grounding detector -------------------------------------------------------------
  bool pcs_ready = false;
  bool text_causal = false;  // MobileCLIP-B masks its text self-attention causally (esam3_set_text_causal)
  PackedGemm* pk_rows(const std::string& wname, const std::string& bname, int row0, int nrows, const std::string& key,
                      int zero_from = -1);
  void* pcs_pos_table(const std::string& key, PackedGemm* g);
  int pcs_prepare();
  bool pcs_geo_pack();
  int ground(const esam3_ground_in* in, const esam3_ground_out* out);
  int text_repmixer(const std::string& p, void* x, int B, int S, int D, void* out);
  int encode_text(const int64_t* tokens, int B, int S, float* memory_sbd, float* embeds_sbd);
  int ensure_arena(size_t need);
};

using E = esam3_engine;

// MBConv (ops.py:315-367): 1x1 expand (+BN/bias, Hardswish) -> dw3x3 (stride) -> 1x1 project (+BN),
// optional identity shortcut fused into the projection's epilogue (ops.py:761-770).
int E::mbconv(const std::string& p, const T4& x, int stride, bool residual, T4* y, const T4* dst) {
  const HostTensor* pw = need(p + "point_conv.conv.weight");
  if (!pw) return -1;
  if (dst) *y = *dst;
  else *y = alloc4(x.B, (x.H + stride - 1) / stride, (x.W + stride - 1) / stride, (int)pw->shape[0]);
  if (!ok(y->p)) return -1;
  // bf16: the high-resolution MBConvs (stages 1-3) run as ONE kernel, the 4x-expanded tensor never leaves the CU
  // (mbconv_fused.hip v2); other shapes and the f32 validation mode run layer by layer
  const HostTensor* ew = need(p + "inverted_conv.conv.weight");
  if (!ew) return -1;
  const int cmid = (int)ew->shape[0];
  // round 4: mbconv3 (evit_fused.hip: depthwise phase on the matrix cores, channels up to 256) also covers the local modules of
  // stages 3-4 and the 128 -> 256 stage entry; ESAM3_MB_V2=1 (dev builds) keeps the round-2 kernel / the layer list for A/B
  const bool use_v3 = esam3_mbconv3_ok(dtype, x.C, cmid, y->C, stride) && x.ld == x.C && !esam3_dev_flag("ESAM3_MB_V2");
  if (use_v3 || (esam3_mbconv_fused2_ok(dtype, x.C, cmid, y->C, stride) && x.ld == x.C)) {
    auto pkc = [&](const std::string& q) {
      return pk_conv(q + ".conv.weight", find(q + ".conv.bias") ? q + ".conv.bias" : "",
                     find(q + ".norm.weight") ? q + ".norm" : "");
    };
    PackedGemm* g1 = pkc(p + "inverted_conv");
    PackedGemm* g2 = pkc(p + "point_conv");
    PackedDw* dw = pk_dw(p + "depth_conv.conv.weight", find(p + "depth_conv.conv.bias") ? p + "depth_conv.conv.bias" : "",
                         find(p + "depth_conv.norm.weight") ? p + "depth_conv.norm" : "");
    if (!g1 || !g2 || !dw) return -1;
    if (!g1->bias || !g2->bias) { esam3_set_error("mbconv %s: missing folded bias", p.c_str()); return -1; }
    if (dry) return 0;
    const double px = (double)x.rows(), opx = (double)y->rows();
    return prof_launch("mbconv_fused_s" + std::to_string(stride) + ":" + p.substr(p.size() > 40 ? p.size() - 40 : 0),
                       2.0 * (px * x.C * cmid + opx * cmid * 9 + opx * cmid * y->C),
                       (px * x.C * (residual ? 2 : 1) + opx * y->C) * (double)esz, [&]() {
                         if (use_v3)
                           return esam3_launch_mbconv3(x.p, y->p, g1->w, g1->Kp, g1->bias, dw->w, dw->bias, g2->w, g2->Kp,
                                                       g2->bias, x.B, x.H, x.W, x.C, cmid, y->C, stride, residual ? 1 : 0, st);
                         return esam3_launch_mbconv_fused(dtype, x.p, y->p, g1->w, g1->Kp, g1->bias, dw->w, dw->bias,
                                                          g2->w, g2->Kp, g2->bias, x.B, x.H, x.W, x.C, cmid, y->C,
                                                          stride, residual ? 1 : 0, st);
                       });
  }
  const size_t mk = arena.mark();
  T4 a, b, yy;
  CK(conv(p + "inverted_conv", true, x, ACT_HSWISH, &a));
  CK(dwconv(p + "depth_conv", true, a, stride, ACT_HSWISH, &b));
  CK(conv(p + "point_conv", true, b, ACT_NONE, &yy, residual ? &x : nullptr, y));
  arena.release(mk);
  return 0;
}

// EfficientViTBlock (ops.py:674-733) = x + LiteMLA(x) ; then x + MBConv(x)
int E::evit_block(const std::string& p, const T4& x, T4* y) {
  const std::string c = p + "context_module.main.";
  const HostTensor* qw = need(c + "qkv.conv.weight");
  if (!qw) return -1;
  const int total3 = (int)qw->shape[0];  // 3 * heads * dim
  const int heads = total3 / (3 * dim);
  T4 x1 = alloc4(x.B, x.H, x.W, x.C);
  *y = alloc4(x.B, x.H, x.W, x.C);
  if (!ok(x1.p) || !ok(y->p)) return -1;
  const size_t mk = arena.mark();
  // round 4 (bf16, dim 16, 128 / 256 channels): the whole context module in three launches (evit_fused.hip: mla1 -> kvprep ->
  // mla2); the 3C-channel qkv tensor, its aggregation and the 6C-channel concat never reach HBM.  ESAM3_NO_MLA_FUSED=1 (dev
  // builds) runs the layer list below.
  if (esam3_mla_fused_ok(dtype, x.C, dim) && x.ld == x.C && total3 == 3 * x.C && !find(c + "qkv.conv.bias") &&
      !find(c + "qkv.norm.weight") && !find(c + "aggreg.0.0.bias") && !find(c + "aggreg.0.1.bias") &&
      !find(c + "proj.conv.bias") && find(c + "proj.norm.weight") && !esam3_dev_flag("ESAM3_NO_MLA_FUSED")) {
    PackedGemm* gq = pk_conv(c + "qkv.conv.weight", "", "");
    PackedDw* d5 = pk_dw(c + "aggreg.0.0.weight", "", "");
    PackedGemm* gg = pk_conv_like_linear(c + "aggreg.0.1.weight", "");
    PackedGemm* gp = pk_conv(c + "proj.conv.weight", "", c + "proj.norm");
    if (!gq || !d5 || !gg || !gp) return -1;
    if (d5->ks != 5 || gg->K != 16 || gp->K != 2 * x.C) { esam3_set_error("evit_block %s: unexpected LiteMLA shapes", p.c_str()); return -1; }
    size_t qb, kb, tb;
    esam3_mla_fused_scratch(x.B, x.H, x.W, x.C, &qb, &kb, &tb);
    void* qms = allocb(qb);
    float* kvp = (float*)allocb(kb);
    void* tab = allocb(tb);
    if (!ok(qms) || !ok(kvp) || !ok(tab)) return -1;
    if (!dry) {
      const double px = (double)x.rows(), C_ = (double)x.C;
      CK(prof_launch("lite_mla_fused", 2.0 * px * (3 * C_ * C_ + 3 * C_ * 25 + 3 * C_ * 16 + 4 * C_ * 17 + 2 * C_ * C_),
                     px * C_ * 3.0 * (double)esz, [&]() {
                       return esam3_launch_mla_fused(x.p, x1.p, gq->w, gq->Kp, d5->w, gg->w, gg->Kp, gp->w, gp->Kp, gp->bias, qms, kvp,
                                                     tab, x.B, x.H, x.W, x.C, st);
                     }));
    }
  } else {
    // multi-scale qkv tensor [B,H,W, 2*total3]: [qkv | aggreg(qkv)]   (ops.py:656-662)
    T4 ms = alloc4(x.B, x.H, x.W, 2 * total3);
    if (!ok(ms.p)) return -1;
    T4 qkv = ms;
    qkv.C = total3;
    T4 tmp;
    CK(conv(c + "qkv", true, x, ACT_NONE, &tmp, nullptr, &qkv));
    T4 agg;
    CK(dwconv(c + "aggreg.0.0", false, qkv, 1, ACT_NONE, &agg));
    {
      // grouped 1x1 (groups = 3*heads, dim -> dim each): executed as ONE dense GEMM with the
      // block-diagonal [3C x 3C] weight matrix -- the zero blocks cost MFMA flops that are free
      // here (the op is HBM-bound), and the GEMM kernel streams at several TB/s where a
      // per-pixel 16x16 mat-vec kernel is bound by its weight reads.
      const std::string wn = c + "aggreg.0.1.weight";
      const std::string key = wn + "#blockdiag";
      if (!find(key)) {
        const HostTensor* gwt = need(wn);
        if (!gwt) return -1;
        const int gs = (int)gwt->shape[1];
        HostTensor bd;
        bd.shape = {total3, total3};
        bd.d.assign((size_t)total3 * total3, 0.f);
        for (int co = 0; co < total3; ++co)
          for (int ci = 0; ci < gs; ++ci)
            bd.d[(size_t)co * total3 + (co / gs) * gs + ci] = gwt->d[(size_t)co * gs + ci];
        raw[key] = std::move(bd);
      }
      PackedGemm* gbd = pk_conv_like_linear(key, "");
      if (!gbd) return -1;
      CK(gemm(gbd, agg.p, agg.ld, ms.rows(), 1, 1, (char*)ms.p + (size_t)total3 * esz, ms.ld, ACT_NONE));
    }
    T4 att = alloc4(x.B, x.H, x.W, 2 * heads * dim);
    float* kv = (float*)allocb(sizeof(float) * (size_t)esam3_lite_mla_scratch_floats(x.B, x.H * x.W, 2 * heads, dim));
    if (!ok(att.p) || !ok(kv)) return -1;
    if (!dry)
      CK(prof_launch("lite_mla", 4.0 * (double)ms.rows() * 2 * heads * (dim + 1) * dim,
                     (double)ms.rows() * (2.0 * total3 + 2.0 * heads * dim) * (double)esz, [&]() {
                       return esam3_launch_lite_mla(dtype, ms.p, ms.ld, att.p, att.ld, kv, x.B, x.H * x.W, 2 * heads, dim, st);
                     }));
    T4 t2;
    CK(conv(c + "proj", true, att, ACT_NONE, &t2, &x, &x1));
  }
  arena.release(mk);
  arena.release(mk);
  {  // local module = Residual(MBConv): same fused kernel, writing into y
    const std::string m = p + "local_module.main.";
    T4 ynew;
    CK(mbconv(m, x1, 1, true, &ynew, y));
    arena.release(mk);
  }
  return 0;
}

// EfficientViTBackbone.forward -> stage_final (backbone.py:150-156)
int E::tap(const esam3_image_features* out, int i, const T4& t) {
  if (out && out->stages_dev[i] && !dry)
    HIP_CHECK_RET(hipMemcpyAsync(out->stages_dev[i], t.p, (size_t)t.rows() * t.C * esz, hipMemcpyDeviceToDevice, st));
  return 0;
}

// 3x3 stride-2 pad-1 conv from the NCHW fp32 image (Cin = 3) + folded BN + activation -> NHWC
// stem conv weights [Cout][3][3][3] (+BN) -> fp32 [27][Cout] (k = tap*3 + c) and bias, cached on the device
int E::stem_weights(const std::string& wname, const std::string& bn, int cout, float** sw_out, float** sb_out) {
  const std::string key = wname + "#stem_packed";
  float *sw = nullptr, *sb = nullptr;
  auto it = fbufs.find(key);
  if (it == fbufs.end()) {
    const HostTensor* w = need(wname);
    if (!w) return -1;
    if ((int)w->shape[0] != cout || w->shape[1] != 3 || w->shape[2] != 3) { esam3_set_error("stem %s: unexpected shape", wname.c_str()); return -1; }
    std::vector<float> scale, shift;
    if (!bn_fold(bn, cout, scale, shift)) return -1;
    std::vector<float> pk(27 * (size_t)cout);
    for (int co = 0; co < cout; ++co)
      for (int c = 0; c < 3; ++c)
        for (int t = 0; t < 9; ++t)
          pk[(size_t)(t * 3 + c) * cout + co] = w->d[((size_t)co * 3 + c) * 9 + t] * scale[co];
    sw = fvec_raw(key, pk);
    sb = fvec_raw(key + ".bias", shift);
  } else {
    sw = it->second;
    sb = fbufs[key + ".bias"];
  }
  if (!sw || !sb) return -1;
  *sw_out = sw;
  *sb_out = sb;
  return 0;
}

int E::stem(const std::string& wname, const std::string& bn, int cout, int act, const float* img, int B, T4* y) {
  *y = alloc4(B, IMG / 2, IMG / 2, cout);
  if (!ok(y->p)) return -1;
  float *sw = nullptr, *sb = nullptr;
  CK(stem_weights(wname, bn, cout, &sw, &sb));
  if (!dry) CK(prof_launch("stem", 0.0, 0.0, [&]() { return esam3_launch_stem(dtype, img, sw, sb, y->p, B, IMG, IMG, cout, act, st); }));
  return 0;
}

// EfficientViT input stem with one Residual(DSConv) block and 16 channels (B1): stem conv + depthwise + pointwise +
// identity in one kernel (backbone.py:48-70, ops.py:273-312).
int E::stem_dsconv_fused(const std::string& stem_w, const std::string& stem_bn, const std::string& ds, const float* img, int B, T4* y) {
  *y = alloc4(B, IMG / 2, IMG / 2, 16);
  if (!ok(y->p)) return -1;
  float *sw = nullptr, *sb = nullptr;
  CK(stem_weights(stem_w, stem_bn, 16, &sw, &sb));
  auto has = [&](const std::string& n) { return find(n) != nullptr; };
  PackedDw* d = pk_dw(ds + "depth_conv.conv.weight", has(ds + "depth_conv.conv.bias") ? ds + "depth_conv.conv.bias" : "",
                      has(ds + "depth_conv.norm.weight") ? ds + "depth_conv.norm" : "");
  PackedGemm* g = pk_conv(ds + "point_conv.conv.weight", has(ds + "point_conv.conv.bias") ? ds + "point_conv.conv.bias" : "",
                          has(ds + "point_conv.norm.weight") ? ds + "point_conv.norm" : "");
  if (!d || !g) return -1;
  if (d->C != 16 || d->ks != 3 || g->N != 16 || g->cin != 16 || g->ksize != 1) { esam3_set_error("stem_dsconv_fused: unexpected shapes"); return -1; }
  if (dry) return 0;
  const double px = (double)B * (IMG / 2) * (IMG / 2);
  return prof_launch("stem+dsconv", 2.0 * px * 16 * (27 + 9 + 16), (double)B * 3 * IMG * IMG * 4 + px * 16 * esz, [&]() {
    return esam3_launch_stem_dsconv(dtype, img, sw, sb, d->w, d->bias, g->w, g->Kp, g->bias, y->p, B, IMG, IMG, st);
  });
}

// Conv2d_BN (repvit.py:29-37, tiny_vit.py:38-64): bias-free 1x1 / 3x3 conv `<p>.c` + BatchNorm `<p>.bn`
int E::conv_bn(const std::string& p, const T4& x, int stride, int act, T4* y, const T4* res, const T4* dst,
               int res_after_act) {
  PackedGemm* g = pk_conv(p + ".c.weight", "", p + ".bn");
  if (!g) return -1;
  if (g->cin != x.C) { esam3_set_error("conv_bn %s: Cin %d != %d", p.c_str(), g->cin, x.C); return -1; }
  if (dst) *y = *dst;
  else *y = alloc4(x.B, (x.H + stride - 1) / stride, (x.W + stride - 1) / stride, g->N);
  if (!ok(y->p)) return -1;
  return gemm(g, x.p, x.ld, y->rows(), x.H, x.W, y->p, y->ld, act, res ? res->p : nullptr, res ? res->ld : 0,
              res_after_act, 0, nullptr, x.pad, 0, stride);
}

int E::dw_launch(PackedDw* d, const T4& in, int s_, int act, T4* o) {
  if (!d) return -1;
  if (d->C != in.C) { esam3_set_error("depthwise conv: C %d != %d", d->C, in.C); return -1; }
  *o = alloc4(in.B, (in.H + s_ - 1) / s_, (in.W + s_ - 1) / s_, in.C);
  if (!ok(o->p)) return -1;
  if (dry) return 0;
  const double px_in = (double)in.rows() * in.C, px_out = (double)o->rows() * o->C;
  return prof_launch("dwconv" + std::to_string(d->ks) + "s" + std::to_string(s_), 2.0 * d->ks * d->ks * px_out,
                     (px_in + px_out) * (double)esz, [&]() {
                       return esam3_launch_dwconv(dtype, in.p, in.ld, d->w, d->bias, o->p, o->ld, in.B, in.H, in.W, in.C,
                                                  d->ks, s_, act, st);
                     });
}
// depthwise Conv2d_BN `<p>.c` + `<p>.bn`
int E::dw_bn(const std::string& p, const T4& x, int stride, int act, T4* y) {
  return dw_launch(pk_dw(p + ".c.weight", "", p + ".bn"), x, stride, act, y);
}

// timm SqueezeExcite, in place on x (`p` ends with '.')
int E::squeeze_excite(const std::string& p, T4& x) {
  const HostTensor* w1 = need(p + "fc1.weight");
  if (!w1) return -1;
  const int R = (int)w1->shape[0];
  float *d1 = fvec(p + "fc1.weight"), *b1 = fvec(p + "fc1.bias"), *d2 = fvec(p + "fc2.weight"), *b2 = fvec(p + "fc2.bias");
  if (!d1 || !b1 || !d2 || !b2) return -1;
  const size_t mk = arena.mark();
  float* sums = (float*)allocb(sizeof(float) * (size_t)esam3_squeeze_excite_scratch_floats(x.B, x.H * x.W, x.C));
  float* gate = (float*)allocb(sizeof(float) * (size_t)x.B * x.C);
  if (!ok(sums) || !ok(gate)) return -1;
  int rc = 0;
  if (!dry)
    rc = prof_launch("squeeze_excite", 0.0, 3.0 * (double)x.rows() * x.C * (double)esz, [&]() {
      return esam3_launch_squeeze_excite(dtype, x.p, x.ld, sums, gate, d1, b1, d2, b2, x.B, x.H * x.W, x.C, R, st);
    });
  arena.release(mk);
  return rc;
}

// RepVGGDW (repvit.py:84-123): bn(dw3x3_bn(x) + dw1x1(x) + x) is ONE depthwise 3x3 with bias -- the
// reference's own fuse() algebra, evaluated in fp64 at load time.
PackedDw* E::pk_repvggdw(const std::string& q) {
  const std::string key = q + "#fused";
  if (!find(key + ".weight")) {
    const HostTensor *w = need(q + "conv.c.weight"), *w1 = need(q + "conv1.weight"), *b1 = need(q + "conv1.bias");
    if (!w || !w1 || !b1) return nullptr;
    const int C = (int)w->shape[0];
    std::vector<float> s1, t1, s2, t2;
    if (!bn_fold(q + "conv.bn", C, s1, t1) || !bn_fold(q + "bn", C, s2, t2)) return nullptr;
    HostTensor fw, fb;
    fw.shape = {C, 1, 3, 3};
    fw.d.resize((size_t)C * 9);
    fb.shape = {C};
    fb.d.resize(C);
    for (int c = 0; c < C; ++c) {
      for (int t = 0; t < 9; ++t) {
        double v = (double)w->d[(size_t)c * 9 + t] * s1[c];
        if (t == 4) v += (double)w1->d[c] + 1.0;  // 1x1 branch and identity sit on the centre tap
        fw.d[(size_t)c * 9 + t] = (float)(v * s2[c]);
      }
      // bn(y) = y*s2 + t2 with y's bias = t1 + b1
      fb.d[c] = (float)(((double)t1[c] + b1->d[c]) * s2[c] + t2[c]);
    }
    raw[key + ".weight"] = std::move(fw);
    raw[key + ".bias"] = std::move(fb);
  }
  return pk_dw(key + ".weight", key + ".bias", "");
}

// RepViTBlock (repvit.py:125-161): token mixer (RepVGGDW [+SE] | dw3x3 s2 + 1x1) then
// Residual(1x1 C->2C, GELU, 1x1 2C->C) with the shortcut fused into the last GEMM's epilogue.
int E::repvit_block(const std::string& p, const T4& x, bool use_se, int stride, T4* y) {
  const HostTensor* ow = need(p + "channel_mixer.m.2.c.weight");
  if (!ow) return -1;
  const T4 dst = alloc4(x.B, (x.H + stride - 1) / stride, (x.W + stride - 1) / stride, (int)ow->shape[0]);
  if (!ok(dst.p)) return -1;
  const size_t mk = arena.mark();  // everything below is scratch of this block
  T4 tm;
  if (stride == 2) {
    if (use_se) { esam3_set_error("RepViT: SE in a stride-2 block is not supported"); return -1; }
    T4 d;
    CK(dw_bn(p + "token_mixer.0", x, 2, ACT_NONE, &d));
    CK(conv_bn(p + "token_mixer.2", d, 1, ACT_NONE, &tm));
  } else {
    CK(dw_launch(pk_repvggdw(p + "token_mixer.0."), x, 1, ACT_NONE, &tm));
    if (use_se) CK(squeeze_excite(p + "token_mixer.1.", tm));
  }
  static const bool no_fmlp = esam3_dev_flag("ESAM3_NO_FUSED_MLP") != 0;  // A/B timing
  if (!no_fmlp && esam3_fused_mlp_ok(dtype, tm.C, 2 * tm.C, dst.C) && tm.ld == tm.C) {
    // channel mixer Residual(1x1 C -> 2C, GELU, 1x1 2C -> C) in one launch: the 2C-channel tensor stays in registers
    PackedGemm* g1 = pk_conv(p + "channel_mixer.m.0.c.weight", "", p + "channel_mixer.m.0.bn");
    PackedGemm* g2 = pk_mlp2(p + "channel_mixer.m.2.c.weight", "", p + "channel_mixer.m.2.bn");
    if (!g1 || !g2) return -1;
    if (g1->Kp == g1->K && g1->N == 2 * tm.C && g1->bias) {
      *y = dst;
      CK(fused_mlp(g1, g2, tm.p, tm.ld, tm.rows(), tm.p, tm.ld, dst.p, dst.ld, ACT_GELU, p + "channel_mixer"));
      arena.release(mk);
      return 0;
    }
  }
  T4 h;
  CK(conv_bn(p + "channel_mixer.m.0", tm, 1, ACT_GELU, &h));
  CK(conv_bn(p + "channel_mixer.m.2", h, 1, ACT_NONE, y, &tm, &dst));
  arena.release(mk);
  return 0;
}

// RepViTTrunkWrapper.forward (model_builder.py:862-865): patch embed (2x Conv3x3 s2 + BN, GELU
// between) then every block of model.features.
int E::backbone_repvit(const float* img, int B, const esam3_image_features* out, T4* feat) {
  const std::string p = EVBB + "features.";
  const int c0 = rv_cfg[0].c;
  T4 s1, x;
  CK(stem(p + "0.0.c.weight", p + "0.0.bn", c0 / 2, ACT_GELU, img, B, &s1));
  CK(conv_bn(p + "0.2", s1, 2, ACT_NONE, &x));
  int stage = 0;
  for (size_t i = 0; i < rv_cfg.size(); ++i) {
    if (rv_cfg[i].stride == 2) CK(tap(out, stage++, x));
    T4 y;
    CK(repvit_block(p + std::to_string(i + 1) + ".", x, rv_cfg[i].se != 0, rv_cfg[i].stride, &y));
    x = y;
  }
  CK(tap(out, stage, x));
  *feat = x;
  return 0;
}

// TinyViT MBConv (tiny_vit.py:87-125): 1x1 + BN, GELU, dw3x3 + BN, GELU, 1x1 + BN, + shortcut, GELU
int E::tv_mbconv(const std::string& p, const T4& x, T4* y) {
  const T4 dst = alloc4(x.B, x.H, x.W, x.C);
  if (!ok(dst.p)) return -1;
  // bf16, 64 channels (TinyViT-5M / -11M layer 0 at 252^2): conv1 -> GELU -> dw3x3 -> GELU -> conv3 -> + x -> GELU in one kernel, the
  // 4x-expanded tensor (1 GB at B = 32) stays in the CU (mbconv_fused.hip v2, GELU variant)
  static const bool no_fused_tv = esam3_dev_flag("ESAM3_NO_FUSED_TV_MBCONV") != 0;  // A/B timing
  if (!no_fused_tv && x.C == 64 && x.ld == x.C && esam3_mbconv_fused2_ok(dtype, 64, 256, 64, 1)) {
    PackedGemm* g1 = pk_conv(p + "conv1.c.weight", "", p + "conv1.bn");
    PackedGemm* g2 = pk_conv(p + "conv3.c.weight", "", p + "conv3.bn");
    PackedDw* dw = pk_dw(p + "conv2.c.weight", "", p + "conv2.bn");
    if (!g1 || !g2 || !dw) return -1;
    if (g1->N == 256 && g1->cin == 64 && g2->N == 64 && g2->cin == 256 && dw->C == 256 && dw->ks == 3 && g1->bias && g2->bias) {
      *y = dst;
      if (dry) return 0;
      const double px = (double)x.rows();
      // round 6: the persistent matrix-core kernel of the EfficientViT MBConvs (evit_fused.hip: mbconv3s) with GELU epilogues; the
      // round-2 kernel (mbconv_fused.hip v2) was the largest launch of a TinyViT step (2 x 1.18 ms, 0.08 of its floor).  A/B: ESAM3_MB_V2
      const bool v3 = esam3_mbconv3_ok(dtype, 64, 256, 64, 1) && !esam3_dev_flag("ESAM3_MB_V2");
      return prof_launch("mbconv_fused_gelu:" + p.substr(p.size() > 40 ? p.size() - 40 : 0), 2.0 * px * 256 * (64 + 9 + 64),
                         px * 64 * 3 * (double)esz, [&]() {
                           if (v3)
                             return esam3_launch_mbconv3(x.p, y->p, g1->w, g1->Kp, g1->bias, dw->w, dw->bias, g2->w, g2->Kp, g2->bias,
                                                         x.B, x.H, x.W, 64, 256, 64, 1, /*shortcut + GELU variant*/ 3, st);
                           return esam3_launch_mbconv_fused(dtype, x.p, y->p, g1->w, g1->Kp, g1->bias, dw->w, dw->bias, g2->w, g2->Kp,
                                                            g2->bias, x.B, x.H, x.W, 64, 256, 64, 1, /*shortcut + GELU variant*/ 3, st);
                         });
    }
  }
  const size_t mk = arena.mark();
  T4 a, d;
  CK(conv_bn(p + "conv1", x, 1, ACT_GELU, &a));
  CK(dw_bn(p + "conv2", a, 1, ACT_GELU, &d));
  CK(conv_bn(p + "conv3", d, 1, ACT_GELU, y, &x, &dst, /*res_after_act=*/0));
  arena.release(mk);
  return 0;
}

// PatchMerging (tiny_vit.py:128-154): 1x1 + BN, GELU, dw3x3 s2 + BN, GELU, 1x1 + BN
int E::tv_patch_merging(const std::string& p, const T4& x, T4* y) {
  const HostTensor* w = need(p + "conv3.c.weight");
  if (!w) return -1;
  const T4 dst = alloc4(x.B, (x.H + 1) / 2, (x.W + 1) / 2, (int)w->shape[0]);
  if (!ok(dst.p)) return -1;
  // round 6, bf16, 64 -> 128 and 128 -> 256 (TinyViT-5M / -11M): conv1 -> GELU -> depthwise 3x3 stride 2 -> GELU -> conv3 in ONE launch of the
  // 8-wave LDS-weight MBConv kernel (evit_fused.hip: mbconv3b<S = 2> with the middle width = the output width); the layer-by-layer path wrote
  // and re-read conv1's output at the INPUT resolution (520 MB + 520 MB at 252^2 x 128, B = 32).  A/B (dev builds): ESAM3_NO_FUSED_TV_PM
  const int cout = (int)w->shape[0];
  if (x.ld == x.C && esam3_patch_merging_fused_ok(dtype, x.C, cout) && !esam3_dev_flag("ESAM3_NO_FUSED_TV_PM")) {
    PackedGemm* g1 = pk_conv(p + "conv1.c.weight", "", p + "conv1.bn");
    PackedGemm* g2 = pk_conv(p + "conv3.c.weight", "", p + "conv3.bn");
    PackedDw* dw = pk_dw(p + "conv2.c.weight", "", p + "conv2.bn");
    if (!g1 || !g2 || !dw) return -1;
    if (g1->N == cout && g1->cin == x.C && g2->N == cout && g2->cin == cout && dw->C == cout && dw->ks == 3 && g1->bias && g2->bias) {
      *y = dst;
      if (dry) return 0;
      const double px = (double)x.rows(), opx = (double)dst.rows();
      return prof_launch("patch_merging_fused:" + p.substr(p.size() > 40 ? p.size() - 40 : 0), 2.0 * px * x.C * cout + 2.0 * opx * cout * (9 + cout),
                         (px * x.C + opx * cout) * (double)esz, [&]() {
                           return esam3_launch_mbconv3(x.p, y->p, g1->w, g1->Kp, g1->bias, dw->w, dw->bias, g2->w, g2->Kp, g2->bias, x.B, x.H,
                                                       x.W, x.C, cout, cout, 2, /*PatchMerging variant*/ 4, st);
                         });
    }
  }
  const size_t mk = arena.mark();
  T4 a, d;
  CK(conv_bn(p + "conv1", x, 1, ACT_GELU, &a));
  CK(dw_bn(p + "conv2", a, 2, ACT_GELU, &d));
  CK(conv_bn(p + "conv3", d, 1, ACT_NONE, y, nullptr, &dst));
  arena.release(mk);
  return 0;
}

// TinyViTBlock (tiny_vit.py:339-380) on NHWC tokens: x + proj(window_attention(qkv(LN(x)))), dw3x3 + BN
// local conv, x + fc2(GELU(fc1(LN(x)))).  The window partition (with its zero padding) is index
// arithmetic inside the attention kernel; padded positions use the constant qkv(LN(0)).
int E::tv_block(const std::string& p, const T4& x, int heads, int ws, T4* y) {
  const int C = x.C;
  if (C != heads * 32) { esam3_set_error("TinyViT block %s: head dim %d != 32", p.c_str(), C / heads); return -1; }
  const T4 dst = alloc4(x.B, x.H, x.W, C);
  if (!ok(dst.p)) return -1;
  const size_t mk = arena.mark();
  const int64_t rows = x.rows();
  // constant qkv of a padded token: W_qkv . LN(0) + b = W_qkv . ln_bias + b   (fp64 on the host)
  const std::string padkey = p + "attn.qkv#pad";
  void* pad_qkv = nullptr;
  {
    auto it = tbufs.find(padkey);
    if (it == tbufs.end()) {
      const HostTensor *w = need(p + "attn.qkv.weight"), *b = need(p + "attn.qkv.bias"), *lb = need(p + "attn.norm.bias");
      if (!w || !b || !lb) return -1;
      std::vector<float> v(3 * (size_t)C);
      for (int o = 0; o < 3 * C; ++o) {
        double a = b->d[o];
        for (int c = 0; c < C; ++c) a += (double)w->d[(size_t)o * C + c] * lb->d[c];
        v[o] = (float)a;
      }
      pad_qkv = upload_T(v);
      if (!pad_qkv) return -1;
      tbufs[padkey] = pad_qkv;
    } else {
      pad_qkv = it->second;
    }
  }
  float* bias = fvec(p + "attn.attention_biases");
  if (!bias) return -1;
  void* ln1 = allocb((size_t)rows * C * esz);
  void* qkv = allocb((size_t)rows * 3 * C * esz);
  void* att = allocb((size_t)rows * C * esz);
  T4 x1 = alloc4(x.B, x.H, x.W, C);
  if (!ok(ln1) || !ok(qkv) || !ok(att) || !ok(x1.p)) return -1;
  CK(layernorm(p + "attn.norm", x.p, ln1, rows, C, 1e-5f));
  CK(linear(p + "attn.qkv", ln1, C, rows, qkv, 3 * C, ACT_NONE));
  if (!dry)
    CK(prof_launch("window_attn" + std::to_string(ws), 4.0 * (double)rows * ws * ws * C, 5.0 * (double)rows * C * (double)esz, [&]() {
      return esam3_launch_window_attn(dtype, qkv, 3 * C, pad_qkv, bias, att, C, x.B, x.H, x.W, heads, ws, st);
    }));
  CK(linear(p + "attn.proj", att, C, rows, x1.p, C, ACT_NONE, x.p, x.ld));
  T4 x2;
  CK(dw_bn(p + "local_conv", x1, 1, ACT_NONE, &x2));
  CK(layernorm(p + "mlp.norm", x2.p, ln1, rows, C, 1e-5f));
  static const bool no_fmlp = esam3_dev_flag("ESAM3_NO_FUSED_MLP") != 0;  // A/B timing
  bool fused = false;
  if (!no_fmlp && esam3_fused_mlp_ok(dtype, C, 4 * C, C)) {  // fc1 -> GELU -> fc2 + shortcut in one launch (layer 1: C = 128)
    PackedGemm* g1 = pk_linear(p + "mlp.fc1");
    PackedGemm* g2 = pk_mlp2(p + "mlp.fc2.weight", p + "mlp.fc2.bias", "");
    if (!g1 || !g2) return -1;
    if (g1->Kp == g1->K && g1->bias) {
      CK(fused_mlp(g1, g2, ln1, C, rows, x2.p, x2.ld, dst.p, dst.ld, ACT_GELU, p + "mlp"));
      fused = true;
    }
  }
  if (!fused) {
    void* hid = allocb((size_t)rows * 4 * C * esz);
    if (!ok(hid)) return -1;
    CK(linear(p + "mlp.fc1", ln1, C, rows, hid, 4 * C, ACT_GELU));
    CK(linear(p + "mlp.fc2", hid, 4 * C, rows, dst.p, C, ACT_NONE, x2.p, x2.ld));
  }
  *y = dst;
  arena.release(mk);
  return 0;
}

// TinyViTTrunkWrapper.forward (model_builder.py:883-896): patch_embed, ConvLayer, 3 BasicLayers
int E::backbone_tinyvit(const float* img, int B, const esam3_image_features* out, T4* feat) {
  const std::string p = EVBB;
  T4 s1, x, y;
  CK(stem(p + "patch_embed.seq.0.c.weight", p + "patch_embed.seq.0.bn", tv_dims[0] / 2, ACT_GELU, img, B, &s1));
  CK(conv_bn(p + "patch_embed.seq.2", s1, 2, ACT_NONE, &x));
  CK(tap(out, 0, x));
  for (size_t li = 0; li < tv_dims.size(); ++li) {
    const std::string q = p + "layers." + std::to_string(li) + ".";
    for (int bi = 0; bi < tv_depths[li]; ++bi) {
      const std::string bp = q + "blocks." + std::to_string(bi) + ".";
      if (li == 0) CK(tv_mbconv(bp, x, &y));
      else CK(tv_block(bp, x, tv_heads[li], tv_windows[li], &y));
      x = y;
    }
    if (li + 1 < tv_dims.size()) {
      CK(tv_patch_merging(q + "downsample.", x, &y));
      x = y;
    }
    CK(tap(out, (int)li + 1, x));
  }
  *feat = x;
  return 0;
}

// compute_axial_cis (vitdet.py:41-57) as an interleaved (cos, sin) fp32 table [end*end][32]
float* E::vit_rope_table(int end, float scale) {
  const std::string key = "vit_rope#" + std::to_string(end) + "#" + std::to_string(scale);
  auto it = fbufs.find(key);
  if (it != fbufs.end()) return it->second;
  const int hd = 64, nf = hd / 4;
  std::vector<float> freq(nf);
  for (int i = 0; i < nf; ++i) freq[i] = 1.0f / std::pow(10000.0f, (float)(4 * i) / (float)hd);
  std::vector<float> cs((size_t)end * end * 2 * nf * 2);
  for (int t = 0; t < end * end; ++t) {
    const float tx = (float)(t % end) * scale, ty = (float)(t / end) * scale;
    for (int i = 0; i < 2 * nf; ++i) {
      const float ang = (i < nf ? tx * freq[i] : ty * freq[i - nf]);
      cs[((size_t)t * 2 * nf + i) * 2] = std::cos(ang);
      cs[((size_t)t * 2 * nf + i) * 2 + 1] = std::sin(ang);
    }
  }
  return fvec_raw(key, cs);
}

// ViT.forward of the SAM3 teacher (vitdet.py:796-839 with the configuration of model_builder.py:70-97):
// 14x14 patch embedding as a GEMM (+ the tiled absolute position table as a batch-broadcast residual),
// ln_pre, 32 pre-norm blocks: 24x24-window attention (global in blocks 7/15/23/31) with axial RoPE,
// Mlp 1024 -> 4736 -> 1024.  Tokens stay [B][72][72][1024]; window partition is index arithmetic.
int E::backbone_vit(const float* img, int B, const esam3_image_features* out, T4* feat) {
  const std::string p = NECK + "trunk.";
  const int D = 1024, heads = 16, P = 14, G = IMG / P, ws = 24, G0 = 24, depth = 32;
  const int64_t rows = (int64_t)B * G * G;
  const HostTensor* pw = need(p + "patch_embed.proj.weight");
  const HostTensor* pe = need(p + "pos_embed");
  if (!pw || !pe) return -1;
  const int K = 3 * P * P, ldk = (K + 7) / 8 * 8;
  if (!find(p + "patch_embed.proj#flat.weight")) {
    HostTensor f = *pw;  // [D][3][P][P] is already [D][K] row-major
    f.shape = {D, K};
    raw[p + "patch_embed.proj#flat.weight"] = std::move(f);
  }
  PackedGemm* gpe = pk_conv_like_linear(p + "patch_embed.proj#flat.weight", "");
  if (!gpe) return -1;
  void* pos_full = nullptr;  // get_abs_pos(tiling=True): drop cls, tile the G0 x G0 table over G x G
  {
    auto it = tbufs.find("vit_pos_full");
    if (it == tbufs.end()) {
      std::vector<float> v((size_t)G * G * D);
      for (int y = 0; y < G; ++y)
        for (int x = 0; x < G; ++x)
          memcpy(&v[((size_t)y * G + x) * D], &pe->d[((size_t)1 + (y % G0) * G0 + (x % G0)) * D], sizeof(float) * D);
      pos_full = upload_T(v);
      if (!pos_full) return -1;
      tbufs["vit_pos_full"] = pos_full;
    } else {
      pos_full = it->second;
    }
  }
  float* rope_win = vit_rope_table(ws, 1.0f);
  float* rope_glob = vit_rope_table(G, (float)ws / (float)G);  // rope_interp: scaled to the window extent
  if (!rope_win || !rope_glob) return -1;

  // bf16 engine: the residual stream x / y of the blocks is kept in fp32, as the reference's autocast keeps it (fp32 pos_embed + bf16
  // conv output -> fp32; every `x + branch` adds a bf16 branch to the fp32 stream, vitdet.py:339-515): LayerNorm reads fp32 rows and
  // writes the bf16 GEMM input, the two residual GEMMs of a block (attn.proj, mlp.fc2) read and write fp32 rows.
  static const bool bf16_stream = esam3_dev_flag("ESAM3_BF16_STREAM") != 0;  // A/B: round-1 behaviour
  const bool s32 = dtype == 1 && !bf16_stream && rows >= 1024;
  const int sdt = s32 ? 0 : dtype;  // dtype of the stream rows
  T4 x = alloc4(B, G, G, D), y = alloc4(B, G, G, D);  // engine-dtype views: patch embedding, stage taps, the trunk output
  void* xs = s32 ? allocb((size_t)rows * D * 4) : x.p;
  void* ys = s32 ? allocb((size_t)rows * D * 4) : y.p;
  void* ln = allocb((size_t)rows * D * esz);
  void* qkv = allocb((size_t)rows * 3 * D * esz);   // also holds the patch rows and the attention output
  void* hid = allocb((size_t)rows * 4736 * esz);
  if (!ok(x.p) || !ok(y.p) || !ok(xs) || !ok(ys) || !ok(ln) || !ok(qkv) || !ok(hid)) return -1;
  auto stream_view = [&]() -> int {  // x (engine dtype) <- xs
    if (!s32 || dry) return 0;
    return prof_launch("cast", 0.0, (double)rows * D * 6.0, [&]() { return esam3_launch_cast_from_f32(dtype, (const float*)xs, x.p, rows * D, st); });
  };
  if (!dry) CK(prof_launch("patchify", 0.0, 0.0, [&]() { return esam3_launch_patchify(dtype, img, qkv, B, IMG, P, ldk, st); }));
  CK(gemm(gpe, qkv, ldk, rows, 1, 1, y.p, D, ACT_NONE, pos_full, D, 1, G * G));
  CK(layernorm_io(dtype, sdt, p + "ln_pre", y.p, xs, rows, D, 1e-5f));
  CK(stream_view());
  CK(tap(out, 0, x));
  int stage = 1;
  for (int i = 0; i < depth; ++i) {
    const std::string q = p + "blocks." + std::to_string(i) + ".";
    const bool global = (i % 8) == 7;
    CK(layernorm_io(sdt, dtype, q + "norm1", xs, ln, rows, D, 1e-5f));
    CK(linear(q + "attn.qkv", ln, D, rows, qkv, 3 * D, ACT_NONE));
    if (!dry) {
      const double keys = global ? (double)G * G : (double)ws * ws;
      CK(prof_launch(global ? "vit_attn_global" : "vit_attn_window", 4.0 * (double)rows * keys * D,
                     4.0 * (double)rows * D * (double)esz, [&]() {
                       return esam3_launch_attn_window(dtype, qkv, 3 * D, 0, D, 2 * D, ln, D, B, G, G, global ? G : ws,
                                                       heads, 64, global ? rope_glob : rope_win, st);
                     }));
    }
    CK(linear(q + "attn.proj", ln, D, rows, ys, D, ACT_NONE, xs, D, 0, s32 ? 1 : 0));
    CK(layernorm_io(sdt, dtype, q + "norm2", ys, ln, rows, D, 1e-5f));
    CK(linear(q + "mlp.fc1", ln, D, rows, hid, 4736, ACT_GELU));
    CK(linear(q + "mlp.fc2", hid, 4736, rows, xs, D, ACT_NONE, ys, D, 0, s32 ? 1 : 0));
    if (global) {
      CK(stream_view());
      CK(tap(out, stage++, x));
    }
  }
  *feat = x;
  return 0;
}

int E::backbone(const float* img, int B, const esam3_image_features* out, T4* feat) {
  if (cfg.backbone == ESAM3_BACKBONE_VIT) return backbone_vit(img, B, out, feat);
  if (cfg.backbone == ESAM3_BACKBONE_REPVIT) return backbone_repvit(img, B, out, feat);
  if (cfg.backbone == ESAM3_BACKBONE_TINYVIT) return backbone_tinyvit(img, B, out, feat);
  auto tap = [&](int i, const T4& t) -> int { return this->tap(out, i, t); };
  // E0 stem: 3x3 s2 conv + BN + Hardswish, straight from the NCHW fp32 input
  T4 x;
  static const bool no_fused_stem = esam3_dev_flag("ESAM3_NO_FUSED_STEM") != 0;  // A/B timing
  const bool fused_stem = widths[0] == 16 && depths[0] == 1 && !no_fused_stem;
  if (fused_stem) {
    CK(stem_dsconv_fused(EVBB + "input_stem.op_list.0.conv.weight", EVBB + "input_stem.op_list.0.norm",
                         EVBB + "input_stem.op_list.1.main.", img, B, &x));
  } else {
    CK(stem(EVBB + "input_stem.op_list.0.conv.weight", EVBB + "input_stem.op_list.0.norm", widths[0], ACT_HSWISH, img, B, &x));
  }
  for (int i = 0; i < (fused_stem ? 0 : depths[0]); ++i) {  // Residual(DSConv)  ops.py:273-312
    const std::string p = EVBB + "input_stem.op_list." + std::to_string(i + 1) + ".main.";
    T4 y = alloc4(x.B, x.H, x.W, x.C);
    if (!ok(y.p)) return -1;
    const size_t mk = arena.mark();
    T4 a, t;
    CK(dwconv(p + "depth_conv", true, x, 1, ACT_HSWISH, &a));
    CK(conv(p + "point_conv", true, a, ACT_NONE, &t, &x, &y));
    arena.release(mk);
    x = y;
  }
  CK(tap(0, x));
  for (int si = 0; si < 2; ++si) {
    for (int i = 0; i < depths[si + 1]; ++i) {
      const std::string p = EVBB + "stages." + std::to_string(si) + ".op_list." + std::to_string(i) + ".main.";
      T4 y;
      CK(mbconv(p, x, i == 0 ? 2 : 1, i != 0, &y));
      x = y;
    }
    CK(tap(si + 1, x));
  }
  for (int si = 2; si < 4; ++si) {
    const std::string sp = EVBB + "stages." + std::to_string(si) + ".op_list.";
    T4 y;
    CK(mbconv(sp + "0.main.", x, 2, false, &y));
    x = y;
    for (int i = 0; i < depths[si + 1]; ++i) {
      CK(evit_block(sp + std::to_string(i + 1) + ".", x, &y));
      x = y;
    }
    CK(tap(si + 1, x));
  }
  *feat = x;
  return 0;
}

// One SimpleFPN neck (necks.py:100-125), levels x4, x2, x1; level x0.5 is computed and
// then dropped by the reference (vl_combiner.py:94-104, scalp=1) and has no observable
// output, so it is not executed.  sam2=true additionally applies conv_s0 / conv_s1
// (sam3_image_processor.py:62-75) and writes the 32/64-channel projections.
int E::neck(const std::string& which, const T4& trunk, void* const outs[3], bool sam2, const T4* pre) {
  const std::string p = NECK + which + ".";
  const int B = trunk.B;
  auto outT = [&](void* ptr, int H, int C) {
    T4 t;
    t.p = ptr; t.B = B; t.H = H; t.W = H; t.C = C; t.ld = C;
    return t;
  };
  const size_t mk = arena.mark();
  const bool fuse = cfg.fuse_linear_chains != 0;
  // First layer of a level: a ConvT k2s2 (prefix = its state-dict or composed key) or a 1x1 conv on the trunk.  With `pre`
  // (the student head's output BEFORE its bilinear 32 -> 72 resize) the layer runs on the small map and the interpolation
  // is applied to its output (resize_shuffle_kernel: a per-pixel linear map commutes with the interpolation, bias
  // included) -- 5 x fewer GEMM rows for the same result; without it, on the resized trunk as the reference does.
  auto first = [&](const std::string& prefix, bool is_convT, int act, bool out_pad, T4* y) -> int {
    if (!pre) {
      if (is_convT) return convT(prefix, trunk, act, y, nullptr, 0, 1, nullptr, out_pad);
      return conv(prefix, false, trunk, act, y, nullptr, nullptr, out_pad);
    }
    PackedGemm* g = is_convT ? pk_convT(prefix + ".weight", prefix + ".bias")
                             : pk_conv(prefix + ".weight", find(prefix + ".bias") ? prefix + ".bias" : "", "");
    if (!g) return -1;
    const int taps = is_convT ? 4 : 1, cout = is_convT ? g->convt_cout : g->N, sc = is_convT ? 2 : 1;
    if (g->cin != pre->C || cout % 8) { esam3_set_error("neck %s: unexpected shape", prefix.c_str()); return -1; }
    void* tmp = allocb((size_t)pre->rows() * g->N * esz);
    if (!ok(tmp)) return -1;
    if (out_pad) { if (alloc4_padded(B, sc * EMB, sc * EMB, cout, y)) return -1; }
    else *y = alloc4(B, sc * EMB, sc * EMB, cout);
    if (!ok(y->p)) return -1;
    if (dry) return 0;
    GemmParams q{};
    q.A = pre->p; q.Wt = g->w; q.bias = nullptr; q.out = tmp;   // the bias is added after the interpolation (it commutes)
    q.M = pre->rows(); q.N = g->N; q.K = g->K; q.Kp = g->Kp; q.H = pre->H; q.W = pre->W; q.Cin = g->cin; q.ksize = 1;
    q.lda = pre->ld; q.ldc = g->N; q.act = ACT_NONE; q.out_mode = OUT_PLAIN; q.res_after_act = 1;
    const double fl = 2.0 * (double)q.M * g->N * g->K;
    const double by = ((double)q.M * g->cin + (double)g->N * g->K + (double)q.M * g->N) * (double)esz;
    if (!prof && !watch_tag.empty() && g->tag == watch_tag) { CK(timed_gemm(g->tag, fl, by, q, st)); }
    else CK(prof_launch(g->tag, fl, by, [&]() { return esam3_launch_gemm(dtype, q, st); }));
    const double opx = (double)B * sc * EMB * sc * EMB * cout;
    return prof_launch("resize_shuffle", 8.0 * opx, ((double)q.M * g->N + opx) * (double)esz, [&]() {
      return esam3_launch_resize_shuffle(dtype, tmp, g->bias, y->p, B, pre->H, pre->W, EMB, EMB, cout, taps, act, y->pad, st);
    });
  };
  if (outs[0]) {  // level 0: ConvT -> GELU -> ConvT -> 1x1 -> 3x3   @288
    T4 a, b, c, d, t;
    // SAM3 side, bf16: dconv_2x2_1 -> conv_1x1 -> conv_3x3 is linear and is composed into ONE up-conv GEMM on the 144^2
    // input (pk_upconv): 27 % fewer MACs than ConvT' + 3x3 and the 288^2 x 256 intermediate (written once, read 1.3 x)
    // never exists.  The SAM2 side keeps ConvT' -> (3x3 o conv_s0) on the narrow kernel: its 32 output channels per
    // parity class do not fill a 256-wide tile.
    static const bool no_upconv = esam3_dev_flag("ESAM3_NO_UPCONV") != 0;  // A/B timing
    const bool upconv = fuse && !sam2 && dtype == 1 && !no_upconv && (2 * EMB) % 16 == 0 && ((int64_t)B * 4 * EMB * EMB) % 256 == 0;
    const bool upnarrow = fuse && sam2 && dtype == 1 && !no_upconv && esam3_upconv_narrow_ok(dtype, 32, DM * 2, 2 * EMB, 2 * EMB) &&
                          ((int64_t)B * 4 * EMB * EMB) % 256 == 0;
    if (upnarrow) {
      // SAM2 side: dconv_2x2_1 -> conv_1x1 -> conv_3x3 -> conv_s0 composed into the narrow up-conv (32 channels per parity class)
      CK(first(p + "0.dconv_2x2_0", true, ACT_GELU, true, &a));
      const std::string kt = p + "0.dconv_2x2_1+conv_1x1", k3 = p + "0.conv_3x3+conv_s0", k = p + "0.dconv_2x2_1+conv_1x1+conv_3x3+conv_s0";
      if (!compose_convT_1x1(p + "0.dconv_2x2_1", p + "0.conv_1x1", kt)) return -1;
      if (!compose_conv_1x1(p + "0.conv_3x3", MD + "conv_s0", k3)) return -1;
      PackedGemm* g = pk_upconv_narrow(kt, k3, k);
      if (!g) return -1;
      if (g->cin != a.C || g->convt_cout != 32) { esam3_set_error("narrow up-conv: unexpected shape"); return -1; }
      if (!dry) {
        GemmParams q{};
        q.A = a.p; q.Wt = g->wn; q.bias = g->bias; q.border_corr = g->border_corr; q.out = outs[0];
        q.M = a.rows(); q.N = g->N; q.K = g->K; q.Kp = g->Kp; q.H = a.H; q.W = a.W; q.Cin = g->cin; q.ksize = 2;
        q.lda = a.ld; q.ldc = 32; q.act = ACT_NONE; q.out_mode = OUT_CONVT2X2; q.convt_cout = 32; q.in_pad = 1; q.res_after_act = 1;
        const double fl = 2.0 * (double)q.M * g->N * g->K;
        const double by = ((double)q.M * g->cin + (double)g->N * g->K + (double)q.M * g->N) * (double)esz;
        CK(prof_launch(g->tag, fl, by, [&]() { return esam3_launch_upconv_narrow(q, st); }));
      }
      arena.release(mk);
    } else if (upconv) {
      CK(first(p + "0.dconv_2x2_0", true, ACT_GELU, true, &a));
      const std::string kt = p + "0.dconv_2x2_1+conv_1x1", k = p + "0.dconv_2x2_1+conv_1x1+conv_3x3";
      if (!compose_convT_1x1(p + "0.dconv_2x2_1", p + "0.conv_1x1", kt)) return -1;
      PackedGemm* g = pk_upconv(kt, p + "0.conv_3x3", k);
      if (!g) return -1;
      T4 o = outT(outs[0], 4 * EMB, DM);
      CK(gemm(g, a.p, a.ld, a.rows(), a.H, a.W, o.p, o.ld, ACT_NONE, nullptr, 0, 1, 0, nullptr, 1, 0));
      arena.release(mk);
    } else {
    CK(first(p + "0.dconv_2x2_0", true, ACT_GELU, false, &a));
    if (fuse) {  // ConvT o 1x1 composed into one ConvT 512 -> 256
      const std::string k = p + "0.dconv_2x2_1+conv_1x1";
      if (!compose_convT_1x1(p + "0.dconv_2x2_1", p + "0.conv_1x1", k)) return -1;
      CK(convT(k, a, ACT_NONE, &c, nullptr, 0, 1, nullptr, true));
    } else {
      CK(convT(p + "0.dconv_2x2_1", a, ACT_NONE, &b));
      CK(conv(p + "0.conv_1x1", false, b, ACT_NONE, &c, nullptr, nullptr, true));
    }
    if (sam2) {
      T4 o = outT(outs[0], 4 * EMB, 32);
      if (fuse) {  // 3x3 o conv_s0 composed into one 3x3 conv 256 -> 32
        const std::string k = p + "0.conv_3x3+conv_s0";
        if (!compose_conv_1x1(p + "0.conv_3x3", MD + "conv_s0", k)) return -1;
        CK(conv(k, false, c, ACT_NONE, &t, nullptr, &o));
      } else {
        CK(conv(p + "0.conv_3x3", false, c, ACT_NONE, &d));
        CK(conv(MD + "conv_s0", false, d, ACT_NONE, &t, nullptr, &o));
      }
    } else {
      T4 o = outT(outs[0], 4 * EMB, DM);
      CK(conv(p + "0.conv_3x3", false, c, ACT_NONE, &t, nullptr, &o));
    }
    arena.release(mk);
    }
  }
  if (outs[1]) {  // level 1: ConvT -> 1x1 -> 3x3   @144
    T4 a, c, d, t;
    if (fuse) {  // ConvT o 1x1 composed into one ConvT 1024 -> 256
      const std::string k = p + "1.dconv_2x2+conv_1x1";
      if (!compose_convT_1x1(p + "1.dconv_2x2", p + "1.conv_1x1", k)) return -1;
      CK(first(k, true, ACT_NONE, true, &c));
    } else {
      CK(first(p + "1.dconv_2x2", true, ACT_NONE, false, &a));
      CK(conv(p + "1.conv_1x1", false, a, ACT_NONE, &c, nullptr, nullptr, true));
    }
    if (sam2) {
      T4 o = outT(outs[1], 2 * EMB, 64);
      if (fuse) {
        const std::string k = p + "1.conv_3x3+conv_s1";
        if (!compose_conv_1x1(p + "1.conv_3x3", MD + "conv_s1", k)) return -1;
        CK(conv(k, false, c, ACT_NONE, &t, nullptr, &o));
      } else {
        CK(conv(p + "1.conv_3x3", false, c, ACT_NONE, &d));
        CK(conv(MD + "conv_s1", false, d, ACT_NONE, &t, nullptr, &o));
      }
    } else {
      T4 o = outT(outs[1], 2 * EMB, DM);
      CK(conv(p + "1.conv_3x3", false, c, ACT_NONE, &t, nullptr, &o));
    }
    arena.release(mk);
  }
  if (outs[2]) {  // level 2: 1x1 -> 3x3   @72
    T4 c, t;
    CK(first(p + "2.conv_1x1", false, ACT_NONE, true, &c));
    T4 o = outT(outs[2], EMB, DM);
    CK(conv(p + "2.conv_3x3", false, c, ACT_NONE, &t, nullptr, &o));
    arena.release(mk);
  }
  return 0;
}

// SAM3VLBackbone.forward_image (vl_combiner.py:81-124) for a student trunk
// (ImageStudentEncoder, model_builder.py:764-787).
int E::encode(const float* img, int B, const esam3_image_features* out) {
  arena.top = 0;
  T4 feat;
  CK(backbone(img, B, out, &feat));
  T4 trunk, head_small;
  const T4* pre = nullptr;  // student head output before its 32 -> 72 resize, when the necks consume that
  if (cfg.backbone == ESAM3_BACKBONE_VIT) {
    trunk = feat;  // the teacher's ViT output is the neck input (necks.py:100-125), no student head
  } else {
  // head: 1x1 (no bias) + BN + GELU, 3x3 + bias, bilinear 32 -> 72
  T4 h1, h2;
  {
    PackedGemm* g = pk_conv(TRUNK + "head.0.weight", "", TRUNK + "head.1");
    if (!g) return -1;
    if (alloc4_padded(feat.B, feat.H, feat.W, g->N, &h1)) return -1;
    CK(gemm(g, feat.p, feat.ld, feat.rows(), feat.H, feat.W, h1.p, h1.ld, ACT_GELU, nullptr, 0, 1, 0, nullptr, 0, 1));
  }
  CK(conv(TRUNK + "head.3", false, h1, ACT_NONE, &h2));
  trunk = h2;
  if (h2.H != EMB || h2.W != EMB) {
    // The necks take the head's output BEFORE the bilinear resize (see neck(): the first layer of every level commutes
    // with the interpolation); the resized trunk itself is only materialised when the caller asks for it.
    static const bool no_pre = esam3_dev_flag("ESAM3_NO_PRERESIZE") != 0;  // A/B timing
    if (cfg.fuse_linear_chains && !no_pre) { head_small = h2; pre = &head_small; }
    if (!pre || out->trunk_dev) {
      trunk = alloc4(B, EMB, EMB, h2.C);
      if (!ok(trunk.p)) return -1;
      if (!dry) CK(prof_launch("resize_bilinear", 0.0, 0.0, [&]() { return esam3_launch_resize_bilinear(dtype, h2.p, trunk.p, B, h2.H, h2.W, EMB, EMB, h2.C, st); }));
    } else {
      trunk.B = B; trunk.H = EMB; trunk.W = EMB; trunk.p = nullptr;  // shape only: nothing reads it
    }
  }
  }
  if (out->trunk_dev && !dry)
    HIP_CHECK_RET(hipMemcpyAsync(out->trunk_dev, trunk.p, (size_t)trunk.rows() * trunk.C * esz,
                                 hipMemcpyDeviceToDevice, st));
  if (out->sam3_fpn_dev[0] || out->sam3_fpn_dev[1] || out->sam3_fpn_dev[2])
    CK(neck("convs", trunk, out->sam3_fpn_dev, false, pre));
  if (out->sam2_fpn_dev[0] || out->sam2_fpn_dev[1] || out->sam2_fpn_dev[2]) {
    if (!cfg.interactive) { esam3_set_error("sam2 features requested but engine built with interactive=0"); return -1; }
    CK(neck("sam2_convs", trunk, out->sam2_fpn_dev, true, pre));
  }
  return 0;
}

// RepMixerBlock (mobile_clip.py:647-702) with every BatchNorm branch and layer scale folded at load
// (the reference's own reparameterize() algebra, mobile_clip.py:131-196,600-640, in fp64):
//   token mixer  x + ls*(BN_s(x) + BN_c(dw(x)) - BN_n(x))  ==  one depthwise 1x11 conv with bias
//   ConvFFN      x + ls2 * fc2(GELU(fc1(BN(dw(x)))))        ==  dw 1x11 + bias, GEMM+GELU, GEMM(+res) with ls2 in fc2
int E::text_repmixer(const std::string& p, void* x, int B, int S, int D, void* out) {
  const std::string kt = p + "#tm", kc = p + "#cf";
  if (!find(kt + ".w")) {
    const std::string t = p + "token_mixer.";
    const HostTensor *cw = need(t + "mixer.rbr_conv.0.conv.weight"), *ls = need(t + "layer_scale"),
                     *fw = need(p + "convffn.conv.conv.weight"), *ls2 = need(p + "layer_scale"),
                     *w2 = need(p + "convffn.fc2.weight"), *b2 = need(p + "convffn.fc2.bias");
    if (!cw || !ls || !fw || !ls2 || !w2 || !b2) return -1;
    const int KW = (int)cw->shape[3];
    std::vector<float> ss, st_, cs, ct, ns, nt, fs, ft;
    if (!bn_fold(t + "mixer.rbr_skip", D, ss, st_) || !bn_fold(t + "mixer.rbr_conv.0.bn", D, cs, ct) ||
        !bn_fold(t + "norm.rbr_skip", D, ns, nt) || !bn_fold(p + "convffn.conv.bn", D, fs, ft))
      return -1;
    HostTensor tw, tb, fw2, fb2, gw, gb;
    tw.shape = {KW, D}; tw.d.resize((size_t)KW * D);
    tb.shape = {D}; tb.d.resize(D);
    fw2.shape = {KW, D}; fw2.d.resize((size_t)KW * D);
    fb2.shape = {D}; fb2.d.resize(D);
    for (int c = 0; c < D; ++c) {
      for (int k = 0; k < KW; ++k) {
        double mix = (double)cw->d[(size_t)c * KW + k] * cs[c];
        double v = (double)ls->d[c] * mix;
        if (k == KW / 2) v = 1.0 + (double)ls->d[c] * (mix + (double)ss[c] - (double)ns[c]);
        tw.d[(size_t)k * D + c] = (float)v;
        fw2.d[(size_t)k * D + c] = (float)((double)fw->d[(size_t)c * KW + k] * fs[c]);
      }
      tb.d[c] = (float)((double)ls->d[c] * ((double)st_[c] + ct[c] - nt[c]));
      fb2.d[c] = ft[c];
    }
    const int Hd = (int)w2->shape[1];
    gw.shape = {D, Hd}; gw.d.resize((size_t)D * Hd);
    gb.shape = {D}; gb.d.resize(D);
    for (int c = 0; c < D; ++c) {
      for (int k = 0; k < Hd; ++k) gw.d[(size_t)c * Hd + k] = (float)((double)w2->d[(size_t)c * Hd + k] * ls2->d[c]);
      gb.d[c] = (float)((double)b2->d[c] * ls2->d[c]);
    }
    raw[kt + ".w"] = std::move(tw); raw[kt + ".b"] = std::move(tb);
    raw[kc + ".w"] = std::move(fw2); raw[kc + ".b"] = std::move(fb2);
    raw[p + "convffn.fc2#ls.weight"] = std::move(gw); raw[p + "convffn.fc2#ls.bias"] = std::move(gb);
  }
  const int KW = (int)raw[kt + ".w"].shape[0];
  float *tw = fvec(kt + ".w"), *tb = fvec(kt + ".b"), *fw = fvec(kc + ".w"), *fb = fvec(kc + ".b");
  PackedGemm* g1 = pk_conv(p + "convffn.fc1.weight", p + "convffn.fc1.bias", "");
  PackedGemm* g2 = pk_conv_like_linear(p + "convffn.fc2#ls.weight", p + "convffn.fc2#ls.bias");
  if (!tw || !tb || !fw || !fb || !g1 || !g2) return -1;
  const int64_t rows = (int64_t)B * S;
  const size_t mk = arena.mark();
  void* tm = allocb((size_t)rows * D * esz);
  void* cf = allocb((size_t)rows * D * esz);
  void* hid = allocb((size_t)rows * g1->N * esz);
  if (!ok(tm) || !ok(cf) || !ok(hid)) return -1;
  if (!dry) {
    CK(prof_launch("seq_dwconv", 0.0, 0.0, [&]() { return esam3_launch_seq_dwconv(dtype, x, tw, tb, tm, B, S, D, KW, st); }));
    CK(prof_launch("seq_dwconv", 0.0, 0.0, [&]() { return esam3_launch_seq_dwconv(dtype, tm, fw, fb, cf, B, S, D, KW, st); }));
  }
  CK(gemm(g1, cf, D, rows, 1, 1, hid, g1->N, ACT_GELU));
  CK(gemm(g2, hid, g1->N, rows, 1, 1, out, D, ACT_NONE, tm, D));
  arena.release(mk);
  return 0;
}

// TextStudentEncoder.forward after tokenisation (text_encoder_student.py:40-58) for the MobileCLIP-S0
// ("mct") text transformer: embeddings, RepMixerBlock, N x pre-norm transformer layers, RepMixerBlock,
// final LayerNorm, projector.  Outputs fp32 [S][B][256] (language_features) and [S][B][512]
// (language_embeds), the reference's sequence-first layouts.
int E::encode_text(const int64_t* tokens, int B, int S, float* memory_sbd, float* embeds_sbd) {
  arena.top = 0;
  const std::string e = TEXTP + "encoder.";
  const HostTensor *tab = need(e + "embedding_layer.weight"), *pos = need(e + "positional_embedding.pos_embed.pos_embed");
  if (!tab || !pos) return -1;
  const int D = (int)tab->shape[1], vocab = (int)tab->shape[0];
  const int pos_len = (int)pos->shape[2];
  if (S > pos_len) { esam3_set_error("encode_text: sequence length %d exceeds the positional table (%d)", S, pos_len); return -1; }
  // "mct" (MobileCLIP-S0): transformer.0 and transformer.N+1 are RepMixer blocks around N encoder layers;
  // "base" (the other students, model_builder.py:525-546): transformer.0 .. N-1 are all encoder layers
  const bool mct = !find(e + "transformer.0.pre_norm_mha.0.weight");
  const int first = mct ? 1 : 0;
  int n_layers = 0;
  while (find(e + "transformer." + std::to_string(n_layers + first) + ".pre_norm_mha.0.weight")) ++n_layers;
  const HostTensor* qw = need(e + "transformer." + std::to_string(first) + ".pre_norm_mha.1.qkv_proj.weight");
  if (!qw || n_layers == 0) { esam3_set_error("encode_text: no transformer layers in the state dict"); return -1; }
  if (D % 64) { esam3_set_error("encode_text: model dim %d is not a multiple of the 64-wide heads", D); return -1; }
  const int heads = D / 64;
  float *dtab = fvec(e + "embedding_layer.weight"), *dpos = fvec(e + "positional_embedding.pos_embed.pos_embed");
  if (!dtab || !dpos) return -1;
  const int64_t rows = (int64_t)B * S;
  void* x = allocb((size_t)rows * D * esz);
  void* y = allocb((size_t)rows * D * esz);
  void* ln = allocb((size_t)rows * D * esz);
  void* qkv = allocb((size_t)rows * 3 * D * esz);
  void* att = allocb((size_t)rows * D * esz);
  void* hid = allocb((size_t)rows * 4 * D * esz);
  if (!ok(x) || !ok(y) || !ok(ln) || !ok(qkv) || !ok(att) || !ok(hid)) return -1;
  if (!dry)
    CK(prof_launch("text_embed", 0.0, 0.0, [&]() { return esam3_launch_text_embed(dtype, tokens, dtab, dpos, x, embeds_sbd, B, S, D, vocab, st); }));
  if (mct) {
    CK(text_repmixer(e + "transformer.0.", x, B, S, D, y));
    std::swap(x, y);
  }
  // Round 5, bf16 engine: the pre-norm transformer layers keep their residual stream in fp32 (`xs`), as the reference's autocast does
  // (`x = x + dropout(mha(norm(x)))` adds a bf16 branch to an fp32 tensor, mobile_clip.py TransformerEncoder): the text memory enters
  // every class logit of the detector, and config 4's logits were the last output above the reference's own bf16 distance
  // (tests/test_pcs.py::test_config4_batch_8_on_the_yardstick_pairs: "dog" 1.68 x the reference's worst draw).
  const bool ts32 = dtype == 1;
  float* xs = ts32 ? (float*)allocb(sizeof(float) * (size_t)rows * D) : nullptr;
  if (ts32 && !ok(xs)) return -1;
  if (ts32 && !dry) CK(esam3_launch_cast_to_f32(dtype, x, xs, rows * D, st));
  for (int i = first; i < n_layers + first; ++i) {
    const std::string q = e + "transformer." + std::to_string(i) + ".";
    if (ts32) {
      CK(layernorm_io(0, dtype, q + "pre_norm_mha.0", xs, ln, rows, D, 1e-5f));
      CK(linear(q + "pre_norm_mha.1.qkv_proj", ln, D, rows, qkv, 3 * D, ACT_NONE));
      if (!dry) CK(prof_launch("text_attn", 0.0, 0.0, [&]() { return esam3_launch_text_attn(dtype, qkv, att, B, S, heads, 64, text_causal ? 1 : 0, st); }));
      CK(linear(q + "pre_norm_mha.1.out_proj", att, D, rows, xs, D, ACT_NONE, xs, D, 0, 1));
      CK(layernorm_io(0, dtype, q + "pre_norm_ffn.0", xs, ln, rows, D, 1e-5f));
      CK(linear(q + "pre_norm_ffn.1", ln, D, rows, hid, 4 * D, ACT_GELU));
      CK(linear(q + "pre_norm_ffn.4", hid, 4 * D, rows, xs, D, ACT_NONE, xs, D, 0, 1));
      continue;
    }
    CK(layernorm(q + "pre_norm_mha.0", x, ln, rows, D, 1e-5f));
    CK(linear(q + "pre_norm_mha.1.qkv_proj", ln, D, rows, qkv, 3 * D, ACT_NONE));
    if (!dry) CK(prof_launch("text_attn", 0.0, 0.0, [&]() { return esam3_launch_text_attn(dtype, qkv, att, B, S, heads, 64, text_causal ? 1 : 0, st); }));
    CK(linear(q + "pre_norm_mha.1.out_proj", att, D, rows, y, D, ACT_NONE, x, D));
    CK(layernorm(q + "pre_norm_ffn.0", y, ln, rows, D, 1e-5f));
    CK(linear(q + "pre_norm_ffn.1", ln, D, rows, hid, 4 * D, ACT_GELU));
    CK(linear(q + "pre_norm_ffn.4", hid, 4 * D, rows, x, D, ACT_NONE, y, D));
  }
  if (ts32 && !dry) CK(esam3_launch_cast_from_f32(dtype, xs, x, rows * D, st));   // the closing RepMixer / final norm read the GEMM type
  if (mct) {
    CK(text_repmixer(e + "transformer." + std::to_string(n_layers + 1) + ".", x, B, S, D, y));
  } else {
    std::swap(x, y);
  }
  CK(layernorm(e + "final_layer_norm", y, ln, rows, D, 1e-5f));
  PackedGemm* gp = pk_linear(TEXTP + "projector");
  if (!gp) return -1;
  void* mem = allocb((size_t)rows * gp->N * esz);
  if (!ok(mem)) return -1;
  CK(gemm(gp, ln, D, rows, 1, 1, mem, gp->N, ACT_NONE));
  if (!dry) CK(prof_launch("bsc_to_sbc", 0.0, 0.0, [&]() { return esam3_launch_bsc_to_sbc_f32(dtype, mem, memory_sbd, B, S, gp->N, st); }));
  return 0;
}

// =======================================================================================
// PCS text-grounding detector: Sam3Image.forward_grounding (sam3_image.py:442-493) for the prompt that
// Sam3Processor.set_text_prompt issues (one text per image + the dummy geometric prompt).  Every tensor is
// token-major [B][L][256]; image tokens are the NHWC 72x72 level of the sam3 neck as it is.
// =======================================================================================
// rows [row0, row0+nrows) of a [R][K] weight (+ bias) as their own packed GEMM; rows >= zero_from are zeroed
PackedGemm* E::pk_rows(const std::string& wname, const std::string& bname, int row0, int nrows, const std::string& key,
                       int zero_from) {
  auto it = gemms.find(key + ".weight");
  if (it != gemms.end()) return &it->second;
  const HostTensor* w = need(wname);
  if (!w) return nullptr;
  const int K = (int)w->shape[1];
  HostTensor sw;
  sw.shape = {nrows, K};
  sw.d.assign(w->d.begin() + (size_t)row0 * K, w->d.begin() + (size_t)(row0 + nrows) * K);
  if (zero_from >= 0) std::fill(sw.d.begin() + (size_t)zero_from * K, sw.d.end(), 0.f);
  raw[key + ".weight"] = std::move(sw);
  if (!bname.empty()) {
    const HostTensor* b = need(bname);
    if (!b) return nullptr;
    HostTensor sb;
    sb.shape = {nrows};
    sb.d.assign(b->d.begin() + row0, b->d.begin() + row0 + nrows);
    raw[key + ".bias"] = std::move(sb);
  }
  return pk_conv_like_linear(key + ".weight", bname.empty() ? "" : key + ".bias");
}

// pos72 . W^T (no bias) as a persistent [5184][N] table: (x + pos) W^T = x W^T + table, added as a
// batch-broadcast residual in the GEMM epilogue
void* E::pcs_pos_table(const std::string& key, PackedGemm* g) {
  auto it = tbufs.find(key);
  if (it != tbufs.end()) return it->second;
  if (!g) return nullptr;
  void* o = nullptr;
  if (hipMalloc(&o, (size_t)EMB * EMB * g->N * esz) != hipSuccess) { esam3_set_error("hipMalloc failed (%s)", key.c_str()); return nullptr; }
  owned.push_back(o);
  const bool was_dry = dry;
  dry = false;
  const int rc = gemm(g, tbufs["pcs_pos72"], DM, (int64_t)EMB * EMB, 1, 1, o, g->N, ACT_NONE);
  dry = was_dry;
  if (rc) return nullptr;
  tbufs[key] = o;
  return o;
}

// one-time constants: the sine position encoding of the 72x72 level (position_encoding.py:92-127), its
// projections through every attention that adds it to queries / keys, and the geometry encoder's CLS token
int E::pcs_prepare() {
  if (pcs_ready) return 0;
  if (!find("transformer.decoder.query_embed.weight")) { esam3_set_error("esam3_ground: no PCS detector weights were loaded"); return -1; }
  {
    std::vector<float> pos((size_t)EMB * EMB * DM);
    const int half = DM / 2;
    const float eps = 1e-6f, scale = 6.283185307179586f;
    std::vector<float> dim_t(half);
    for (int i = 0; i < half; ++i) dim_t[i] = std::pow(10000.0f, (float)(2 * (i / 2)) / (float)half);
    for (int y = 0; y < EMB; ++y)
      for (int x = 0; x < EMB; ++x) {
        const float ye = (float)(y + 1) / ((float)EMB + eps) * scale, xe = (float)(x + 1) / ((float)EMB + eps) * scale;
        float* o = &pos[((size_t)y * EMB + x) * DM];
        for (int i = 0; i < half; ++i) {
          const float py = ye / dim_t[i], px = xe / dim_t[i];
          o[i] = (i % 2 == 0) ? std::sin(py) : std::cos(py);          // channels [0,128): y
          o[half + i] = (i % 2 == 0) ? std::sin(px) : std::cos(px);   // channels [128,256): x
        }
      }
    void* d = upload_T(pos);
    if (!d) return -1;
    tbufs["pcs_pos72"] = d;
  }
  // geometry encoder: x0 = norm(final_proj(cls)) is a constant of the weights
  {
    const std::string g = "geometry_encoder.";
    const HostTensor *cls = need(g + "cls_embed.weight"), *w = need(g + "final_proj.weight"), *b = need(g + "final_proj.bias"),
                     *nw = need(g + "norm.weight"), *nb = need(g + "norm.bias");
    if (!cls || !w || !b || !nw || !nb) return -1;
    std::vector<double> y(DM);
    double mean = 0;
    for (int o = 0; o < DM; ++o) {
      double a = b->d[o];
      for (int c = 0; c < DM; ++c) a += (double)w->d[(size_t)o * DM + c] * cls->d[c];
      y[o] = a; mean += a;
    }
    mean /= DM;
    double var = 0;
    for (int o = 0; o < DM; ++o) var += (y[o] - mean) * (y[o] - mean);
    var /= DM;
    std::vector<float> x0(DM);
    for (int o = 0; o < DM; ++o) x0[o] = (float)((y[o] - mean) / std::sqrt(var + 1e-5) * nw->d[o] + nb->d[o]);
    if (!fvec_raw("pcs_geo_x0", x0)) return -1;
    std::vector<float> zero(DM, 0.f);
    if (!fvec_raw("pcs_zero_row", zero)) return -1;
  }
  const bool was_dry = dry;
  for (int i = 0; i < 3; ++i) {  // geometry layers: keys = Wk (img + pos)
    const std::string p = "geometry_encoder.encode." + std::to_string(i) + ".cross_attn_image.";
    if (!pcs_pos_table(p + "#posk", pk_rows(p + "in_proj_weight", "", DM, DM, p + "#posk_w"))) return -1;
  }
  for (int i = 0; i < 6; ++i) {
    const std::string e = "transformer.encoder.layers." + std::to_string(i) + ".self_attn.";
    // fused [q|k|v] projection of LN(x): the position encoding enters q and k only
    if (!pcs_pos_table(e + "#posqkv", pk_rows(e + "in_proj_weight", "", 0, 3 * DM, e + "#posqkv_w", 2 * DM))) return -1;
    const std::string d = "transformer.decoder.layers." + std::to_string(i) + ".cross_attn.";
    if (!pcs_pos_table(d + "#posk", pk_rows(d + "in_proj_weight", "", DM, DM, d + "#posk_w"))) return -1;
  }
  dry = was_dry;
  HIP_CHECK_RET(hipStreamSynchronize(st));
  pcs_ready = true;
  return 0;
}

// weights of the geometric-prompt token assembly (only needed when a point / box prompt arrives): summed biases of
// the projections that land on a point / box token, the 7x7 conv as a [256][49*256] tap-major linear, and the
// 258-wide box sine projection zero padded to 264 columns
bool E::pcs_geo_pack() {
  if (fbufs.count("pcs_geo_b_pt")) return true;
  const std::string g = "geometry_encoder.";
  const HostTensor *b1 = need(g + "points_direct_project.bias"), *b2 = need(g + "points_pool_project.bias"),
                   *b3 = need(g + "points_pos_enc_project.bias"), *c1 = need(g + "boxes_direct_project.bias"),
                   *c2 = need(g + "boxes_pool_project.bias"), *c3 = need(g + "boxes_pos_enc_project.bias"),
                   *wc = need(g + "boxes_pool_project.weight"), *wp = need(g + "boxes_pos_enc_project.weight");
  if (!b1 || !b2 || !b3 || !c1 || !c2 || !c3 || !wc || !wp) return false;
  std::vector<float> bp(DM), bb(DM);
  for (int c = 0; c < DM; ++c) {
    bp[c] = (float)((double)b1->d[c] + b2->d[c] + b3->d[c]);
    bb[c] = (float)((double)c1->d[c] + c2->d[c] + c3->d[c]);
  }
  HostTensor flat;
  flat.shape = {DM, 49 * DM};
  flat.d.resize((size_t)DM * 49 * DM);
  for (int n = 0; n < DM; ++n)
    for (int c = 0; c < DM; ++c)
      for (int t = 0; t < 49; ++t) flat.d[((size_t)n * 49 + t) * DM + c] = wc->d[((size_t)n * DM + c) * 49 + t];
  raw[g + "boxes_pool_project#flat.weight"] = std::move(flat);
  HostTensor pad;
  pad.shape = {DM, 264};
  pad.d.assign((size_t)DM * 264, 0.f);
  for (int n = 0; n < DM; ++n)
    for (int k = 0; k < DM + 2; ++k) pad.d[(size_t)n * 264 + k] = wp->d[(size_t)n * (DM + 2) + k];
  raw[g + "boxes_pos_enc_project#pad.weight"] = std::move(pad);
  if (!pk_conv_like_linear(g + "boxes_pool_project#flat.weight", "") || !pk_conv_like_linear(g + "boxes_pos_enc_project#pad.weight", "") ||
      !pk_conv_like_linear(g + "points_pool_project.weight", "") || !pk_conv_like_linear(g + "points_pos_enc_project.weight", "") ||
      !fvec(g + "points_direct_project.weight") || !fvec(g + "boxes_direct_project.weight") || !fvec(g + "label_embed.weight") ||
      !fvec(g + "cls_embed.weight"))
    return false;
  return fvec_raw("pcs_geo_b_pt", bp) && fvec_raw("pcs_geo_b_bx", bb);
}

int E::ground(const esam3_ground_in* in, const esam3_ground_out* out) {
  arena.top = 0;
  const int B = in->n_images, S = in->n_tokens, Np = in->n_points, Nb = in->n_boxes, Lg = Np + Nb + 1, Sp = S + Lg, Q = 200, QR = Q + 1, HEADS = 8, FF = 2048;
  const int64_t P = (int64_t)EMB * EMB;
  const void* img = in->sam3_fpn_dev[2];
  const void* pos = tbufs["pcs_pos72"];
  auto L = [&](const std::string& w, const std::string& b, int r0, int n, const std::string& key) { return pk_rows(w, b, r0, n, key); };
  auto lin = [&](PackedGemm* g, const void* A, int lda, int64_t M, void* o, int ldc, int act, const void* res = nullptr,
                 int ldr = 0, int res_mod = 0, int out_f32 = 0) -> int {
    return gemm(g, A, lda, M, 1, 1, o, ldc, act, res, ldr, 1, res_mod, nullptr, 0, 0, 1, out_f32);
  };
  auto attn = [&](const void* q, int ldq, int qo, const void* kv, int ldk, int ko, int vo, void* o, int Nq, int Nk,
                  const uint8_t* mask, const float* by = nullptr, const float* bx = nullptr, int q0 = 0) -> int {
    if (dry) return 0;
    if (dtype == 1 && !mask && !by && Nq >= 1024) {  // big plain attention (fusion encoder self-attention): MFMA
      int rc = 1;
      CK(prof_launch("pcs_attn_mfma", 4.0 * (double)B * Nq * Nk * DM, 0.0, [&]() {
        rc = esam3_launch_attn_mfma32(q, ldq, qo, kv, ldk, ko, vo, o, DM, B, Nq, Nk, HEADS, st);
        return rc < 0 ? -1 : 0;
      }));
      if (rc == 0) return 0;
    }
    if (dtype == 1 && Nq <= 1024 && Nk >= 128) {  // few queries x many keys (decoder / geometry cross-attention)
      int rc = 1;
      CK(prof_launch("pcs_attn_splitk", 4.0 * (double)B * Nq * Nk * DM, 0.0, [&]() {
        rc = esam3_launch_attn_mfma32_splitk(q, ldq, qo, kv, ldk, ko, vo, o, DM, B, Nq, Nk, HEADS, mask, by, bx, EMB, EMB, q0, st);
        return rc < 0 ? -1 : 0;
      }));
      if (rc == 0) return 0;
    }
    return prof_launch("pcs_attn", 4.0 * (double)B * Nq * Nk * DM, 0.0, [&]() {
      return esam3_launch_mha_core(dtype, q, ldq, qo, kv, ldk, ko, vo, o, DM, B, Nq, Nk, HEADS, mask, by, bx, EMB, EMB, q0, st);
    });
  };
  auto addk = [&](const void* a, const void* b, void* o, int64_t n) -> int { return dry ? 0 : esam3_launch_add(dtype, a, b, o, n, st); };
  auto LN = [&](const std::string& name, const void* x, void* o, int64_t rows) -> int { return layernorm(name, x, o, rows, DM, 1e-5f); };

  ScopeSeq scope;
  if (!dry) scope.next("SAM3Image._encode_prompt");
  // ---- prompt = [text tokens ; geometry tokens] ------------------------------------------------------
  void* prompt = allocb((size_t)B * Sp * DM * esz);
  uint8_t* pmask = (uint8_t*)allocb((size_t)B * Sp);
  if (!ok(prompt) || !ok(pmask)) return -1;
  if (!dry) CK(esam3_launch_pcs_prompt(dtype, in->language_features_dev, in->language_mask_dev, prompt, pmask, B, S, Lg, DM, st));

  // scratch shared by all stages (image-sized)
  void* t2 = allocb((size_t)B * P * DM * esz);        // LayerNorm output / attention output
  void* qkv = allocb((size_t)B * P * 3 * DM * esz);   // projections of image tokens
  void* hid = allocb((size_t)B * P * FF * esz);       // FFN hidden
  void* mem = allocb((size_t)B * P * DM * esz);       // encoder state -> memory
  void* pkv = allocb((size_t)B * Sp * 2 * DM * esz);  // k|v of the prompt tokens
  if (!ok(t2) || !ok(qkv) || !ok(hid) || !ok(mem) || !ok(pkv)) return -1;

  // ---- geometry encoder (geometry_encoders.py:732-853): [points ; boxes ; CLS] through 3 layers ---------
  {
    const std::string g = "geometry_encoder.";
    const int64_t R = (int64_t)B * Lg;
    void* x = allocb((size_t)R * DM * esz);
    void* a = allocb((size_t)R * DM * esz);
    void* b_ = allocb((size_t)R * 3 * DM * esz);
    void* h2 = allocb((size_t)R * FF * esz);
    uint8_t* geo_mask = (uint8_t*)allocb((size_t)R);
    if (!ok(x) || !ok(a) || !ok(b_) || !ok(h2) || !ok(geo_mask)) return -1;
    if (Lg == 1) {
      if (!dry) CK(esam3_launch_bcast_rows(dtype, fbufs["pcs_geo_x0"], 1, x, 1, 0, B, DM, st));
    } else {
      // token assembly: direct projections / label embeddings in one kernel, the pooled-feature and sine
      // projections as bias-free GEMMs summed onto it, then final_proj + norm
      void* a_samp = allocb((size_t)R * DM * esz);
      void* a_encp = allocb((size_t)R * DM * esz);
      void* a_roi = allocb((size_t)R * 49 * DM * esz);
      void* a_encb = allocb((size_t)R * 264 * esz);
      if (!ok(a_samp) || !ok(a_encp) || !ok(a_roi) || !ok(a_encb)) return -1;
      CK(LN(g + "img_pre_norm", img, t2, B * P));
      if (!pcs_geo_pack()) return -1;
      if (!dry)
        CK(prof_launch("pcs_geo_tokens", 0.0, 0.0, [&]() {
          return esam3_launch_geo_tokens(dtype, in->points_dev, in->point_labels_dev, in->point_mask_dev, Np, in->boxes_dev,
                                         in->box_labels_dev, in->box_mask_dev, Nb, t2, EMB, EMB,
                                         fvec(g + "points_direct_project.weight"), fbufs["pcs_geo_b_pt"],
                                         fvec(g + "boxes_direct_project.weight"), fbufs["pcs_geo_b_bx"],
                                         fvec(g + "label_embed.weight"), fvec(g + "cls_embed.weight"), x, a_samp, a_encp, a_roi,
                                         a_encb, pmask, Sp, S, geo_mask, B, st);
        }));
      if (Np > 0) {
        CK(lin(pk_conv_like_linear(g + "points_pool_project.weight", ""), a_samp, DM, R, x, DM, ACT_NONE, x, DM));
        CK(lin(pk_conv_like_linear(g + "points_pos_enc_project.weight", ""), a_encp, DM, R, x, DM, ACT_NONE, x, DM));
      }
      if (Nb > 0) {
        CK(lin(pk_conv_like_linear(g + "boxes_pool_project#flat.weight", ""), a_roi, 49 * DM, R, x, DM, ACT_NONE, x, DM));
        CK(lin(pk_conv_like_linear(g + "boxes_pos_enc_project#pad.weight", ""), a_encb, 264, R, x, DM, ACT_NONE, x, DM));
      }
      CK(lin(pk_linear(g + "final_proj"), x, DM, R, a, DM, ACT_NONE));
      CK(LN(g + "norm", a, x, R));
    }
    for (int i = 0; i < 3; ++i) {
      const std::string p = g + "encode." + std::to_string(i) + ".";
      CK(LN(p + "norm1", x, a, R));
      if (Lg == 1) {
        // self-attention over a single token: softmax over one key = 1 -> out_proj(v_proj(LN(x)))
        CK(lin(L(p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", 2 * DM, DM, p + "self_attn.#v"), a, DM, R, b_, DM, ACT_NONE));
        CK(lin(pk_linear(p + "self_attn.out_proj"), b_, DM, R, x, DM, ACT_NONE, x, DM));
      } else {
        CK(lin(L(p + "self_attn.in_proj_weight", p + "self_attn.in_proj_bias", 0, 3 * DM, p + "self_attn.#qkv"), a, DM, R, b_, 3 * DM, ACT_NONE));
        CK(attn(b_, 3 * DM, 0, b_, 3 * DM, DM, 2 * DM, a, Lg, Lg, geo_mask));
        CK(lin(pk_linear(p + "self_attn.out_proj"), a, DM, R, x, DM, ACT_NONE, x, DM));
      }
      // cross-attention to the image tokens, keys carry the position encoding
      CK(LN(p + "norm2", x, a, R));
      const std::string c = p + "cross_attn_image.";
      CK(lin(L(c + "in_proj_weight", c + "in_proj_bias", 0, DM, c + "#q"), a, DM, R, b_, DM, ACT_NONE));
      CK(lin(L(c + "in_proj_weight", c + "in_proj_bias", DM, DM, c + "#k"), img, DM, B * P, qkv, 2 * DM, ACT_NONE, tbufs[c + "#posk"], DM, (int)P));
      CK(lin(L(c + "in_proj_weight", c + "in_proj_bias", 2 * DM, DM, c + "#v"), img, DM, B * P, (char*)qkv + DM * esz, 2 * DM, ACT_NONE));
      CK(attn(b_, DM, 0, qkv, 2 * DM, 0, DM, a, Lg, (int)P, nullptr));
      CK(lin(pk_linear(c + "out_proj"), a, DM, R, x, DM, ACT_NONE, x, DM));
      CK(LN(p + "norm3", x, a, R));
      CK(lin(pk_linear(p + "linear1"), a, DM, R, h2, FF, ACT_RELU));
      CK(lin(pk_linear(p + "linear2"), h2, FF, R, x, DM, ACT_NONE, x, DM));
    }
    CK(LN(g + "encode_norm", x, a, R));
    if (!dry) CK(esam3_launch_copy_rows(dtype, a, Lg, prompt, Sp, S, B, DM, st));
  }

  if (!dry) scope.next("SAM3Image._run_encoder");
  // ---- fusion encoder (encoder.py:139-201,513-577): 6 pre-norm layers over the 5184 image tokens -------
  // Round 5, bf16 engine: the residual stream of the six layers is kept in fp32 (`ms`), as the reference's autocast keeps it
  // (`tgt = tgt + dropout(tgt2)` adds the bf16 branch to an fp32 tensor: 18 additions that rounded the 5184 x 256 stream to bf16 each
  // here; the box outputs, the most sensitive to the memory, sat at 1.3 x the reference's own bf16 distance in
  // tests/test_pcs.py::test_pcs_bf16_distribution_vs_reference_draws).  Every LayerNorm reads the fp32 stream and writes the GEMM
  // input type; `mem` (the bf16 view the decoder and the pixel decoder read) is cast once, after the last layer.
  static const bool enc_bf16_stream = esam3_dev_flag("ESAM3_BF16_STREAM") != 0;  // A/B: the round-4 behaviour
  const bool es32 = dtype == 1 && !enc_bf16_stream;
  float* ms = es32 ? (float*)allocb(sizeof(float) * (size_t)B * P * DM) : nullptr;
  if (es32 && !ok(ms)) return -1;
  if (!dry) {
    if (es32) CK(esam3_launch_cast_to_f32(dtype, img, ms, (int64_t)B * P * DM, st));
    else HIP_CHECK_RET(hipMemcpyAsync(mem, img, (size_t)B * P * DM * esz, hipMemcpyDeviceToDevice, st));
  }
  auto LNm = [&](const std::string& name) -> int {   // t2 = LN(stream)
    return es32 ? layernorm_io(0, dtype, name, ms, t2, B * P, DM, 1e-5f) : LN(name, mem, t2, B * P);
  };
  auto linres_m = [&](PackedGemm* g_, const void* A, int lda) -> int {   // stream += A . W^T + b
    return es32 ? lin(g_, A, lda, B * P, ms, DM, ACT_NONE, ms, DM, 0, 1) : lin(g_, A, lda, B * P, mem, DM, ACT_NONE, mem, DM);
  };
  for (int i = 0; i < 6; ++i) {
    const std::string p = "transformer.encoder.layers." + std::to_string(i) + ".";
    const std::string sa = p + "self_attn.", ca = p + "cross_attn_image.";
    CK(LNm(p + "norm1"));
    CK(lin(L(sa + "in_proj_weight", sa + "in_proj_bias", 0, 3 * DM, sa + "#qkv"), t2, DM, B * P, qkv, 3 * DM, ACT_NONE,
           tbufs[sa + "#posqkv"], 3 * DM, (int)P));
    CK(attn(qkv, 3 * DM, 0, qkv, 3 * DM, DM, 2 * DM, t2, (int)P, (int)P, nullptr));
    CK(linres_m(pk_linear(sa + "out_proj"), t2, DM));
    CK(LNm(p + "norm2"));
    CK(lin(L(ca + "in_proj_weight", ca + "in_proj_bias", 0, DM, ca + "#q"), t2, DM, B * P, qkv, DM, ACT_NONE));
    CK(lin(L(ca + "in_proj_weight", ca + "in_proj_bias", DM, 2 * DM, ca + "#kv"), prompt, DM, B * Sp, pkv, 2 * DM, ACT_NONE));
    CK(attn(qkv, DM, 0, pkv, 2 * DM, 0, DM, t2, (int)P, Sp, pmask));
    CK(linres_m(pk_linear(ca + "out_proj"), t2, DM));
    CK(LNm(p + "norm3"));
    CK(lin(pk_linear(p + "linear1"), t2, DM, B * P, hid, FF, ACT_RELU));
    CK(linres_m(pk_linear(p + "linear2"), hid, FF));
  }
  if (es32 && !dry) CK(esam3_launch_cast_from_f32(dtype, ms, mem, (int64_t)B * P * DM, st));

  if (!dry) scope.next("SAM3Image._run_decoder");
  // ---- decoder (decoder.py:33-191,417-618): [presence ; 200 queries] per image, post-norm layers, box refinement
  const std::string t = "transformer.decoder.";
  const int64_t R = (int64_t)B * QR;
  void* x = allocb((size_t)R * DM * esz);      // rows: presence token, then the queries
  void* xp = allocb((size_t)R * DM * esz);     // x + query_pos
  void* qpos = allocb((size_t)R * DM * esz);
  void* sine = allocb((size_t)R * 2 * DM * esz);
  void* da = allocb((size_t)R * DM * esz);
  void* dq = allocb((size_t)R * 3 * DM * esz);
  void* dh = allocb((size_t)R * FF * esz);
  void* hs = allocb((size_t)R * DM * esz);
  float* ref = (float*)allocb(sizeof(float) * (size_t)R * 4);
  float* by = (float*)allocb(sizeof(float) * (size_t)R * EMB * HEADS);
  float* bx = (float*)allocb(sizeof(float) * (size_t)R * EMB * HEADS);
  void* mkv = allocb((size_t)B * P * 2 * DM * esz);  // [k (with position) | v] of the memory
  if (!ok(x) || !ok(xp) || !ok(qpos) || !ok(sine) || !ok(da) || !ok(dq) || !ok(dh) || !ok(hs) || !ok(ref) || !ok(by) || !ok(bx) || !ok(mkv))
    return -1;
  {
    float* qe = fvec(t + "query_embed.weight");
    float* pt = fvec(t + "presence_token.weight");
    if (!qe || !pt) return -1;
    auto it = fbufs.find("pcs_ref0");
    if (it == fbufs.end()) {  // sigmoid(reference_points) with a dummy row for the presence token
      const HostTensor* rp = need(t + "reference_points.weight");
      if (!rp) return -1;
      std::vector<float> r0((size_t)QR * 4, 0.5f);
      for (int i = 0; i < Q * 4; ++i) r0[4 + i] = 1.f / (1.f + std::exp(-rp->d[i]));
      if (!fvec_raw("pcs_ref0", r0)) return -1;
    }
    if (!dry) {
      CK(esam3_launch_bcast_rows(dtype, pt, 1, x, QR, 0, B, DM, st));
      CK(esam3_launch_bcast_rows(dtype, qe, Q, x, QR, 1, B, DM, st));
      CK(esam3_launch_bcast_rows(0, fbufs["pcs_ref0"], QR, ref, QR, 0, B, 4, st));
    }
  }
  const float* rpbx[4] = {fvec(t + "boxRPB_embed_x.layers.0.weight"), fvec(t + "boxRPB_embed_x.layers.0.bias"),
                          fvec(t + "boxRPB_embed_x.layers.1.weight"), fvec(t + "boxRPB_embed_x.layers.1.bias")};
  const float* rpby[4] = {fvec(t + "boxRPB_embed_y.layers.0.weight"), fvec(t + "boxRPB_embed_y.layers.0.bias"),
                          fvec(t + "boxRPB_embed_y.layers.1.weight"), fvec(t + "boxRPB_embed_y.layers.1.bias")};
  for (int k = 0; k < 4; ++k) if (!rpbx[k] || !rpby[k]) return -1;
  void* ph = allocb((size_t)B * DM * esz);
  float* presence_logits = out->presence_logit_dev;
  // bf16 engine: the decoder's query stream is kept in fp32, as it is in the reference under autocast (post-norm layers: every
  // LayerNorm returns fp32, `tgt + branch` adds the bf16 branch to it, decoder.py:120-200); `x` is its bf16 view = the GEMM input.
  static const bool bf16_stream = esam3_dev_flag("ESAM3_BF16_STREAM") != 0;  // A/B: round-1 behaviour
  const bool ds32 = dtype == 1 && !bf16_stream;
  float* xs = ds32 ? (float*)allocb(sizeof(float) * (size_t)R * DM) : nullptr;
  if (ds32 && !ok(xs)) return -1;
  auto view = [&]() -> int {  // x <- xs
    if (!ds32 || dry) return 0;
    return esam3_launch_cast_from_f32(dtype, xs, x, R * DM, st);
  };
  auto LNs = [&](const std::string& name) -> int {  // stream = LN(stream)
    if (!ds32) return LN(name, x, x, R);
    CK(layernorm_io(0, 0, name, xs, xs, R, DM, 1e-5f));
    return view();
  };
  auto LNout = [&](const std::string& name, void* o) -> int {  // o (engine dtype) = LN(stream)
    return ds32 ? layernorm_io(0, dtype, name, xs, o, R, DM, 1e-5f) : LN(name, x, o, R);
  };
  auto linres = [&](PackedGemm* g, const void* A, int lda) -> int {  // stream += A . W^T + b
    return ds32 ? lin(g, A, lda, R, xs, DM, ACT_NONE, xs, DM, 0, 1) : lin(g, A, lda, R, x, DM, ACT_NONE, x, DM);
  };
  if (ds32 && !dry) CK(esam3_launch_cast_to_f32(dtype, x, xs, R * DM, st));
  // Round 5: two more rounding points of the query stream removed in the bf16 engine (tests/test_pcs.py's distribution test showed
  // the boxes' median distance to fp32 at 1.32 - 1.36 x the reference's own bf16 run while logits and masks sat at 0.8 x): the
  // conditional position is kept in fp32 and added to the fp32 stream BEFORE the one rounding to the GEMM input (the reference adds
  // `tgt + query_pos` in fp32 under autocast; x + qpos on the bf16 view rounded twice), and the box delta of bbox_embed's last
  // layer is read in fp32 by the refinement (six accumulated logit-space deltas per box).
  float* qpos32 = ds32 ? (float*)allocb(sizeof(float) * (size_t)R * DM) : nullptr;
  float* dl32 = ds32 ? (float*)allocb(sizeof(float) * (size_t)R * 8) : nullptr;
  if (ds32 && (!ok(qpos32) || !ok(dl32))) return -1;
  auto add_pos = [&]() -> int {  // xp = stream + query_pos, in the GEMM input type
    if (!ds32) return addk(x, qpos, xp, R * DM);
    return dry ? 0 : esam3_launch_add_f32_to_bf16(xs, qpos32, xp, R * DM, st);
  };
  for (int i = 0; i < 6; ++i) {
    const std::string p = t + "layers." + std::to_string(i) + ".";
    // conditional query position: MLP(sine(reference box)); zero for the presence token
    if (!dry) CK(esam3_launch_box_sine(dtype, ref, sine, R, QR, st));
    CK(lin(pk_linear(t + "ref_point_head.layers.0"), sine, 2 * DM, R, da, DM, ACT_RELU));
    if (ds32) CK(lin(pk_linear(t + "ref_point_head.layers.1"), da, DM, R, qpos32, DM, ACT_NONE, nullptr, 0, 0, 1));
    else CK(lin(pk_linear(t + "ref_point_head.layers.1"), da, DM, R, qpos, DM, ACT_NONE));
    if (!dry) {
      if (ds32) CK(esam3_launch_bcast_rows(0, fbufs["pcs_zero_row"], 1, qpos32, QR, 0, B, DM, st));
      else CK(esam3_launch_bcast_rows(dtype, fbufs["pcs_zero_row"], 1, qpos, QR, 0, B, DM, st));
      CK(prof_launch("pcs_rpb", 0.0, 0.0, [&]() { return esam3_launch_rpb_mlp(ref, rpbx, rpby, by, bx, R, QR, EMB, EMB, HEADS, st); }));
    }
    // self-attention among presence + queries: q = k = x + pos, v = x
    const std::string sa = p + "self_attn.", ct = p + "ca_text.", ci = p + "cross_attn.";
    CK(add_pos());
    CK(lin(L(sa + "in_proj_weight", sa + "in_proj_bias", 0, 2 * DM, sa + "#qk"), xp, DM, R, dq, 3 * DM, ACT_NONE));
    CK(lin(L(sa + "in_proj_weight", sa + "in_proj_bias", 2 * DM, DM, sa + "#v"), x, DM, R, (char*)dq + 2 * DM * esz, 3 * DM, ACT_NONE));
    CK(attn(dq, 3 * DM, 0, dq, 3 * DM, DM, 2 * DM, da, QR, QR, nullptr));
    CK(linres(pk_linear(sa + "out_proj"), da, DM));
    CK(LNs(p + "norm2"));
    // cross-attention to the prompt tokens
    CK(add_pos());
    CK(lin(L(ct + "in_proj_weight", ct + "in_proj_bias", 0, DM, ct + "#q"), xp, DM, R, dq, DM, ACT_NONE));
    CK(lin(L(ct + "in_proj_weight", ct + "in_proj_bias", DM, 2 * DM, ct + "#kv"), prompt, DM, B * Sp, pkv, 2 * DM, ACT_NONE));
    CK(attn(dq, DM, 0, pkv, 2 * DM, 0, DM, da, QR, Sp, pmask));
    CK(linres(pk_linear(ct + "out_proj"), da, DM));
    CK(LNs(p + "catext_norm"));
    // cross-attention to the image memory with the box-relative position bias (none for the presence token)
    CK(add_pos());
    CK(lin(L(ci + "in_proj_weight", ci + "in_proj_bias", 0, DM, ci + "#q"), xp, DM, R, dq, DM, ACT_NONE));
    CK(lin(L(ci + "in_proj_weight", ci + "in_proj_bias", DM, DM, ci + "#k"), mem, DM, B * P, mkv, 2 * DM, ACT_NONE, tbufs[ci + "#posk"], DM, (int)P));
    CK(lin(L(ci + "in_proj_weight", ci + "in_proj_bias", 2 * DM, DM, ci + "#v"), mem, DM, B * P, (char*)mkv + DM * esz, 2 * DM, ACT_NONE));
    CK(attn(dq, DM, 0, mkv, 2 * DM, 0, DM, da, QR, (int)P, nullptr, by, bx, 1));
    CK(linres(pk_linear(ci + "out_proj"), da, DM));
    CK(LNs(p + "norm1"));
    CK(lin(pk_linear(p + "linear1"), x, DM, R, dh, FF, ACT_RELU));
    CK(linres(pk_linear(p + "linear2"), dh, FF));
    CK(LNs(p + "norm3"));
    // box refinement from the normed queries (the update after the last layer IS pred_boxes)
    CK(LNout(t + "norm", hs));
    CK(lin(pk_linear(t + "bbox_embed.layers.0"), hs, DM, R, da, DM, ACT_RELU));
    CK(lin(pk_linear(t + "bbox_embed.layers.1"), da, DM, R, dq, DM, ACT_RELU));
    if (ds32) {
      CK(lin(pk_linear(t + "bbox_embed.layers.2"), dq, DM, R, dl32, 8, ACT_NONE, nullptr, 0, 0, 1));
      if (!dry) CK(esam3_launch_box_refine(0, dl32, 8, ref, R, st));
    } else {
      CK(lin(pk_linear(t + "bbox_embed.layers.2"), dq, DM, R, da, 8, ACT_NONE));
      if (!dry) CK(esam3_launch_box_refine(dtype, da, 8, ref, R, st));
    }
  }
  // presence logit of the last layer: MLP(LN(presence token)) (row 0 of every image)
  {
    void* pr = allocb((size_t)R * DM * esz);
    void* p2 = allocb((size_t)R * DM * esz);
    if (!ok(pr) || !ok(p2)) return -1;
    CK(LNout(t + "presence_token_out_norm", pr));
    CK(lin(pk_linear(t + "presence_token_head.layers.0"), pr, DM, R, p2, DM, ACT_RELU));
    CK(lin(pk_linear(t + "presence_token_head.layers.1"), p2, DM, R, pr, DM, ACT_RELU));
    CK(lin(pk_linear(t + "presence_token_head.layers.2"), pr, DM, R, p2, 8, ACT_NONE));
    if (!dry) CK(esam3_launch_strided_to_f32(dtype, p2, 8 * QR, presence_logits, B, st));
  }
  if (!dry)  // rows 1..200 of every image: the boxes after the last refinement = pred_boxes (cxcywh in [0,1])
    HIP_CHECK_RET(hipMemcpy2DAsync(out->pred_boxes_dev, (size_t)Q * 16, ref + 4, (size_t)QR * 16, (size_t)Q * 16, B,
                                   hipMemcpyDeviceToDevice, st));

  // ---- dot-product scoring (model_misc.py:37-91) ---------------------------------------------------------
  {
    const std::string d = "dot_prod_scoring.";
    void* pm = allocb((size_t)B * Sp * FF * esz);
    void* p1 = allocb((size_t)B * Sp * DM * esz);
    void* pool = allocb((size_t)B * DM * esz);
    void* hp = allocb((size_t)R * DM * esz);
    if (!ok(pm) || !ok(p1) || !ok(pool) || !ok(hp)) return -1;
    CK(lin(pk_linear(d + "prompt_mlp.layers.0"), prompt, DM, B * Sp, pm, FF, ACT_RELU));
    CK(lin(pk_linear(d + "prompt_mlp.layers.1"), pm, FF, B * Sp, p1, DM, ACT_NONE, prompt, DM));
    if (dtype == 1) {
      // bf16 engine: the normed prompt stays fp32 up to the pooled vector (LayerNorm returns fp32 under the reference's autocast and
      // mean_pool_text averages it in fp32, model_misc.py:60-81): ONE rounding, at prompt_proj's input, instead of three.  The pooled
      // vector enters every one of the 200 class logits of an image (config 4's logits sat at 1.4 x the reference's own bf16 distance)
      float* p1f = (float*)allocb(sizeof(float) * (size_t)B * Sp * DM);
      float* poolf = (float*)allocb(sizeof(float) * (size_t)B * DM);
      if (!ok(p1f) || !ok(poolf)) return -1;
      CK(layernorm_io(dtype, 0, d + "prompt_mlp.out_norm", p1, p1f, B * Sp, DM, 1e-5f));
      if (!dry) {
        CK(esam3_launch_masked_mean(0, p1f, pmask, poolf, B, Sp, DM, st));
        CK(esam3_launch_cast_from_f32(dtype, poolf, pool, (int64_t)B * DM, st));
      }
    } else {
      CK(LN(d + "prompt_mlp.out_norm", p1, p1, B * Sp));
      if (!dry) CK(esam3_launch_masked_mean(dtype, p1, pmask, pool, B, Sp, DM, st));
    }
    CK(lin(pk_linear(d + "prompt_proj"), pool, DM, B, ph, DM, ACT_NONE));
    CK(lin(pk_linear(d + "hs_proj"), hs, DM, R, hp, DM, ACT_NONE));
    if (!dry) CK(esam3_launch_dot_score(dtype, hp, QR, 1, Q, ph, out->pred_logits_dev, B, DM, 1.0f / 16.0f, 12.0f, st));
  }

  if (!dry) scope.next("SAM3Image._run_segmentation_heads");
  // ---- segmentation head (maskformer_segmentation.py:172-323) ------------------------------------------------
  {
    const std::string h = "segmentation_head.";
    const std::string ca = h + "cross_attend_prompt.";
    CK(LN(h + "cross_attn_norm", mem, t2, B * P));
    CK(lin(L(ca + "in_proj_weight", ca + "in_proj_bias", 0, DM, ca + "#q"), t2, DM, B * P, qkv, DM, ACT_NONE));
    CK(lin(L(ca + "in_proj_weight", ca + "in_proj_bias", DM, 2 * DM, ca + "#kv"), prompt, DM, B * Sp, pkv, 2 * DM, ACT_NONE));
    CK(attn(qkv, DM, 0, pkv, 2 * DM, 0, DM, t2, (int)P, Sp, pmask));
    CK(lin(pk_linear(ca + "out_proj"), t2, DM, B * P, mem, DM, ACT_NONE, mem, DM));
    // pixel decoder: 72 -> 144 -> 288, fine + nearest(coarse), 3x3 conv, GroupNorm(8), ReLU
    float* gn = (float*)allocb(sizeof(float) * (size_t)esam3_groupnorm_scratch_floats(B, 8));
    if (!ok(gn)) return -1;
    T4 prev;
    prev.p = mem; prev.B = B; prev.H = EMB; prev.W = EMB; prev.C = DM; prev.ld = DM;
    for (int li = 0; li < 2; ++li) {
      const void* fine = in->sam3_fpn_dev[1 - li];
      T4 u, c;
      if (alloc4_padded(B, 2 * prev.H, 2 * prev.W, DM, &u)) return -1;
      if (!dry) CK(prof_launch("pcs_upsample_add", 0.0, 0.0, [&]() { return esam3_launch_upsample_add(dtype, fine, prev.p, u.p, B, prev.H, prev.W, DM, st); }));
      CK(conv(h + "pixel_decoder.conv_layers." + std::to_string(li), false, u, ACT_NONE, &c));
      float* gw = fvec(h + "pixel_decoder.norms." + std::to_string(li) + ".weight");
      float* gb = fvec(h + "pixel_decoder.norms." + std::to_string(li) + ".bias");
      if (!gw || !gb) return -1;
      if (!dry) CK(prof_launch("pcs_groupnorm", 0.0, 0.0, [&]() { return esam3_launch_groupnorm_relu(dtype, c.p, gn, gw, gb, B, c.H * c.W, DM, 8, 1e-5f, st); }));
      prev = c;
    }
    const int64_t PX = (int64_t)prev.H * prev.W;  // 288 * 288
    T4 inst;
    CK(conv(h + "instance_seg_head", false, prev, ACT_NONE, &inst));
    if (out->semantic_seg_dev) {
      T4 sem;
      CK(conv(h + "semantic_seg_head", false, prev, ACT_NONE, &sem));
      if (!dry) CK(esam3_launch_cast_to_f32(dtype, sem.p, out->semantic_seg_dev, (int64_t)B * PX, st));
    }
    // mask embedding of the last layer's queries, then masks[b][q][pixel] = <embed[b][q], inst[b][pixel]>:
    // a GEMM per image whose "weight" operand is that image's pixel embedding
    void* me1 = allocb((size_t)R * DM * esz);
    void* me2 = allocb((size_t)R * DM * esz);
    void* mt = allocb((size_t)Q * PX * esz);
    if (!ok(me1) || !ok(me2) || !ok(mt)) return -1;
    CK(lin(pk_linear(h + "mask_predictor.mask_embed.layers.0"), hs, DM, R, me1, DM, ACT_RELU));
    CK(lin(pk_linear(h + "mask_predictor.mask_embed.layers.1"), me1, DM, R, me2, DM, ACT_RELU));
    CK(lin(pk_linear(h + "mask_predictor.mask_embed.layers.2"), me2, DM, R, me1, DM, ACT_NONE));
    for (int b = 0; b < B; ++b) {
      PackedGemm g;
      g.w = (char*)inst.p + (size_t)b * PX * DM * esz;
      g.N = (int)PX; g.Np = (int)PX; g.K = DM; g.Kp = DM; g.cin = DM; g.ksize = 1;
      g.tag = "segmentation_head.mask_einsum";
      CK(gemm(&g, (char*)me1 + ((size_t)b * QR + 1) * DM * esz, DM, Q, 1, 1, mt, (int)PX, ACT_NONE));
      if (!dry) CK(esam3_launch_cast_to_f32(dtype, mt, out->pred_masks_dev + (size_t)b * Q * PX, (int64_t)Q * PX, st));
    }
  }
  return 0;
}

// PositionEmbeddingRandom on the 72x72 grid (prompt_encoder.py:223-234) and its projections
// through the k_proj / q_proj weights of the image-side cross attentions, so that
// proj(keys + pe) = proj(keys) + PEproj is a residual in the GEMM epilogue.
// k_proj and v_proj of a token -> image attention concatenated into one [256][256] linear (rows: k then v)
PackedGemm* E::pk_kv_cat(const std::string& ap) {
  const std::string key = ap + "kv_proj#cat";
  auto it = gemms.find(key + ".weight");
  if (it != gemms.end()) return &it->second;
  if (!find(key + ".weight")) {
    const HostTensor *kw = need(ap + "k_proj.weight"), *vw = need(ap + "v_proj.weight"), *kbias = need(ap + "k_proj.bias"),
                     *vbias = need(ap + "v_proj.bias");
    if (!kw || !vw || !kbias || !vbias) return nullptr;
    HostTensor cw, cb;
    cw.shape = {kw->shape[0] + vw->shape[0], kw->shape[1]};
    cw.d = kw->d;
    cw.d.insert(cw.d.end(), vw->d.begin(), vw->d.end());
    cb.shape = {(int64_t)(kbias->d.size() + vbias->d.size())};
    cb.d = kbias->d;
    cb.d.insert(cb.d.end(), vbias->d.begin(), vbias->d.end());
    // the separate projections stay available to the fallback t2i() (T > 16 tokens or Bp > 65535: more than ~10 points per
    // prompt): pack them now, while the host weights exist -- after esam3_release_host_weights() a cache miss would be fatal
    if (!pk_linear(ap + "k_proj") || !pk_linear(ap + "v_proj")) return nullptr;
    raw[key + ".weight"] = std::move(cw);
    raw[key + ".bias"] = std::move(cb);
  }
  return pk_conv_like_linear(key + ".weight", key + ".bias");
}

int E::precompute_pe() {
  const HostTensor* g = need(PE + "pe_layer.positional_encoding_gaussian_matrix");
  if (!g) return -1;
  std::vector<float> pe((size_t)EMB * EMB * DM);
  for (int y = 0; y < EMB; ++y)
    for (int x = 0; x < EMB; ++x) {
      const float cx = 2.f * (((float)x + 0.5f) / (float)EMB) - 1.f;
      const float cy = 2.f * (((float)y + 0.5f) / (float)EMB) - 1.f;
      for (int f = 0; f < 128; ++f) {
        const float ang = 6.283185307179586f * (cx * g->d[f] + cy * g->d[128 + f]);
        pe[((size_t)y * EMB + x) * DM + f] = std::sin(ang);
        pe[((size_t)y * EMB + x) * DM + 128 + f] = std::cos(ang);
      }
    }
  void* pe_dev = upload_T(pe);
  if (!pe_dev) return -1;
  tbufs["dense_pe"] = pe_dev;
  const std::string names[5] = {
      MD + "transformer.layers.0.cross_attn_token_to_image.k_proj",
      MD + "transformer.layers.0.cross_attn_image_to_token.q_proj",
      MD + "transformer.layers.1.cross_attn_token_to_image.k_proj",
      MD + "transformer.layers.1.cross_attn_image_to_token.q_proj",
      MD + "transformer.final_attn_token_to_image.k_proj"};
  for (const auto& n : names) {
    // weight-only copy of the projection (bias belongs to the main GEMM)
    const HostTensor* w = need(n + ".weight");
    if (!w) return -1;
    const std::string key = n + ".weight#nobias";
    raw[key] = *w;
    PackedGemm* gm = pk_conv_like_linear(key, "");
    if (!gm) return -1;
    void* o = nullptr;
    HIP_CHECK_RET(hipMalloc(&o, (size_t)EMB * EMB * gm->N * esz));
    owned.push_back(o);
    const bool was_dry = dry;
    dry = false;
    const int rc = gemm(gm, pe_dev, DM, (int64_t)EMB * EMB, 1, 1, o, gm->N, ACT_NONE);
    dry = was_dry;
    if (rc) return -1;
    tbufs[n + "#pe"] = o;
  }
  // bf16 engine: k_proj and v_proj of a token -> image attention as ONE N = 256 GEMM over the image tokens (they are read once);
  // its position table is pe . Wk^T in the k half and zero in the v half (v takes no position encoding, transformer.py:165-170)
  if (dtype == 1) {
    const std::string aps[3] = {MD + "transformer.layers.0.cross_attn_token_to_image.", MD + "transformer.layers.1.cross_attn_token_to_image.",
                                MD + "transformer.final_attn_token_to_image."};
    for (const auto& ap : aps) {
      PackedGemm* gcat = pk_kv_cat(ap);
      if (!gcat) return -1;
      const int nkv = gcat->N, nk = nkv / 2;
      PackedGemm* gm = pk_rows(ap + "kv_proj#cat.weight", "", 0, nkv, ap + "kv_proj#pe_w", nk);   // v rows zeroed, no bias
      if (!gm) return -1;
      void* o = nullptr;
      HIP_CHECK_RET(hipMalloc(&o, (size_t)EMB * EMB * gm->N * esz));
      owned.push_back(o);
      const bool was_dry = dry;
      dry = false;
      const int rc = gemm(gm, pe_dev, DM, (int64_t)EMB * EMB, 1, 1, o, gm->N, ACT_NONE);
      dry = was_dry;
      if (rc) return -1;
      tbufs[ap + "kv_proj#pe"] = o;
    }
  }
  HIP_CHECK_RET(hipStreamSynchronize(st));
  return 0;
}

// Prompt encoder + MaskDecoder.forward (sam1_task_predictor.py:385-421, mask_decoder.py:107-242)
int E::decode(const esam3_prompts* pr, const esam3_decode_out* out) {
  arena.top = 0;
  ScopeSeq scope;
  if (!dry) scope.next("sam_mask_decoder");   // the reference's scope around the SAM heads (sam3_tracker_base.py:314)
  const int Bp = pr->n_prompts, Np = pr->n_points;
  const int pad = Np > 0 ? 1 : 0;  // _embed_points appends a pad point (prompt_encoder.py:84-88)
  const int T = 6 + Np + pad;      // output tokens + points (boxes are passed as points) + pad
  const int64_t P = (int64_t)EMB * EMB;
  const int64_t TR = (int64_t)Bp * T;
  const std::string tp = MD + "transformer.";

  // ---- tokens (sparse prompt embeddings) ------------------------------------------------
  void* tokens0 = allocb((size_t)TR * DM * esz);
  void* queries = allocb((size_t)TR * DM * esz);
  if (!ok(tokens0) || !ok(queries)) return -1;
  // bf16 engine: the TOKEN stream of the two-way transformer is kept in fp32 between layers, as the reference's autocast
  // keeps it (every LayerNorm returns fp32 and `queries + attn_out` adds a bf16 branch to that fp32 tensor,
  // sam/transformer.py:155-182; the prompt tokens and `query_pe` are fp32 parameters / embeddings): q32 is the stream,
  // t32 the positional tokens, `queries` its bf16 copy that feeds the GEMMs.  The image-side stream (5184 tokens per
  // prompt) stays bf16.  A few hundred rows: the cost is launches, not bytes.
  static const bool bf16_tokens = esam3_dev_flag("ESAM3_BF16_TOKENS") != 0;  // A/B: round-2 behaviour
  const bool tok32 = dtype == 1 && !bf16_tokens;
  float* q32 = tok32 ? (float*)allocb((size_t)TR * DM * sizeof(float)) : nullptr;
  float* t32 = tok32 ? (float*)allocb((size_t)TR * DM * sizeof(float)) : nullptr;
  if (tok32 && (!ok(q32) || !ok(t32))) return -1;
  {
    const std::string key = "out_tokens_cat";
    float* ot = nullptr;
    auto it = fbufs.find(key);
    if (it == fbufs.end()) {
      const HostTensor *a = need(MD + "obj_score_token.weight"), *b = need(MD + "iou_token.weight"),
                       *c = need(MD + "mask_tokens.weight");
      if (!a || !b || !c) return -1;
      std::vector<float> cat;
      cat.insert(cat.end(), a->d.begin(), a->d.end());
      cat.insert(cat.end(), b->d.begin(), b->d.end());
      cat.insert(cat.end(), c->d.begin(), c->d.end());
      ot = fvec_raw(key, cat);
      std::vector<float> pemb;
      for (int i = 0; i < 4; ++i) {
        const HostTensor* e = need(PE + "point_embeddings." + std::to_string(i) + ".weight");
        if (!e) return -1;
        pemb.insert(pemb.end(), e->d.begin(), e->d.end());
      }
      fvec_raw("point_emb_cat", pemb);
      const HostTensor *nm = need(SAM + "no_mem_embed"), *nk = need(PE + "no_mask_embed.weight");
      if (!nm || !nk) return -1;
      std::vector<float> cb(DM);
      for (int i = 0; i < DM; ++i) cb[i] = nm->d[i] + nk->d[i];
      fvec_raw("src_cbias", cb);
      fvec_raw("src_cbias_nomask", nm->d);  // with a mask prompt the dense embedding replaces no_mask_embed
    } else {
      ot = it->second;
    }
    float* gauss = fvec(PE + "pe_layer.positional_encoding_gaussian_matrix");
    float* nap = fvec(PE + "not_a_point_embed.weight");
    if (!ot || !gauss || !nap) return -1;
    if (!dry) {
      if (tok32) {
        CK(esam3_launch_build_tokens(0, ot, pr->coords_dev, pr->labels_dev, gauss, fbufs["point_emb_cat"], nap, t32, Bp, Np, pad,
                                     (float)IMG, st));
        HIP_CHECK_RET(hipMemcpyAsync(q32, t32, (size_t)TR * DM * sizeof(float), hipMemcpyDeviceToDevice, st));
        CK(esam3_launch_add_f32_to_bf16(t32, nullptr, queries, TR * DM, st));
      } else {
      CK(esam3_launch_build_tokens(dtype, ot, pr->coords_dev, pr->labels_dev, gauss, fbufs["point_emb_cat"],
                                   nap, tokens0, Bp, Np, pad, (float)IMG, st));
      HIP_CHECK_RET(hipMemcpyAsync(queries, tokens0, (size_t)TR * DM * esz, hipMemcpyDeviceToDevice, st));
      }
    }
  }
  // ---- src = image_embed[img] + no_mem_embed + dense(no_mask) ------------------------------
  void* keys = allocb((size_t)Bp * P * DM * esz);
  if (!ok(keys)) return -1;
  void* dense = nullptr;
  if (pr->mask_input_dev) {  // PromptEncoder._embed_masks (prompt_encoder.py:131-134)
    dense = allocb((size_t)Bp * P * DM * esz);
    if (!ok(dense)) return -1;
    const std::string mp = PE + "mask_downscaling.";
    const float* w[10] = {fvec(mp + "0.weight"), fvec(mp + "0.bias"), fvec(mp + "1.weight"), fvec(mp + "1.bias"),
                          fvec(mp + "3.weight"), fvec(mp + "3.bias"), fvec(mp + "4.weight"), fvec(mp + "4.bias"),
                          fvec(mp + "6.weight"), fvec(mp + "6.bias")};
    for (const float* q : w) if (!q) return -1;
    if (!dry)
      CK(prof_launch("mask_embed", 0.0, 0.0, [&]() {
        return esam3_launch_mask_embed(dtype, pr->mask_input_dev, w, dense, Bp, 4 * EMB, EMB, st);
      }));
  }
  if (!dry)
    CK(prof_launch("gather_add", (double)Bp * P * DM, 2.0 * (double)Bp * P * DM * (double)esz, [&]() {
      return esam3_launch_gather_add(dtype, pr->sam2_fpn_dev[2], pr->prompt_image_dev,
                                     fbufs[dense ? "src_cbias_nomask" : "src_cbias"], dense, keys, Bp, P, DM, st);
    }));

  // scratch
  void* qin = allocb((size_t)TR * DM * esz);
  void* tq = allocb((size_t)TR * DM * esz);
  void* tk = allocb((size_t)TR * DM * esz);
  void* tv = allocb((size_t)TR * DM * esz);
  void* ta = allocb((size_t)TR * DM * esz);
  void* th = allocb((size_t)TR * 2048 * esz);
  void* ikv = allocb((size_t)Bp * P * 256 * esz);   // [k | v] rows of the merged projection, or ik then iv
  if (!ok(qin) || !ok(tq) || !ok(tk) || !ok(tv) || !ok(ta) || !ok(th) || !ok(ikv)) return -1;
  void* ik = ikv;
  void* iv = (char*)ikv + (size_t)Bp * P * 128 * esz;
  static const bool no_t2i_mfma = esam3_dev_flag("ESAM3_NO_T2I_MFMA") != 0;  // A/B: separate k / v GEMMs + the VALU attention
  const bool t2i_mfma = !no_t2i_mfma && Bp <= 65535 && esam3_t2i_mfma_ok(dtype, T, (int)P, 8, 16);   // grid.y = prompt
  float* t2i_scratch = nullptr;  // per-chunk softmax partials of the token -> image attention
  {
    int64_t nf = esam3_attn_scratch_floats(Bp, T, (int)P, 8, 16);
    if (t2i_mfma) nf = std::max(nf, esam3_t2i_mfma_scratch_floats(Bp, T, (int)P));
    if (nf) {
      t2i_scratch = (float*)allocb((size_t)nf * sizeof(float));
      if (!ok(t2i_scratch)) return -1;
    }
  }

  // [k | v] rows of the image tokens = keys . Wkv^T + bias + position table: the weights-resident row kernel where the shape fits
  // (256 -> 256, bf16), the tile GEMM otherwise
  // (measured at parity with the tile GEMM, 0.066-0.069 against 0.064-0.073 ms at 32 prompts: 39 us of it are MFMA + LDS + epilogue, 18 us
  // its 16-byte-per-row stores, 8 us its 32-byte-per-row loads -- profiles/r04/rowlin256_abl.txt; the tile GEMM stays the default)
  static const bool use_rowlin = esam3_dev_flag("ESAM3_ROWLIN") != 0;  // A/B
  auto kv_rows = [&](PackedGemm* gkv, const void* table) -> int {
    if (use_rowlin && esam3_rowlin256_ok(dtype, Bp * P, gkv->N, gkv->K, (int)P)) {
      if (dry) return 0;
      const double rows = (double)Bp * P;
      return prof_launch(gkv->tag, 2.0 * rows * gkv->N * gkv->K, (2.0 * rows * 256 + (double)gkv->N * gkv->K) * (double)esz, [&]() {
        return esam3_launch_rowlin256(keys, gkv->w, gkv->Kp, gkv->bias, table, (int)P, ikv, Bp * P, st); });
    }
    return gemm(gkv, keys, DM, Bp * P, 1, 1, ikv, 256, ACT_NONE, table, 256, 1, (int)P);
  };
  auto add = [&](const void* a, const void* b, void* o, int64_t n) -> int {
    if (dry) return 0;
    return esam3_launch_add(dtype, a, b, o, n, st);
  };
  auto ln = [&](const std::string& name, void* x, int64_t rows) -> int {
    return layernorm(name, x, x, rows, DM, 1e-5f);
  };
  // token stream helpers: q + query_pe as GEMM input; residual projection into the stream; the stream's LayerNorm
  auto q_plus_pe = [&]() -> int {  // -> qin
    if (!tok32) return add(queries, tokens0, qin, TR * DM);
    return dry ? 0 : esam3_launch_add_f32_to_bf16(q32, t32, qin, TR * DM, st);
  };
  auto tok_proj = [&](const std::string& name, const void* A, int lda, bool residual) -> int {  // stream (+)= Linear(A)
    if (!tok32) return linear(name, A, lda, TR, queries, DM, ACT_NONE, residual ? queries : nullptr, DM);
    return linear(name, A, lda, TR, q32, DM, ACT_NONE, residual ? q32 : nullptr, DM, 0, 1);
  };
  auto tok_ln = [&](const std::string& name) -> int {  // stream = LN(stream); bf16 copy refreshed
    if (!tok32) return ln(name, queries, TR);
    CK(layernorm_io(0, 1, name, q32, queries, TR, DM, 1e-5f));   // bf16 copy of LN(pre-norm stream) for the GEMMs ...
    return layernorm_io(0, 0, name, q32, q32, TR, DM, 1e-5f);    // ... then the stream itself, in place
  };
  // token -> image cross attention, result added to queries and normalised
  auto t2i = [&](const std::string& ap, const std::string& norm) -> int {
    CK(q_plus_pe());
    CK(linear(ap + "q_proj", qin, DM, TR, tq, 128, ACT_NONE));
    if (t2i_mfma) {
      // one N = 256 GEMM writes [k | v] rows; the attention runs on the matrix cores out of the two halves
      PackedGemm* gkv = pk_kv_cat(ap);
      if (!gkv) return -1;
      CK(kv_rows(gkv, tbufs[ap + "kv_proj#pe"]));
      if (!dry) CK(prof_launch("attn_t2i", 4.0 * Bp * T * (double)P * 128, 2.0 * (double)Bp * P * 128 * (double)esz, [&]() {
        return esam3_launch_t2i_mfma(tq, 128, ikv, 256, (const char*)ikv + 128 * esz, 256, ta, Bp, T, (int)P, t2i_scratch, st); }));
    } else {
    CK(linear(ap + "k_proj", keys, DM, Bp * P, ik, 128, ACT_NONE, tbufs[ap + "k_proj#pe"], 128, (int)P));
    CK(linear(ap + "v_proj", keys, DM, Bp * P, iv, 128, ACT_NONE));
    if (!dry) CK(prof_launch("attn_t2i", 4.0 * Bp * T * (double)P * 128, 2.0 * (double)Bp * P * 128 * (double)esz, [&]() {
      return esam3_launch_attn(dtype, tq, 128, ik, 128, iv, 128, ta, 128, Bp, T, (int)P, 8, 16, t2i_scratch, st); }));
    }
    CK(tok_proj(ap + "out_proj", ta, 128, true));
    return tok_ln(norm);
  };

  // bf16 engine, <= 16 tokens: the token side runs as three per-prompt kernels per stage (decoder_fused.hip) between the fused
  // image-side kernels -- 5 token-side launches instead of 57
  static const bool no_tok_fused = esam3_dev_flag("ESAM3_NO_TOK_FUSED") != 0;  // A/B: the layer-by-layer token path
  const bool tok_fused = tok32 && !no_tok_fused && t2i_mfma && esam3_tok_fused_ok(dtype, T) && esam3_i2t_fused_ok(dtype, (int)P, T, 8, 16, DM) &&
                         !esam3_dev_flag("ESAM3_NO_I2T_FUSED");
  void* hyper = allocb((size_t)Bp * 4 * 32 * esz);
  float* iou4 = (float*)allocb((size_t)Bp * 8 * sizeof(float));  // the four predicted IoUs stay fp32 (see the IoU head below)
  void* obj = allocb((size_t)Bp * 8 * esz);
  if (!ok(hyper) || !ok(iou4) || !ok(obj)) return -1;
  void* tokb_scratch = nullptr;
  if (tok_fused) {
    const size_t nb = (size_t)esam3_tok_b_scratch_bytes(Bp);
    tokb_scratch = allocb(nb);
    if (!ok(tokb_scratch)) return -1;
  }
  if (tok_fused) {
    auto TL = [&](const std::string& prefix, esam3_tok_lin* o) -> int {
      PackedGemm* g = pk_linear(prefix);
      if (!g) return -1;
      o->w = g->w; o->bias = g->bias; o->ldw = g->Kp;
      return 0;
    };
    auto kv_and_attn = [&](const std::string& ap) -> int {   // [k | v] rows of the image tokens, then tq -> ta
      PackedGemm* gkv = pk_kv_cat(ap);
      if (!gkv) return -1;
      CK(kv_rows(gkv, tbufs[ap + "kv_proj#pe"]));
      if (!dry) CK(prof_launch("attn_t2i", 4.0 * Bp * T * (double)P * 128, 2.0 * (double)Bp * P * 128 * (double)esz, [&]() {
        return esam3_launch_t2i_mfma(tq, 128, ikv, 256, (const char*)ikv + 128 * esz, 256, ta, Bp, T, (int)P, t2i_scratch, st); }));
      return 0;
    };
    const std::string fa = tp + "final_attn_token_to_image.";
    for (int li = 0; li < 2; ++li) {
      const std::string lp = tp + "layers." + std::to_string(li) + ".";
      const std::string ap = lp + "cross_attn_token_to_image.", ip = lp + "cross_attn_image_to_token.";
      esam3_tok_lin la[5], lb[6];
      CK(TL(lp + "self_attn.q_proj", &la[0])); CK(TL(lp + "self_attn.k_proj", &la[1])); CK(TL(lp + "self_attn.v_proj", &la[2]));
      CK(TL(lp + "self_attn.out_proj", &la[3])); CK(TL(ap + "q_proj", &la[4]));
      CK(TL(ap + "out_proj", &lb[0])); CK(TL(lp + "mlp.lin1", &lb[1])); CK(TL(lp + "mlp.lin2", &lb[2]));
      CK(TL(ip + "k_proj", &lb[3])); CK(TL(ip + "v_proj", &lb[4])); CK(TL(fa + "q_proj", &lb[5]));
      float *g1 = fvec(lp + "norm1.weight"), *b1 = fvec(lp + "norm1.bias"), *g2 = fvec(lp + "norm2.weight"), *b2 = fvec(lp + "norm2.bias"),
            *g3 = fvec(lp + "norm3.weight"), *b3 = fvec(lp + "norm3.bias"), *g4 = fvec(lp + "norm4.weight"), *b4 = fvec(lp + "norm4.bias");
      PackedGemm *gq = pk_linear(ip + "q_proj"), *go = pk_linear(ip + "out_proj");
      if (!g1 || !b1 || !g2 || !b2 || !g3 || !b3 || !g4 || !b4 || !gq || !go) return -1;
      if (!dry) CK(prof_launch("tok_a", 0.0, 0.0, [&]() { return esam3_launch_tok_a(q32, t32, tq, la, g1, b1, 1e-5f, Bp, T, li == 0, st); }));
      CK(kv_and_attn(ap));
      if (!dry) CK(prof_launch("tok_b", 0.0, 0.0, [&]() {
        return esam3_launch_tok_b(q32, t32, ta, tk, tv, li == 1 ? tq : nullptr, lb, g2, b2, g3, b3, 1e-5f, tokb_scratch, Bp, T, st); }));
      if (!dry) {
        const double rows = (double)Bp * P;
        CK(prof_launch("i2t_fused", 2.0 * rows * (2.0 * DM * 128 + 2.0 * T * 128), 2.0 * rows * DM * (double)esz, [&]() {
          return esam3_launch_i2t_fused(keys, keys, gq->w, gq->Kp, gq->bias, tbufs[ip + "q_proj#pe"], go->w, go->Kp, go->bias, g4, b4,
                                        1e-5f, tk, 128, tv, 128, Bp, (int)P, T, st);
        }));
      }
    }
    CK(kv_and_attn(fa));
    esam3_tok_lin xo, mlps[18];
    CK(TL(fa + "out_proj", &xo));
    for (int r = 0; r < 6; ++r)
      for (int l = 0; l < 3; ++l) {
        const std::string hp = r < 4 ? MD + "output_hypernetworks_mlps." + std::to_string(r) + ".layers."
                                     : (r == 4 ? MD + "iou_prediction_head.layers." : MD + "pred_obj_score_head.layers.");
        CK(TL(hp + std::to_string(l), &mlps[r * 3 + l]));
      }
    float *gf = fvec(tp + "norm_final_attn.weight"), *bfin = fvec(tp + "norm_final_attn.bias");
    if (!gf || !bfin) return -1;
    if (!dry) CK(prof_launch("tok_d", 0.0, 0.0, [&]() {
      return esam3_launch_tok_d(q32, ta, queries, xo, gf, bfin, 1e-5f, mlps, hyper, iou4, obj, Bp, T, st); }));
  } else {
  for (int li = 0; li < 2; ++li) {
    const std::string lp = tp + "layers." + std::to_string(li) + ".";
    // (1) token self attention (transformer.py:155-163)
    const void* qk_in = queries;
    if (li > 0) { CK(q_plus_pe()); qk_in = qin; }
    CK(linear(lp + "self_attn.q_proj", qk_in, DM, TR, tq, DM, ACT_NONE));
    CK(linear(lp + "self_attn.k_proj", qk_in, DM, TR, tk, DM, ACT_NONE));
    CK(linear(lp + "self_attn.v_proj", queries, DM, TR, tv, DM, ACT_NONE));
    if (!dry) CK(prof_launch("attn", 0.0, 0.0, [&]() { return esam3_launch_attn(dtype, tq, DM, tk, DM, tv, DM, ta, DM, Bp, T, T, 8, 32, nullptr, st); }));
    CK(tok_proj(lp + "self_attn.out_proj", ta, DM, li != 0));  // layer 0 replaces the tokens (skip_first_layer_pe)
    CK(tok_ln(lp + "norm1"));
    // (2) tokens attend to the image (transformer.py:165-170)
    CK(t2i(lp + "cross_attn_token_to_image.", lp + "norm2"));
    // (3) MLP on tokens (transformer.py:172-175)
    CK(linear(lp + "mlp.lin1", queries, DM, TR, th, 2048, ACT_RELU));
    CK(tok_proj(lp + "mlp.lin2", th, 2048, true));
    CK(tok_ln(lp + "norm3"));
    // (4) image attends to the tokens (transformer.py:177-182)
    const std::string ip = lp + "cross_attn_image_to_token.";
    CK(q_plus_pe());
    CK(linear(ip + "k_proj", qin, DM, TR, tk, 128, ACT_NONE));
    CK(linear(ip + "v_proj", queries, DM, TR, tv, 128, ACT_NONE));
    static const bool no_i2t_fused = esam3_dev_flag("ESAM3_NO_I2T_FUSED") != 0;  // A/B: q_proj / attn_fewkeys / out_proj / layernorm
    if (!no_i2t_fused && esam3_i2t_fused_ok(dtype, (int)P, T, 8, 16, DM)) {
      // q_proj + attention over the T prompt tokens + out_proj + residual + norm4 in one pass over the image tokens
      PackedGemm *gq = pk_linear(ip + "q_proj"), *go = pk_linear(ip + "out_proj");
      float *gam = fvec(lp + "norm4.weight"), *bet = fvec(lp + "norm4.bias");
      if (!gq || !go || !gam || !bet) return -1;
      if (!dry) {
        const double rows = (double)Bp * P;
        CK(prof_launch("i2t_fused", 2.0 * rows * (2.0 * DM * 128 + 2.0 * T * 128), 2.0 * rows * DM * (double)esz, [&]() {
          return esam3_launch_i2t_fused(keys, keys, gq->w, gq->Kp, gq->bias, tbufs[ip + "q_proj#pe"], go->w, go->Kp, go->bias, gam, bet,
                                        1e-5f, tk, 128, tv, 128, Bp, (int)P, T, st);
        }));
      }
    } else {
      CK(linear(ip + "q_proj", keys, DM, Bp * P, ik, 128, ACT_NONE, tbufs[ip + "q_proj#pe"], 128, (int)P));
      if (!dry) CK(prof_launch("attn_fewkeys", 0.0, 0.0, [&]() { return esam3_launch_attn_fewkeys(dtype, ik, 128, tk, 128, tv, 128, iv, 128, Bp, (int)P, T, 8, 16, st); }));
      CK(linear(ip + "out_proj", iv, 128, Bp * P, keys, DM, ACT_NONE, keys, DM));
      CK(ln(lp + "norm4", keys, Bp * P));
    }
  }
  CK(t2i(tp + "final_attn_token_to_image.", tp + "norm_final_attn"));
  }
  // queries == hs [Bp][T][256]; keys == src [Bp][72*72][256]

  // ---- upscaling with high-res features (mask_decoder.py:213-222) -------------------------
  T4 src;
  src.p = keys; src.B = Bp; src.H = EMB; src.W = EMB; src.C = DM; src.ld = DM;
  T4 u1, u2;
  CK(convT(MD + "output_upscaling.0", src, ACT_NONE, &u1, pr->sam2_fpn_dev[1], 64, 1, pr->prompt_image_dev));
  CK(layernorm(MD + "output_upscaling.1", u1.p, u1.p, u1.rows(), u1.C, 1e-6f, ACT_GELU));
  // bf16: output_upscaling.3 + GELU + the hypernetwork product run as one kernel below (the 288^2 x 32 tensor is never stored)
  static const bool no_fused_up = esam3_dev_flag("ESAM3_NO_FUSED_UPSCALE") != 0;  // A/B timing
  const bool fused_up = dtype == 1 && !no_fused_up;
  if (!fused_up) CK(convT(MD + "output_upscaling.3", u1, ACT_GELU, &u2, pr->sam2_fpn_dev[0], 32, 0, pr->prompt_image_dev));

  // ---- hypernetwork MLPs, IoU head, object-score head (mask_decoder.py:224-242) ------------
  void* h1 = allocb((size_t)Bp * DM * esz);
  void* h2 = allocb((size_t)Bp * DM * esz);
  float* all_masks = (float*)allocb(sizeof(float) * (size_t)Bp * 4 * 16 * P);
  int* counters = (int*)allocb(sizeof(int) * 2 * (size_t)Bp);
  if (!ok(h1) || !ok(h2) || !ok(all_masks) || !ok(counters)) return -1;
  const size_t tok_stride = (size_t)DM * esz;
  if (!tok_fused) {
  for (int i = 0; i < 4; ++i) {
    const std::string hp = MD + "output_hypernetworks_mlps." + std::to_string(i) + ".layers.";
    const char* tok = (const char*)queries + (size_t)(2 + i) * tok_stride;  // mask token i of every prompt
    CK(linear(hp + "0", tok, T * DM, Bp, h1, DM, ACT_RELU));
    CK(linear(hp + "1", h1, DM, Bp, h2, DM, ACT_RELU));
    CK(linear(hp + "2", h2, DM, Bp, (char*)hyper + (size_t)i * 32 * esz, 4 * 32, ACT_NONE));
  }
  {
    const std::string hp = MD + "iou_prediction_head.layers.";
    const char* tok = (const char*)queries + (size_t)1 * tok_stride;
    CK(linear(hp + "0", tok, T * DM, Bp, h1, DM, ACT_RELU));
    CK(linear(hp + "1", h1, DM, Bp, h2, DM, ACT_RELU));
    // The four scores leave the last layer in fp32 (fp32 accumulators + sigmoid, no bf16 rounding of a value in [0, 1)):
    // they decide the single-mask fallback by argmax and are returned to the caller as float32 anyway
    // (mask_decoder.py:236-242, 256-290).
    // (the fp32-output GEMM kernel for few rows takes <= 2048 rows per launch)
    for (int r0 = 0; r0 < Bp; r0 += 2048) {
      const int rows = Bp - r0 < 2048 ? Bp - r0 : 2048;
      CK(linear(hp + "2", (const char*)h2 + (size_t)r0 * DM * esz, DM, rows, iou4 + (size_t)r0 * 8, 8, ACT_SIGMOID, nullptr, 0, 0,
                dtype == 1 ? 1 : 0));
    }
  }
  {
    const std::string hp = MD + "pred_obj_score_head.layers.";
    CK(linear(hp + "0", queries, T * DM, Bp, h1, DM, ACT_RELU));
    CK(linear(hp + "1", h1, DM, Bp, h2, DM, ACT_RELU));
    CK(linear(hp + "2", h2, DM, Bp, obj, 8, ACT_NONE));
  }
  }
  const int64_t P4 = 16 * P;  // 288 * 288
  PackedGemm* gu = nullptr;  // packed in the dry pass too: weight uploads must not happen while a graph is being captured
  if (fused_up && !(gu = pk_convT(MD + "output_upscaling.3.weight", MD + "output_upscaling.3.bias"))) return -1;
  if (!dry) {
    if (fused_up) {
      if (gu->N != 128 || gu->K != 64 || gu->convt_cout != 32) { esam3_set_error("output_upscaling.3: unexpected shape"); return -1; }
      const double px = (double)Bp * 4 * P;  // pixels of the 144^2 map
      CK(prof_launch("upscale+mask", 2.0 * px * 128 * 64 + 2.0 * px * 4 * 32 * 4, (px * 64 + px * 4 * 32) * esz + px * 16 * 4.0, [&]() {
        return esam3_launch_upscale_mask(u1.p, gu->w, gu->Kp, gu->bias, pr->sam2_fpn_dev[0], pr->prompt_image_dev, hyper, 32,
                                         all_masks, Bp, 2 * EMB, st);
      }));
    } else {
      CK(esam3_launch_mask_product(dtype, hyper, 32, u2.p, all_masks, Bp, P4, 32, st));
    }
    CK(esam3_launch_select_masks(0 /* fp32 scores */, all_masks, iou4, 8, out->low_res_dev, out->iou_dev, counters, Bp, P4,
                                 pr->multimask_output, 0.05f, 0.98f, st));
    if (out->obj_score_dev) CK(esam3_launch_strided_to_f32(dtype, obj, 8, out->obj_score_dev, Bp, st));
  }
  return 0;
}

int E::ensure_arena(size_t need_bytes) {
  if (need_bytes <= arena.cap) return 0;
  if (arena.base) {
    HIP_CHECK_RET(hipDeviceSynchronize());
    HIP_CHECK_RET(hipFree(arena.base));
    arena.base = nullptr;
    arena.cap = 0;
  }
  const size_t cap = need_bytes + (need_bytes >> 4) + (1 << 20);
  void* p = nullptr;
  if (hipMalloc(&p, cap) != hipSuccess) {
    esam3_set_error("hipMalloc of %zu-byte workspace failed", cap);
    return -1;
  }
  arena.base = (char*)p;
  arena.cap = cap;
  return 0;
}

// --------------------------------------------------------------------------------------
// C ABI
// --------------------------------------------------------------------------------------
namespace {
// Selects the engine's device for the duration of a C-ABI call and gives the caller's current device back afterwards
// (two engines on different GPUs in one process must not change each other's -- or the host framework's -- device).
struct DeviceGuard {
  int prev = -1;
  bool ok = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
}  // namespace

extern "C" {

int esam3_create(const esam3_config* cfg, esam3_engine** out) {
  if (!cfg || !out) { esam3_set_error("esam3_create: null argument"); return -1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    esam3_set_error("no HIP device available (the engine has no CPU fallback)");
    return -1;
  }
  if (cfg->device < 0 || cfg->device >= ndev) { esam3_set_error("bad device ordinal %d", cfg->device); return -1; }
  DeviceGuard guard(cfg->device);
  if (!guard.ok) { esam3_set_error("esam3_create: hipSetDevice(%d) failed", cfg->device); return -1; }
  if (cfg->backbone != ESAM3_BACKBONE_EFFICIENTVIT && cfg->backbone != ESAM3_BACKBONE_REPVIT &&
      cfg->backbone != ESAM3_BACKBONE_TINYVIT && cfg->backbone != ESAM3_BACKBONE_VIT) {
    esam3_set_error("unsupported backbone %d", cfg->backbone);
    return -1;
  }
  esam3_engine* e = new esam3_engine();
  e->cfg = *cfg;
  e->dtype = cfg->dtype == ESAM3_F32 ? 0 : 1;
  e->esz = e->dtype == 0 ? 4 : 2;
  const std::string mn(cfg->model_name);
  if (cfg->backbone == ESAM3_BACKBONE_VIT) {  // one configuration: the ViT-H of _create_vit_backbone
    *out = e;
    return 0;
  }
  if (cfg->backbone == ESAM3_BACKBONE_TINYVIT) {
    e->tv_depths = {2, 2, 6, 2};
    e->tv_windows = {7, 7, 14, 7};
    if (mn == "5m") { e->tv_dims = {64, 128, 160, 320}; e->tv_heads = {2, 4, 5, 10}; }
    else if (mn == "11m") { e->tv_dims = {64, 128, 256, 448}; e->tv_heads = {2, 4, 8, 14}; }
    else if (mn == "21m") { e->tv_dims = {96, 192, 384, 576}; e->tv_heads = {3, 6, 12, 18}; }
    else { esam3_set_error("unknown TinyViT model '%s'", mn.c_str()); delete e; return -1; }
    *out = e;
    return 0;
  }
  if (cfg->backbone == ESAM3_BACKBONE_REPVIT) {
    // (channels, SE, stride) per block: repvit.py:320-350 (m0_9), :386-416 (m1_1), :446-506 (m2_3)
    auto stage = [&](int c, int n_s1, bool first_stage, int tail_plain) {
      if (!first_stage) e->rv_cfg.push_back({c, 0, 2});
      for (int i = 0; i < n_s1; ++i) e->rv_cfg.push_back({c, (i % 2 == 0 && i < n_s1 - tail_plain) ? 1 : 0, 1});
    };
    if (mn == "m1.1" || mn == "m1_1") { stage(64, 3, true, 1); stage(128, 3, false, 1); stage(256, 13, false, 1); stage(512, 2, false, 0); }
    else if (mn == "m2.3" || mn == "m2_3") { stage(80, 7, true, 1); stage(160, 7, false, 1); stage(320, 35, false, 1); stage(640, 2, false, 0); }
    else if (mn == "m0.9" || mn == "m0_9") { stage(48, 3, true, 1); stage(96, 3, false, 1); stage(192, 15, false, 1); stage(384, 2, false, 0); }
    else { esam3_set_error("unknown RepViT model '%s'", mn.c_str()); delete e; return -1; }
    *out = e;
    return 0;
  }
  if (mn == "b0") { e->widths = {8, 16, 32, 64, 128}; e->depths = {1, 2, 2, 2, 2}; e->dim = 16; }
  else if (mn == "b1") { e->widths = {16, 32, 64, 128, 256}; e->depths = {1, 2, 3, 3, 4}; e->dim = 16; }
  else if (mn == "b2") { e->widths = {24, 48, 96, 192, 384}; e->depths = {1, 3, 4, 4, 6}; e->dim = 32; }
  else { esam3_set_error("unknown EfficientViT model '%s'", mn.c_str()); delete e; return -1; }
  *out = e;
  return 0;
}

void esam3_destroy(esam3_engine* e) {
  if (!e) return;
  DeviceGuard guard(e->cfg.device);
  (void)hipDeviceSynchronize();

  for (void* p : e->owned) (void)hipFree(p);
  if (e->arena.base) (void)hipFree(e->arena.base);
  delete e;
}

int esam3_load_weight(esam3_engine* e, const char* name, const float* data, const int64_t* shape, int ndim) {
  if (!e || !name || !data) { esam3_set_error("esam3_load_weight: null argument"); return -1; }
  if (e->finalized) { esam3_set_error("esam3_load_weight after esam3_finalize"); return -1; }
  HostTensor t;
  int64_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
  t.d.assign(data, data + n);
  e->raw[name] = std::move(t);
  return 0;
}

int esam3_finalize(esam3_engine* e) {
  if (!e) { esam3_set_error("null engine"); return -1; }
  DeviceGuard guard(e->cfg.device);
  if (!guard.ok) { esam3_set_error("esam3_finalize: hipSetDevice(%d) failed", e->cfg.device); return -1; }
  // dry run of both graphs at B = 1: packs every weight the graphs touch
  e->dry = true;
  e->arena.dry = true;
  esam3_image_features f{};
  void* dummy = reinterpret_cast<void*>(4096);
  for (int i = 0; i < 3; ++i) f.sam3_fpn_dev[i] = dummy;
  if (e->cfg.interactive) for (int i = 0; i < 3; ++i) f.sam2_fpn_dev[i] = dummy;
  int rc = e->encode(nullptr, 1, &f);
  if (rc == 0 && e->cfg.interactive) {
    esam3_prompts pr{};
    pr.n_images = 1; pr.n_prompts = 1; pr.n_points = 1;
    esam3_decode_out o{};
    rc = e->decode(&pr, &o);
  }
  e->dry = false;
  e->arena.dry = false;
  e->arena.top = e->arena.peak = 0;
  if (rc) return -1;
  if (e->cfg.interactive) CK(e->precompute_pe());
  e->finalized = true;
  return 0;
}

// fp32 host copies of image-encoder / mask-decoder weights that a packer has consumed and that a complete dry pass of
// encode + decode does not look at again (neither data nor shape) are freed; everything else stays.
int64_t esam3_release_host_weights(esam3_engine* e) {
  if (!e) { esam3_set_error("null engine"); return -1; }
  if (!e->finalized) { esam3_set_error("esam3_release_host_weights before esam3_finalize"); return -1; }
  DeviceGuard guard(e->cfg.device);
  if (!guard.ok) { esam3_set_error("esam3_release_host_weights: hipSetDevice(%d) failed", e->cfg.device); return -1; }
  ++e->epoch;
  e->dry = true;
  e->arena.dry = true;
  esam3_image_features f{};
  void* dummy = reinterpret_cast<void*>(4096);
  for (int i = 0; i < 3; ++i) f.sam3_fpn_dev[i] = dummy;
  if (e->cfg.interactive) for (int i = 0; i < 3; ++i) f.sam2_fpn_dev[i] = dummy;
  int rc = e->encode(nullptr, 1, &f);
  if (rc == 0 && e->cfg.interactive) {
    esam3_prompts pr{};
    pr.n_images = 1; pr.n_prompts = 1; pr.n_points = 1;
    esam3_decode_out o{};
    rc = e->decode(&pr, &o);
  }
  e->dry = false;
  e->arena.dry = false;
  if (rc) return -1;
  int64_t freed = 0;
  for (auto& kv : e->raw) {
    HostTensor& t = kv.second;
    const bool in_scope = kv.first.rfind("backbone.vision_backbone.", 0) == 0 || kv.first.rfind("inst_interactive_predictor.", 0) == 0;
    if (in_scope && t.packed && !t.released && t.touched != e->epoch) {
      freed += (int64_t)t.d.size() * 4;
      std::vector<float>().swap(t.d);
      t.released = true;
    }
  }
  return freed;
}

static int run_sized(esam3_engine* e, void* stream, const std::function<int()>& graph) {
  // Every engine-bound compute entry point runs on the engine's device whatever the caller's current device is
  // (arena growth, lazily packed text / grounding weights and all launches would otherwise land on the wrong GPU);
  // the caller's device is restored on return.  `stream` must belong to the engine's device.
  DeviceGuard guard(e->cfg.device);
  if (!guard.ok) { esam3_set_error("hipSetDevice(%d) failed", e->cfg.device); return -1; }
  // pass 1 (dry): measure arena peak; pass 2: launch
  e->st = (hipStream_t)stream;
  e->dry = true; e->arena.dry = true; e->arena.peak = 0;
  int rc = graph();
  e->dry = false; e->arena.dry = false;
  if (rc) return -1;
  CK(e->ensure_arena(e->arena.peak));
  return graph();
}

int esam3_encode_image(esam3_engine* e, const float* img, int B, const esam3_image_features* out, void* stream) {
  if (!e || !out || !img || B <= 0) { esam3_set_error("esam3_encode_image: bad argument"); return -1; }
  if (!e->finalized) { esam3_set_error("esam3_finalize has not been called"); return -1; }
  return run_sized(e, stream, [&]() { return e->encode(img, B, out); });
}

int esam3_decode(esam3_engine* e, const esam3_prompts* pr, const esam3_decode_out* out, void* stream) {
  if (!e || !pr || !out) { esam3_set_error("esam3_decode: null argument"); return -1; }
  if (!e->finalized || !e->cfg.interactive) { esam3_set_error("engine not finalized / not interactive"); return -1; }
  if (pr->n_prompts <= 0 || pr->n_points < 0 || !pr->prompt_image_dev || !out->low_res_dev || !out->iou_dev) {
    esam3_set_error("esam3_decode: bad prompt/out description");
    return -1;
  }
  if (pr->n_points > 0 && (!pr->coords_dev || !pr->labels_dev)) { esam3_set_error("esam3_decode: null coords/labels"); return -1; }
  return run_sized(e, stream, [&]() { return e->decode(pr, out); });
}

int esam3_encode_text(esam3_engine* e, const int64_t* tokens, int B, int S, float* memory, float* embeds, void* stream) {
  if (!e || !tokens || !memory || B <= 0 || S <= 0) { esam3_set_error("esam3_encode_text: bad argument"); return -1; }
  if (!e->finalized) { esam3_set_error("esam3_finalize has not been called"); return -1; }
  if (!e->find(TEXTP + "projector.weight")) { esam3_set_error("esam3_encode_text: no text-encoder weights were loaded"); return -1; }
  return run_sized(e, stream, [&]() { return e->encode_text(tokens, B, S, memory, embeds); });
}

int esam3_ground(esam3_engine* e, const esam3_ground_in* in, const esam3_ground_out* out, void* stream) {
  if (!e || !in || !out) { esam3_set_error("esam3_ground: null argument"); return -1; }
  if (!e->finalized) { esam3_set_error("esam3_finalize has not been called"); return -1; }
  if (in->n_images <= 0 || in->n_tokens <= 0 || !in->sam3_fpn_dev[0] || !in->sam3_fpn_dev[1] || !in->sam3_fpn_dev[2] ||
      !in->language_features_dev || !in->language_mask_dev || !out->pred_logits_dev || !out->pred_boxes_dev ||
      !out->presence_logit_dev || !out->pred_masks_dev) {
    esam3_set_error("esam3_ground: bad input/output description");
    return -1;
  }
  if (in->n_points < 0 || in->n_boxes < 0 || (in->n_points > 0 && (!in->points_dev || !in->point_labels_dev)) ||
      (in->n_boxes > 0 && (!in->boxes_dev || !in->box_labels_dev)) || in->n_points + in->n_boxes > 64) {
    esam3_set_error("esam3_ground: bad geometric prompt (n_points=%d n_boxes=%d, at most 64 in total)", in->n_points, in->n_boxes);
    return -1;
  }
  e->st = (hipStream_t)stream;
  CK(e->pcs_prepare());
  return run_sized(e, stream, [&]() { return e->ground(in, out); });
}

int esam3_postprocess_masks(esam3_engine* e, const float* low_res, int n, int oh, int ow, float max_hole_area,
                            float thr, uint8_t* masks_u8, float* masks_logits, void* stream) {
  if (!e || !low_res || n <= 0) { esam3_set_error("esam3_postprocess_masks: bad argument"); return -1; }
  if (oh <= 0 || ow <= 0) { esam3_set_error("esam3_postprocess_masks: output size %d x %d", oh, ow); return -1; }
  if (!masks_u8 && !masks_logits) { esam3_set_error("esam3_postprocess_masks: no output buffer given"); return -1; }
  const int LR = 4 * EMB;
  const size_t px = (size_t)n * LR * LR;
  return run_sized(e, stream, [&]() -> int {
    e->arena.top = 0;
    const float* src = low_res;
    if (max_hole_area > 0.f) {
      float* filled = (float*)e->allocb(px * 4);
      int* labels = (int*)e->allocb(px * 4);
      int* areas = (int*)e->allocb(px * 4);
      if (!e->ok(filled) || !e->ok(labels) || !e->ok(areas)) return -1;
      if (!e->dry) CK(esam3_launch_fill_holes(low_res, filled, labels, areas, n, LR, LR, thr, max_hole_area, e->st));
      src = filled;
    }
    if (!e->dry) CK(esam3_launch_upsample_masks(src, masks_logits, masks_u8, n, LR, LR, oh, ow, thr, e->st));
    return 0;
  });
}

int esam3_clamp_f32(esam3_engine* e, float* x, int64_t n, float lo, float hi, void* stream) {
  if (!e) return esam3_launch_clamp(x, n, lo, hi, (hipStream_t)stream);  // engine-less call: the caller's current device
  DeviceGuard guard(e->cfg.device);
  if (!guard.ok) { esam3_set_error("hipSetDevice(%d) failed", e->cfg.device); return -1; }
  return esam3_launch_clamp(x, n, lo, hi, (hipStream_t)stream);
}

int esam3_preprocess_resize_u8(const uint8_t* in, int H, int W, float* out, int out_h, int out_w, void* stream) {
  if (!in || !out || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) {
    esam3_set_error("esam3_preprocess_resize_u8: bad argument");
    return -1;
  }
  return esam3_launch_resize_aa_u8(in, 1, H, W, out, out_h, out_w, (hipStream_t)stream);
}

int esam3_preprocess_resize_u8_batch(const uint8_t* in, int B, int H, int W, float* out, int out_h, int out_w, void* stream) {
  if (!in || !out || B <= 0 || B > 65535 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) {
    esam3_set_error("esam3_preprocess_resize_u8_batch: bad argument");
    return -1;
  }
  return esam3_launch_resize_aa_u8(in, B, H, W, out, out_h, out_w, (hipStream_t)stream);
}

int esam3_preprocess_resize_rgbx_batch(const uint8_t* in, int B, int H, int W, float* out, int out_h, int out_w, void* stream) {
  if (!in || !out || B <= 0 || B > 65535 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) {
    esam3_set_error("esam3_preprocess_resize_rgbx_batch: bad argument");
    return -1;
  }
  return esam3_launch_resize_aa_u8(in, B, H, W, out, out_h, out_w, (hipStream_t)stream, 4);
}

int esam3_preprocess_u8(const uint8_t* in, float* out, int B, int H, int W, void* stream) {
  if (!in || !out || B <= 0) { esam3_set_error("esam3_preprocess_u8: bad argument"); return -1; }
  return esam3_launch_preprocess_u8(in, out, B, H, W, (hipStream_t)stream);
}

int esam3_profile_enable(esam3_engine* e, int on) {
  if (!e) return -1;
  for (auto& r : e->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  e->recs.clear();
  e->prof = on != 0;
  return 0;
}

int esam3_set_text_causal(esam3_engine* e, int causal_masking) {
  if (!e) { esam3_set_error("esam3_set_text_causal: null engine"); return -1; }
  e->text_causal = causal_masking != 0;
  return 0;
}

int esam3_profile_tag(esam3_engine* e, const char* tag) {
  if (!e) return -1;
  e->watch_tag = tag ? tag : "";
  for (auto& r : e->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  e->recs.clear();
  return 0;
}

// JSON array, one object per tag sorted by total time:
// {"tag":..., "launches":n, "ms":total, "algorithmic_flops":of the LAST launch of the tag, "algorithmic_bytes":likewise,
//  "algorithmic_flops_total":sum over the tag's launches, "algorithmic_bytes_total":likewise}
// Both figures are ALGORITHMIC (each operand and result counted once), not HBM traffic: a tensor that lives in the
// 256 MiB Infinity Cache between two launches makes bytes / time exceed the HBM peak.
int esam3_profile_report(esam3_engine* e, char* buf, int64_t buf_size) {
  if (!e || !buf || buf_size <= 2) { esam3_set_error("esam3_profile_report: bad argument"); return -1; }
  DeviceGuard guard(e->cfg.device);
  if (!guard.ok) { esam3_set_error("hipSetDevice(%d) failed", e->cfg.device); return -1; }
  HIP_CHECK_RET(hipDeviceSynchronize());
  struct Agg { double ms = 0, flops = 0, bytes = 0, flops_total = 0, bytes_total = 0; int n = 0; const char* kernel = nullptr; };
  std::unordered_map<std::string, Agg> agg;
  for (auto& r : e->recs) {
    float ms = 0.f;
    HIP_CHECK_RET(hipEventElapsedTime(&ms, r.a, r.b));
    Agg& a = agg[r.tag];
    a.ms += ms; a.n += 1; a.flops = r.flops; a.bytes = r.bytes; a.kernel = r.kernel;
    a.flops_total += r.flops; a.bytes_total += r.bytes;   // launches that share a tag may differ in shape: the sums are exact
    (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
  }
  e->recs.clear();
  std::vector<std::pair<std::string, Agg>> v(agg.begin(), agg.end());
  std::sort(v.begin(), v.end(), [](const auto& x, const auto& y) { return x.second.ms > y.second.ms; });
  std::string out = "[";
  for (size_t i = 0; i < v.size(); ++i) {
    char line[1280];
    const int len = snprintf(line, sizeof(line),
                             "%s{\"tag\":\"%s\",\"launches\":%d,\"ms\":%.6f,\"algorithmic_flops\":%.6e,\"algorithmic_bytes\":%.6e,"
                             "\"algorithmic_flops_total\":%.6e,\"algorithmic_bytes_total\":%.6e,\"kernel\":\"%s\"}",
                             i ? "," : "", v[i].first.c_str(), v[i].second.n, v[i].second.ms, v[i].second.flops, v[i].second.bytes,
                             v[i].second.flops_total, v[i].second.bytes_total,
                             v[i].second.kernel ? v[i].second.kernel : v[i].first.c_str());
    if (len < 0 || len >= (int)sizeof(line)) continue;  // never emit a truncated (invalid) entry
    if ((int64_t)(out.size() + strlen(line) + 2) >= buf_size) break;
    out += line;
  }
  out += "]";
  memcpy(buf, out.c_str(), out.size() + 1);
  return 0;
}

int64_t esam3_workspace_bytes(const esam3_engine* e) { return e ? (int64_t)e->arena.cap : 0; }
int esam3_elem_size(const esam3_engine* e) { return e ? (int)e->esz : 0; }

}  // extern "C"
