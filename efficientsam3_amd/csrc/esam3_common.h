// Shared device/host definitions for the EfficientSAM3 gfx950 engine.
// Activations are NHWC; element type T is `float` (validation mode) or bf16
// stored as uint16_t (throughput mode).  All accumulation is fp32.
#pragma once
#include <cstdlib>

// Development A/B and bisecting switches (ESAM3_NO_SKINNY, ESAM3_GEMM256_CLASSIC, ESAM3_BF16_STREAM, ...) are read from
// the environment only in a -DESAM3_DEV build (tools/dev_variants.sh); the release library has no getenv in it and every
// switch is its default.
#ifdef ESAM3_DEV
inline int esam3_dev_flag(const char* name, int dflt = 0) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}
#else
constexpr int esam3_dev_flag(const char*, int dflt = 0) { return dflt; }
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_v;
typedef __attribute__((ext_vector_type(16))) float f32x16_v;
typedef __attribute__((ext_vector_type(4))) float f32x4_v;

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_HSWISH = 3, ACT_SIGMOID = 4 };

__host__ __device__ inline float bf16_to_f32(bf16_t v) {
  union { uint32_t u; float f; } x;
  x.u = ((uint32_t)v) << 16;
  return x.f;
}
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
typedef float f32x2_v __attribute__((ext_vector_type(2)));
__host__ __device__ inline bf16_t f32_to_bf16(float f) {  // round-to-nearest-even
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint16_t, (__bf16)f);  // one v_cvt_pk_bf16_f32
#endif
  union { uint32_t u; float f; } x;
  x.f = f;
  uint32_t u = x.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);  // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}

#if defined(__HIPCC__)
// two floats -> packed bf16 pair (lo in bits 0..15): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_v f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_v));
}
#endif

template <typename T> struct ElemOps;
template <> struct ElemOps<float> {
  static __host__ __device__ inline float ld(float v) { return v; }
  static __host__ __device__ inline float st(float v) { return v; }
};
template <> struct ElemOps<bf16_t> {
  static __host__ __device__ inline float ld(bf16_t v) { return bf16_to_f32(v); }
  static __host__ __device__ inline bf16_t st(float v) { return f32_to_bf16(v); }
};
template <typename T> __host__ __device__ inline float to_f32(T v) { return ElemOps<T>::ld(v); }
template <typename T> __host__ __device__ inline T from_f32(float v) { return ElemOps<T>::st(v); }

// erf GELU, 0.5 x (1 + erf(x / sqrt 2)), in the form
//   GELU(x) = max(x, 0) + u P(u) exp(-x^2 / 2),  u = min(|x|, 5.9),  P(u) = -0.5 exp(u^2 / 2) erfc(u / sqrt 2):
// both signs share one branch-free expression (no 1 - (1 - small) cancellation on the negative side), P is smooth and
// bounded on [0, inf) (0.5 ... 0.067 at 5.9) and is fitted by a degree-9 polynomial with P(0) = -0.5 exact (weighted
// minimax for the GELU error; beyond 5.9 the exponential is < 3e-8 and the clamp keeps inf / huge inputs finite).
// |GELU error| <= 5e-7 for all x, relative error <= 6e-6 for |x| < 1: an order of magnitude below the bf16 output step
// and the f32-mode test tolerances.  ONE transcendental (v_exp_f32, quarter rate) and 15 full-rate operations; written
// on pairs so that the polynomial and the products become v_pk_fma_f32 / v_pk_mul_f32 (8.5 issue slots per element).
// History: libm's erff is a two-branch polynomial of about thirty operations; round 2 used Abramowitz & Stegun 7.1.26
// (rcp + exp2 + a dozen operations); TinyViT's GELU MBConvs and the ViT-H / TinyViT MLP epilogues are VALU-bound on it.
typedef float f32x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_v gelu_fast2(f32x2_v x) {
  const f32x2_v u = {fminf(fabsf(x.x), 5.9f), fminf(fabsf(x.y), 5.9f)};
  f32x2_v p = {5.145186606e-06f, 5.145186606e-06f};
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(-1.068554411e-04f));
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(9.848108748e-04f));
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(-5.413614679e-03f));
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(2.033651061e-02f));
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(-5.733343959e-02f));
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(1.303861737e-01f));
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(-2.493070066e-01f));
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(3.988702297e-01f));
  p = __builtin_elementwise_fma(p, u, (f32x2_v)(-5.000000000e-01f));
  const f32x2_v w = (x * (f32x2_v)(-0.72134752044448170f)) * x;  // -x^2 / 2 x log2(e)
  const f32x2_v e = {__builtin_amdgcn_exp2f(w.x), __builtin_amdgcn_exp2f(w.y)};
  const f32x2_v m = {fmaxf(x.x, 0.f), fmaxf(x.y, 0.f)};
  return __builtin_elementwise_fma(u * p, e, m);
}
__device__ __forceinline__ float gelu_fast(float x) {
  const f32x2_v r = gelu_fast2(f32x2_v{x, x});
  return r.x;
}
// N values in place, pairwise
template <int N>
__device__ __forceinline__ void gelu_fast_n(float (&v)[N]) {
#pragma unroll
  for (int i = 0; i + 1 < N; i += 2) {
    const f32x2_v r = gelu_fast2(f32x2_v{v[i], v[i + 1]});
    v[i] = r.x;
    v[i + 1] = r.y;
  }
  if constexpr (N & 1) v[N - 1] = gelu_fast(v[N - 1]);
}

__device__ inline float act_apply(float x, int act) {
  switch (act) {
    case ACT_RELU: return x > 0.f ? x : 0.f;
    case ACT_GELU: return gelu_fast(x);  // erf GELU
    case ACT_HSWISH: {  // x * relu6(x + 3) / 6
      float r = fminf(fmaxf(x + 3.f, 0.f), 6.f);
      return x * r * (1.f / 6.f);
    }
    case ACT_SIGMOID: return 1.f / (1.f + expf(-x));
    default: return x;
  }
}

// Same activation on N values with the (wave-uniform) selection hoisted out of the loop: one
// branch per call instead of one switch per element (a per-element switch in an unrolled
// epilogue multiplies the code size and thrashes the instruction cache).
template <int N>
__device__ __forceinline__ void act_apply_n(float (&v)[N], int act) {
  if (act == ACT_NONE) return;
  if (act == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
  } else if (act == ACT_GELU) {
    gelu_fast_n<N>(v);
  } else if (act == ACT_HSWISH) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = v[i] * fminf(fmaxf(v[i] + 3.f, 0.f), 6.f) * (1.f / 6.f);
  } else if (act == ACT_SIGMOID) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = 1.f / (1.f + expf(-v[i]));
  }
}

// ---- implicit-GEMM convolution / linear ------------------------------------------
// out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] ) (+ res)
//   m = flattened (b, oh, ow) output pixel (or token row), n = output channel,
//   k = (tap, cin) with cin fastest; Wt is packed [Np][Kp] K-contiguous, zero padded.
enum OutMode { OUT_PLAIN = 0, OUT_CONVT2X2 = 1 };

struct GemmParams {
  const void* A;      // activations, NHWC, row stride `lda` elements
  const void* Wt;     // packed weights [Np][Kp]
  const float* bias;  // [N] fp32 (BN folded) or nullptr
  const void* res;    // residual, indexed like `out` (or by m % res_mod) or nullptr
  void* out;
  int64_t M;          // rows (B*OH*OW)
  int N, K, Kp;       // logical out channels, logical K, padded K
  int H, W, Cin;      // input spatial dims and channels (3x3 gather); H*W = pixels/img
  int ksize;          // 1 or 3 (stride 1, pad ksize/2); 2 = the up-conv gather of gemm256p (ConvT k2s2 composed with the
                      // following 3x3: N = 4 classes x convt_cout, K = 4 taps x Cin, zero-bordered input, ConvT store)
  int lda;            // input row stride in elements (>= Cin)
  int ldc;            // output row stride in elements
  int ldr;            // residual row stride in elements
  int act;            // Act
  int res_mod;        // >0: residual row = m % res_mod (batch-broadcast residual)
  int out_mode;       // OutMode
  int convt_cout;     // OUT_CONVT2X2: Cout per tap (N == 4*Cout)
  int res_after_act;  // 1: out = act(acc+bias) + res ; 0: out = act(acc+bias+res)
  const int* res_bidx;  // optional: residual batch index per output batch item (gather)
  int in_pad;         // 1 (ksize 3 only): A is [B][H+2][W+2][lda] with a zero border -> no bounds checks
  int out_pad;        // 1: write the output inside a 1-pixel border ([B][OH+2][OW+2][ldc])
  int korder;         // ksize 3: 0 -> k = tap*Cin + c ; 1 -> k = (c/BKE)*9*BKE + tap*BKE + c%BKE
                      // (channel-chunk major: the 9 taps of one 128-byte channel chunk are
                      //  consecutive K tiles, so shifted re-reads of the same pixels hit in L2)
  int stride;         // ksize 3 only: 0/1 -> stride 1; 2 -> H, W are the INPUT dims, M = B*ceil(H/2)*ceil(W/2)
  const float* border_corr;  // ksize 2 only: [4 classes][3: row edge, column edge, both][convt_cout] fp32, added to the
                      // accumulators of the output image's ring pixels (the bias share of the 3x3 taps that fall outside)
  int out_f32;        // 1 (bf16 GEMMs on gemm256p, plain rows, no activation): `out` and `res` are fp32 (ldc / ldr in fp32
                      // elements) -- the residual stream of a LayerNorm / attention stack kept in fp32 as the reference's
                      // autocast does (fp32 x + bf16 branch -> fp32)
};

// ---- 8 consecutive activations <-> 8 floats (one 16-byte access for bf16, two for f32) -------
template <typename T> struct Vec8;
template <> struct Vec8<bf16_t> {
  static __device__ inline void load(const bf16_t* p, float* v) {
    const uint4 u = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ inline void store(bf16_t* p, const float* v) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]);
    o.y = pack_bf16x2(v[2], v[3]);
    o.z = pack_bf16x2(v[4], v[5]);
    o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
};
template <> struct Vec8<float> {
  static __device__ inline void load(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ inline void store(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};

// ---- error handling ---------------------------------------------------------------
void esam3_set_error(const char* fmt, ...);
// Profiler scopes with the reference's record_function names (sam3_image.py:449-479, sam3_tracker_base.py:314) around the launches of each
// phase: forwarded to the hooks of esam3_set_scope_hooks (the Python layer opens torch.profiler.record_function ranges) and to roctx
// (rocprofv3 --marker-trace) when libroctx64.so is loadable.  ScopeSeq: next(name) closes the previous phase and opens the next one, the
// destructor closes the last (also on an error return).
void esam3_scope_push(const char* name);
void esam3_scope_pop();
struct ScopeSeq {
  bool open = false;
  void next(const char* name) {
    if (open) esam3_scope_pop();
    esam3_scope_push(name);
    open = true;
  }
  ~ScopeSeq() {
    if (open) esam3_scope_pop();
  }
};
#define HIP_CHECK_RET(expr)                                                        \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess) {                                                        \
      esam3_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),       \
                      __FILE__, __LINE__);                                         \
      return -1;                                                                   \
    }                                                                              \
  } while (0)
