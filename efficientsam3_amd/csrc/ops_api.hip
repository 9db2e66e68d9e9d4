// Single-operator C entry points (include/esam3.h, "esam3_op_*").  They pack host fp32
// weights exactly like the engine does and launch the same kernels, so unit parity tests
// can compare one operator at a time against the oracle.  Test-sized: every call uploads
// its weights, synchronises and frees them.
#if !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/esam3.h"
#include "kernels.h"

namespace {

struct Tmp {  // scoped device allocations
  std::vector<void*> ptrs;
  ~Tmp() {
    for (void* p : ptrs) (void)hipFree(p);
  }
  void* up(const void* src, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    if (bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return p;
  }
  void* upT(int dtype, const std::vector<float>& v) {
    if (dtype == 0) return up(v.data(), v.size() * 4);
    std::vector<bf16_t> h(v.size());
    for (size_t i = 0; i < v.size(); ++i) h[i] = f32_to_bf16(v[i]);
    return up(h.data(), h.size() * 2);
  }
  void* raw(size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 256) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return p;
  }
};

// -DESAM3_DEV builds: ESAM3_OP_REPEAT=N re-launches the operator N times between two HIP events and prints the average
// (tools/evit_fused_bench.py); the release library launches once
template <typename F>
int op_timed(const char* what, hipStream_t s, F&& launch) {
  const int reps = esam3_dev_flag("ESAM3_OP_REPEAT");
  if (reps <= 0) return launch();
  hipEvent_t a, b;
  HIP_CHECK_RET(hipEventCreate(&a));
  HIP_CHECK_RET(hipEventCreate(&b));
  if (launch()) return -1;   // warm-up
  HIP_CHECK_RET(hipEventRecord(a, s));
  for (int i = 0; i < reps; ++i)
    if (launch()) return -1;
  HIP_CHECK_RET(hipEventRecord(b, s));
  HIP_CHECK_RET(hipEventSynchronize(b));
  float ms = 0.f;
  HIP_CHECK_RET(hipEventElapsedTime(&ms, a, b));
  fprintf(stderr, "[op_timed] %s: %.4f ms per launch (%d launches)\n", what, ms / reps, reps);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return 0;
}

int fail(const char* what) {
  esam3_set_error("%s: device allocation/upload failed", what);
  return -1;
}

}  // namespace

extern "C" {

int esam3_op_linear(int dtype, const void* a, const float* w, const float* bias, const void* res, void* out,
                    int64_t M, int N, int K, int act, void* stream) {
  Tmp t;
  const int esz = dtype == 0 ? 4 : 2;
  const int Kp = esam3_gemm_pad_k(K, esz), Np = esam3_gemm_pad_n(N);
  std::vector<float> pk((size_t)Np * Kp, 0.f);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) pk[(size_t)n * Kp + k] = w[(size_t)n * K + k];
  GemmParams p{};
  p.A = a; p.Wt = t.upT(dtype, pk); p.bias = bias ? (float*)t.up(bias, (size_t)N * 4) : nullptr;
  if (!p.Wt || (bias && !p.bias)) return fail("op_linear");
  p.res = res; p.out = out; p.M = M; p.N = N; p.K = K; p.Kp = Kp; p.H = 1; p.W = 1; p.Cin = K; p.ksize = 1;
  p.lda = K; p.ldc = N; p.ldr = N; p.act = act; p.res_after_act = 1;
  if (esam3_launch_gemm(dtype, p, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_fused_mlp(const void* x, const float* w1, const float* b1, const float* w2, const float* b2, const void* res, void* out,
                       int64_t M, int Cin, int Hid, int Cout, int act, void* stream) {
  Tmp t;
  if (!esam3_fused_mlp_ok(1, Cin, Hid, Cout)) { esam3_set_error("op_fused_mlp: shape %d -> %d -> %d is not instantiated", Cin, Hid, Cout); return -1; }
  std::vector<float> p1(w1, w1 + (size_t)Hid * Cin), p2((size_t)Cout * Hid);
  for (int n = 0; n < Cout; ++n)
    for (int k = 0; k < Hid; ++k) p2[(size_t)n * Hid + k] = w2[(size_t)n * Hid + (k & ~31) + esam3_fused_mlp_kperm(k & 31)];
  void* d1 = t.upT(1, p1);
  void* d2 = t.upT(1, p2);
  const float* db1 = static_cast<const float*>(t.up(b1, (size_t)Hid * 4));
  const float* db2 = static_cast<const float*>(t.up(b2, (size_t)Cout * 4));
  if (!d1 || !d2 || !db1 || !db2) return fail("op_fused_mlp");
  if (esam3_launch_fused_mlp(x, Cin, d1, db1, d2, db2, res, Cout, out, Cout, M, Cin, Hid, Cout, act, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_resize_axis_tables(int in_size, int out_size, int* first, int* count, float* frac) {
  return esam3_resize_axis_tables_host(in_size, out_size, first, count, frac);
}

int esam3_op_resize_shuffle(int dtype, const void* in, const float* bias, void* out, int B, int IH, int IW, int OH, int OW, int C, int taps,
                            int act, int out_pad, void* stream) {
  Tmp t;
  const float* db = bias ? static_cast<const float*>(t.up(bias, (size_t)C * 4)) : nullptr;
  if (bias && !db) return fail("op_resize_shuffle");
  if (op_timed("resize_shuffle", (hipStream_t)stream, [&]() {
        return esam3_launch_resize_shuffle(dtype, in, db, out, B, IH, IW, OH, OW, C, taps, act, out_pad, (hipStream_t)stream);
      }))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_conv2d(int dtype, const void* x, const float* w, const float* bias, const void* res, void* out,
                    int B, int H, int W, int Cin, int Cout, int ks, int act, void* stream) {
  Tmp t;
  const int esz = dtype == 0 ? 4 : 2;
  const int K = Cin * ks * ks;
  const int Kp = esam3_gemm_pad_k(K, esz), Np = esam3_gemm_pad_n(Cout);
  std::vector<float> pk((size_t)Np * Kp, 0.f);
  for (int n = 0; n < Cout; ++n)
    for (int c = 0; c < Cin; ++c)
      for (int tp = 0; tp < ks * ks; ++tp)
        pk[(size_t)n * Kp + esam3_conv_k_index(Cin, ks, esz, tp, c)] = w[((size_t)n * Cin + c) * ks * ks + tp];
  GemmParams p{};
  p.A = x; p.Wt = t.upT(dtype, pk); p.bias = bias ? (float*)t.up(bias, (size_t)Cout * 4) : nullptr;
  if (!p.Wt || (bias && !p.bias)) return fail("op_conv2d");
  p.res = res; p.out = out; p.M = (int64_t)B * H * W; p.N = Cout; p.K = K; p.Kp = Kp; p.H = H; p.W = W;
  p.Cin = Cin; p.ksize = ks; p.lda = Cin; p.ldc = Cout; p.ldr = Cout; p.act = act; p.res_after_act = 1;
  p.korder = esam3_conv_korder(Cin, ks, esz);
  if (esam3_launch_gemm(dtype, p, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_upconv(const void* xpad, const float* wt, const float* bt, const float* w3, const float* b3, void* out, int B, int H, int W,
                    int Cin, int Cmid, int Cout, int narrow, void* stream) {
  Tmp t;
  if (!xpad || !wt || !bt || !w3 || !b3 || !out || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cmid <= 0 || Cout <= 0) {
    esam3_set_error("op_upconv: bad argument");
    return -1;
  }
  std::vector<float> w, bias, corr;
  esam3_compose_upconv_host(wt, bt, w3, b3, Cin, Cmid, Cout, w, bias, corr);  // [class][o][tap][ci], [Cout], [class][3][Cout]
  GemmParams p{};
  p.A = xpad; p.out = out; p.M = (int64_t)B * H * W; p.N = 4 * Cout; p.K = 4 * Cin; p.Kp = p.K; p.H = H; p.W = W; p.Cin = Cin;
  p.ksize = 2; p.lda = Cin; p.ldc = Cout; p.ldr = Cout; p.act = ACT_NONE; p.out_mode = OUT_CONVT2X2; p.convt_cout = Cout; p.in_pad = 1;
  p.res_after_act = 1;
  p.bias = static_cast<const float*>(t.up(bias.data(), bias.size() * 4));
  p.border_corr = static_cast<const float*>(t.up(corr.data(), corr.size() * 4));
  if (!p.bias || !p.border_corr) return fail("op_upconv");
  if (narrow) {  // engine.hip: pk_upconv_narrow
    if (!esam3_upconv_narrow_ok(1, Cout, Cin, H, W)) { esam3_set_error("op_upconv: shape %d -> 4 x %d @%dx%d is not a narrow up-conv", Cin, Cout, H, W); return -1; }
    std::vector<float> pn((size_t)4 * Cout * 4 * Cin);
    for (int cls = 0; cls < 4; ++cls)
      for (int o = 0; o < Cout; ++o)
        for (int tap = 0; tap < 4; ++tap)
          for (int ci = 0; ci < Cin; ++ci)
            pn[(size_t)esam3_upconv_narrow_windex(o, cls, tap, ci)] = w[(((size_t)cls * Cout + o) * 4 + tap) * Cin + ci];
    p.Wt = t.upT(1, pn);
    if (!p.Wt) return fail("op_upconv");
    if (op_timed("upconv_narrow", (hipStream_t)stream, [&]() { return esam3_launch_upconv_narrow(p, (hipStream_t)stream); })) return -1;
  } else {  // engine.hip: pk_upconv
    if (Cin % 64 != 0 || Cout % 256 != 0) { esam3_set_error("op_upconv: shape %d -> 4 x %d is not an up-conv gather of gemm256p", Cin, Cout); return -1; }
    const int Np = esam3_gemm_pad_n(p.N);
    std::vector<float> pk((size_t)Np * p.Kp, 0.f);
    for (int n = 0; n < 4 * Cout; ++n)
      for (int tap = 0; tap < 4; ++tap)
        for (int ci = 0; ci < Cin; ++ci) pk[(size_t)n * p.Kp + esam3_upconv_kindex(tap, ci)] = w[((size_t)n * 4 + tap) * Cin + ci];
    p.Wt = t.upT(1, pk);
    if (!p.Wt) return fail("op_upconv");
    if (op_timed("upconv_gather", (hipStream_t)stream, [&]() { return esam3_launch_gemm(1, p, (hipStream_t)stream); })) return -1;
    const char* kn = esam3_take_last_gemm_kernel();
    if (!kn || !strstr(kn, "gemm256p")) {  // the point of the entry is the dominant kernel: refuse a silent fall-back to another GEMM
      esam3_set_error("op_upconv: the launch did not run on gemm256p (%s)", kn ? kn : "no kernel noted");
      return -1;
    }
  }
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_conv3x3_s2(int dtype, const void* x, const float* w, const float* bias, void* out, int B, int H, int W,
                        int Cin, int Cout, int act, void* stream) {
  Tmp t;
  const int esz = dtype == 0 ? 4 : 2;
  const int K = Cin * 9;
  const int Kp = esam3_gemm_pad_k(K, esz), Np = esam3_gemm_pad_n(Cout);
  std::vector<float> pk((size_t)Np * Kp, 0.f);
  for (int n = 0; n < Cout; ++n)
    for (int c = 0; c < Cin; ++c)
      for (int tp = 0; tp < 9; ++tp) pk[(size_t)n * Kp + esam3_conv_k_index(Cin, 3, esz, tp, c)] = w[((size_t)n * Cin + c) * 9 + tp];
  GemmParams p{};
  p.A = x; p.Wt = t.upT(dtype, pk); p.bias = bias ? (float*)t.up(bias, (size_t)Cout * 4) : nullptr;
  if (!p.Wt || (bias && !p.bias)) return fail("op_conv3x3_s2");
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  p.out = out; p.M = (int64_t)B * OH * OW; p.N = Cout; p.K = K; p.Kp = Kp; p.H = H; p.W = W;
  p.Cin = Cin; p.ksize = 3; p.lda = Cin; p.ldc = Cout; p.ldr = Cout; p.act = act; p.res_after_act = 1;
  p.korder = esam3_conv_korder(Cin, 3, esz);
  p.stride = 2;
  if (esam3_launch_gemm(dtype, p, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_window_attention(int dtype, const void* qkv, const float* pad_qkv, const float* bias, void* out, int B,
                              int H, int W, int heads, int ws, void* stream) {
  Tmp t;
  const int C = heads * 32;
  std::vector<float> pv(pad_qkv, pad_qkv + 3 * (size_t)C);
  void* pd = t.upT(dtype, pv);
  float* bd = (float*)t.up(bias, (size_t)heads * ws * ws * 4);
  if (!pd || !bd) return fail("op_window_attention");
  if (esam3_launch_window_attn(dtype, qkv, 3 * C, pd, bd, out, C, B, H, W, heads, ws, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_attn_window(int dtype, void* qkv, const float* cos_sin, void* out, int B, int H, int W, int ws, int heads,
                         void* stream) {
  Tmp t;
  const int D = heads * 64;
  float* cs = nullptr;
  if (cos_sin) {
    cs = (float*)t.up(cos_sin, (size_t)ws * ws * 32 * 2 * 4);
    if (!cs) return fail("op_attn_window");
  }
  if (esam3_launch_attn_window(dtype, qkv, 3 * D, 0, D, 2 * D, out, D, B, H, W, ws, heads, 64, cs, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_mha(int dtype, int path, const void* q, const void* k, const void* v, void* out, int B, int Nq, int Nk, int heads,
                 const uint8_t* key_mask, const float* bias_y, const float* bias_x, int Hk, int Wk, int bias_q0, void* stream) {
  const int D = heads * 32;
  hipStream_t s = (hipStream_t)stream;
  // k and v are separate [B*Nk][D] buffers: address v relative to k (element offset) like the engine's packed kv rows
  const size_t es = dtype == 0 ? 4 : 2;
  const int64_t voff = ((const char*)v - (const char*)k) / (int64_t)es;
  if (voff < INT32_MIN || voff > INT32_MAX) { esam3_set_error("op_mha: k and v too far apart"); return -1; }
  int rc;
  if (path == 1) rc = bias_y || key_mask ? 1 : esam3_launch_attn_mfma32(q, D, 0, k, D, 0, (int)voff, out, D, B, Nq, Nk, heads, s);
  else if (path == 2) rc = esam3_launch_attn_mfma32_splitk(q, D, 0, k, D, 0, (int)voff, out, D, B, Nq, Nk, heads, key_mask, bias_y,
                                                          bias_x, Hk, Wk, bias_q0, s);
  else rc = esam3_launch_mha_core(dtype, q, D, 0, k, D, 0, (int)voff, out, D, B, Nq, Nk, heads, key_mask, bias_y, bias_x, Hk, Wk,
                                  bias_q0, s);
  if (rc > 0) { esam3_set_error("op_mha: path %d does not take this shape", path); return -1; }
  if (rc) return -1;
  HIP_CHECK_RET(hipStreamSynchronize(s));
  return 0;
}

int esam3_op_vit_rope(int dtype, void* qkv, const float* cos_sin, int64_t rows, int H, int W, int ws, int heads,
                      void* stream) {
  Tmp t;
  float* cs = (float*)t.up(cos_sin, (size_t)ws * ws * 32 * 2 * 4);
  if (!cs) return fail("op_vit_rope");
  if (esam3_launch_vit_rope(dtype, qkv, cs, rows, H, W, ws, heads, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_squeeze_excite(int dtype, void* x, const float* w1, const float* b1, const float* w2, const float* b2,
                            int B, int HW, int C, int R, void* stream) {
  Tmp t;
  float* d1 = (float*)t.up(w1, (size_t)R * C * 4);
  float* e1 = (float*)t.up(b1, (size_t)R * 4);
  float* d2 = (float*)t.up(w2, (size_t)R * C * 4);
  float* e2 = (float*)t.up(b2, (size_t)C * 4);
  float* sums = (float*)t.raw((size_t)esam3_squeeze_excite_scratch_floats(B, HW, C) * 4);
  float* gate = (float*)t.raw((size_t)B * C * 4);
  if (!d1 || !e1 || !d2 || !e2 || !sums || !gate) return fail("op_squeeze_excite");
  if (esam3_launch_squeeze_excite(dtype, x, C, sums, gate, d1, e1, d2, e2, B, HW, C, R, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_conv3x3_padded(int dtype, const void* x_padded, const float* w, const float* bias, void* out, int B,
                            int H, int W, int Cin, int Cout, int act, int out_pad, void* stream) {
  Tmp t;
  const int esz = dtype == 0 ? 4 : 2;
  const int K = Cin * 9;
  const int Kp = esam3_gemm_pad_k(K, esz), Np = esam3_gemm_pad_n(Cout);
  std::vector<float> pk((size_t)Np * Kp, 0.f);
  for (int n = 0; n < Cout; ++n)
    for (int c = 0; c < Cin; ++c)
      for (int tp = 0; tp < 9; ++tp) pk[(size_t)n * Kp + esam3_conv_k_index(Cin, 3, esz, tp, c)] = w[((size_t)n * Cin + c) * 9 + tp];
  GemmParams p{};
  p.A = x_padded; p.Wt = t.upT(dtype, pk); p.bias = bias ? (float*)t.up(bias, (size_t)Cout * 4) : nullptr;
  if (!p.Wt || (bias && !p.bias)) return fail("op_conv3x3_padded");
  p.out = out; p.M = (int64_t)B * H * W; p.N = Cout; p.K = K; p.Kp = Kp; p.H = H; p.W = W;
  p.Cin = Cin; p.ksize = 3; p.lda = Cin; p.ldc = Cout; p.ldr = Cout; p.act = act; p.res_after_act = 1;
  p.korder = esam3_conv_korder(Cin, 3, esz);
  p.in_pad = 1; p.out_pad = out_pad;
  if (out_pad && esam3_launch_zero_border(dtype, out, B, H + 2, W + 2, Cout, (hipStream_t)stream)) return -1;
  if (esam3_conv3x3_narrow_ok(dtype, Cout, Cin, H, W, 1, out_pad, 1, false)) {  // the engine's choice for these shapes
    std::vector<float> pn((size_t)Cout * K);
    for (int n = 0; n < Cout; ++n)
      for (int c = 0; c < Cin; ++c)
        for (int tp = 0; tp < 9; ++tp) pn[(size_t)esam3_conv3x3_narrow_windex(Cout, n, tp, c)] = w[((size_t)n * Cin + c) * 9 + tp];
    p.Wt = t.upT(dtype, pn);
    if (!p.Wt) return fail("op_conv3x3_padded");
    if (esam3_launch_conv3x3_narrow(p, (hipStream_t)stream)) return -1;
    HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
    return 0;
  }
  if (esam3_launch_gemm(dtype, p, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_conv_transpose2x2(int dtype, const void* x, const float* w, const float* bias, const void* res,
                               void* out, int B, int H, int W, int Cin, int Cout, int act, int res_after_act,
                               void* stream) {
  Tmp t;
  const int esz = dtype == 0 ? 4 : 2;
  const int N = 4 * Cout;
  const int Kp = esam3_gemm_pad_k(Cin, esz), Np = esam3_gemm_pad_n(N);
  std::vector<float> pk((size_t)Np * Kp, 0.f);
  for (int ci = 0; ci < Cin; ++ci)
    for (int co = 0; co < Cout; ++co)
      for (int tp = 0; tp < 4; ++tp) pk[((size_t)tp * Cout + co) * Kp + ci] = w[((size_t)ci * Cout + co) * 4 + tp];
  GemmParams p{};
  p.A = x; p.Wt = t.upT(dtype, pk); p.bias = bias ? (float*)t.up(bias, (size_t)Cout * 4) : nullptr;
  if (!p.Wt || (bias && !p.bias)) return fail("op_conv_transpose2x2");
  p.res = res; p.out = out; p.M = (int64_t)B * H * W; p.N = N; p.K = Cin; p.Kp = Kp; p.H = H; p.W = W;
  p.Cin = Cin; p.ksize = 1; p.lda = Cin; p.ldc = Cout; p.ldr = Cout; p.act = act;
  p.out_mode = OUT_CONVT2X2; p.convt_cout = Cout; p.res_after_act = res_after_act;
  if (esam3_launch_gemm(dtype, p, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

// fused MBConv: w1 [Cmid][Cin], wd [Cmid][1][3][3], w2 [Cout][Cmid] (PyTorch layouts, BN already folded)
int esam3_op_mbconv_fused(int dtype, const void* x, const float* w1, const float* b1, const float* wd,
                          const float* bd, const float* w2, const float* b2, void* out, int B, int H, int W,
                          int Cin, int Cmid, int Cout, int stride, int residual, void* stream) {
  Tmp t;
  const int esz = dtype == 0 ? 4 : 2;
  const int Kp1 = esam3_gemm_pad_k(Cin, esz), Np1 = esam3_gemm_pad_n(Cmid);
  const int Kp2 = esam3_gemm_pad_k(Cmid, esz), Np2 = esam3_gemm_pad_n(Cout);
  std::vector<float> p1((size_t)Np1 * Kp1, 0.f), p2((size_t)Np2 * Kp2, 0.f), pd((size_t)9 * Cmid);
  for (int n = 0; n < Cmid; ++n)
    for (int k = 0; k < Cin; ++k) p1[(size_t)n * Kp1 + k] = w1[(size_t)n * Cin + k];
  for (int n = 0; n < Cout; ++n)
    for (int k = 0; k < Cmid; ++k) p2[(size_t)n * Kp2 + k] = w2[(size_t)n * Cmid + k];
  for (int c = 0; c < Cmid; ++c)
    for (int tp = 0; tp < 9; ++tp) pd[(size_t)tp * Cmid + c] = wd[(size_t)c * 9 + tp];
  void* d1 = t.upT(dtype, p1);
  void* d2 = t.upT(dtype, p2);
  float* dd = (float*)t.up(pd.data(), pd.size() * 4);
  float* db1 = (float*)t.up(b1, (size_t)Cmid * 4);
  float* dbd = bd ? (float*)t.up(bd, (size_t)Cmid * 4) : nullptr;
  float* db2 = (float*)t.up(b2, (size_t)Cout * 4);
  if (!d1 || !d2 || !dd || !db1 || !db2) return fail("op_mbconv_fused");
  if (op_timed("mbconv_fused", (hipStream_t)stream, [&]() {
        return esam3_launch_mbconv_fused(dtype, x, out, d1, Kp1, db1, dd, dbd, d2, Kp2, db2, B, H, W, Cin, Cmid, Cout, stride, residual,
                                         (hipStream_t)stream);
      }))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_mbconv3(const void* x, const float* w1, const float* b1, const float* wd, const float* bd, const float* w2,
                     const float* b2, void* out, int B, int H, int W, int Cin, int Cmid, int Cout, int stride, int residual,
                     void* stream) {
  Tmp t;
  const int Kp1 = esam3_gemm_pad_k(Cin, 2), Np1 = esam3_gemm_pad_n(Cmid);
  const int Kp2 = esam3_gemm_pad_k(Cmid, 2), Np2 = esam3_gemm_pad_n(Cout);
  std::vector<float> p1((size_t)Np1 * Kp1, 0.f), p2((size_t)Np2 * Kp2, 0.f), pd((size_t)9 * Cmid);
  for (int n = 0; n < Cmid; ++n)
    for (int k = 0; k < Cin; ++k) p1[(size_t)n * Kp1 + k] = w1[(size_t)n * Cin + k];
  for (int n = 0; n < Cout; ++n)
    for (int k = 0; k < Cmid; ++k) p2[(size_t)n * Kp2 + k] = w2[(size_t)n * Cmid + k];
  for (int c = 0; c < Cmid; ++c)
    for (int tp = 0; tp < 9; ++tp) pd[(size_t)tp * Cmid + c] = wd[(size_t)c * 9 + tp];
  void* d1 = t.upT(1, p1);
  void* d2 = t.upT(1, p2);
  float* dd = (float*)t.up(pd.data(), pd.size() * 4);
  float* db1 = (float*)t.up(b1, (size_t)Cmid * 4);
  float* dbd = bd ? (float*)t.up(bd, (size_t)Cmid * 4) : nullptr;
  float* db2 = (float*)t.up(b2, (size_t)Cout * 4);
  if (!d1 || !d2 || !dd || !db1 || !db2) return fail("op_mbconv3");
  if (op_timed("mbconv3", (hipStream_t)stream, [&]() {
        return esam3_launch_mbconv3(x, out, d1, Kp1, db1, dd, dbd, d2, Kp2, db2, B, H, W, Cin, Cmid, Cout, stride, residual,
                                    (hipStream_t)stream);
      }))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_lite_mla_block(const void* x, const float* wqkv, const float* wdw, const float* wgrp, const float* wproj,
                            const float* bproj, void* out, int B, int H, int W, int C, void* stream) {
  Tmp t;
  if (!esam3_mla_fused_ok(1, C, 16)) { esam3_set_error("op_lite_mla_block: C = %d is not instantiated", C); return -1; }
  const int C3 = 3 * C;
  const int Kpq = esam3_gemm_pad_k(C, 2), Npq = esam3_gemm_pad_n(C3);
  const int Kpg = esam3_gemm_pad_k(16, 2);
  const int Kpp = esam3_gemm_pad_k(2 * C, 2), Npp = esam3_gemm_pad_n(C);
  std::vector<float> pq((size_t)Npq * Kpq, 0.f), pg((size_t)C3 * Kpg, 0.f), pp((size_t)Npp * Kpp, 0.f), pd((size_t)25 * C3);
  for (int n = 0; n < C3; ++n)
    for (int k = 0; k < C; ++k) pq[(size_t)n * Kpq + k] = wqkv[(size_t)n * C + k];
  for (int n = 0; n < C3; ++n)
    for (int k = 0; k < 16; ++k) pg[(size_t)n * Kpg + k] = wgrp[(size_t)n * 16 + k];
  for (int n = 0; n < C; ++n)
    for (int k = 0; k < 2 * C; ++k) pp[(size_t)n * Kpp + k] = wproj[(size_t)n * 2 * C + k];
  for (int c = 0; c < C3; ++c)
    for (int tp = 0; tp < 25; ++tp) pd[(size_t)tp * C3 + c] = wdw[(size_t)c * 25 + tp];
  void* dq = t.upT(1, pq);
  void* dg = t.upT(1, pg);
  void* dp = t.upT(1, pp);
  float* dd = (float*)t.up(pd.data(), pd.size() * 4);
  float* db = (float*)t.up(bproj, (size_t)C * 4);
  size_t qb, kb, tb;
  esam3_mla_fused_scratch(B, H, W, C, &qb, &kb, &tb);
  void* qms = t.raw(qb);
  float* kvp = (float*)t.raw(kb);
  void* tab = t.raw(tb);
  if (!dq || !dg || !dp || !dd || !db || !qms || !kvp || !tab) return fail("op_lite_mla_block");
  if (op_timed("lite_mla_block", (hipStream_t)stream, [&]() {
        return esam3_launch_mla_fused(x, out, dq, Kpq, dd, dg, Kpg, dp, Kpp, db, qms, kvp, tab, B, H, W, C, (hipStream_t)stream);
      }))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_rowlin256(const void* x, const float* w, const float* bias, const float* table, int P, void* out, int64_t rows, void* stream) {
  Tmp t;
  if (!esam3_rowlin256_ok(1, rows, 256, 256, table ? P : 0)) { esam3_set_error("op_rowlin256: rows = %lld, P = %d unsupported", (long long)rows, P); return -1; }
  const int Kp = esam3_gemm_pad_k(256, 2);
  std::vector<float> pk((size_t)256 * Kp, 0.f);
  for (int n = 0; n < 256; ++n)
    for (int k = 0; k < 256; ++k) pk[(size_t)n * Kp + k] = w[(size_t)n * 256 + k];
  void* dw = t.upT(1, pk);
  float* db = bias ? (float*)t.up(bias, 256 * 4) : nullptr;
  void* dt = table ? t.upT(1, std::vector<float>(table, table + (size_t)P * 256)) : nullptr;
  if (!dw || (bias && !db) || (table && !dt)) return fail("op_rowlin256");
  if (op_timed("rowlin256", (hipStream_t)stream, [&]() { return esam3_launch_rowlin256(x, dw, Kp, db, dt, P, out, rows, (hipStream_t)stream); }))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_i2t_block(const void* x, const float* wq, const float* bq, const float* peq, const float* wo, const float* bo,
                       const float* gamma, const float* beta, const float* tk, const float* tv, void* out, int Bp, int P, int T,
                       void* stream) {
  Tmp t;
  if (!esam3_i2t_fused_ok(1, P, T, 8, 16, 256)) { esam3_set_error("op_i2t_block: P = %d, T = %d unsupported", P, T); return -1; }
  const int Kpq = esam3_gemm_pad_k(256, 2), Kpo = esam3_gemm_pad_k(128, 2);
  std::vector<float> pq((size_t)128 * Kpq, 0.f), po((size_t)256 * Kpo, 0.f);
  for (int n = 0; n < 128; ++n)
    for (int k = 0; k < 256; ++k) pq[(size_t)n * Kpq + k] = wq[(size_t)n * 256 + k];
  for (int n = 0; n < 256; ++n)
    for (int k = 0; k < 128; ++k) po[(size_t)n * Kpo + k] = wo[(size_t)n * 128 + k];
  void* dq = t.upT(1, pq);
  void* d_o = t.upT(1, po);
  void* dpe = t.upT(1, std::vector<float>(peq, peq + (size_t)P * 128));
  void* dk = t.upT(1, std::vector<float>(tk, tk + (size_t)Bp * T * 128));
  void* dv = t.upT(1, std::vector<float>(tv, tv + (size_t)Bp * T * 128));
  float* dbq = (float*)t.up(bq, 128 * 4);
  float* dbo = (float*)t.up(bo, 256 * 4);
  float* dg = (float*)t.up(gamma, 256 * 4);
  float* db = (float*)t.up(beta, 256 * 4);
  if (!dq || !d_o || !dpe || !dk || !dv || !dbq || !dbo || !dg || !db) return fail("op_i2t_block");
  if (op_timed("i2t_block", (hipStream_t)stream, [&]() {
        return esam3_launch_i2t_fused(x, out, dq, Kpq, dbq, dpe, d_o, Kpo, dbo, dg, db, 1e-5f, dk, 128, dv, 128, Bp, P, T, (hipStream_t)stream);
      }))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_dwconv(int dtype, const void* x, const float* w, const float* bias, void* out, int B, int H, int W,
                    int C, int ks, int stride, int act, void* stream) {
  Tmp t;
  std::vector<float> pk((size_t)ks * ks * C);
  for (int c = 0; c < C; ++c)
    for (int tp = 0; tp < ks * ks; ++tp) pk[(size_t)tp * C + c] = w[(size_t)c * ks * ks + tp];
  float* dw = (float*)t.up(pk.data(), pk.size() * 4);
  float* db = bias ? (float*)t.up(bias, (size_t)C * 4) : nullptr;
  if (!dw || (bias && !db)) return fail("op_dwconv");
  if (esam3_launch_dwconv(dtype, x, C, dw, db, out, C, B, H, W, C, ks, stride, act, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_stem(int dtype, const float* img, const float* w, const float* bias, void* out, int B, int H, int W,
                  int Cout, int act, void* stream) {
  Tmp t;
  std::vector<float> pk(27 * (size_t)Cout);
  for (int co = 0; co < Cout; ++co)
    for (int c = 0; c < 3; ++c)
      for (int tp = 0; tp < 9; ++tp) pk[(size_t)(tp * 3 + c) * Cout + co] = w[((size_t)co * 3 + c) * 9 + tp];
  float* dw = (float*)t.up(pk.data(), pk.size() * 4);
  float* db = bias ? (float*)t.up(bias, (size_t)Cout * 4) : nullptr;
  if (!dw || (bias && !db)) return fail("op_stem");
  if (esam3_launch_stem(dtype, img, dw, db, out, B, H, W, Cout, act, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_stem_dsconv(int dtype, const float* img, const float* w0, const float* b0, const float* wd, const float* bd,
                         const float* wp, const float* bp, void* out, int B, int H, int W, int variant, void* stream) {
  Tmp t;
  constexpr int C = 16;
  std::vector<float> p0(27 * (size_t)C), pd(9 * (size_t)C);
  for (int co = 0; co < C; ++co)
    for (int c = 0; c < 3; ++c)
      for (int tp = 0; tp < 9; ++tp) p0[(size_t)(tp * 3 + c) * C + co] = w0[((size_t)co * 3 + c) * 9 + tp];
  for (int c = 0; c < C; ++c)
    for (int tp = 0; tp < 9; ++tp) pd[(size_t)tp * C + c] = wd[(size_t)c * 9 + tp];
  const int Kp = esam3_gemm_pad_k(C, dtype == 1 ? 2 : 4), Np = esam3_gemm_pad_n(C);
  std::vector<float> pp((size_t)Np * Kp, 0.f);
  for (int n = 0; n < C; ++n)
    for (int k = 0; k < C; ++k) pp[(size_t)n * Kp + k] = wp[(size_t)n * C + k];
  float* d0 = (float*)t.up(p0.data(), p0.size() * 4);
  float* db0 = b0 ? (float*)t.up(b0, C * 4) : nullptr;
  float* dd = (float*)t.up(pd.data(), pd.size() * 4);
  float* dbd = bd ? (float*)t.up(bd, C * 4) : nullptr;
  void* dp = t.upT(dtype, pp);
  float* dbp = bp ? (float*)t.up(bp, C * 4) : nullptr;
  if (!d0 || !dd || !dp || (b0 && !db0) || (bd && !dbd) || (bp && !dbp)) return fail("op_stem_dsconv");
  if (op_timed("stem_dsconv", (hipStream_t)stream, [&]() {
        return esam3_launch_stem_dsconv(dtype, img, d0, db0, dd, dbd, dp, Kp, dbp, out, B, H, W, (hipStream_t)stream, variant);
      }))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_lite_mla(int dtype, const void* ms, void* out, int B, int N, int groups, int dim, void* stream) {
  Tmp t;
  float* kv = (float*)t.raw(sizeof(float) * (size_t)esam3_lite_mla_scratch_floats(B, N, groups, dim));
  if (!kv) return fail("op_lite_mla");
  if (esam3_launch_lite_mla(dtype, ms, groups * 3 * dim, out, groups * dim, kv, B, N, groups, dim,
                            (hipStream_t)stream))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_grouped_pw(int dtype, const void* x, const float* w, void* out, int64_t rows, int C, int gs,
                        void* stream) {
  Tmp t;
  float* dw = (float*)t.up(w, (size_t)C * gs * 4);
  if (!dw) return fail("op_grouped_pw");
  if (esam3_launch_grouped_pw(dtype, x, C, dw, out, C, rows, C, gs, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_resize_bilinear(int dtype, const void* x, void* out, int B, int IH, int IW, int OH, int OW, int C,
                             void* stream) {
  if (esam3_launch_resize_bilinear(dtype, x, out, B, IH, IW, OH, OW, C, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_layernorm(int dtype, const void* x, const void* res, const float* gamma, const float* beta,
                       void* out, int64_t rows, int C, float eps, int act, void* stream) {
  Tmp t;
  float* g = (float*)t.up(gamma, (size_t)C * 4);
  float* b = (float*)t.up(beta, (size_t)C * 4);
  if (!g || !b) return fail("op_layernorm");
  if (esam3_launch_layernorm(dtype, x, res, g, b, out, rows, C, eps, act, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_attention(int dtype, const void* q, const void* k, const void* v, void* out, int B, int Nq, int Nk,
                       int heads, int hd, int few_keys, void* stream) {
  const int D = heads * hd;
  int rc;
  Tmp t;
  if (few_keys == 2 || few_keys == 3) {   // token -> image attention on the matrix cores; 3: k_dev holds [k | v] rows of 2 D
    if (!esam3_t2i_mfma_ok(dtype, Nq, Nk, heads, hd)) { esam3_set_error("op_attention: the MFMA token -> image path does not take this shape"); return -1; }
    float* scratch = (float*)t.raw((size_t)esam3_t2i_mfma_scratch_floats(B, Nq, Nk) * sizeof(float));
    if (!scratch) return fail("op_attention");
    const int ld = few_keys == 3 ? 2 * D : D;
    const void* vp = few_keys == 3 ? (const void*)((const char*)k + (size_t)D * 2) : v;
    rc = op_timed("attn_t2i_mfma", (hipStream_t)stream, [&]() {
      return esam3_launch_t2i_mfma(q, D, k, ld, vp, ld, out, B, Nq, Nk, scratch, (hipStream_t)stream);
    });
  } else if (few_keys) {
    rc = esam3_launch_attn_fewkeys(dtype, q, D, k, D, v, D, out, D, B, Nq, Nk, heads, hd, (hipStream_t)stream);
  } else {
    float* scratch = nullptr;  // the engine hands the tiled token -> image kernel its workspace; do the same here
    if (const int64_t nf = esam3_attn_scratch_floats(B, Nq, Nk, heads, hd)) {
      scratch = (float*)t.raw((size_t)nf * sizeof(float));
      if (!scratch) return fail("op_attention");
    }
    rc = op_timed("attn", (hipStream_t)stream, [&]() {
      return esam3_launch_attn(dtype, q, D, k, D, v, D, out, D, B, Nq, Nk, heads, hd, scratch, (hipStream_t)stream);
    });
  }
  if (rc) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_fill_holes(const float* in, float* out, int n, int H, int W, float thr, float max_area,
                        void* stream) {
  Tmp t;
  int* labels = (int*)t.raw((size_t)n * H * W * 4);
  int* areas = (int*)t.raw((size_t)n * H * W * 4);
  if (!labels || !areas) return fail("op_fill_holes");
  if (op_timed("fill_holes", (hipStream_t)stream, [&]() { return esam3_launch_fill_holes(in, out, labels, areas, n, H, W, thr, max_area, (hipStream_t)stream); }))
    return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_upsample_masks(const float* in, float* out_f32, uint8_t* out_u8, int n, int IH, int IW, int OH,
                            int OW, float thr, void* stream) {
  if (esam3_launch_upsample_masks(in, out_f32, out_u8, n, IH, IW, OH, OW, thr, (hipStream_t)stream)) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

int esam3_op_cast(int dtype, int to_f32, const void* in, void* out, int64_t n, void* stream) {
  int rc = to_f32 ? esam3_launch_cast_to_f32(dtype, in, (float*)out, n, (hipStream_t)stream)
                  : esam3_launch_cast_from_f32(dtype, (const float*)in, out, n, (hipStream_t)stream);
  if (rc) return -1;
  HIP_CHECK_RET(hipStreamSynchronize((hipStream_t)stream));
  return 0;
}

}  // extern "C"

// ---- kernel micro-benchmark (development aid): times one GEMM/conv shape on random data ----
extern "C" int esam3_bench_gemm(int dtype, int B, int H, int W, int Cin, int N, int ksize, int convt, int iters,
                                float* avg_ms) {
  Tmp t;
  const int esz = dtype == 0 ? 4 : 2;
  const int K = Cin * ksize * ksize;
  const int Kp = esam3_gemm_pad_k(K, esz), Np = esam3_gemm_pad_n(N);
  const int64_t M = (int64_t)B * H * W;
  const int pad = ksize >= 2 ? 1 : 0;  // ksize 2: the up-conv gather (N = 4 classes x Cout, convt store)
  const size_t a_elems = (size_t)B * (H + 2 * pad) * (W + 2 * pad) * Cin;
  const size_t o_elems = (size_t)M * N;
  std::vector<float> hw((size_t)Np * Kp), ha(1 << 20);
  uint32_t s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1.0f; };
  for (auto& v : hw) v = rnd() * 0.05f;
  for (auto& v : ha) v = rnd();
  // ESAM3_BENCH_DATA = zeros | gelu: the chip clocks to its power budget, which depends on the operand bits (MI355X_MICROARCH.md,
  // "DVFS give-back") -- zeros bound what the schedule alone could reach, "gelu" is GELU of a unit normal (what the dominant launch reads)
  if (const char* mode = getenv("ESAM3_BENCH_DATA")) {
    if (!strcmp(mode, "zeros")) {
      for (auto& v : hw) v = 0.f;
      for (auto& v : ha) v = 0.f;
    } else if (!strcmp(mode, "gelu")) {
      for (size_t i = 0; i + 1 < ha.size(); i += 2) {  // Box-Muller on the same LCG
        const float u1 = (rnd() + 1.0f) * 0.5f * 0.99998f + 1e-5f, u2 = (rnd() + 1.0f) * 0.5f;
        const float r = sqrtf(-2.f * logf(u1));
        const float z[2] = {r * cosf(6.2831853f * u2), r * sinf(6.2831853f * u2)};
        for (int j = 0; j < 2; ++j) ha[i + j] = 0.5f * z[j] * (1.f + erff(z[j] * 0.70710678f));
      }
    }
  }
  void* w = t.upT(dtype, hw);
  std::vector<float> hb((size_t)Np, 0.01f);
  const float* bias_dev = static_cast<const float*>(t.up(hb.data(), hb.size() * 4));
  void* a = t.raw(a_elems * esz);
  void* o = t.raw(o_elems * esz);
  void* chunk = t.upT(dtype, ha);
  if (!w || !a || !o || !chunk) return fail("bench_gemm");
  for (size_t off = 0; off < a_elems; off += ha.size()) {
    const size_t n = std::min(ha.size(), a_elems - off);
    (void)hipMemcpy((char*)a + off * esz, chunk, n * esz, hipMemcpyDeviceToDevice);
  }
  GemmParams p{};
  p.A = a; p.Wt = w; p.bias = bias_dev; p.out = o; p.M = M; p.N = N; p.K = K; p.Kp = Kp; p.H = H; p.W = W; p.Cin = Cin;
  p.ksize = ksize; p.lda = Cin; p.ldc = convt ? N / 4 : N; p.ldr = p.ldc; p.in_pad = pad; p.res_after_act = 1;
  p.korder = esam3_conv_korder(Cin, ksize, esz);
  if (convt) { p.out_mode = OUT_CONVT2X2; p.convt_cout = N / 4; }
  if (ksize == 2) {
    std::vector<float> hc((size_t)4 * 3 * (N / 4), 0.02f);
    p.border_corr = static_cast<const float*>(t.up(hc.data(), hc.size() * 4));
    if (!p.border_corr) return fail("bench_gemm");
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const bool narrow = !convt && ksize == 3 && esam3_conv3x3_narrow_ok(dtype, N, Cin, H, W, 1, 0, 1, false);  // the engine's choice
  auto launch = [&]() { return narrow ? esam3_launch_conv3x3_narrow(p, nullptr) : esam3_launch_gemm(dtype, p, nullptr); };
  for (int i = 0; i < 3; ++i) if (launch()) return -1;
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters; ++i) if (launch()) return -1;
  (void)hipEventRecord(e1, nullptr);
  HIP_CHECK_RET(hipDeviceSynchronize());
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *avg_ms = ms / iters;
  return 0;
}


// Development aid (tools/bench_dw.py): average time of one depthwise-conv launch on device-resident data.
extern "C" int esam3_bench_dwconv(int dtype, int B, int H, int W, int C, int ks, int stride, int act, int iters, float* avg_ms) {
  Tmp t;
  const int esz = dtype == 0 ? 4 : 2;
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
  const size_t n_in = (size_t)B * H * W * C, n_out = (size_t)B * OH * OW * C;
  std::vector<float> hw((size_t)ks * ks * C, 0.05f), hb((size_t)C, 0.01f), ha(1 << 20);
  uint32_t s = 777u;
  for (auto& v : ha) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  float* w = (float*)t.up(hw.data(), hw.size() * 4);
  float* b = (float*)t.up(hb.data(), hb.size() * 4);
  void* a = t.raw(n_in * esz);
  void* o = t.raw(n_out * esz);
  void* chunk = t.upT(dtype, ha);
  if (!w || !b || !a || !o || !chunk) return fail("bench_dwconv");
  for (size_t off = 0; off < n_in; off += ha.size()) {
    const size_t n = std::min(ha.size(), n_in - off);
    (void)hipMemcpy((char*)a + off * esz, chunk, n * esz, hipMemcpyDeviceToDevice);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (esam3_launch_dwconv(dtype, a, C, w, b, o, C, B, H, W, C, ks, stride, act, nullptr)) return -1;
  (void)hipEventRecord(e0, nullptr);
  for (int i = 0; i < iters; ++i) if (esam3_launch_dwconv(dtype, a, C, w, b, o, C, B, H, W, C, ks, stride, act, nullptr)) return -1;
  (void)hipEventRecord(e1, nullptr);
  HIP_CHECK_RET(hipDeviceSynchronize());
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  *avg_ms = ms / iters;
  return 0;
}


// ---- COCO run-length encoding (eval writers: eval_efficientsam3_all_subsets.py:124-135, masks_ops.py:161-230) ----
int esam3_rle_encode(const uint8_t* masks_dev, int n, int H, int W, uint32_t* counts_dev, int64_t capacity,
                     int32_t* offsets_dev, void* scratch_dev, int64_t scratch_bytes, void* stream) {
  if (!masks_dev || !counts_dev || !offsets_dev || !scratch_dev || n <= 0 || H <= 0 || W <= 0 || capacity <= 0) {
    esam3_set_error("esam3_rle_encode: bad argument");
    return -1;
  }
  if (scratch_bytes < esam3_rle_scratch_bytes(n, H, W, capacity)) {
    esam3_set_error("esam3_rle_encode: scratch of %lld bytes, need %lld", (long long)scratch_bytes,
                    (long long)esam3_rle_scratch_bytes(n, H, W, capacity));
    return -1;
  }
  return esam3_launch_rle_encode(masks_dev, n, H, W, counts_dev, capacity, offsets_dev, scratch_dev, (hipStream_t)stream);
}

// cocoapi's compressed string form of the counts (maskApi.c rleToString / rleFrString: every count after the third is
// stored as the difference to the count two places before; 5 payload bits per character with a continuation bit,
// sign-extended, + 48).  Host code, no device work.  Returns the string length (without a terminator) or -1.
// Host side of the mask return path (sam1_task_predictor.py:293-295: the reference hands back float32 numpy masks): uint8 0 / 1 masks in
// a pinned staging buffer -> the caller's float32 array.  Plain C so that a few Python worker threads can run it concurrently through
// ctypes (the GIL is released for the call); the loop vectorises, the stores are the cost (4 bytes written per byte read).
int esam3_host_widen_u8_f32(const uint8_t* src_host, float* dst_host, int64_t n) {
  if (!src_host || !dst_host || n < 0) { esam3_set_error("esam3_host_widen_u8_f32: bad argument"); return -1; }
  int64_t i = 0;
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SSE2__)
  // non-temporal stores: the 4 n bytes written are not read back by this thread, and a write-allocate would fetch every line first
  // (half of the memory traffic of the plain loop)
  for (; i < n && ((uintptr_t)(dst_host + i) & 15); ++i) dst_host[i] = (float)src_host[i];
  const __m128i zero = _mm_setzero_si128();
  for (; i + 16 <= n; i += 16) {
    const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src_host + i));
    const __m128i lo = _mm_unpacklo_epi8(b, zero), hi = _mm_unpackhi_epi8(b, zero);
    _mm_stream_ps(dst_host + i, _mm_cvtepi32_ps(_mm_unpacklo_epi16(lo, zero)));
    _mm_stream_ps(dst_host + i + 4, _mm_cvtepi32_ps(_mm_unpackhi_epi16(lo, zero)));
    _mm_stream_ps(dst_host + i + 8, _mm_cvtepi32_ps(_mm_unpacklo_epi16(hi, zero)));
    _mm_stream_ps(dst_host + i + 12, _mm_cvtepi32_ps(_mm_unpackhi_epi16(hi, zero)));
  }
  _mm_sfence();
#endif
  for (; i < n; ++i) dst_host[i] = (float)src_host[i];
  return 0;
}

int64_t esam3_rle_to_string(const uint32_t* counts_host, int64_t n_counts, char* out, int64_t capacity) {
  if (!counts_host || !out || n_counts < 0) { esam3_set_error("esam3_rle_to_string: bad argument"); return -1; }
  int64_t p = 0;
  for (int64_t i = 0; i < n_counts; ++i) {
    long long x = (long long)counts_host[i];
    if (i > 2) x -= (long long)counts_host[i - 2];
    bool more = true;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? x != -1 : x != 0;
      if (more) c |= 0x20;
      c += 48;
      if (p >= capacity) { esam3_set_error("esam3_rle_to_string: output buffer too small"); return -1; }
      out[p++] = c;
    }
  }
  return p;
}
int64_t esam3_rle_from_string(const char* s, int64_t len, uint32_t* counts_host, int64_t capacity) {
  if (!s || !counts_host || len < 0) { esam3_set_error("esam3_rle_from_string: bad argument"); return -1; }
  int64_t m = 0, p = 0;
  while (p < len) {
    long long x = 0;
    int k = 0;
    bool more = true;
    while (more) {
      if (p >= len) { esam3_set_error("esam3_rle_from_string: truncated string"); return -1; }
      const long long c = (long long)s[p] - 48;
      // cocoapi's alphabet is the 64 characters from '0' (48) up; a count has at most 7 groups of 5 payload bits here
      // (32-bit counts + sign), so a longer continuation run is a malformed string, not something to shift by >= 64
      if (c < 0 || c > 63) { esam3_set_error("esam3_rle_from_string: character %d outside the RLE alphabet", (int)s[p]); return -1; }
      if (k > 12) { esam3_set_error("esam3_rle_from_string: more than 13 continuation characters in one count"); return -1; }
      x |= (c & 0x1f) << (5 * k);
      more = (c & 0x20) != 0;
      ++p; ++k;
      if (!more && (c & 0x10) && 5 * k < 64) x |= -1LL << (5 * k);
    }
    if (m > 2) x += (long long)counts_host[m - 2];
    if (x < 0 || x > 0xffffffffLL) { esam3_set_error("esam3_rle_from_string: count %lld does not fit a run length", x); return -1; }
    if (m >= capacity) { esam3_set_error("esam3_rle_from_string: output buffer too small"); return -1; }
    counts_host[m++] = (uint32_t)x;
  }
  return m;
}
