// Stage-1 distillation loss, forward and dL/dpreds (stage1/train_image_encoder_stage1.py:186-210,271-307):
// masked MSE and masked cosine loss between the student embedding and the teacher embedding, both [B][HW][C] token-major
// (NHWC), and the gradient of  masked_mse + w * masked_cosine_loss  with respect to the student embedding.
// HBM-bound: each embedding is read once (twice by the backward kernel, the second time out of L2); fixed-order reductions.
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/esam3.h"
#include "esam3_common.h"
#include "kernels.h"
#include "resize_aa.h"
#include "train_act.h"

namespace {

template <int DT> struct Elem;  // 0 f32, 1 bf16, 2 f16
template <> struct Elem<0> {
  using type = float;
  struct raw8 { float4 a, b; };   // 8 elements as loaded: several rows' loads can be in flight before the first is unpacked
  static __device__ inline raw8 loadraw(const float* p) { return raw8{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)}; }
  static __device__ inline void unpack(const raw8& q, float* v) {
    v[0] = q.a.x; v[1] = q.a.y; v[2] = q.a.z; v[3] = q.a.w; v[4] = q.b.x; v[5] = q.b.y; v[6] = q.b.z; v[7] = q.b.w;
  }
  static __device__ inline void load8(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
};
template <> struct Elem<1> {
  using type = uint16_t;
  typedef uint4 raw8;
  static __device__ inline raw8 loadraw(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ inline void unpack(const raw8& q, float* v) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ inline void load8(const uint16_t* p, float* v) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
};
template <> struct Elem<2> {
  using type = __half;
  static __device__ inline void load8(const __half* p, float* v) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const __half2* h = reinterpret_cast<const __half2*>(&q);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 f = __half22float2(h[i]);
      v[2 * i] = f.x;
      v[2 * i + 1] = f.y;
    }
  }
};

// one wavefront per pixel: sum_c (p - t)^2 and 1 - cos(p, t) (F.cosine_similarity: each norm clamped at 1e-8)
template <int DP, int DTT>
__global__ __launch_bounds__(256) void distill_pixel_kernel(const typename Elem<DP>::type* __restrict__ preds,
                                                            const typename Elem<DTT>::type* __restrict__ teacher,
                                                            float* __restrict__ px, int64_t n_pix, int C) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= n_pix) return;
  const auto* p = preds + pix * C;
  const auto* t = teacher + pix * C;
  float sq = 0.f, dot = 0.f, np_ = 0.f, nt = 0.f;
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
    float a[8], b[8];
    Elem<DP>::load8(p + c0, a);
    Elem<DTT>::load8(t + c0, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = a[e] - b[e];
      sq = fmaf(d, d, sq);
      dot = fmaf(a[e], b[e], dot);
      np_ = fmaf(a[e], a[e], np_);
      nt = fmaf(b[e], b[e], nt);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    sq += __shfl_xor(sq, o, 64);
    dot += __shfl_xor(dot, o, 64);
    np_ += __shfl_xor(np_, o, 64);
    nt += __shfl_xor(nt, o, 64);
  }
  if (lane == 0) {
    px[pix * 2] = sq;
    px[pix * 2 + 1] = 1.f - dot / (fmaxf(sqrtf(np_), 1e-8f) * fmaxf(sqrtf(nt), 1e-8f));
  }
}

// per image: masked sums over its HW pixels in a fixed order -> out[b] = {mse_b, cos_b} (each divided by
// max(#valid, 1), masked_mse / masked_cosine_loss before their batch mean)
__global__ __launch_bounds__(256) void distill_reduce_kernel(const float* __restrict__ px, const uint8_t* __restrict__ valid,
                                                             int HW, float* __restrict__ out) {
  __shared__ float s0[256], s1[256], s2[256];
  const int b = blockIdx.x;
  float a0 = 0.f, a1 = 0.f, n = 0.f;
  for (int i = threadIdx.x; i < HW; i += 256) {
    if (valid[(int64_t)b * HW + i]) {
      a0 += px[((int64_t)b * HW + i) * 2];
      a1 += px[((int64_t)b * HW + i) * 2 + 1];
      n += 1.f;
    }
  }
  s0[threadIdx.x] = a0; s1[threadIdx.x] = a1; s2[threadIdx.x] = n;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      s0[threadIdx.x] += s0[threadIdx.x + o];
      s1[threadIdx.x] += s1[threadIdx.x + o];
      s2[threadIdx.x] += s2[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float d = fmaxf(s2[0], 1.f);
    out[b * 2] = s0[0] / d;
    out[b * 2 + 1] = s1[0] / d;
  }
}

template <int DP>
int launch_pixel(int teacher_dtype, const void* preds, const void* teacher, float* px, int64_t n_pix, int C, hipStream_t s) {
  const dim3 grid((unsigned)((n_pix + 3) / 4));
  using P = typename Elem<DP>::type;
  if (teacher_dtype == 0) hipLaunchKernelGGL((distill_pixel_kernel<DP, 0>), grid, dim3(256), 0, s, (const P*)preds, (const float*)teacher, px, n_pix, C);
  else if (teacher_dtype == 1) hipLaunchKernelGGL((distill_pixel_kernel<DP, 1>), grid, dim3(256), 0, s, (const P*)preds, (const uint16_t*)teacher, px, n_pix, C);
  else hipLaunchKernelGGL((distill_pixel_kernel<DP, 2>), grid, dim3(256), 0, s, (const P*)preds, (const __half*)teacher, px, n_pix, C);
  return 0;
}

// per image: 1 / max(#valid, 1)
__global__ __launch_bounds__(256) void valid_recip_kernel(const uint8_t* __restrict__ valid, int HW, float* __restrict__ out,
                                                          const float* __restrict__ scale_dev = nullptr) {
  __shared__ int sc[256];
  const int b = blockIdx.x;
  int n = 0;
  for (int i = threadIdx.x; i < HW; i += 256) n += valid[(int64_t)b * HW + i] ? 1 : 0;
  sc[threadIdx.x] = n;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sc[threadIdx.x] += sc[threadIdx.x + o];
    __syncthreads();
  }
  // scale_dev: the loss scale read ON the device (a power of two: exact wherever it multiplies) -- no host read-back in a training step
  if (threadIdx.x == 0) out[b] = (1.f / fmaxf((float)sc[0], 1.f)) * (scale_dev ? *scale_dev : 1.f);
}

template <int DP> struct Store8;
template <> struct Store8<0> {
  static __device__ inline void st(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct Store8<1> {
  static __device__ inline void st(uint16_t* p, const float* v) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
};

// dL/dpreds of  L = mean_b [ sum_valid |p - t|^2 / n_b ] + w * mean_b [ sum_valid (1 - cos(p, t)) / n_b ]  (times grad_scale,
// e.g. 1 / ACCUMULATION_STEPS): for a valid pixel of image b
//   g = s_b * ( 2 (p - t)  -  w * ( t - (p.t / |p|^2) p ) / (|p| |t|) ),   s_b = grad_scale / (B n_b);   0 for masked pixels.
// (F.cosine_similarity clamps each norm at 1e-8: a clamped |p| is a constant, so its p-term vanishes.)  One wavefront per pixel.
template <int DP, int DTT>
__global__ __launch_bounds__(256) void distill_backward_kernel(const typename Elem<DP>::type* __restrict__ preds,
                                                               const typename Elem<DTT>::type* __restrict__ teacher,
                                                               const uint8_t* __restrict__ valid, const float* __restrict__ recip_n,
                                                               typename Elem<DP>::type* __restrict__ grad, int64_t n_pix, int HW,
                                                               int C, float w_cos, float scale_over_b) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= n_pix) return;
  const auto* p = preds + pix * C;
  const auto* t = teacher + pix * C;
  auto* g = grad + pix * C;
  if (!valid[pix]) {
    const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int c0 = lane * 8; c0 < C; c0 += 512) Store8<DP>::st(g + c0, z);
    return;
  }
  float dot = 0.f, np_ = 0.f, nt = 0.f;
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
    float a[8], b[8];
    Elem<DP>::load8(p + c0, a);
    Elem<DTT>::load8(t + c0, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dot = fmaf(a[e], b[e], dot);
      np_ = fmaf(a[e], a[e], np_);
      nt = fmaf(b[e], b[e], nt);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dot += __shfl_xor(dot, o, 64);
    np_ += __shfl_xor(np_, o, 64);
    nt += __shfl_xor(nt, o, 64);
  }
  const float s = scale_over_b * recip_n[pix / HW];
  const float lp = sqrtf(np_), lt = sqrtf(nt);
  const float inv = 1.f / (fmaxf(lp, 1e-8f) * fmaxf(lt, 1e-8f));
  const float kt = -w_cos * inv;                                   // coefficient of t
  const float kp = lp > 1e-8f ? w_cos * inv * dot / np_ : 0.f;     // coefficient of p from d|p|
  for (int c0 = lane * 8; c0 < C; c0 += 512) {
    float a[8], b[8], o[8];
    Elem<DP>::load8(p + c0, a);
    Elem<DTT>::load8(t + c0, b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = s * (2.f * (a[e] - b[e]) + kt * b[e] + kp * a[e]);
    Store8<DP>::st(g + c0, o);
  }
}

template <int DP>
int launch_backward(int teacher_dtype, const void* preds, const void* teacher, const uint8_t* valid, const float* recip, void* grad,
                    int64_t n_pix, int HW, int C, float w, float sb, hipStream_t s) {
  const dim3 grid((unsigned)((n_pix + 3) / 4));
  using P = typename Elem<DP>::type;
  if (teacher_dtype == 0) hipLaunchKernelGGL((distill_backward_kernel<DP, 0>), grid, dim3(256), 0, s, (const P*)preds, (const float*)teacher, valid, recip, (P*)grad, n_pix, HW, C, w, sb);
  else if (teacher_dtype == 1) hipLaunchKernelGGL((distill_backward_kernel<DP, 1>), grid, dim3(256), 0, s, (const P*)preds, (const uint16_t*)teacher, valid, recip, (P*)grad, n_pix, HW, C, w, sb);
  else hipLaunchKernelGGL((distill_backward_kernel<DP, 2>), grid, dim3(256), 0, s, (const P*)preds, (const __half*)teacher, valid, recip, (P*)grad, n_pix, HW, C, w, sb);
  return 0;
}

}  // namespace

static int distill_loss_backward_impl(int preds_dtype, const void* preds_dev, int teacher_dtype, const void* teacher_dev,
                                const uint8_t* valid_dev, int B, int HW, int C, float cosine_weight, float grad_scale, const float* scale_dev,
                                void* grad_preds_dev, float* scratch_dev, void* stream);

int esam3_distill_loss_backward(int preds_dtype, const void* preds_dev, int teacher_dtype, const void* teacher_dev,
                                const uint8_t* valid_dev, int B, int HW, int C, float cosine_weight, float grad_scale,
                                void* grad_preds_dev, float* scratch_dev, void* stream) {
  return distill_loss_backward_impl(preds_dtype, preds_dev, teacher_dtype, teacher_dev, valid_dev, B, HW, C, cosine_weight, grad_scale, nullptr,
                                    grad_preds_dev, scratch_dev, stream);
}

// the same gradient times a scale that lives in DEVICE memory (the AMP loss scale of the updater's state, a power of two): the training step
// no longer reads it back to the host -- that read was the step's only synchronisation point (1 ms of idle GPU per step behind it)
int esam3_distill_loss_backward_ds(int preds_dtype, const void* preds_dev, int teacher_dtype, const void* teacher_dev,
                                   const uint8_t* valid_dev, int B, int HW, int C, float cosine_weight, float grad_scale,
                                   const float* scale_dev, void* grad_preds_dev, float* scratch_dev, void* stream) {
  if (!scale_dev) {
    esam3_set_error("esam3_distill_loss_backward_ds: scale_dev is NULL");
    return -1;
  }
  return distill_loss_backward_impl(preds_dtype, preds_dev, teacher_dtype, teacher_dev, valid_dev, B, HW, C, cosine_weight, grad_scale, scale_dev,
                                    grad_preds_dev, scratch_dev, stream);
}

static int distill_loss_backward_impl(int preds_dtype, const void* preds_dev, int teacher_dtype, const void* teacher_dev,
                                const uint8_t* valid_dev, int B, int HW, int C, float cosine_weight, float grad_scale, const float* scale_dev,
                                void* grad_preds_dev, float* scratch_dev, void* stream) {
  if (!preds_dev || !teacher_dev || !valid_dev || !grad_preds_dev || !scratch_dev || B <= 0 || HW <= 0 || C <= 0 || C % 8 ||
      preds_dtype < 0 || preds_dtype > 1 || teacher_dtype < 0 || teacher_dtype > 2) {
    esam3_set_error("esam3_distill_loss_backward: bad argument (B=%d HW=%d C=%d dtypes %d/%d; C must be a multiple of 8)", B, HW, C,
                    preds_dtype, teacher_dtype);
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(valid_recip_kernel, dim3((unsigned)B), dim3(256), 0, s, valid_dev, HW, scratch_dev, scale_dev);
  const int64_t n_pix = (int64_t)B * HW;
  const float sb = grad_scale / (float)B;
  if (preds_dtype == 0) launch_backward<0>(teacher_dtype, preds_dev, teacher_dev, valid_dev, scratch_dev, grad_preds_dev, n_pix, HW, C, cosine_weight, sb, s);
  else launch_backward<1>(teacher_dtype, preds_dev, teacher_dev, valid_dev, scratch_dev, grad_preds_dev, n_pix, HW, C, cosine_weight, sb, s);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

namespace {
}  // namespace

int esam3_distill_loss(int preds_dtype, const void* preds_dev, int teacher_dtype, const void* teacher_dev,
                       const uint8_t* valid_dev, int B, int HW, int C, float* per_image_dev, float* scratch_dev,
                       void* stream) {
  if (!preds_dev || !teacher_dev || !valid_dev || !per_image_dev || !scratch_dev || B <= 0 || HW <= 0 || C <= 0 || C % 8 ||
      preds_dtype < 0 || preds_dtype > 1 || teacher_dtype < 0 || teacher_dtype > 2) {
    esam3_set_error("esam3_distill_loss: bad argument (B=%d HW=%d C=%d dtypes %d/%d; C must be a multiple of 8)", B, HW, C,
                    preds_dtype, teacher_dtype);
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_pix = (int64_t)B * HW;
  if (preds_dtype == 0) launch_pixel<0>(teacher_dtype, preds_dev, teacher_dev, scratch_dev, n_pix, C, s);
  else launch_pixel<1>(teacher_dtype, preds_dev, teacher_dev, scratch_dev, n_pix, C, s);
  hipLaunchKernelGGL(distill_reduce_kernel, dim3((unsigned)B), dim3(256), 0, s, scratch_dev, valid_dev, HW, per_image_dev);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---- stage-1 input pipeline (BASELINE config 5) ----------------------------------------------------------
// SA1BDataset.__getitem__ (stage1/data/sa1b_dataset.py:163,170-171,217-228) turns a uint8 image into the network input
// with  ResizeLongestSide(img_size).apply_image_torch (stage1/data/transforms.py:48-55: fp32 antialiased bilinear resize
// so that the longer side becomes img_size, NOT rounded back to uint8)  ->  (x - pixel_mean) / pixel_std  ->  zero
// padding at the bottom / right to img_size x img_size.  One thread per pixel of the padded square: inside the resized
// (new_h, new_w) rectangle the taps of torch's upsample_bilinear2d_aa in its summation order (horizontal pass, then
// vertical, separate multiply and add), outside 0.  HBM-bound (each source byte is read by ~support^2 threads out of L2).
namespace {
__global__ __launch_bounds__(256) void stage1_preprocess_kernel(const uint8_t* __restrict__ in, int H, int W, float* __restrict__ out,
                                                                int S, int NH, int NW, float m0, float m1, float m2, float s0,
                                                                float s1, float s2) {
#pragma clang fp contract(off)
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (ox >= S) return;
  const int64_t plane = (int64_t)S * S;
  float* o = out + (int64_t)oy * S + ox;
  if (oy >= NH || ox >= NW) {  // F.pad(x, (0, padw, 0, padh)) AFTER the normalisation: the padding is 0, not -mean/std
    o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f;
    return;
  }
  const float sy = (float)H / (float)NH, sx = (float)W / (float)NW;
  const float sup_y = sy >= 1.f ? sy : 1.f, sup_x = sx >= 1.f ? sx : 1.f;
  const float inv_y = sy >= 1.f ? 1.f / sy : 1.f, inv_x = sx >= 1.f ? 1.f / sx : 1.f;
  int y0, ny, x0, nx;
  float ym, xm;
  aa_span(oy, H, sy, sup_y, y0, ny, ym);
  aa_span(ox, W, sx, sup_x, x0, nx, xm);
  float ty = 0.f, tx = 0.f;
  for (int j = 0; j < ny; ++j) ty += aa_tap(j, y0, ym, inv_y);
  for (int j = 0; j < nx; ++j) tx += aa_tap(j, x0, xm, inv_x);
  float acc[3] = {0.f, 0.f, 0.f};
  for (int jy = 0; jy < ny; ++jy) {
    float wy = aa_tap(jy, y0, ym, inv_y);
    if (ty != 0.f) wy /= ty;
    const uint8_t* row = in + ((int64_t)(y0 + jy) * W + x0) * 3;
    float r[3] = {0.f, 0.f, 0.f};
    for (int jx = 0; jx < nx; ++jx) {
      float wx = aa_tap(jx, x0, xm, inv_x);
      if (tx != 0.f) wx /= tx;
#pragma unroll
      for (int c = 0; c < 3; ++c) r[c] += (float)row[jx * 3 + c] * wx;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += r[c] * wy;
  }
  o[0] = (acc[0] - m0) / s0;
  o[plane] = (acc[1] - m1) / s1;
  o[2 * plane] = (acc[2] - m2) / s2;
}

// ------------------------------------------------------------------------------------------------------------------
// Update half of the stage-1 training step: what NativeScalerWithGradNormCount.__call__ does after
// backward (stage1/utils.py:347-362: GradScaler.unscale_, clip_grad_norm_ or ampscaler_get_grad_norm :324-338,
// GradScaler.step(optimizer), GradScaler.update()) with the optimizer stage1/optimizer.py:6-29 builds (torch.optim.AdamW, two
// weight-decay groups from set_weight_decay :32-46, lr_scale groups from utils.py:557-620), on ONE flat fp32 arena holding every
// trainable parameter (each padded to a multiple of 256 elements; per-256-chunk tables carry the group: decay on/off, lr
// scale).  Three launches, no host synchronisation -- the scale, the growth tracker, the step count, the found-inf flag, the
// gradient norm and the clip coefficient live in a 16-float device state:
//   1. update_norm_kernel     reads the gradients once: non-finite check on the raw values, sum of squares of g / scale
//   2. update_finalize_kernel one workgroup: total norm (fp64 sum of the partials), clip coefficient, found_inf, AdamW step
//                             count and bias corrections, loss-scale growth / backoff
//   3. update_adamw_kernel    p, m, v <- AdamW(g / scale * clip) in torch's operation order; skipped (gradients still zeroed on
//                             request) when a gradient was non-finite; optionally also writes the bf16 copy of the weights
// HBM-bound: 4 bytes per parameter in pass 1, 16 read + 12 (+2) written in pass 3.
// ------------------------------------------------------------------------------------------------------------------
enum { US_SCALE = 0, US_TRACKER = 1, US_FOUND_INF = 2, US_GRAD_NORM = 3, US_STEP = 4, US_CLIP_COEF = 5, US_BC1 = 6, US_BC2_SQRT = 7 };

__device__ __forceinline__ float update_inv_scale(const float* state, int amp) {
  return amp ? (float)(1.0 / (double)state[US_SCALE]) : 1.f;  // GradScaler: scale.double().reciprocal().float()
}

__global__ __launch_bounds__(256) void update_norm_kernel(const float* __restrict__ grads, int64_t n, const float* __restrict__ state,
                                                          int amp, float* __restrict__ partial /* [grid][2] */) {
  const float inv = update_inv_scale(state, amp);
  float acc = 0.f;
  int bad = 0;
  const int64_t n4 = n >> 2;  // n is a multiple of 256
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 g = reinterpret_cast<const float4*>(grads)[i];
    const float v[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      bad |= !isfinite(v[e]);
      const float u = inv == 1.f ? v[e] : v[e] * inv;
      acc = fmaf(u, u, acc);
    }
  }
  __shared__ float sacc[256];
  __shared__ int sbad[256];
  sacc[threadIdx.x] = acc;
  sbad[threadIdx.x] = bad;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sacc[threadIdx.x] += sacc[threadIdx.x + o];
      sbad[threadIdx.x] |= sbad[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = sacc[0];
    partial[2 * blockIdx.x + 1] = sbad[0] ? 1.f : 0.f;
  }
}

__global__ __launch_bounds__(256) void update_finalize_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ state,
                                                              int amp, float clip_grad, double beta1, double beta2, float growth,
                                                              float backoff, int growth_interval) {
  __shared__ double ssum[256];
  __shared__ int sbad[256];
  double a = 0.0;
  int bad = 0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {  // fixed order: deterministic
    a += (double)partial[2 * i];
    bad |= partial[2 * i + 1] != 0.f;
  }
  ssum[threadIdx.x] = a;
  sbad[threadIdx.x] = bad;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      ssum[threadIdx.x] += ssum[threadIdx.x + o];
      sbad[threadIdx.x] |= sbad[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const float total = (float)sqrt(ssum[0]);
  const bool found = amp && sbad[0];
  state[US_GRAD_NORM] = total;  // what clip_grad_norm_ / ampscaler_get_grad_norm return (inf / nan when a gradient is)
  state[US_FOUND_INF] = found ? 1.f : 0.f;
  float coef = 1.f;
  if (clip_grad > 0.f) {  // clip_grad_norm_: clip_coef = max_norm / (total_norm + 1e-6), clamped to 1
    coef = clip_grad / (total + 1e-6f);
    coef = coef > 1.f ? 1.f : coef;
  }
  state[US_CLIP_COEF] = coef;
  if (!found) {  // GradScaler.step runs optimizer.step(): AdamW's step count and bias corrections (python doubles in torch)
    const double step = (double)state[US_STEP] + 1.0;
    state[US_STEP] = (float)step;
    state[US_BC1] = (float)(1.0 - pow(beta1, step));
    state[US_BC2_SQRT] = (float)sqrt(1.0 - pow(beta2, step));
  }
  if (amp) {  // GradScaler.update (_amp_update_scale_)
    if (found) {
      state[US_SCALE] *= backoff;
      state[US_TRACKER] = 0.f;
    } else {
      const float ok = state[US_TRACKER] + 1.f;
      if ((int)ok == growth_interval) {
        const float grown = state[US_SCALE] * growth;
        if (isfinite(grown)) state[US_SCALE] = grown;
        state[US_TRACKER] = 0.f;
      } else {
        state[US_TRACKER] = ok;
      }
    }
  }
}

// inv_scale is passed in: the finalize kernel has already grown / backed off state[US_SCALE]
__global__ __launch_bounds__(256) void update_adamw_kernel(float* __restrict__ params, float* __restrict__ grads, float* __restrict__ m,
                                                           float* __restrict__ v, int64_t n, const float* __restrict__ chunk_lr_scale,
                                                           const uint8_t* __restrict__ chunk_decay, const float* __restrict__ state,
                                                           const float* __restrict__ inv_scale_p, double lr, double beta1, double beta2,
                                                           double eps_d, double wd, int zero_grads, bf16_t* __restrict__ bf16_out) {
  const bool skip = state[US_FOUND_INF] != 0.f;
  const float inv = *inv_scale_p, coef = state[US_CLIP_COEF];
  const float bc1 = state[US_BC1], bc2s = state[US_BC2_SQRT];
  // torch hands its python doubles to the fp32 kernels as fp32 scalars: 1 - beta is formed in double FIRST (1 - fp32(0.999)
  // is 1.3e-5 away from fp32(1 - 0.999))
  const float w1 = (float)(1.0 - beta1), w2 = (float)(1.0 - beta2), b2 = (float)beta2, eps = (float)eps_d;
  const int64_t nchunks = n >> 8;
  for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {  // one 256-element chunk per iteration: one group per chunk
    const int64_t i = (ch << 8) + threadIdx.x;
    const float g0 = grads[i];
    if (zero_grads) grads[i] = 0.f;
    if (skip) continue;
    const double lr_g = lr * (double)(chunk_lr_scale ? chunk_lr_scale[ch] : 1.f);
    const float decay_mul = (chunk_decay == nullptr || chunk_decay[ch]) ? (float)(1.0 - lr_g * wd) : 1.f;
    const float step_size = (float)(lr_g / (double)bc1);
    float g = inv == 1.f ? g0 : g0 * inv;  // unscale_
    g = g * coef;                          // clip_grad_norm_ (coef == 1 when not clipping)
    float p = params[i] * decay_mul;       // param.mul_(1 - lr * weight_decay)
    float mm = m[i], vv = v[i];
    mm = mm + w1 * (g - mm);               // exp_avg.lerp_(grad, 1 - beta1)
    vv = vv * b2 + (w2 * g) * g;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = sqrtf(vv) / bc2s + eps;
    p = p - step_size * (mm / denom);      // param.addcdiv_(exp_avg, denom, value = -step_size)
    params[i] = p;
    m[i] = mm;
    v[i] = vv;
    if (bf16_out) bf16_out[i] = f32_to_bf16(p);
  }
}

__global__ void update_save_inv_scale_kernel(const float* __restrict__ state, int amp, float* __restrict__ out) {
  *out = update_inv_scale(state, amp);
}


// ------------------------------------------------------------------------------------------------------------------
// BatchNorm2d in TRAINING mode, forward and backward, on NHWC rows [rows = B H W][C]: every ConvLayer of the
// EfficientViT / RepViT / TinyViT students normalises with batch statistics while stage 1 trains
// (backbones/efficientvit/nn/ops.py:69-77 norm="bn2d" = nn.BatchNorm2d, nn/norm.py:47; stage1/train_image_encoder_stage1.py:165
// model.train(), :310-314 EVAL_BN_WHEN_TRAINING False in every shipped config).  Every ConvLayer / Conv2d_BN of the stage-1 trainer runs them
// (efficientsam3_amd/train_blocks.py: ConvLayerTrain); checked against torch's batch_norm + autograd.  The inference engine folds BN instead.
//   forward : mean_c, biased var_c over the rows; y = (x - mean) rstd gamma + beta; running_mean / running_var updated with
//             `momentum` (running_var takes the UNBIASED variance, as torch does); mean and rstd saved for the backward
//   backward: dbeta = sum dy, dgamma = sum dy xhat, dx = gamma rstd (dy - dbeta / n - xhat dgamma / n)
// Two passes over the tensor each way (statistics, then the elementwise map): HBM-bound, 16-byte accesses, fixed-order
// reductions (per-split partial sums in fp32, combined in fp64 by one small kernel: deterministic).  The per-split sums are the
// quantities a SyncBatchNorm (train_image_encoder_stage1.py:60-63) would all-reduce across ranks before the finalize step.
// ------------------------------------------------------------------------------------------------------------------
constexpr int BN_SPLITS = 512;

// partial[split][2][C]: (sum a, sum b) per channel with  FWD: a = x - p, b = (x - p)^2;  BWD: a = dy, b = dy (x - mean) rstd.
// p is a per-channel PIVOT, the tensor's first row (stored after the partials, partial[splits][2][C] .. + C): E[x^2] - E[x]^2 on the
// raw values cancels catastrophically when |mean| >> std (a large conv bias in front of the norm); shifted by a sample of the
// channel, the two sums are of the order of the spread, not of the offset, and the fp64 finalize loses nothing.
// the BatchNorm's output for one element, in ONE written-down form: the forward map stores it, the recomputing backward forms it again
__device__ __forceinline__ float bn_affine(float v, float mean, float rstd, float gamma, float beta) { return fmaf((v - mean) * rstd, gamma, beta); }
template <int DT> __device__ __forceinline__ float bn_stored(float v) {   // the value as the storage type holds it
  if constexpr (DT == 1) return __uint_as_float(pack_bf16x2(v, 0.f) << 16);
  else return v;
}

template <int DT, bool BWD>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const typename Elem<DT>::type* __restrict__ x,
                                                        const typename Elem<DT>::type* __restrict__ dy, int64_t rows, int C,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        float* __restrict__ partial,
                                                        const typename Elem<DT>::type* __restrict__ pre = nullptr, int act = 0,
                                                        const float* __restrict__ gamma = nullptr, const float* __restrict__ beta = nullptr) {
  // BWD with `pre` (round 5): dy is the gradient of act(pre), pre = the BatchNorm's own output -- g = dy act'(pre) is formed here instead of
  // in a separate elementwise pass that wrote it out (the ConvLayer's activation derivative fused into the BatchNorm backward).
  // BWD with act and NO `pre` but gamma / beta (round 6): pre is RECOMPUTED from x exactly as the forward map formed and stored it
  // (bn_affine, rounded to the storage type) -- the forward need not write the BatchNorm's output and the backward does not read it: two of
  // the seven tensor passes of a fused BatchNorm + activation backward, one of the four of its forward.
  extern __shared__ float red[];  // [RL][2][C]
  const bool rc = BWD && !pre && act != 0 && gamma && beta;
  const int CG = C >> 3, RL = 256 / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
  const int64_t r0 = (int64_t)blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  float a[8], b[8], mu[8], rs[8], pv[8], gm[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    a[e] = 0.f; b[e] = 0.f; pv[e] = 0.f;
    mu[e] = BWD ? mean[cg * 8 + e] : 0.f;
    rs[e] = BWD ? rstd[cg * 8 + e] : 0.f;
    gm[e] = rc ? gamma[cg * 8 + e] : 0.f;
    bt[e] = rc ? beta[cg * 8 + e] : 0.f;
  }
  if constexpr (!BWD) {
    if (rl < RL) Elem<DT>::load8(x + cg * 8, pv);   // row 0
    if (blockIdx.x == 0 && rl == 0) {
#pragma unroll
      for (int e = 0; e < 8; ++e) partial[(int64_t)gridDim.x * 2 * C + cg * 8 + e] = pv[e];
    }
  }
  // one row: the same operations in the same order whichever way the loop below groups the loads
  auto accumulate = [&](const float* v, float* g, const float* pv2) {
    if constexpr (BWD) {
      if (pre) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= act_grad(pv2[e], act);
      } else if (rc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= act_grad(bn_stored<DT>(bn_affine(v[e], mu[e], rs[e], gm[e], bt[e])), act);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a[e] += g[e];
        b[e] = fmaf(g[e], (v[e] - mu[e]) * rs[e], b[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = v[e] - pv[e];
        a[e] += d;
        b[e] = fmaf(d, d, b[e]);
      }
    }
  };
  if (rl < RL) {
    // round 6: U rows' loads in flight per thread (issued before the first is used) -- with wide channel counts a workgroup has few row
    // lanes and each walked its rows one dependent load at a time (C = 512, 127 k rows: 1.5 TB/s); rows still accumulate in ascending order
    int64_t r = r0 + rl;
    constexpr int U = BWD ? 2 : 4;   // rows in flight: the backward loads two or three tensors per row
    for (; r + (U - 1) * RL < r1; r += U * RL) {
      typename Elem<DT>::raw8 qv[U], qg[U], qp[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        qv[u] = Elem<DT>::loadraw(x + (r + u * RL) * C + cg * 8);
        if constexpr (BWD) {
          qg[u] = Elem<DT>::loadraw(dy + (r + u * RL) * C + cg * 8);
          if (pre) qp[u] = Elem<DT>::loadraw(pre + (r + u * RL) * C + cg * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[8], g[8], pv2[8];
        Elem<DT>::unpack(qv[u], v);
        if constexpr (BWD) {
          Elem<DT>::unpack(qg[u], g);
          if (pre) Elem<DT>::unpack(qp[u], pv2);
        }
        accumulate(v, g, pv2);
      }
    }
    for (; r < r1; r += RL) {
      float v[8], g[8], pv2[8];
      Elem<DT>::load8(x + r * C + cg * 8, v);
      if constexpr (BWD) {
        Elem<DT>::load8(dy + r * C + cg * 8, g);
        if (pre) Elem<DT>::load8(pre + r * C + cg * 8, pv2);
      }
      accumulate(v, g, pv2);
    }
  }
  if (rl < RL) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(rl * 2 + 0) * C + cg * 8 + e] = a[e];
      red[(rl * 2 + 1) * C + cg * 8 + e] = b[e];
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {  // fixed order over the row lanes
    float t = 0.f;
    for (int l = 0; l < RL; ++l) t += red[l * 2 * C + i];
    partial[(int64_t)blockIdx.x * 2 * C + i] = t;
  }
}

// FWD: mean, rstd, running statistics.  BWD: dgamma, dbeta.  A workgroup owns 64 channels; its four waves sum contiguous quarters of the
// splits in fp64, two interleaved chains each, and the partial sums are combined in one fixed order (deterministic).  The first version
// -- one thread per channel walking all 512 splits -- took 128 us per call, 9 ms of a 54 ms stage-1 step over its 70 calls
// (profiles/r04/stage1_step_kernel_stats.csv): a chain of 1024 dependent loads + fp64 adds on a handful of lanes.
template <bool BWD>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int splits, int C, int64_t rows, double eps,
                                                           double momentum, float* __restrict__ o0, float* __restrict__ o1,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ var_out = nullptr) {
  // 64 channels x 16 wavefronts: each wave sums a contiguous sixteenth of the splits (round 5: four waves of 128 splits were 20 us per call,
  // 1.5 ms of a 21.6 ms B1 step over its 70 calls, profiles/r05/stage1_step_b1_kernel_stats_final.csv)
  constexpr int Q = 16;
  __shared__ double sq[2][Q][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + e;
  const int per = (splits + Q - 1) / Q, z0 = q * per, z1 = splits < z0 + per ? splits : z0 + per;
  double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
  if (c < C) {
    int z = z0;
    for (; z + 1 < z1; z += 2) {
      a0 += (double)partial[(int64_t)z * 2 * C + c];
      b0 += (double)partial[(int64_t)z * 2 * C + C + c];
      a1 += (double)partial[(int64_t)(z + 1) * 2 * C + c];
      b1 += (double)partial[(int64_t)(z + 1) * 2 * C + C + c];
    }
    if (z < z1) {
      a0 += (double)partial[(int64_t)z * 2 * C + c];
      b0 += (double)partial[(int64_t)z * 2 * C + C + c];
    }
  }
  sq[0][q][e] = a0 + a1;
  sq[1][q][e] = b0 + b1;
  __syncthreads();
  if (q != 0 || c >= C) return;
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int k = 0; k < Q; k += 4) {   // one written-down order
    s0 += (sq[0][k][e] + sq[0][k + 1][e]) + (sq[0][k + 2][e] + sq[0][k + 3][e]);
    s1 += (sq[1][k][e] + sq[1][k + 1][e]) + (sq[1][k + 2][e] + sq[1][k + 3][e]);
  }
  if constexpr (BWD) {
    o0[c] = (float)s1;  // dgamma
    o1[c] = (float)s0;  // dbeta
  } else {
    const double n = (double)rows, shift = s0 / n, mean = (double)partial[(int64_t)splits * 2 * C + c] + shift;
    double var = s1 / n - shift * shift;   // sums of (x - pivot): see bn_reduce_kernel
    var = var > 0.0 ? var : 0.0;
    o0[c] = (float)mean;
    o1[c] = (float)(1.0 / sqrt(var + eps));
    if (var_out) var_out[c] = (float)var;   // the biased variance itself: what a SyncBatchNorm gathers from every rank
    if (running_mean) running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
    if (running_var) running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * var * (n > 1.0 ? n / (n - 1.0) : 1.0));
  }
}

// FWD: y = (x - mean) rstd gamma + beta.   BWD: dx = gamma rstd (dy - dbeta / n - xhat dgamma / n)
template <int DT, bool BWD, bool HOIST>
__global__ __launch_bounds__(256) void bn_map_kernel(const typename Elem<DT>::type* __restrict__ x,
                                                     const typename Elem<DT>::type* __restrict__ dy,
                                                     typename Elem<DT>::type* __restrict__ out, int64_t rows, int C,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ dgamma, const float* __restrict__ dbeta, float invn,
                                                     const typename Elem<DT>::type* __restrict__ pre = nullptr,
                                                     typename Elem<DT>::type* __restrict__ out_act = nullptr, int act = 0) {
  // round 5, the ConvLayer's activation in the same pass.  FWD with `out_act`: also writes act(y) (of the value as stored: rounded to bf16
  // first in bf16 mode, so the result equals the separate pass's); `out` may then be NULL (round 6: the backward recomputes y, see
  // bn_reduce_kernel).  BWD with `pre`: dy is the gradient of act(pre), g = dy act'(pre); BWD with act, no `pre` and `beta`: pre recomputed.
  // HOIST (round 6): with 256 % (C / 8) == 0 a thread's channel group is the same in every iteration of the grid-stride loop, so its 8
  // channels' constants live in registers -- the general form re-reads 24 - 40 of them per 16 bytes of tensor.
  const int CG = C >> 3;
  const int64_t total = rows * CG;   // invn = 1 / (rows of the whole batch): `rows` of this rank, or of all ranks under SyncBatchNorm
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  int cg = (int)(i0 % CG);
  const int dcg = (int)(stride % CG);  // the channel group advances by a fixed amount per iteration: no 64-bit modulo in the loop
  const bool rc = BWD && !pre && act != 0 && beta;
  float pm[8], pr[8], pg[8], pb[8], pdg[8], pdb[8];
  // a channel group's constants as 16-byte loads (written element by element they became 48 single-dword loads, each touching 16 cache lines
  // of the wave: 3 ms of a B1 step, profiles/r06/bn_map_hoist_trace.txt)
  auto ld8 = [](const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  };
  auto load_params = [&](int g8) {
    ld8(mean + g8 * 8, pm);
    ld8(rstd + g8 * 8, pr);
    ld8(gamma + g8 * 8, pg);
    if (!BWD || rc) ld8(beta + g8 * 8, pb);
    if constexpr (BWD) {
      ld8(dgamma + g8 * 8, pdg);
      ld8(dbeta + g8 * 8, pdb);
#pragma unroll
      for (int e = 0; e < 8; ++e) pdb[e] *= invn;
    }
  };
#pragma unroll
  for (int e = 0; e < 8; ++e) pb[e] = 0.f;
  if constexpr (HOIST) load_params(cg);
  for (int64_t i = i0; i < total; i += stride, cg = cg + dcg >= CG ? cg + dcg - CG : cg + dcg) {
    if constexpr (!HOIST) load_params(cg);
    float v[8], o[8];
    Elem<DT>::load8(x + i * 8, v);
    if constexpr (BWD) {
      float g[8];
      Elem<DT>::load8(dy + i * 8, g);
      if (pre) {
        float pv2[8];
        Elem<DT>::load8(pre + i * 8, pv2);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= act_grad(pv2[e], act);
      } else if (rc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] *= act_grad(bn_stored<DT>(bn_affine(v[e], pm[e], pr[e], pg[e], pb[e])), act);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xh = (v[e] - pm[e]) * pr[e];
        o[e] = pg[e] * pr[e] * (g[e] - pdb[e] - xh * pdg[e] * invn);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = bn_affine(v[e], pm[e], pr[e], pg[e], pb[e]);
      if (out_act) {
        float oa[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) oa[e] = act_fwd(bn_stored<DT>(o[e]), act);   // of what `out` holds (or would hold)
        Store8<DT>::st(out_act + i * 8, oa);
      }
    }
    if (BWD || out) Store8<DT>::st(out + i * 8, o);
  }
}

// launches the map with the hoisted form when the channel count allows it
template <int DT, bool BWD>
void bn_map_launch(unsigned grid, hipStream_t s, const void* x, const void* dy, void* out, int64_t rows, int C, const float* gamma, const float* beta,
                   const float* mean, const float* rstd, const float* dgamma, const float* dbeta, float invn, const void* pre = nullptr,
                   void* out_act = nullptr, int act = 0) {
  typedef typename Elem<DT>::type T;
  if (256 % (C >> 3) == 0)   // a thread pays its constants once: fewer, longer threads (16 workgroups per CU)
    hipLaunchKernelGGL((bn_map_kernel<DT, BWD, true>), dim3(grid < 4096 ? grid : 4096), dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out, rows, C, gamma, beta, mean, rstd,
                       dgamma, dbeta, invn, (const T*)pre, (T*)out_act, act);
  else
    hipLaunchKernelGGL((bn_map_kernel<DT, BWD, false>), dim3(grid), dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out, rows, C, gamma, beta, mean, rstd,
                       dgamma, dbeta, invn, (const T*)pre, (T*)out_act, act);
}

template <int DT>
int bn_forward_t(const void* x, void* y, int64_t rows, int C, const float* gamma, const float* beta, float* rm, float* rv,
                 double momentum, double eps, float* save_mean, float* save_rstd, float* partial, hipStream_t s, void* y_act = nullptr,
                 int act = 0) {
  typedef typename Elem<DT>::type T;
  const int RL = 256 / (C / 8);
  const int splits = (int)(rows < BN_SPLITS ? rows : BN_SPLITS);
  hipLaunchKernelGGL((bn_reduce_kernel<DT, false>), dim3((unsigned)splits), dim3(256), sizeof(float) * 2 * (size_t)RL * C, s, (const T*)x,
                     (const T*)nullptr, rows, C, (const float*)nullptr, (const float*)nullptr, partial);
  hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, s, partial, splits, C, rows, eps, momentum,
                     save_mean, save_rstd, rm, rv);
  const int64_t total = rows * (C / 8);
  const unsigned grid = (unsigned)(total / 256 + 1 < 16384 ? total / 256 + 1 : 16384);
  bn_map_launch<DT, false>(grid, s, x, nullptr, y, rows, C, gamma, beta, save_mean, save_rstd, nullptr, nullptr, 0.f, nullptr, y_act, act);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <int DT>
int bn_backward_t(const void* x, const void* dy, void* dx, int64_t rows, int C, const float* gamma, const float* save_mean,
                  const float* save_rstd, float* dgamma, float* dbeta, float* partial, hipStream_t s, const void* pre = nullptr, int act = 0,
                  const float* beta = nullptr) {
  // act with `pre` NULL and `beta` given: the BatchNorm's output is recomputed from x in both passes (bn_reduce_kernel)
  typedef typename Elem<DT>::type T;
  const int RL = 256 / (C / 8);
  const int splits = (int)(rows < BN_SPLITS ? rows : BN_SPLITS);
  hipLaunchKernelGGL((bn_reduce_kernel<DT, true>), dim3((unsigned)splits), dim3(256), sizeof(float) * 2 * (size_t)RL * C, s, (const T*)x,
                     (const T*)dy, rows, C, save_mean, save_rstd, partial, (const T*)pre, act, gamma, beta);
  hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, s, partial, splits, C, rows, 0.0, 0.0, dgamma,
                     dbeta, (float*)nullptr, (float*)nullptr);
  const int64_t total = rows * (C / 8);
  const unsigned grid = (unsigned)(total / 256 + 1 < 16384 ? total / 256 + 1 : 16384);
  bn_map_launch<DT, true>(grid, s, x, dy, dx, rows, C, gamma, beta, save_mean, save_rstd, dgamma, dbeta, 1.f / (float)rows, pre, nullptr, act);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// The four halves of the two calls above, for SyncBatchNorm (train_image_encoder_stage1.py:62-63 --use-sync-bn): between the statistics and
// the elementwise map the host combines the ranks' (mean, biased variance, rows) (forward) or sums their (sum dy, sum dy xhat, rows)
// (backward) with one collective each -- torch.nn.SyncBatchNorm's own protocol (efficientsam3_amd/train_blocks.py: bn_train_forward).
template <int DT>
int bn_stats_t(const void* x, int64_t rows, int C, double eps, float* mean, float* rstd, float* var, float* partial, hipStream_t s) {
  typedef typename Elem<DT>::type T;
  const int RL = 256 / (C / 8);
  const int splits = (int)(rows < BN_SPLITS ? rows : BN_SPLITS);
  hipLaunchKernelGGL((bn_reduce_kernel<DT, false>), dim3((unsigned)splits), dim3(256), sizeof(float) * 2 * (size_t)RL * C, s, (const T*)x,
                     (const T*)nullptr, rows, C, (const float*)nullptr, (const float*)nullptr, partial);
  hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, s, partial, splits, C, rows, eps, 0.0, mean, rstd,
                     (float*)nullptr, (float*)nullptr, var);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <int DT>
int bn_apply_t(const void* x, void* y, int64_t rows, int C, const float* gamma, const float* beta, const float* mean, const float* rstd,
               hipStream_t s) {
  const int64_t total = rows * (C / 8);
  const unsigned grid = (unsigned)(total / 256 + 1 < 16384 ? total / 256 + 1 : 16384);
  bn_map_launch<DT, false>(grid, s, x, nullptr, y, rows, C, gamma, beta, mean, rstd, nullptr, nullptr, 0.f);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <int DT>
int bn_backward_sums_t(const void* x, const void* dy, int64_t rows, int C, const float* mean, const float* rstd, float* sum_dy_xhat, float* sum_dy,
                       float* partial, hipStream_t s) {
  typedef typename Elem<DT>::type T;
  const int RL = 256 / (C / 8);
  const int splits = (int)(rows < BN_SPLITS ? rows : BN_SPLITS);
  hipLaunchKernelGGL((bn_reduce_kernel<DT, true>), dim3((unsigned)splits), dim3(256), sizeof(float) * 2 * (size_t)RL * C, s, (const T*)x, (const T*)dy, rows,
                     C, mean, rstd, partial);
  hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3((unsigned)((C + 63) / 64)), dim3(1024), 0, s, partial, splits, C, rows, 0.0, 0.0, sum_dy_xhat, sum_dy,
                     (float*)nullptr, (float*)nullptr);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
template <int DT>
int bn_backward_apply_t(const void* x, const void* dy, void* dx, int64_t rows, int C, const float* gamma, const float* mean, const float* rstd,
                        const float* sum_dy_xhat, const float* sum_dy, double total_rows, hipStream_t s) {
  const int64_t total = rows * (C / 8);
  const unsigned grid = (unsigned)(total / 256 + 1 < 16384 ? total / 256 + 1 : 16384);
  bn_map_launch<DT, true>(grid, s, x, dy, dx, rows, C, gamma, nullptr, mean, rstd, sum_dy_xhat, sum_dy, (float)(1.0 / total_rows));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

bool bn_args_ok(const char* who, int dtype, int64_t rows, int C) {
  if ((dtype != 0 && dtype != 1) || rows <= 0 || C <= 0 || C % 8 || C > 2048) {
    esam3_set_error("%s: dtype %d, rows %lld, C %d (fp32 / bf16, C a multiple of 8 up to 2048)", who, dtype, (long long)rows, C);
    return false;
  }
  return true;
}

}  // namespace

void esam3_stage1_preprocess_shape(int H, int W, int img_size, int* new_h, int* new_w) {
  // ResizeLongestSide.get_preprocess_shape (transforms.py:81-88): double arithmetic, round half up
  const double scale = (double)img_size * 1.0 / (double)(H > W ? H : W);
  if (new_h) *new_h = (int)((double)H * scale + 0.5);
  if (new_w) *new_w = (int)((double)W * scale + 0.5);
}

int esam3_stage1_preprocess_u8(const uint8_t* img_hwc_u8_dev, int H, int W, float* out_chw_f32_dev, int img_size,
                               const float* pixel_mean3, const float* pixel_std3, int* new_h, int* new_w, void* stream) {
  if (!img_hwc_u8_dev || !out_chw_f32_dev || !pixel_mean3 || !pixel_std3 || H <= 0 || W <= 0 || img_size <= 0) {
    esam3_set_error("esam3_stage1_preprocess_u8: bad argument");
    return -1;
  }
  for (int c = 0; c < 3; ++c)
    if (!(pixel_std3[c] > 0.f)) { esam3_set_error("esam3_stage1_preprocess_u8: pixel_std[%d] must be positive", c); return -1; }
  int nh = 0, nw = 0;
  esam3_stage1_preprocess_shape(H, W, img_size, &nh, &nw);
  if (nh <= 0 || nw <= 0 || nh > img_size || nw > img_size) {
    esam3_set_error("esam3_stage1_preprocess_u8: %d x %d does not resize into %d", H, W, img_size);
    return -1;
  }
  if (new_h) *new_h = nh;
  if (new_w) *new_w = nw;
  hipLaunchKernelGGL(stage1_preprocess_kernel, dim3((unsigned)((img_size + 255) / 256), (unsigned)img_size), dim3(256), 0,
                     (hipStream_t)stream, img_hwc_u8_dev, H, W, out_chw_f32_dev, img_size, nh, nw, pixel_mean3[0], pixel_mean3[1],
                     pixel_mean3[2], pixel_std3[0], pixel_std3[1], pixel_std3[2]);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int64_t esam3_stage1_update_workspace(int64_t n) {
  (void)n;
  return (int64_t)((2 * 1024 + 1) * sizeof(float));  // [1024 blocks][2] partials + the pre-update 1 / scale
}

int esam3_stage1_update(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const float* chunk_lr_scale,
                        const uint8_t* chunk_decay, double lr, double beta1, double beta2, double eps, double weight_decay,
                        float clip_grad, float* state16, float growth_factor, float backoff_factor, int growth_interval,
                        int amp_enabled, int zero_grads, void* bf16_params_out, void* workspace, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !state16 || !workspace || n <= 0 || (n & 255)) {
    esam3_set_error("esam3_stage1_update: bad argument (the arena length must be a positive multiple of 256)");
    return -1;
  }
  if (!(beta1 >= 0.0 && beta1 < 1.0) || !(beta2 >= 0.0 && beta2 < 1.0) || !(eps >= 0.0) || !(lr >= 0.0) || !(weight_decay >= 0.0) ||
      growth_interval < 1) {
    esam3_set_error("esam3_stage1_update: invalid hyper-parameter");  // the checks torch.optim.AdamW / GradScaler make
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)workspace;
  float* inv_scale = partial + 2 * 1024;
  const int64_t nchunks = n >> 8;
  const int nb = (int)(nchunks < 1024 ? nchunks : 1024);
  hipLaunchKernelGGL(update_save_inv_scale_kernel, dim3(1), dim3(1), 0, s, state16, amp_enabled, inv_scale);
  hipLaunchKernelGGL(update_norm_kernel, dim3((unsigned)nb), dim3(256), 0, s, grads, n, state16, amp_enabled, partial);
  hipLaunchKernelGGL(update_finalize_kernel, dim3(1), dim3(256), 0, s, partial, nb, state16, amp_enabled, clip_grad, beta1, beta2,
                     growth_factor, backoff_factor, growth_interval);
  const int64_t gb = nchunks < 8192 ? nchunks : 8192;
  hipLaunchKernelGGL(update_adamw_kernel, dim3((unsigned)gb), dim3(256), 0, s, params, grads, exp_avg, exp_avg_sq, n, chunk_lr_scale,
                     chunk_decay, state16, inv_scale, lr, beta1, beta2, eps, weight_decay, zero_grads, (bf16_t*)bf16_params_out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int64_t esam3_bn_train_workspace(int C) { return (int64_t)sizeof(float) * (2 * BN_SPLITS + 1) * (int64_t)(C > 0 ? C : 0); }  // + the pivots

int esam3_bn_train_forward(int dtype, const void* x, void* y, int64_t rows, int C, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, double momentum, double eps, float* save_mean, float* save_rstd,
                           void* workspace, void* stream) {
  if (!bn_args_ok("esam3_bn_train_forward", dtype, rows, C)) return -1;
  if (!x || !y || !gamma || !beta || !save_mean || !save_rstd || !workspace || !(eps >= 0.0)) {
    esam3_set_error("esam3_bn_train_forward: bad argument");
    return -1;
  }
  return dtype == 0 ? bn_forward_t<0>(x, y, rows, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd,
                                      (float*)workspace, (hipStream_t)stream)
                    : bn_forward_t<1>(x, y, rows, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd,
                                      (float*)workspace, (hipStream_t)stream);
}

// The ConvLayer's activation in the BatchNorm's own passes (round 5; ops.py:39-81 Conv2d -> BatchNorm2d -> act): forward also writes
// y_act = act(y); backward takes dy = the gradient of act(y) with pre = y and forms dy act'(pre) on the fly -- one elementwise pass less each way.
int esam3_bn_act_train_forward(int dtype, const void* x, void* y, void* y_act, int act, int64_t rows, int C, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, double momentum, double eps, float* save_mean, float* save_rstd,
                               void* workspace, void* stream) {
  if (!bn_args_ok("esam3_bn_act_train_forward", dtype, rows, C)) return -1;
  if (!x || !y_act || !gamma || !beta || !save_mean || !save_rstd || !workspace || !(eps >= 0.0) || act < ACT_RELU || act > ACT_SIGMOID) {   // y may be NULL
    esam3_set_error("esam3_bn_act_train_forward: bad argument (act relu | gelu | hswish | sigmoid)");
    return -1;
  }
  return dtype == 0 ? bn_forward_t<0>(x, y, rows, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd,
                                      (float*)workspace, (hipStream_t)stream, y_act, act)
                    : bn_forward_t<1>(x, y, rows, C, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd,
                                      (float*)workspace, (hipStream_t)stream, y_act, act);
}

int esam3_bn_act_train_backward(int dtype, const void* x, const void* dy, const void* pre, int act, void* dx, int64_t rows, int C,
                                const float* gamma, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, void* workspace,
                                void* stream) {
  if (!bn_args_ok("esam3_bn_act_train_backward", dtype, rows, C)) return -1;
  if (!x || !dy || !pre || !dx || !gamma || !save_mean || !save_rstd || !dgamma || !dbeta || !workspace || act < ACT_RELU || act > ACT_SIGMOID) {
    esam3_set_error("esam3_bn_act_train_backward: bad argument (act relu | gelu | hswish | sigmoid)");
    return -1;
  }
  return dtype == 0 ? bn_backward_t<0>(x, dy, dx, rows, C, gamma, save_mean, save_rstd, dgamma, dbeta, (float*)workspace, (hipStream_t)stream, pre, act)
                    : bn_backward_t<1>(x, dy, dx, rows, C, gamma, save_mean, save_rstd, dgamma, dbeta, (float*)workspace, (hipStream_t)stream, pre, act);
}

// the same backward WITHOUT the BatchNorm's saved output: pre = y is recomputed from x, save_mean / save_rstd, gamma and beta exactly as the
// forward formed and stored it (the forward may then be called with y = NULL) -- two tensor reads less, and one tensor less kept per layer
int esam3_bn_act_train_backward_rc(int dtype, const void* x, const void* dy, int act, void* dx, int64_t rows, int C, const float* gamma,
                                   const float* beta, const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta,
                                   void* workspace, void* stream) {
  if (!bn_args_ok("esam3_bn_act_train_backward_rc", dtype, rows, C)) return -1;
  if (!x || !dy || !dx || !gamma || !beta || !save_mean || !save_rstd || !dgamma || !dbeta || !workspace || act < ACT_RELU || act > ACT_SIGMOID) {
    esam3_set_error("esam3_bn_act_train_backward_rc: bad argument (act relu | gelu | hswish | sigmoid)");
    return -1;
  }
  return dtype == 0 ? bn_backward_t<0>(x, dy, dx, rows, C, gamma, save_mean, save_rstd, dgamma, dbeta, (float*)workspace, (hipStream_t)stream, nullptr, act, beta)
                    : bn_backward_t<1>(x, dy, dx, rows, C, gamma, save_mean, save_rstd, dgamma, dbeta, (float*)workspace, (hipStream_t)stream, nullptr, act, beta);
}

int esam3_bn_train_stats(int dtype, const void* x, int64_t rows, int C, double eps, float* mean, float* rstd, float* var, void* workspace,
                         void* stream) {
  if (!bn_args_ok("esam3_bn_train_stats", dtype, rows, C)) return -1;
  if (!x || !mean || !rstd || !var || !workspace || !(eps >= 0.0)) {
    esam3_set_error("esam3_bn_train_stats: bad argument");
    return -1;
  }
  return dtype == 0 ? bn_stats_t<0>(x, rows, C, eps, mean, rstd, var, (float*)workspace, (hipStream_t)stream)
                    : bn_stats_t<1>(x, rows, C, eps, mean, rstd, var, (float*)workspace, (hipStream_t)stream);
}

int esam3_bn_train_apply(int dtype, const void* x, void* y, int64_t rows, int C, const float* gamma, const float* beta, const float* mean,
                         const float* rstd, void* stream) {
  if (!bn_args_ok("esam3_bn_train_apply", dtype, rows, C)) return -1;
  if (!x || !y || !gamma || !beta || !mean || !rstd) {
    esam3_set_error("esam3_bn_train_apply: bad argument");
    return -1;
  }
  return dtype == 0 ? bn_apply_t<0>(x, y, rows, C, gamma, beta, mean, rstd, (hipStream_t)stream)
                    : bn_apply_t<1>(x, y, rows, C, gamma, beta, mean, rstd, (hipStream_t)stream);
}

int esam3_bn_train_backward_sums(int dtype, const void* x, const void* dy, int64_t rows, int C, const float* mean, const float* rstd,
                                 float* sum_dy_xhat, float* sum_dy, void* workspace, void* stream) {
  if (!bn_args_ok("esam3_bn_train_backward_sums", dtype, rows, C)) return -1;
  if (!x || !dy || !mean || !rstd || !sum_dy_xhat || !sum_dy || !workspace) {
    esam3_set_error("esam3_bn_train_backward_sums: bad argument");
    return -1;
  }
  return dtype == 0 ? bn_backward_sums_t<0>(x, dy, rows, C, mean, rstd, sum_dy_xhat, sum_dy, (float*)workspace, (hipStream_t)stream)
                    : bn_backward_sums_t<1>(x, dy, rows, C, mean, rstd, sum_dy_xhat, sum_dy, (float*)workspace, (hipStream_t)stream);
}

int esam3_bn_train_backward_apply(int dtype, const void* x, const void* dy, void* dx, int64_t rows, int C, const float* gamma, const float* mean,
                                  const float* rstd, const float* sum_dy_xhat, const float* sum_dy, double total_rows, void* stream) {
  if (!bn_args_ok("esam3_bn_train_backward_apply", dtype, rows, C)) return -1;
  if (!x || !dy || !dx || !gamma || !mean || !rstd || !sum_dy_xhat || !sum_dy || !(total_rows > 0.0)) {
    esam3_set_error("esam3_bn_train_backward_apply: bad argument (total_rows > 0: the rows of all ranks, or 1 when the sums are already divided by it)");
    return -1;
  }
  return dtype == 0 ? bn_backward_apply_t<0>(x, dy, dx, rows, C, gamma, mean, rstd, sum_dy_xhat, sum_dy, total_rows, (hipStream_t)stream)
                    : bn_backward_apply_t<1>(x, dy, dx, rows, C, gamma, mean, rstd, sum_dy_xhat, sum_dy, total_rows, (hipStream_t)stream);
}

int esam3_bn_train_backward(int dtype, const void* x, const void* dy, void* dx, int64_t rows, int C, const float* gamma,
                            const float* save_mean, const float* save_rstd, float* dgamma, float* dbeta, void* workspace,
                            void* stream) {
  if (!bn_args_ok("esam3_bn_train_backward", dtype, rows, C)) return -1;
  if (!x || !dy || !dx || !gamma || !save_mean || !save_rstd || !dgamma || !dbeta || !workspace) {
    esam3_set_error("esam3_bn_train_backward: bad argument");
    return -1;
  }
  return dtype == 0 ? bn_backward_t<0>(x, dy, dx, rows, C, gamma, save_mean, save_rstd, dgamma, dbeta, (float*)workspace, (hipStream_t)stream)
                    : bn_backward_t<1>(x, dy, dx, rows, C, gamma, save_mean, save_rstd, dgamma, dbeta, (float*)workspace, (hipStream_t)stream);
}
