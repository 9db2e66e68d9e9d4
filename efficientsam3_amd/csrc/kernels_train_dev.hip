// Stage-1 training path, device-resident weights (stage1/train_image_encoder_stage1.py:165-226).
//
// The block tests of round 3 drove the forward operators through the esam3_op_* TEST entry points: host fp32 weights, packed on
// the CPU, uploaded, launched, synchronised -- per call.  A training step cannot work like that: the parameters live in the
// optimizer's flat fp32 arena ON THE DEVICE (esam3_stage1_update) and change every step.  The entry points below take the fp32
// master weight where it lies (a view of that arena), re-pack it on the device into the layout the forward kernel wants (one
// small launch), and launch the same kernels as the inference engine -- no host copy, no synchronisation:
//   esam3_train_linear     1x1 conv / Linear (and its data gradient: `transpose`), esam3_launch_gemm
//   esam3_train_conv3x3    dense 3x3 conv, padding 1 (the student head's second conv, stage1/model.py:197-200) and its data
//                          gradient (the same conv with the 180-degree-rotated, channel-transposed weight)
//   esam3_train_conv3x3_s2 the same conv with stride 2 (RepViT / TinyViT patch embedding, forward only; see there)
//   esam3_train_dwconv     depthwise k x k
//   esam3_train_stem       the 3 -> C0 stride-2 stem conv on the fp32 NCHW image
//   esam3_resize_bilinear_backward   adjoint of F.interpolate(bilinear, align_corners=False) (stage1/model.py:205-210)
#include "../../include/esam3.h"
#include "kernels.h"

namespace {

constexpr int VEC = 8;  // channels per thread of the vectorised NHWC kernels

// fp32 master weight -> the implicit-GEMM kernels' packed [Np][Kp] activation-dtype layout (zero padded), one thread per element.
//   mode 0  Linear      logical W[n][k] = w[n * K + k]
//   mode 1  Linear^T    logical W[n][k] = w[k * N + n]                                  (data gradient of a Linear [K][N])
//   mode 2  conv 3x3    logical W[n][(tap, c)] = w[((n * cin + c) * 9 + tap)]           (weight [N][cin][3][3])
//   mode 3  conv 3x3 data gradient: W[n][(tap, c)] = w[((c * N + n) * 9 + (8 - tap))]   (weight [cin][N][3][3]: forward Cout = cin)
// (tap, c) -> k as esam3_conv_k_index: korder 0 k = tap * cin + c; korder 1 k = (c / bke) * 9 * bke + tap * bke + c % bke
template <typename T>
__global__ void pack_gemm_weight_kernel(const float* __restrict__ w, T* __restrict__ out, int N, int K, int Np, int Kp, int mode, int cin,
                                        int korder, int bke) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)Np * Kp) return;
  const int n = (int)(i / Kp), kp = (int)(i - (int64_t)n * Kp);
  float v = 0.f;
  if (n < N && kp < K) {
    if (mode == 0) v = w[(int64_t)n * K + kp];
    else if (mode == 1) v = w[(int64_t)kp * N + n];
    else {
      int tap, c;
      if (korder) {
        const int chunk = kp / (9 * bke), rem = kp - chunk * 9 * bke;
        tap = rem / bke;
        c = chunk * bke + (rem - tap * bke);
      } else {
        tap = kp / cin;
        c = kp - tap * cin;
      }
      v = mode == 2 ? w[((int64_t)n * cin + c) * 9 + tap] : w[((int64_t)c * N + n) * 9 + (8 - tap)];
    }
  }
  out[i] = from_f32<T>(v);
}

// [C][kk] -> [kk][C] (depthwise) ; [Cout][3 c][9 tap] -> [27 = tap * 3 + c][Cout] (stem)
__global__ void pack_dw_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int kk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * kk) return;
  const int t = i / C, c = i - t * C;
  out[i] = w[c * kk + t];
}
// the same with the taps reversed: the data gradient of a stride-1 "same" depthwise conv IS that conv of dy with the
// 180-degree-rotated kernel
__global__ void pack_dw_weight_flipped_kernel(const float* __restrict__ w, float* __restrict__ out, int C, int kk) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * kk) return;
  const int t = i / C, c = i - t * C;
  out[i] = w[c * kk + (kk - 1 - t)];
}
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 27 * Cout) return;
  const int k = i / Cout, co = i - k * Cout;
  const int tap = k / 3, c = k - tap * 3;
  out[i] = w[(co * 3 + c) * 9 + tap];
}

// Round 6: the `x` operand of the stem's weight gradient, im2col of the 3x3 / stride 2 / padding 1 stem on the NCHW fp32 image: row = output
// pixel, column k = c * 9 + kh * 3 + kw (torch.nn.functional.unfold's order), columns 27..31 zero, in the activation dtype.  A thread writes 8
// columns (16 bytes in bf16) of one row; 64 threads = 16 consecutive rows = 1 KB of contiguous output.  Replaces unfold + permute + pad + cast
// (four ATen passes, 2.1 ms at batch 32, profiles/r06/roofline_stage1_step_b1_b32.md).
template <typename T>
__global__ void stem_im2col_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H, int W, int OH, int OW) {
  const int64_t total = (int64_t)B * OH * OW * 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int q = (int)(i & 3);
    const int64_t row = i >> 2;
    const int ox = (int)(row % OW), oy = (int)((row / OW) % OH);
    const int64_t b = row / ((int64_t)OW * OH);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = q * 8 + e;
      float val = 0.f;
      if (k < 27) {
        const int c = k / 9, t = k - c * 9, kh = t / 3, kw = t - kh * 3;
        const int iy = oy * 2 + kh - 1, ix = ox * 2 + kw - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) val = img[((b * 3 + c) * H + iy) * (int64_t)W + ix];
      }
      v[e] = val;
    }
    if constexpr (sizeof(T) == 2) {
      uint4 o;
      o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<uint4*>(out + row * 32 + q * 8) = o;
    } else {
      *reinterpret_cast<float4*>(out + row * 32 + q * 8) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out + row * 32 + q * 8 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  }
}

// dx[b][iy][ix][c] = sum over the output pixels (oy, ox) whose bilinear footprint contains (iy, ix) of weight * dy[b][oy][ox][c]:
// the exact adjoint of resize_bilinear_kernel (same source coordinate, clamping and fraction arithmetic); gather form, fixed order.
template <typename T>
__global__ void resize_bilinear_backward_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int IH, int IW, int OH, int OW, int C) {
  const int CG = C / VEC;
  const int64_t total = (int64_t)B * IH * IW * CG;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % CG);
  const int64_t pix = idx / CG;
  const int ix = (int)(pix % IW), iy = (int)((pix / IW) % IH);
  const int64_t b = pix / ((int64_t)IW * IH);
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  // candidate output rows / columns: source coordinate within (i - 1, i + 1)
  const int oy_lo = max(0, (int)floorf(((float)iy - 1.f + 0.5f) / sy - 0.5f) - 1), oy_hi = min(OH - 1, (int)ceilf(((float)iy + 1.f + 0.5f) / sy - 0.5f) + 1);
  const int ox_lo = max(0, (int)floorf(((float)ix - 1.f + 0.5f) / sx - 0.5f) - 1), ox_hi = min(OW - 1, (int)ceilf(((float)ix + 1.f + 0.5f) / sx - 0.5f) + 1);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  const T* base = dy + b * OH * (int64_t)OW * C + cg * VEC;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    float fy = ((float)oy + 0.5f) * sy - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    const int y0 = (int)fy, y1 = y0 + (y0 < IH - 1 ? 1 : 0);
    const float ly = fy - (float)y0;
    const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      float fx = ((float)ox + 0.5f) * sx - 0.5f;
      fx = fx < 0.f ? 0.f : fx;
      const int x0 = (int)fx, x1 = x0 + (x0 < IW - 1 ? 1 : 0);
      const float lx = fx - (float)x0;
      const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
      if (wx == 0.f) continue;
      float g[VEC];
      Vec8<T>::load(base + ((int64_t)oy * OW + ox) * C, g);
      const float wgt = wy * wx;
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] = fmaf(wgt, g[e], acc[e]);
    }
  }
  Vec8<T>::store(dx + pix * C + cg * VEC, acc);
}

// interior copy into a buffer with a 1-pixel border (the border itself is zeroed by esam3_launch_zero_border): row (b, y) of W * C elements ->
// row (b, y + 1) at column 1; 16 bytes per thread, grid.y = y, grid.z = b
__global__ void pad_copy_kernel(const char* __restrict__ in, char* __restrict__ out, int H, int W, int row_bytes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 16 >= W * row_bytes) return;
  const int y = blockIdx.y, b = blockIdx.z;
  const int64_t src = ((int64_t)b * H + y) * W * row_bytes + (int64_t)i * 16;
  const int64_t dst = (((int64_t)b * (H + 2) + y + 1) * (W + 2) + 1) * row_bytes + (int64_t)i * 16;
  *reinterpret_cast<uint4*>(out + dst) = *reinterpret_cast<const uint4*>(in + src);
}

int bad(const char* what) {
  esam3_set_error("%s: bad arguments", what);
  return -1;
}

int pack_gemm(int dtype, const float* w, void* ws, int N, int K, int mode, int cin, hipStream_t s) {
  const int esz = dtype == 0 ? 4 : 2;
  const int Kp = esam3_gemm_pad_k(K, esz), Np = esam3_gemm_pad_n(N);
  const int64_t tot = (int64_t)Np * Kp;
  const int korder = mode >= 2 ? esam3_conv_korder(cin, 3, esz) : 0;
  const unsigned grid = (unsigned)((tot + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL(pack_gemm_weight_kernel<float>, dim3(grid), dim3(256), 0, s, w, (float*)ws, N, K, Np, Kp, mode, cin, korder, 128 / esz);
  else hipLaunchKernelGGL(pack_gemm_weight_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, w, (bf16_t*)ws, N, K, Np, Kp, mode, cin, korder, 128 / esz);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int64_t esam3_train_pack_bytes(int dtype, int N, int K) {
  const int esz = dtype == 0 ? 4 : 2;
  if (N <= 0 || K <= 0) return 0;
  return (int64_t)esam3_gemm_pad_n(N) * esam3_gemm_pad_k(K, esz) * esz;
}

int esam3_train_linear(int dtype, const void* x, const float* w, const float* bias, void* out, int64_t M, int N, int K, int transpose,
                       void* ws, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !w || !out || !ws || M <= 0 || N <= 0 || K <= 0) return bad("esam3_train_linear");
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == 0 ? 4 : 2;
  if (pack_gemm(dtype, w, ws, N, K, transpose ? 1 : 0, K, s)) return -1;
  GemmParams p{};
  p.A = x; p.Wt = ws; p.bias = bias; p.out = out; p.M = M; p.N = N; p.K = K; p.Kp = esam3_gemm_pad_k(K, esz); p.H = 1; p.W = 1; p.Cin = K;
  p.ksize = 1; p.lda = K; p.ldc = N; p.ldr = N; p.act = ACT_NONE; p.res_after_act = 1;
  return esam3_launch_gemm(dtype, p, s);
}

int esam3_train_conv3x3(int dtype, const void* x, const float* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                        int dgrad, void* ws, void* stream) {
  // forward: x [B][H][W][Cin], w [Cout][Cin][3][3] -> out [B][H][W][Cout].  dgrad: x = dy [B][H][W][Cin] where Cin is the FORWARD conv's
  // Cout, w the forward weight [Cin][Cout][3][3] -> out = dx [B][H][W][Cout] (Cout = the forward conv's Cin); bias must be NULL.
  if ((dtype != 0 && dtype != 1) || !x || !w || !out || !ws || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (dgrad && bias))
    return bad("esam3_train_conv3x3");
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == 0 ? 4 : 2;
  const int K = Cin * 9;
  if (pack_gemm(dtype, w, ws, Cout, K, dgrad ? 3 : 2, Cin, s)) return -1;
  GemmParams p{};
  p.A = x; p.Wt = ws; p.bias = bias; p.out = out; p.M = (int64_t)B * H * W; p.N = Cout; p.K = K; p.Kp = esam3_gemm_pad_k(K, esz);
  p.H = H; p.W = W; p.Cin = Cin; p.ksize = 3; p.lda = Cin; p.ldc = Cout; p.ldr = Cout; p.act = ACT_NONE; p.res_after_act = 1;
  p.korder = esam3_conv_korder(Cin, 3, esz);
  return esam3_launch_gemm(dtype, p, s);
}

// Round 6: the same conv through the zero-bordered-input form of the implicit GEMM -- the engine's 256 x 256 tile kernel stages its 3x3 gather
// by LDS-DMA, which needs the border in memory; the un-bordered form above runs on the register-staged 128-wide kernel (564 TFLOP/s on the head's
// 1024 -> 1024 conv at 32^2 against 1400+ for the tile kernel: 1.1 + 1.05 ms of a B1 training step for forward + data gradient).  The workspace
// carries the packed weights and a bordered copy of the input; with a workspace too small for the copy (or a shape the tile kernel does not
// take) the call is esam3_train_conv3x3.
int64_t esam3_train_conv3x3_workspace(int dtype, int B, int H, int W, int Cin, int Cout) {
  if ((dtype != 0 && dtype != 1) || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  const int esz = dtype == 0 ? 4 : 2;
  const int64_t pack = (esam3_train_pack_bytes(dtype, Cout, 9 * Cin) + 255) & ~(int64_t)255;
  return pack + (int64_t)B * (H + 2) * (W + 2) * Cin * esz;
}

int esam3_train_conv3x3_ws(int dtype, const void* x, const float* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                           int dgrad, void* ws, int64_t ws_bytes, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !w || !out || !ws || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (dgrad && bias))
    return bad("esam3_train_conv3x3_ws");
  const int esz = dtype == 0 ? 4 : 2;
  const int64_t pack = (esam3_train_pack_bytes(dtype, Cout, 9 * Cin) + 255) & ~(int64_t)255;
  const int64_t padded = (int64_t)B * (H + 2) * (W + 2) * Cin * esz;
  const bool tile_kernel = dtype == 1 && Cin % 64 == 0 && Cout >= 192 && ((int64_t)B * H * W) % 256 == 0 && (Cin * esz) % 16 == 0 && B <= 65535 &&
                           H <= 65535 && ws_bytes >= pack + padded;
  if (!tile_kernel) return esam3_train_conv3x3(dtype, x, w, bias, out, B, H, W, Cin, Cout, dgrad, ws, stream);
  hipStream_t s = (hipStream_t)stream;
  const int K = Cin * 9;
  if (pack_gemm(dtype, w, ws, Cout, K, dgrad ? 3 : 2, Cin, s)) return -1;
  char* xp = (char*)ws + pack;
  if (esam3_launch_zero_border(dtype, xp, B, H + 2, W + 2, Cin, s)) return -1;
  const int row_bytes = Cin * esz;
  hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)((W * (row_bytes / 16) + 255) / 256), (unsigned)H, (unsigned)B), dim3(256), 0, s, (const char*)x, xp, H, W,
                     row_bytes);
  HIP_CHECK_RET(hipGetLastError());
  GemmParams p{};
  p.A = xp; p.Wt = ws; p.bias = bias; p.out = out; p.M = (int64_t)B * H * W; p.N = Cout; p.K = K; p.Kp = esam3_gemm_pad_k(K, esz);
  p.H = H; p.W = W; p.Cin = Cin; p.ksize = 3; p.lda = Cin; p.ldc = Cout; p.ldr = Cout; p.act = ACT_NONE; p.res_after_act = 1;
  p.korder = esam3_conv_korder(Cin, 3, esz); p.in_pad = 1;
  return esam3_launch_gemm(dtype, p, s);
}

// Round 5 (RepViT students): the patch embedding's second conv is a dense 3x3 with STRIDE 2 (repvit.py:229-230 Conv2d_BN(C/2, C, 3, 2, 1)):
// the same implicit GEMM with GemmParams::stride = 2; out [B][ceil(H/2)][ceil(W/2)][Cout].  Its data gradient is esam3_train_conv3x3
// (dgrad = 1) on dy spread over the even pixels of a zero H x W grid, its weight gradient nine esam3_linear_wgrad calls on strided views
// (efficientsam3_amd/train_blocks.py: Conv3x3S2Train).
int esam3_train_conv3x3_s2(int dtype, const void* x, const float* w, const float* bias, void* out, int B, int H, int W, int Cin, int Cout,
                           void* ws, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !w || !out || !ws || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0)
    return bad("esam3_train_conv3x3_s2");
  hipStream_t s = (hipStream_t)stream;
  const int esz = dtype == 0 ? 4 : 2;
  const int K = Cin * 9;
  if (pack_gemm(dtype, w, ws, Cout, K, 2, Cin, s)) return -1;
  GemmParams p{};
  p.A = x; p.Wt = ws; p.bias = bias; p.out = out; p.M = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2); p.N = Cout; p.K = K;
  p.Kp = esam3_gemm_pad_k(K, esz); p.H = H; p.W = W; p.Cin = Cin; p.ksize = 3; p.lda = Cin; p.ldc = Cout; p.ldr = Cout; p.act = ACT_NONE;
  p.res_after_act = 1; p.korder = esam3_conv_korder(Cin, 3, esz); p.stride = 2;
  return esam3_launch_gemm(dtype, p, s);
}

int esam3_train_dwconv(int dtype, const void* x, const float* w, const float* bias, void* out, int B, int H, int W, int C, int ksize,
                       int stride, void* ws, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !w || !out || !ws || (ksize != 3 && ksize != 5) || (stride != 1 && stride != 2))
    return bad("esam3_train_dwconv");
  hipStream_t s = (hipStream_t)stream;
  const int kk = ksize * ksize;
  hipLaunchKernelGGL(pack_dw_weight_kernel, dim3((unsigned)((C * kk + 255) / 256)), dim3(256), 0, s, w, (float*)ws, C, kk);
  HIP_CHECK_RET(hipGetLastError());
  return esam3_launch_dwconv(dtype, x, C, (const float*)ws, bias, out, C, B, H, W, C, ksize, stride, ACT_NONE, s);
}

// Round 5: the stride-1 data gradient runs on the FORWARD depthwise kernels (the matrix-core / strip kernels of kernels_backbone.hip:
// 20 - 30 us per layer at batch 8 where the scalar dw_dgrad_kernel took 300 - 490 us, profiles/r04/stage1_step_kernel_stats.csv);
// stride 2 (a transposed convolution) keeps esam3_dwconv_dgrad's kernel.
int esam3_train_dwconv_dgrad(int dtype, const void* dy, const float* w, void* dx, int B, int H, int W, int C, int ksize, int stride,
                             void* ws, void* stream) {
  if ((dtype != 0 && dtype != 1) || !dy || !w || !dx || !ws || (ksize != 3 && ksize != 5) || (stride != 1 && stride != 2))
    return bad("esam3_train_dwconv_dgrad");
  if (stride != 1) return esam3_dwconv_dgrad(dtype, dy, w, dx, B, H, W, C, ksize, stride, stream);
  hipStream_t s = (hipStream_t)stream;
  const int kk = ksize * ksize;
  hipLaunchKernelGGL(pack_dw_weight_flipped_kernel, dim3((unsigned)((C * kk + 255) / 256)), dim3(256), 0, s, w, (float*)ws, C, kk);
  HIP_CHECK_RET(hipGetLastError());
  return esam3_launch_dwconv(dtype, dy, C, (const float*)ws, nullptr, dx, C, B, H, W, C, ksize, 1, ACT_NONE, s);
}

int esam3_train_stem(int dtype, const float* img, const float* w, void* out, int B, int H, int W, int Cout, void* ws, void* stream) {
  if ((dtype != 0 && dtype != 1) || !img || !w || !out || !ws || Cout <= 0) return bad("esam3_train_stem");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((unsigned)((27 * Cout + 255) / 256)), dim3(256), 0, s, w, (float*)ws, Cout);
  HIP_CHECK_RET(hipGetLastError());
  return esam3_launch_stem(dtype, img, (const float*)ws, nullptr, out, B, H, W, Cout, ACT_NONE, s);
}

int esam3_stem_im2col(int dtype, const float* img, void* out, int B, int H, int W, void* stream) {
  if ((dtype != 0 && dtype != 1) || !img || !out || B <= 0 || H <= 0 || W <= 0) return bad("esam3_stem_im2col");
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const int64_t total = (int64_t)B * OH * OW * 4;
  const unsigned grid = (unsigned)(total / 256 + 1 < 65536 ? total / 256 + 1 : 65536);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == 0) hipLaunchKernelGGL(stem_im2col_kernel<float>, dim3(grid), dim3(256), 0, s, img, (float*)out, B, H, W, OH, OW);
  else hipLaunchKernelGGL(stem_im2col_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, img, (bf16_t*)out, B, H, W, OH, OW);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_resize_bilinear_backward(int dtype, const void* dy, void* dx, int B, int IH, int IW, int OH, int OW, int C, void* stream) {
  if ((dtype != 0 && dtype != 1) || !dy || !dx || B <= 0 || IH <= 0 || IW <= 0 || OH <= 0 || OW <= 0 || C <= 0 || C % VEC)
    return bad("esam3_resize_bilinear_backward");
  hipStream_t s = (hipStream_t)stream;
  const int64_t total = (int64_t)B * IH * IW * (C / VEC);
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (dtype == 0) hipLaunchKernelGGL(resize_bilinear_backward_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)dy, (float*)dx, B, IH, IW, OH, OW, C);
  else hipLaunchKernelGGL(resize_bilinear_backward_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, B, IH, IW, OH, OW, C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // extern "C"
