// HBM-bound kernels of the student image encoder (EfficientViT family) and shared
// elementwise helpers.  NHWC activations, 16-byte (8 x bf16 / 2x4 x f32) channel vectors
// per lane so that a wavefront's accesses are contiguous along C.
#include "resize_aa.h"
#include "kernels.h"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // 16-byte register value

constexpr int VEC = 8;  // channels per thread for vectorised NHWC kernels

// Workgroups are dispatched round-robin over the 8 XCDs (linear id % 8), each with a private
// L2.  Stencil kernels whose neighbouring workgroups share input rows remap the linear id so
// that every XCD owns one contiguous band of rows and the shared rows hit in its L2.
__device__ __forceinline__ void xcd_remap_2d(unsigned gx, unsigned gy, unsigned& bx, unsigned& by) {
  const unsigned nb = gx * gy;
  const unsigned L = blockIdx.x;  // launched as a 1-D grid of gx*gy workgroups
  const unsigned q = nb / 8, r = nb % 8, xcd = L % 8, idx = L / 8;
  const unsigned Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  bx = Lp % gx;
  by = Lp / gx;
}


// ------------------------------------------------------------------------------------
// E0 stem: 3x3 stride-2 pad-1 conv, Cin = 3, NCHW fp32 in -> NHWC T out
// (efficientvit/backbone.py:48-56; RepViT / TinyViT patch embedding).  Cout <= 64.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                   const float* __restrict__ bias, T* __restrict__ out, int B,
                                                   int H, int W, int Cout, int act, unsigned gx, unsigned gy) {
  // thread = (output pixel, 8-channel group); weights [27][Cout] + bias in LDS
  __shared__ float sw[27 * 64];
  __shared__ float sb[64];
  for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) sw[i] = w[i];
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) sb[i] = bias ? bias[i] : 0.f;
  __syncthreads();
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const unsigned CG = (unsigned)Cout / VEC;
  unsigned bx, by;
  xcd_remap_2d(gx, gy, bx, by);
  const unsigned xi = bx * blockDim.x + threadIdx.x;
  const unsigned cg = xi % CG, ow = xi / CG;
  if ((int)ow >= OW) return;
  const unsigned b = by / (unsigned)OH;
  const int oh = (int)(by - b * (unsigned)OH);
  const int co0 = (int)cg * VEC;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = sb[co0 + e];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* plane = img + ((int64_t)(b * 3 + c) * H) * W;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 2 * oh + kh - 1;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = 2 * (int)ow + kw - 1;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float xv = plane[(int64_t)ih * W + iw];
        const float* wk = sw + ((kh * 3 + kw) * 3 + c) * Cout + co0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = fmaf(xv, wk[e], acc[e]);
      }
    }
  }
  act_apply_n<VEC>(acc, act);
  Vec8<T>::store(out + (((int64_t)b * OH + oh) * OW + ow) * Cout + co0, acc);
}

// bf16 stem on the matrix cores (RepViT / TinyViT patch embedding, 3 -> 32 channels: the VALU kernel above spends 0.9 ms of
// a B = 32 step on 27 strided scalar loads and 216 FMAs per thread).  A workgroup owns a 16 x 16 output tile, stages its
// 33 x 33 x 3 input halo in LDS as bf16 (the reference's autocast rounds the conv input the same way) and computes 16
// pixels x 16 channels x (27 -> 32 patch values) per v_mfma_f32_16x16x32_bf16; a lane ends with 4 consecutive channels
// of one pixel.  w: [27][COUT] fp32 (k = tap * 3 + c), rounded to bf16 here.
template <int COUT>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                       const float* __restrict__ bias, bf16_t* __restrict__ out, int H, int W, int OH,
                                                       int OW, int tiles_x, int act) {
  constexpr int TS = 16, IR = 2 * TS + 1, IP = 36, NB = COUT / 16;
  __shared__ __attribute__((aligned(16))) bf16_t simg[3 * IR * IP];
  struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int b = blockIdx.y;
  const int oy0 = ty * TS, ox0 = tx * TS;
  const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;  // input pixel of simg[.][0][0]
  for (int i = threadIdx.x; i < 3 * IR * (IP / 4); i += 256) {
    const int cy = i / (IP / 4), xg = i - cy * (IP / 4);
    const int c = cy / IR, y = cy - c * IR;
    const int iy = iy0 + y, ix = ix0 + 4 * xg;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H) {
      const float* rp = img + ((int64_t)(b * 3 + c) * H + iy) * W;
      if (ix >= 0 && ix + 3 < W) {
        const F4 f = *reinterpret_cast<const F4*>(rp + ix);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((unsigned)(ix + e) < (unsigned)W) v[e] = rp[ix + e];
      }
    }
    *reinterpret_cast<uint2*>(simg + cy * IP + 4 * xg) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  int poff[8];
  uint32_t pmask[4], wa[NB][4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    uint32_t m = 0u;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wa[nb][h] = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = 2 * h + u, k = 8 * kg + e;
      const bool valid = k < 27;
      const int tap = valid ? k / 3 : 0, c = valid ? k - tap * 3 : 0;
      const int kh = tap / 3, kw = tap - kh * 3;
      poff[e] = (c * IR + kh) * IP + kw;
      if (valid) {
        m |= 0xffffu << (16 * u);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wa[nb][h] |= (uint32_t)f32_to_bf16(w[k * COUT + nb * 16 + l15]) << (16 * u);
      }
    }
    pmask[h] = m;
  }
  f32x4_v bq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int v = 0; v < 4; ++v) bq[nb][v] = bias ? bias[nb * 16 + 4 * kg + v] : 0.f;
  __syncthreads();
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  for (int py = wave; py < TS; py += 4) {  // one tile row = 16 pixels per MFMA
    const bf16_t* pb = simg + (2 * py) * IP + 2 * l15;
    uint32_t xb[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
      xb[h] = ((uint32_t)pb[poff[2 * h]] | ((uint32_t)pb[poff[2 * h + 1]] << 16)) & pmask[h];
    const u32x4_t bv = {xb[0], xb[1], xb[2], xb[3]};
    const int oy = oy0 + py, ox = ox0 + l15;
    const bool inside = oy < OH && ox < OW;
    bf16_t* op = out + (((int64_t)b * OH + oy) * OW + ox) * COUT + 4 * kg;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const u32x4_t av = {wa[nb][0], wa[nb][1], wa[nb][2], wa[nb][3]};
      const f32x4_v acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_v, av), __builtin_bit_cast(bf16x8_v, bv), bq[nb], 0, 0, 0);
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      act_apply_n<4>(v, act);
      if (inside) *reinterpret_cast<uint2*>(op + nb * 16) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
}

// Round 6: the same stem conv as persistent workgroups (the EfficientViT stem's treatment, stem_dsconv_mfma_p_kernel): the per-lane MFMA
// operands are built once per workgroup, tiles are walked in XCD-contiguous ranges and the next tile's 33 x 33 x 3 image halo is requested
// into registers right after the barrier that publishes the current one.  Same arithmetic in the same order: bit-identical output.
template <int COUT>
__global__ __launch_bounds__(256, 3) void stem_mfma_p_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ out, int H, int W, int OH,
                                                            int OW, int tiles_x, int tiles_img, unsigned ntiles, int act) {
  constexpr int TS = 16, IR = 2 * TS + 1, IP = 36, NB = COUT / 16;
  constexpr int NIT = (3 * IR * (IP / 4) + 255) / 256;   // image items (plane, row, 4 columns) per thread: 4
  __shared__ __attribute__((aligned(16))) bf16_t simg[3 * IR * IP];
  struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
  const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const unsigned nx = nwg / 8 + (xcd < nwg % 8 ? 1u : 0u);
  const unsigned tq = ntiles / 8, tr = ntiles % 8;
  const unsigned first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1u : 0u);
  if (wi >= cnt) return;
  int it_pk[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int i = threadIdx.x + 256 * k;
    const int cy = i / (IP / 4), xg = i - cy * (IP / 4);
    const int c = cy / IR;
    it_pk[k] = i < 3 * IR * (IP / 4) ? (c << 16) | ((cy - c * IR) << 8) | xg : -1;
  }
  float pre[NIT][4];
  auto load_img = [&](unsigned tile) {
    const unsigned b = tile / (unsigned)tiles_img, ti = tile - b * (unsigned)tiles_img;
    const int ty = (int)(ti / (unsigned)tiles_x), tx = (int)(ti - ty * tiles_x);
    const int iy0 = 2 * (ty * TS) - 1, ix0 = 2 * (tx * TS) - 1;
    const float* ib = img + (int64_t)b * 3 * H * W;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int c = it_pk[k] >> 16, y = (it_pk[k] >> 8) & 255, xg = it_pk[k] & 255;
      const int iy = iy0 + y, ix = ix0 + 4 * xg;
      pre[k][0] = pre[k][1] = pre[k][2] = pre[k][3] = 0.f;
      if (it_pk[k] >= 0 && (unsigned)iy < (unsigned)H) {
        const float* rp = ib + (c * H + iy) * W;   // 3 H W < 2^31 (checked by the launcher)
        if (ix >= 0 && ix + 3 < W) {
          const F4 f = *reinterpret_cast<const F4*>(rp + ix);
          pre[k][0] = f.x; pre[k][1] = f.y; pre[k][2] = f.z; pre[k][3] = f.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((unsigned)(ix + e) < (unsigned)W) pre[k][e] = rp[ix + e];
        }
      }
    }
  };
  load_img(first + wi);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  int poff[8];
  uint32_t pmask[4], wa[NB][4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    uint32_t m = 0u;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) wa[nb][h] = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = 2 * h + u, k = 8 * kg + e;
      const bool valid = k < 27;
      const int tap = valid ? k / 3 : 0, c = valid ? k - tap * 3 : 0;
      const int kh = tap / 3, kw = tap - kh * 3;
      poff[e] = (c * IR + kh) * IP + kw;
      if (valid) {
        m |= 0xffffu << (16 * u);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) wa[nb][h] |= (uint32_t)f32_to_bf16(w[k * COUT + nb * 16 + l15]) << (16 * u);
      }
    }
    pmask[h] = m;
  }
  f32x4_v bq[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int v = 0; v < 4; ++v) bq[nb][v] = bias ? bias[nb * 16 + 4 * kg + v] : 0.f;
  typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
  for (unsigned t = wi; t < cnt; t += nx) {
    const unsigned tile = first + t;
    const unsigned b = tile / (unsigned)tiles_img, ti = tile - b * (unsigned)tiles_img;
    const int ty = (int)(ti / (unsigned)tiles_x), tx = (int)(ti - ty * tiles_x);
    const int oy0 = ty * TS, ox0 = tx * TS;
    if (t != wi) __syncthreads();   // every wave is done reading the previous tile's halo
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (it_pk[k] >= 0) {
        const int c = it_pk[k] >> 16, y = (it_pk[k] >> 8) & 255, xg = it_pk[k] & 255;
        *reinterpret_cast<uint2*>(simg + (c * IR + y) * IP + 4 * xg) = make_uint2(pack_bf16x2(pre[k][0], pre[k][1]), pack_bf16x2(pre[k][2], pre[k][3]));
      }
    __syncthreads();
    if (t + nx < cnt) load_img(tile + nx);
    for (int py = wave; py < TS; py += 4) {  // one tile row = 16 pixels per MFMA
      const bf16_t* pb = simg + (2 * py) * IP + 2 * l15;
      uint32_t xb[4];
#pragma unroll
      for (int h = 0; h < 4; ++h)
        xb[h] = ((uint32_t)pb[poff[2 * h]] | ((uint32_t)pb[poff[2 * h + 1]] << 16)) & pmask[h];
      const u32x4_t bv = {xb[0], xb[1], xb[2], xb[3]};
      const int oy = oy0 + py, ox = ox0 + l15;
      const bool inside = oy < OH && ox < OW;
      bf16_t* op = out + (((int64_t)b * OH + oy) * OW + ox) * COUT + 4 * kg;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const u32x4_t av = {wa[nb][0], wa[nb][1], wa[nb][2], wa[nb][3]};
        const f32x4_v acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_v, av), __builtin_bit_cast(bf16x8_v, bv), bq[nb], 0, 0, 0);
        float v[4] = {acc[0], acc[1], acc[2], acc[3]};
        act_apply_n<4>(v, act);
        if (inside) *reinterpret_cast<uint2*>(op + nb * 16) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// EfficientViT input stem in ONE kernel (efficientvit/backbone.py:48-70): ConvLayer 3 -> 16, 3x3 stride 2, BN, Hardswish,
// then ResidualBlock(DSConv 16 -> 16: depthwise 3x3 + BN + Hardswish, pointwise 1x1 + BN) + identity.
// Three launches (stem, depthwise, pointwise) moved the 504^2 x 16-channel tensor through HBM five times (0.88 ms of the
// B = 32 step); here a workgroup keeps an 18 x 18 halo of the stem output of its 16 x 16 pixels in LDS, so the image is
// read once and only the block's output is written.  Every intermediate is rounded to the activation dtype where the
// separate kernels stored it, so the fused path rounds exactly where the unfused one did.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void stem_dsconv_kernel(const float* __restrict__ img, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ wd,
                                                         const float* __restrict__ bd, const T* __restrict__ wp, int ldw,
                                                         const float* __restrict__ bp, T* __restrict__ out, int H, int W, int OH,
                                                         int OW, int tiles_x) {
  constexpr int C = 16, TS = 16, HS = TS + 2, IR = 2 * HS + 1, IP = 40;  // 37 input rows / columns per halo, row pitch 40
  __shared__ __attribute__((aligned(16))) float simg[3 * IR * IP];
  __shared__ __attribute__((aligned(16))) float sstem[HS * HS * C];
  __shared__ __attribute__((aligned(16))) float smid[TS * TS * C];
  __shared__ __attribute__((aligned(16))) float sw0[27 * C];
  __shared__ __attribute__((aligned(16))) float sb0[C];
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int b = blockIdx.y;
  const int oy0 = ty * TS, ox0 = tx * TS;
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;  // input pixel of simg[.][0][0]
  for (int i = threadIdx.x; i < 27 * C; i += 256) sw0[i] = w0[i];
  if (threadIdx.x < C) sb0[threadIdx.x] = b0 ? b0[threadIdx.x] : 0.f;
  for (int i = threadIdx.x; i < 3 * IR * IR; i += 256) {
    const int c = i / (IR * IR), r = i - c * IR * IR;
    const int y = r / IR, x = r - y * IR;
    const int iy = iy0 + y, ix = ix0 + x;
    float v = 0.f;
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = img[((int64_t)(b * 3 + c) * H + iy) * W + ix];
    simg[(c * IR + y) * IP + x] = v;
  }
  const int q4 = (threadIdx.x & 3) * 4;  // this thread's 4 channels in every phase
  float wdr[9][4], bdr[4], wpr[4][C], bpr[4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) wdr[t][e] = wd[t * C + q4 + e];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    bdr[e] = bd ? bd[q4 + e] : 0.f;
    bpr[e] = bp ? bp[q4 + e] : 0.f;
#pragma unroll
    for (int ci = 0; ci < C; ++ci) wpr[e][ci] = to_f32<T>(wp[(int64_t)(q4 + e) * ldw + ci]);
  }
  __syncthreads();
  // ---- phase 1: stem conv on the 18 x 18 halo; item = (2 horizontally adjacent pixels, 4 channels) -------------------
  for (int id = threadIdx.x; id < HS * (HS / 2) * 4; id += 256) {
    const int pair = id >> 2;
    const int hy = pair / (HS / 2), hx = (pair - hy * (HS / 2)) * 2;
    float acc[2][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[0][e] = acc[1][e] = sb0[q4 + e];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const float* row = simg + (c * IR + 2 * hy + kh) * IP + 2 * hx;
        float xin[5];
#pragma unroll
        for (int x = 0; x < 5; ++x) xin[x] = row[x];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float4 wv = *reinterpret_cast<const float4*>(sw0 + ((kh * 3 + kw) * 3 + c) * C + q4);
          const float wk[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[0][e] = fmaf(xin[kw], wk[e], acc[0][e]);
            acc[1][e] = fmaf(xin[kw + 2], wk[e], acc[1][e]);
          }
        }
      }
    const int sy = oy0 - 1 + hy;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int sx = ox0 - 1 + hx + u;
      act_apply_n<4>(acc[u], ACT_HSWISH);
      const bool inside = (unsigned)sy < (unsigned)OH && (unsigned)sx < (unsigned)OW;  // outside = the depthwise conv's zero padding
      float4 o;
      o.x = inside ? to_f32<T>(from_f32<T>(acc[u][0])) : 0.f;
      o.y = inside ? to_f32<T>(from_f32<T>(acc[u][1])) : 0.f;
      o.z = inside ? to_f32<T>(from_f32<T>(acc[u][2])) : 0.f;
      o.w = inside ? to_f32<T>(from_f32<T>(acc[u][3])) : 0.f;
      *reinterpret_cast<float4*>(sstem + (hy * HS + hx + u) * C + q4) = o;
    }
  }
  __syncthreads();
  // ---- phase 2: depthwise 3x3 + Hardswish; item = (pixel, 4 channels) -----------------------------------------------------
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int pix = (threadIdx.x >> 2) + 64 * r;
    const int py = pix >> 4, px = pix & 15;
    float acc[4] = {bdr[0], bdr[1], bdr[2], bdr[3]};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const float4 xv = *reinterpret_cast<const float4*>(sstem + ((py + kh) * HS + px + kw) * C + q4);
        acc[0] = fmaf(xv.x, wdr[kh * 3 + kw][0], acc[0]);
        acc[1] = fmaf(xv.y, wdr[kh * 3 + kw][1], acc[1]);
        acc[2] = fmaf(xv.z, wdr[kh * 3 + kw][2], acc[2]);
        acc[3] = fmaf(xv.w, wdr[kh * 3 + kw][3], acc[3]);
      }
    act_apply_n<4>(acc, ACT_HSWISH);
    *reinterpret_cast<float4*>(smid + pix * C + q4) = make_float4(to_f32<T>(from_f32<T>(acc[0])), to_f32<T>(from_f32<T>(acc[1])),
                                                                 to_f32<T>(from_f32<T>(acc[2])), to_f32<T>(from_f32<T>(acc[3])));
  }
  __syncthreads();
  // ---- phase 3: pointwise 16 -> 16 + identity; item = (pixel, 4 output channels) -------------------------------------------
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int pix = (threadIdx.x >> 2) + 64 * r;
    const int py = pix >> 4, px = pix & 15;
    const int oy = oy0 + py, ox = ox0 + px;
    float acc[4] = {bpr[0], bpr[1], bpr[2], bpr[3]};
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const float4 m = *reinterpret_cast<const float4*>(smid + pix * C + 4 * c4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[e] = fmaf(m.x, wpr[e][4 * c4], acc[e]);
        acc[e] = fmaf(m.y, wpr[e][4 * c4 + 1], acc[e]);
        acc[e] = fmaf(m.z, wpr[e][4 * c4 + 2], acc[e]);
        acc[e] = fmaf(m.w, wpr[e][4 * c4 + 3], acc[e]);
      }
    }
    const float4 idn = *reinterpret_cast<const float4*>(sstem + ((py + 1) * HS + px + 1) * C + q4);
    acc[0] += idn.x; acc[1] += idn.y; acc[2] += idn.z; acc[3] += idn.w;
    if (oy < OH && ox < OW) {
      T* op = out + (((int64_t)b * OH + oy) * OW + ox) * C + q4;
      if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<uint2*>(op) = make_uint2(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]));
      } else {
        *reinterpret_cast<float4*>(op) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      }
    }
  }
}

// bf16 variant of the fused input stem on the matrix cores.  The VALU kernel above issues ~1900 instructions per
// thread and tile (0.92 ms at B = 32: instruction-issue bound); here
//   phase 1  stem conv      = 16 channels x (27 -> 32 patch values) x 16 pixels per v_mfma_f32_16x16x32_bf16,
//   phase 2  depthwise 3x3  = diag(w) blocks on v_mfma_f32_4x4x4_16b_bf16 (see dwconv_mfma_kernel),
//   phase 3  pointwise      = 16 x 16 x 16 pixels per v_mfma_f32_16x16x16_bf16,
// and every phase leaves a lane with 4 consecutive channels of one pixel (one 8-byte LDS / global store).  The image is
// rounded to bf16 when it is staged, as the reference's autocast does to the conv input.
__global__ __launch_bounds__(256) void stem_dsconv_mfma_kernel(const float* __restrict__ img, const float* __restrict__ w0,
                                                              const float* __restrict__ b0, const float* __restrict__ wd,
                                                              const float* __restrict__ bd, const bf16_t* __restrict__ wp, int ldw,
                                                              const float* __restrict__ bp, bf16_t* __restrict__ out, int H, int W,
                                                              int OH, int OW, int tiles_x) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  constexpr int C = 16, TS = 16, HS = TS + 2, IR = 2 * HS + 1, IP = 40;  // 37 input rows / columns per halo, row pitch 40
  __shared__ __attribute__((aligned(16))) bf16_t simg[3 * IR * IP];
  __shared__ __attribute__((aligned(16))) bf16_t sstem[HS * HS * C];
  __shared__ __attribute__((aligned(16))) bf16_t smid[TS * TS * C];
  struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int b = blockIdx.y;
  const int oy0 = ty * TS, ox0 = tx * TS;
  const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;  // input pixel of simg[.][0][0]
  // ---- phase 0: image halo -> LDS as bf16; item = (plane, row, 4 columns) --------------------------------------------
  for (int i = threadIdx.x; i < 3 * IR * (IP / 4); i += 256) {
    const int cy = i / (IP / 4), xg = i - cy * (IP / 4);
    const int c = cy / IR, y = cy - c * IR;
    const int iy = iy0 + y, ix = ix0 + 4 * xg;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned)iy < (unsigned)H) {
      const float* rp = img + ((int64_t)(b * 3 + c) * H + iy) * W;
      if (ix >= 0 && ix + 3 < W) {
        const F4 f = *reinterpret_cast<const F4*>(rp + ix);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if ((unsigned)(ix + e) < (unsigned)W) v[e] = rp[ix + e];
      }
    }
    *reinterpret_cast<uint2*>(simg + cy * IP + 4 * xg) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  // phase-1 constants: patch offsets of this lane's 8 k values (k = tap*3 + c, 27 valid), weights A[co = l15][k]
  int poff[8];
  uint32_t pmask[4], wa[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    uint32_t m = 0u, wv = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = 2 * h + u, k = 8 * kg + e;
      const bool valid = k < 27;
      const int tap = valid ? k / 3 : 0, c = valid ? k - tap * 3 : 0;
      const int kh = tap / 3, kw = tap - kh * 3;
      poff[e] = (c * IR + kh) * IP + kw;
      if (valid) {
        m |= 0xffffu << (16 * u);
        wv |= (uint32_t)f32_to_bf16(w0[k * C + l15]) << (16 * u);
      }
    }
    pmask[h] = m;
    wa[h] = wv;
  }
  f32x4_v bias1, bias2, bias3;
  s16x4 wdg[9];
  const int cgp = (lane >> 2) & 3, pi = lane & 3;  // phase 2: channel group of the lane's block, row / pixel within the block
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    bias1[v] = b0 ? b0[4 * kg + v] : 0.f;
    bias2[v] = bd ? bd[4 * cgp + v] : 0.f;
    bias3[v] = bp ? bp[4 * kg + v] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const short wb = (short)f32_to_bf16(wd[t * C + 4 * cgp + pi]);
    wdg[t] = s16x4{(short)(pi == 0 ? wb : 0), (short)(pi == 1 ? wb : 0), (short)(pi == 2 ? wb : 0), (short)(pi == 3 ? wb : 0)};
  }
  const s16x4 wpa = *reinterpret_cast<const s16x4*>(wp + (int64_t)l15 * ldw + 4 * kg);  // A[co = l15][ci = 4 kg ..]
  __syncthreads();
  // ---- phase 1: stem conv over the 324 halo pixels, 16 per MFMA ---------------------------------------------------------
  for (int grp = wave; grp < (HS * HS + 15) / 16; grp += 4) {
    const int hp_raw = 16 * grp + l15;
    const int hp = hp_raw < HS * HS ? hp_raw : HS * HS - 1;
    const int hy = hp / HS, hx = hp - hy * HS;
    const bf16_t* pb = simg + (2 * hy) * IP + 2 * hx;
    uint32_t xb[4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
      xb[h] = ((uint32_t)pb[poff[2 * h]] | ((uint32_t)pb[poff[2 * h + 1]] << 16)) & pmask[h];
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t av = {wa[0], wa[1], wa[2], wa[3]}, bv = {xb[0], xb[1], xb[2], xb[3]};
    f32x4_v acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_v, av), __builtin_bit_cast(bf16x8_v, bv), bias1, 0, 0, 0);
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    act_apply_n<4>(v, ACT_HSWISH);
    const int sy = oy0 - 1 + hy, sx = ox0 - 1 + hx;
    const bool inside = (unsigned)sy < (unsigned)OH && (unsigned)sx < (unsigned)OW;  // outside = the depthwise conv's zero padding
    const uint2 o = inside ? make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])) : make_uint2(0u, 0u);
    if (hp_raw < HS * HS) *reinterpret_cast<uint2*>(sstem + hp * C + 4 * kg) = o;
  }
  __syncthreads();
  // ---- phase 2: depthwise 3x3 + Hardswish; one MFMA = one tap of 16 pixels (a tile row) x 16 channels ---------------------
  {
    const int px = 4 * (lane >> 4) + pi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int py = 4 * wave + r;
      f32x4_v acc = bias2;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const s16x4 xv = *reinterpret_cast<const s16x4*>(sstem + ((py + kh) * HS + px + kw) * C + 4 * cgp);
          acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wdg[kh * 3 + kw], xv, acc, 0, 0, 0);
        }
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      act_apply_n<4>(v, ACT_HSWISH);
      *reinterpret_cast<uint2*>(smid + (py * TS + px) * C + 4 * cgp) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
    }
  }
  __syncthreads();
  // ---- phase 3: pointwise 16 -> 16 + identity; one MFMA = a tile row of 16 pixels ---------------------------------------------
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int py = 4 * wave + r;
    const s16x4 mv = *reinterpret_cast<const s16x4*>(smid + (py * TS + l15) * C + 4 * kg);
    f32x4_v acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wpa, mv, bias3, 0, 0, 0);
    const uint2 idn = *reinterpret_cast<const uint2*>(sstem + ((py + 1) * HS + l15 + 1) * C + 4 * kg);
    const float o0 = acc[0] + __uint_as_float(idn.x << 16), o1 = acc[1] + __uint_as_float(idn.x & 0xffff0000u);
    const float o2 = acc[2] + __uint_as_float(idn.y << 16), o3 = acc[3] + __uint_as_float(idn.y & 0xffff0000u);
    const int oy = oy0 + py, ox = ox0 + l15;
    if (oy < OH && ox < OW)
      *reinterpret_cast<uint2*>(out + (((int64_t)b * OH + oy) * OW + ox) * C + 4 * kg) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
  }
}

// Round 6: the same kernel as PERSISTENT workgroups.  The one-tile-per-workgroup form lived 12.7 us per 16 x 16 tile (32 768 workgroups at
// B = 32, five resident per CU, 0.33 ms): a tile's life was the HBM latency of its image halo + ~40 scalar-ish global loads and ~300
// VALU instructions of per-lane operand building (patch offsets, stem / depthwise / pointwise weights as MFMA operands), all in
// front of 20 us-scale phases (profiles/r06/pmc_backbone_before.txt: 61 % of the wave time parked, matrix pipes busy 7 %).  Here a
// workgroup builds its operands once, walks the tiles of its XCD's contiguous range, and requests the NEXT tile's image halo (5
// 16-byte loads per thread, into registers) right after the barrier that publishes the current one -- the loads land under the three
// compute phases.  Same arithmetic, same order: bit-identical to stem_dsconv_mfma_kernel.
__global__ __launch_bounds__(256, 3) void stem_dsconv_mfma_p_kernel(const float* __restrict__ img, const float* __restrict__ w0,
                                                                const float* __restrict__ b0, const float* __restrict__ wd,
                                                                const float* __restrict__ bd, const bf16_t* __restrict__ wp, int ldw,
                                                                const float* __restrict__ bp, bf16_t* __restrict__ out, int H, int W,
                                                                int OH, int OW, int tiles_x, int tiles_img, unsigned ntiles) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  constexpr int C = 16, TS = 16, HS = TS + 2, IR = 2 * HS + 1, IP = 40;  // 37 input rows / columns per halo, row pitch 40
  constexpr int NIT = (3 * IR * (IP / 4) + 255) / 256;                   // image items (plane, row, 4 columns) per thread: 5
  __shared__ __attribute__((aligned(16))) bf16_t simg[3 * IR * IP];
  __shared__ __attribute__((aligned(16))) bf16_t sstem[HS * HS * C];
  __shared__ __attribute__((aligned(16))) bf16_t smid[TS * TS * C];
  struct __attribute__((packed, aligned(4))) F4 { float x, y, z, w; };
  // ---- this workgroup's tiles: XCD xcd owns the contiguous range [first, first + cnt), its workgroups stride through it ----
  const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const unsigned nx = nwg / 8 + (xcd < nwg % 8 ? 1u : 0u);
  const unsigned tq = ntiles / 8, tr = ntiles % 8;
  const unsigned first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1u : 0u);
  if (wi >= cnt) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, kg = lane >> 4;
  // this thread's image items: (plane c, halo row y, column group xg), fixed for all tiles; packed c << 16 | y << 8 | xg (one register each)
  int it_pk[NIT];
#pragma unroll
  for (int k = 0; k < NIT; ++k) {
    const int i = threadIdx.x + 256 * k;
    const int cy = i / (IP / 4), xg = i - cy * (IP / 4);
    const int c = cy / IR;
    it_pk[k] = i < 3 * IR * (IP / 4) ? (c << 16) | ((cy - c * IR) << 8) | xg : -1;
  }
  float pre[NIT][4];
  auto load_img = [&](unsigned tile) {
    const unsigned b = tile / (unsigned)tiles_img, ti = tile - b * (unsigned)tiles_img;
    const int ty = (int)(ti / (unsigned)tiles_x), tx = (int)(ti - ty * tiles_x);
    const int iy0 = 2 * (ty * TS) - 3, ix0 = 2 * (tx * TS) - 3;
    const float* ib = img + (int64_t)b * 3 * H * W;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int c = it_pk[k] >> 16, y = (it_pk[k] >> 8) & 255, xg = it_pk[k] & 255;
      const int iy = iy0 + y, ix = ix0 + 4 * xg;
      pre[k][0] = pre[k][1] = pre[k][2] = pre[k][3] = 0.f;
      if (it_pk[k] >= 0 && (unsigned)iy < (unsigned)H) {
        const float* rp = ib + (c * H + iy) * W;   // 3 H W < 2^31 (checked by the launcher)
        if (ix >= 0 && ix + 3 < W) {
          const F4 f = *reinterpret_cast<const F4*>(rp + ix);
          pre[k][0] = f.x; pre[k][1] = f.y; pre[k][2] = f.z; pre[k][3] = f.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if ((unsigned)(ix + e) < (unsigned)W) pre[k][e] = rp[ix + e];
        }
      }
    }
  };
  load_img(first + wi);
  // phase-1 constants: patch offsets of this lane's 8 k values (k = tap*3 + c, 27 valid), weights A[co = l15][k]
  int poff[8];
  uint32_t pmask[4], wa[4];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    uint32_t m = 0u, wv = 0u;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = 2 * h + u, k = 8 * kg + e;
      const bool valid = k < 27;
      const int tap = valid ? k / 3 : 0, c = valid ? k - tap * 3 : 0;
      const int kh = tap / 3, kw = tap - kh * 3;
      poff[e] = (c * IR + kh) * IP + kw;
      if (valid) {
        m |= 0xffffu << (16 * u);
        wv |= (uint32_t)f32_to_bf16(w0[k * C + l15]) << (16 * u);
      }
    }
    pmask[h] = m;
    wa[h] = wv;
  }
  f32x4_v bias1, bias2, bias3;
  s16x4 wdg[9];
  const int cgp = (lane >> 2) & 3, pi = lane & 3;  // phase 2: channel group of the lane's block, row / pixel within the block
#pragma unroll
  for (int v = 0; v < 4; ++v) {
    bias1[v] = b0 ? b0[4 * kg + v] : 0.f;
    bias2[v] = bd ? bd[4 * cgp + v] : 0.f;
    bias3[v] = bp ? bp[4 * kg + v] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const short wb = (short)f32_to_bf16(wd[t * C + 4 * cgp + pi]);
    wdg[t] = s16x4{(short)(pi == 0 ? wb : 0), (short)(pi == 1 ? wb : 0), (short)(pi == 2 ? wb : 0), (short)(pi == 3 ? wb : 0)};
  }
  const s16x4 wpa = *reinterpret_cast<const s16x4*>(wp + (int64_t)l15 * ldw + 4 * kg);  // A[co = l15][ci = 4 kg ..]

  for (unsigned t = wi; t < cnt; t += nx) {
    const unsigned tile = first + t;
    const unsigned b = tile / (unsigned)tiles_img, ti = tile - b * (unsigned)tiles_img;
    const int ty = (int)(ti / (unsigned)tiles_x), tx = (int)(ti - ty * tiles_x);
    const int oy0 = ty * TS, ox0 = tx * TS;
    // ---- phase 0: this tile's image halo (requested one tile ago) -> LDS as bf16 ----
#pragma unroll
    for (int k = 0; k < NIT; ++k)
      if (it_pk[k] >= 0) {
        const int c = it_pk[k] >> 16, y = (it_pk[k] >> 8) & 255, xg = it_pk[k] & 255;
        *reinterpret_cast<uint2*>(simg + (c * IR + y) * IP + 4 * xg) = make_uint2(pack_bf16x2(pre[k][0], pre[k][1]), pack_bf16x2(pre[k][2], pre[k][3]));
      }
    __syncthreads();   // also: every wave is past the previous tile's phase 3 (its reads of sstem / smid)
    if (t + nx < cnt) load_img(tile + nx);
    // ---- phase 1: stem conv over the 324 halo pixels, 16 per MFMA ----
    for (int grp = wave; grp < (HS * HS + 15) / 16; grp += 4) {
      const int hp_raw = 16 * grp + l15;
      const int hp = hp_raw < HS * HS ? hp_raw : HS * HS - 1;
      const int hy = hp / HS, hx = hp - hy * HS;
      const bf16_t* pb = simg + (2 * hy) * IP + 2 * hx;
      uint32_t xb[4];
#pragma unroll
      for (int h = 0; h < 4; ++h)
        xb[h] = ((uint32_t)pb[poff[2 * h]] | ((uint32_t)pb[poff[2 * h + 1]] << 16)) & pmask[h];
      typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t av = {wa[0], wa[1], wa[2], wa[3]}, bv = {xb[0], xb[1], xb[2], xb[3]};
      f32x4_v acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_v, av), __builtin_bit_cast(bf16x8_v, bv), bias1, 0, 0, 0);
      float v[4] = {acc[0], acc[1], acc[2], acc[3]};
      act_apply_n<4>(v, ACT_HSWISH);
      const int sy = oy0 - 1 + hy, sx = ox0 - 1 + hx;
      const bool inside = (unsigned)sy < (unsigned)OH && (unsigned)sx < (unsigned)OW;  // outside = the depthwise conv's zero padding
      const uint2 o = inside ? make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])) : make_uint2(0u, 0u);
      if (hp_raw < HS * HS) *reinterpret_cast<uint2*>(sstem + hp * C + 4 * kg) = o;
    }
    __syncthreads();
    // ---- phase 2: depthwise 3x3 + Hardswish; one MFMA = one tap of 16 pixels (a tile row) x 16 channels ----
    {
      const int px = 4 * (lane >> 4) + pi;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int py = 4 * wave + r;
        f32x4_v acc = bias2;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const s16x4 xv = *reinterpret_cast<const s16x4*>(sstem + ((py + kh) * HS + px + kw) * C + 4 * cgp);
            acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wdg[kh * 3 + kw], xv, acc, 0, 0, 0);
          }
        float v[4] = {acc[0], acc[1], acc[2], acc[3]};
        act_apply_n<4>(v, ACT_HSWISH);
        *reinterpret_cast<uint2*>(smid + (py * TS + px) * C + 4 * cgp) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    }
    __syncthreads();
    // ---- phase 3: pointwise 16 -> 16 + identity; one MFMA = a tile row of 16 pixels ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int py = 4 * wave + r;
      const s16x4 mv = *reinterpret_cast<const s16x4*>(smid + (py * TS + l15) * C + 4 * kg);
      f32x4_v acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wpa, mv, bias3, 0, 0, 0);
      const uint2 idn = *reinterpret_cast<const uint2*>(sstem + ((py + 1) * HS + l15 + 1) * C + 4 * kg);
      const float o0 = acc[0] + __uint_as_float(idn.x << 16), o1 = acc[1] + __uint_as_float(idn.x & 0xffff0000u);
      const float o2 = acc[2] + __uint_as_float(idn.y << 16), o3 = acc[3] + __uint_as_float(idn.y & 0xffff0000u);
      const int oy = oy0 + py, ox = ox0 + l15;
      if (oy < OH && ox < OW)
        *reinterpret_cast<uint2*>(out + (((int64_t)b * OH + oy) * OW + ox) * C + 4 * kg) = make_uint2(pack_bf16x2(o0, o1), pack_bf16x2(o2, o3));
    }
  }
}

// ------------------------------------------------------------------------------------
// depthwise k x k conv (k = 3 or 5), stride 1 or 2, pad k/2, + bias + activation
// (DSConv / MBConv depth_conv, ops.py:290-299,344-353; LiteMLA aggreg.0.0, ops.py:560-567).
// thread = (output pixel, 8-channel group); channel groups are the fastest index.
// ------------------------------------------------------------------------------------
// grid.y = (image, output row): no per-thread divisions by run-time 64-bit values; a thread
// produces NP horizontally adjacent outputs for 8 channels from a sliding window, so every input
// vector is loaded once per (row tap) instead of once per tap.
template <typename T, int KS, int STRIDE, int NP>
__global__ void dwconv_kernel(const T* __restrict__ in, int ld_in, const float* __restrict__ w,
                              const float* __restrict__ bias, T* __restrict__ out, int ld_out, int H,
                              int W, int C, int OH, int OW, int act, unsigned gx, unsigned gy) {
  constexpr int P = KS / 2;
  constexpr int WIN = (NP - 1) * STRIDE + KS;  // input columns touched by NP outputs
  const unsigned CG = (unsigned)C / VEC;
  unsigned bx, by;
  xcd_remap_2d(gx, gy, bx, by);
  const unsigned xi = bx * blockDim.x + threadIdx.x;
  const unsigned cg = xi % CG, pg = xi / CG;  // channel group, group of NP output pixels
  const int ow0 = (int)pg * NP;
  if (ow0 >= OW) return;
  const unsigned b = by / (unsigned)OH;
  const int oh = (int)(by - b * (unsigned)OH);
  const int c0 = (int)cg * VEC;
  float acc[NP][VEC];
#pragma unroll
  for (int j = 0; j < NP; ++j)
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[j][e] = bias ? bias[c0 + e] : 0.f;
  const int iw0 = ow0 * STRIDE - P;
#pragma unroll
  for (int kh = 0; kh < KS; ++kh) {
    const int ih = oh * STRIDE + kh - P;
    if ((unsigned)ih >= (unsigned)H) continue;
    const T* row = in + ((int64_t)(b * (unsigned)H + ih) * W) * ld_in + c0;
    float x[WIN][VEC];
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const int iw = iw0 + t;
      if ((unsigned)iw < (unsigned)W) {
        Vec8<T>::load(row + (int64_t)iw * ld_in, x[t]);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[t][e] = 0.f;
      }
    }
#pragma unroll
    for (int kw = 0; kw < KS; ++kw) {
      const float4 w0 = *reinterpret_cast<const float4*>(w + (kh * KS + kw) * C + c0);
      const float4 w1 = *reinterpret_cast<const float4*>(w + (kh * KS + kw) * C + c0 + 4);
      const float ww[VEC] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[j][e] = fmaf(x[j * STRIDE + kw][e], ww[e], acc[j][e]);
    }
  }
  T* orow = out + ((int64_t)(b * (unsigned)OH + oh) * OW) * ld_out + c0;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    if (ow0 + j < OW) {
      act_apply_n<VEC>(acc[j], act);
      Vec8<T>::store(orow + (int64_t)(ow0 + j) * ld_out, acc[j]);
    }
  }
}

// Depthwise k x k conv, stride 1, LDS-tiled: a workgroup stages the (8J + k - 1) x (16 + k - 1) input halo of a
// 64-channel block once (16-byte vectors, zero-filled outside the image) and every thread produces J runs of four
// horizontally adjacent outputs for 8 channels from it with a sliding window, so an input vector is fetched from
// global memory ~1.5-2 times instead of k times per output row, and the k*k weights come from LDS row by row.
// Workgroups are persistent: the halo of the NEXT tile is already on its way into registers while the current one is
// computed from LDS (a streaming kernel with separate load and compute phases keeps too few bytes in flight).
// LDS pixel pitch = 64 channels + 32 bytes: the four runs a 16-lane ds_read_b128 group touches (pixels 4 apart)
// then fall into four different 64-byte quarters of the 256-byte bank row.
template <typename T, int KS, int J>
__global__ __launch_bounds__(256, 2) void dwconv_tiled_kernel(const T* __restrict__ in, int ld_in, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ out, int ld_out,
                                                             int H, int W, int C, int act, int tiles_x, int tiles_y, int B) {
  constexpr int CB = 64, TW = 16, TH = 8 * J, HW_ = TW + KS - 1, HH = TH + KS - 1, P = KS / 2;
  constexpr int PITCH = CB + 32 / (int)sizeof(T);  // elements
  constexpr int NV = (HH * HW_ * 8 + 255) / 256;   // halo vectors (8 channels each) per thread
  typedef uint32_t raw_t __attribute__((ext_vector_type(2 * sizeof(T))));  // 8 elements of T
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  T* tile = reinterpret_cast<T*>(dsm);                                   // [HH*HW_][PITCH]
  float* sw = reinterpret_cast<float*>(dsm + (size_t)HH * HW_ * PITCH * sizeof(T));  // [KS*KS][CB] then bias [CB]
  const int tiles_img = tiles_x * tiles_y;
  const int per_cb = tiles_img * B;               // tile order: channel block slowest, then image, row, column
  const int total = per_cb * (C / CB);
  const int cg = threadIdx.x & 7, pt = threadIdx.x >> 3;

  struct Pos { int cb, b, oy0, ox0; };
  auto pos_of = [&](int t) {
    Pos p;
    p.cb = t / per_cb;
    int r = t - p.cb * per_cb;
    p.b = r / tiles_img;
    r -= p.b * tiles_img;
    const int ty = r / tiles_x;
    p.oy0 = ty * TH;
    p.ox0 = (r - ty * tiles_x) * TW;
    return p;
  };
  raw_t hv[NV];
  auto fetch = [&](const Pos& p) {  // this thread's halo vectors of tile p -> registers
    const T* img = in + (int64_t)p.b * H * W * ld_in + p.cb * CB;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = threadIdx.x + 256 * i;
      const int px = v >> 3, slot = v & 7;
      const int hy = px / HW_, hx = px - hy * HW_;
      const int iy = p.oy0 + hy - P, ix = p.ox0 + hx - P;
      raw_t r = {};
      if (v < HH * HW_ * 8 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
        r = *reinterpret_cast<const raw_t*>(img + ((int64_t)iy * W + ix) * ld_in + slot * 8);
      hv[i] = r;
    }
  };
  int t = blockIdx.x;
  if (t >= total) return;
  Pos cur = pos_of(t);
  fetch(cur);
  int cb_loaded = -1;
  for (; t < total; t += gridDim.x) {
    if (cur.cb != cb_loaded) {  // weights of this 64-channel block (the previous tile's reads ended at the loop's last barrier)
      for (int i = threadIdx.x; i < KS * KS * CB; i += 256) sw[i] = w[(i / CB) * C + cur.cb * CB + (i % CB)];
      if (threadIdx.x < CB) sw[KS * KS * CB + threadIdx.x] = bias ? bias[cur.cb * CB + threadIdx.x] : 0.f;
      cb_loaded = cur.cb;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = threadIdx.x + 256 * i;
      if (v < HH * HW_ * 8) *reinterpret_cast<raw_t*>(tile + (v >> 3) * PITCH + (v & 7) * 8) = hv[i];
    }
    __syncthreads();
    const Pos me = cur;
    if (t + (int)gridDim.x < total) {
      cur = pos_of(t + gridDim.x);
      fetch(cur);  // in flight during the arithmetic below
    }
    float acc[J][4][8];
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][q][e] = sw[KS * KS * CB + cg * 8 + e];
#pragma unroll 1  // a rolled loop: unrolled, the compiler hoists all k*k weight reads above it and spills them
    for (int ky = 0; ky < KS; ++ky) {
      float wr[KS][8];
#pragma unroll
      for (int kx = 0; kx < KS; ++kx) {
        const float4 a = *reinterpret_cast<const float4*>(sw + (ky * KS + kx) * CB + cg * 8);
        const float4 c = *reinterpret_cast<const float4*>(sw + (ky * KS + kx) * CB + cg * 8 + 4);
        wr[kx][0] = a.x; wr[kx][1] = a.y; wr[kx][2] = a.z; wr[kx][3] = a.w;
        wr[kx][4] = c.x; wr[kx][5] = c.y; wr[kx][6] = c.z; wr[kx][7] = c.w;
      }
#pragma unroll
      for (int j = 0; j < J; ++j) {
        const int run = pt + 32 * j;
        const int row = run >> 2, col0 = (run & 3) * 4;
        const T* lp = tile + ((row + ky) * HW_ + col0) * PITCH + cg * 8;
#pragma unroll
        for (int x = 0; x < 4 + KS - 1; ++x) {
          float xv[8];
          Vec8<T>::load(lp + x * PITCH, xv);
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const int q = x - kx;
            if (q >= 0 && q < 4) {
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[j][q][e] = fmaf(xv[e], wr[kx][e], acc[j][q][e]);
            }
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int run = pt + 32 * j;
      const int oy = me.oy0 + (run >> 2), oxb = me.ox0 + (run & 3) * 4;
      if (oy < H) {
        T* orow = out + ((int64_t)(me.b * H + oy) * W) * ld_out + me.cb * CB + cg * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (oxb + q < W) {
            act_apply_n<8>(acc[j][q], act);
            Vec8<T>::store(orow + (int64_t)(oxb + q) * ld_out, acc[j][q]);
          }
        }
      }
    }
    __syncthreads();  // everyone is done reading the tile (and the weights) before they are overwritten
  }
}

// Depthwise k x k conv, stride 1, bf16, on the matrix cores.  v_mfma_f32_4x4x4_16b_bf16 multiplies 16 independent
// 4x4x4 blocks per instruction (8 cycles): with block b = 4 consecutive channels, the first operand = diag(w[tap][4b..4b+3])
// and the second = those 4 channels of 4 horizontally adjacent pixels, one instruction accumulates one tap for
// 4 pixels x 64 channels -- 4x redundant multiplies instead of the 32x a 32x32 tile would waste, no bf16 -> f32
// unpacking, and k*k instructions per 256 outputs where the VALU kernels above issue ~30 per output element (they are
// VALU-issue-bound at ~2 TB/s).  The weights are rounded to bf16, as the reference's autocast does to its conv weights.
//   workgroup  256 threads = 4 waves, 8J x 16 outputs x 64 channels; halo staged in LDS as in dwconv_tiled_kernel
//   lane       (b = lane/4, p = lane%4): reads 8 bytes = channels 4b..4b+3 of pixel p of a run; its 4 accumulator
//              registers are those 4 channels of that pixel, so the result packs into one 8-byte store
//   LDS pitch  64 channels + 64 bytes: the 4 pixels x 8 blocks a 32-lane ds_read_b64 group touches tile the 256-byte
//              bank row exactly
template <int KS, int J>
__global__ __launch_bounds__(256) void dwconv_mfma_kernel(const bf16_t* __restrict__ in, int ld_in, const float* __restrict__ w,
                                                         const float* __restrict__ bias, bf16_t* __restrict__ out, int ld_out,
                                                         int H, int W, int C, int act, int tiles_x) {
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  constexpr int CB = 64, TW = 16, TH = 8 * J, HW_ = TW + KS - 1, HH = TH + KS - 1, P = KS / 2;
  constexpr int PITCH = CB + 32;  // elements (192 bytes)
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  bf16_t* tile = reinterpret_cast<bf16_t*>(dsm);  // [HH*HW_][PITCH]
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int c0 = blockIdx.y * CB;
  const int b = blockIdx.z;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const bf16_t* img = in + (int64_t)b * H * W * ld_in + c0;
  for (int v = threadIdx.x; v < HH * HW_ * 8; v += 256) {
    const int px = v >> 3, slot = v & 7;
    const int hy = px / HW_, hx = px - hy * HW_;
    const int iy = oy0 + hy - P, ix = ox0 + hx - P;
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
      r = *reinterpret_cast<const uint4*>(img + ((int64_t)iy * W + ix) * ld_in + slot * 8);
    *reinterpret_cast<uint4*>(tile + px * PITCH + slot * 8) = r;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = lane >> 2, pi = lane & 3;
  // diag(w[tap][c0 + 4 blk .. + 3]): this lane is row `pi` of its block, only element k = pi is non-zero
  s16x4 wd[KS * KS];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) {
    const short wb = (short)f32_to_bf16(w[t * C + c0 + 4 * blk + pi]);
    wd[t] = s16x4{(short)(pi == 0 ? wb : 0), (short)(pi == 1 ? wb : 0), (short)(pi == 2 ? wb : 0), (short)(pi == 3 ? wb : 0)};
  }
  f32x4 bs = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    const float4 bb = *reinterpret_cast<const float4*>(bias + c0 + 4 * blk);
    bs = f32x4{bb.x, bb.y, bb.z, bb.w};
  }
  __syncthreads();
#ifdef ESAM3_DW_DEV
  const int dev = act >> 8;  // 1: no arithmetic, 2: no stores
  act &= 255;
#endif
  constexpr int RPW = TH / 4;  // tile rows per wave
  f32x4 acc[RPW][4];
#pragma unroll
  for (int r = 0; r < RPW; ++r)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[r][q] = bs;
  const bf16_t* lbase = tile + (wave * RPW * HW_ + pi) * PITCH + blk * 4;
#ifdef ESAM3_DW_DEV
  if (!(dev & 1))
#endif
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const s16x4 xv = *reinterpret_cast<const s16x4*>(lbase + ((r + ky) * HW_ + q * 4 + kx) * PITCH);
          acc[r][q] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wd[ky * KS + kx], xv, acc[r][q], 0, 0, 0);
        }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int oy = oy0 + wave * RPW + r;
    if (oy >= H) continue;
#ifdef ESAM3_DW_DEV
    if ((dev & 2) && acc[r][0][0] != 12345.f) continue;
#endif
    bf16_t* orow = out + ((int64_t)(b * H + oy) * W) * ld_out + c0 + 4 * blk;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ox = ox0 + q * 4 + pi;
      if (ox < W) {
        float v[4] = {acc[r][q][0], acc[r][q][1], acc[r][q][2], acc[r][q][3]};
        act_apply_n<4>(v, act);
        *reinterpret_cast<uint2*>(orow + (int64_t)ox * ld_out) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    }
  }
}

// 3x3 depthwise conv, column-strip variant: a thread keeps the 9 x 8 weights of its channel group
// in registers and walks R output rows downwards with a rotating 3-row register window, so every
// input vector is loaded (R+2)/R times instead of 3 and the weights once per R*NP outputs.
template <typename T, int STRIDE, int NP, int R>
__global__ __launch_bounds__(256) void dwconv3_strip_kernel(const T* __restrict__ in, int ld_in,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ bias, T* __restrict__ out,
                                                            int ld_out, int H, int W, int C, int OH, int OW,
                                                            int act, unsigned gx, unsigned gy, int strips) {
  constexpr int KS = 3;
  constexpr int WIN = (NP - 1) * STRIDE + KS;
  const unsigned CG = (unsigned)C / VEC;
  unsigned bx, by;
  xcd_remap_2d(gx, gy, bx, by);
  const unsigned xi = bx * blockDim.x + threadIdx.x;
  const unsigned cg = xi % CG, pg = xi / CG;
  const int ow0 = (int)pg * NP;
  if (ow0 >= OW) return;
  const unsigned b = by / (unsigned)strips;
  const int oh0 = (int)(by - b * (unsigned)strips) * R;
  const int c0 = (int)cg * VEC;
  float wt[KS * KS][VEC], bs[VEC];
#pragma unroll
  for (int t = 0; t < KS * KS; ++t) {
    const float4 w0 = *reinterpret_cast<const float4*>(w + t * C + c0);
    const float4 w1 = *reinterpret_cast<const float4*>(w + t * C + c0 + 4);
    wt[t][0] = w0.x; wt[t][1] = w0.y; wt[t][2] = w0.z; wt[t][3] = w0.w;
    wt[t][4] = w1.x; wt[t][5] = w1.y; wt[t][6] = w1.z; wt[t][7] = w1.w;
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) bs[e] = bias ? bias[c0 + e] : 0.f;
  const int iw0 = ow0 * STRIDE - 1;
  const int ih0 = oh0 * STRIDE - 1;  // input row of window-relative row 0
  const T* img = in + (int64_t)b * H * W * ld_in + c0;
  float x[KS][WIN][VEC];
  auto load_row = [&](int rel) {  // window-relative input row -> slot rel % KS
    const int ih = ih0 + rel;
    const bool rok = (unsigned)ih < (unsigned)H;
    const T* row = img + (int64_t)ih * W * ld_in;
#pragma unroll
    for (int t = 0; t < WIN; ++t) {
      const int iw = iw0 + t;
      if (rok && (unsigned)iw < (unsigned)W) {
        Vec8<T>::load(row + (int64_t)iw * ld_in, x[rel % KS][t]);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) x[rel % KS][t][e] = 0.f;
      }
    }
  };
#pragma unroll
  for (int rel = 0; rel < KS - STRIDE; ++rel) load_row(rel);
  T* obase = out + (int64_t)b * OH * OW * ld_out + c0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (oh0 + r >= OH) break;
#pragma unroll
    for (int q = 0; q < STRIDE; ++q) load_row(r * STRIDE + KS - STRIDE + q);
    float acc[NP][VEC];
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[j][e] = bs[e];
#pragma unroll
    for (int kh = 0; kh < KS; ++kh)
#pragma unroll
      for (int kw = 0; kw < KS; ++kw)
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            acc[j][e] = fmaf(x[(r * STRIDE + kh) % KS][j * STRIDE + kw][e], wt[kh * KS + kw][e], acc[j][e]);
    T* orow = obase + (int64_t)(oh0 + r) * OW * ld_out;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      if (ow0 + j < OW) {
        act_apply_n<VEC>(acc[j], act);
        Vec8<T>::store(orow + (int64_t)(ow0 + j) * ld_out, acc[j]);
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// grouped 1x1 conv, gs(=8 or 16 or 32) channels per group in and out, no bias
// (LiteMLA aggreg.0.1, ops.py:568).  thread = (row, group, 8-output-channel chunk).
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void grouped_pw_kernel(const T* __restrict__ in, int ld_in, const float* __restrict__ w,
                                  T* __restrict__ out, int ld_out, int64_t rows, int C, int gs) {
  const int chunks = C / VEC;  // output chunks per row
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * chunks) return;
  const int ch = (int)(idx % chunks);
  const int64_t row = idx / chunks;
  const int co0 = ch * VEC;
  const int grp = co0 / gs;
  const T* x = in + row * ld_in + grp * gs;
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  for (int ci0 = 0; ci0 < gs; ci0 += VEC) {
    float xv[VEC];
    Vec8<T>::load(x + ci0, xv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float* wr = w + (int64_t)(co0 + e) * gs + ci0;  // w[cout][cin_in_group]
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc[e] = fmaf(xv[i], wr[i], acc[e]);
    }
  }
  Vec8<T>::store(out + row * ld_out + co0, acc);
}

// ------------------------------------------------------------------------------------
// LiteMLA ReLU linear attention (ops.py:584-621), two kernels:
//   kv[b][g][dv][dk] = sum_n v1[n][dv] * relu(k[n][dk]),  v1 = [v ; 1]   (dv in [0, dim])
//   out[n][d]       = (sum_dk kv[d][dk] relu(q[n][dk])) / (sum_dk kv[dim][dk] relu(q[n][dk]) + 1e-15)
// fp32 accumulation and fp32 division regardless of T (the reference keeps this island
// out of autocast, ops.py:586-589,616-618).
// ------------------------------------------------------------------------------------
template <typename T, int DIM>
__global__ void mla_kv_generic_kernel(const T* __restrict__ ms, int ld, float* __restrict__ kv, int N,
                              int groups, int n_split) {
  // block = (b, group, split); 256 threads
  constexpr int CH = 64;                 // positions per LDS chunk
  __shared__ float sk[CH][DIM + 1];
  __shared__ float sv[CH][DIM + 1];
  const int split = blockIdx.x % n_split;
  const int bg = blockIdx.x / n_split;
  const int g = bg % groups;
  const int64_t b = bg / groups;
  const T* base = ms + (b * N) * (int64_t)ld + g * 3 * DIM;
  constexpr int PAIRS = (DIM + 1) * DIM;
  constexpr int PPT = (PAIRS + 255) / 256;
  float acc[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) acc[i] = 0.f;
  const int per = (N + n_split - 1) / n_split;
  const int n_begin = split * per, n_end = min(N, n_begin + per);
  for (int nb = n_begin; nb < n_end; nb += CH) {
    for (int i = threadIdx.x; i < CH * DIM; i += 256) {
      const int r = i / DIM, d = i - r * DIM;
      const int n = nb + r;
      float kk = 0.f, vv = 0.f;
      if (n < n_end) {
        kk = to_f32<T>(base[(int64_t)n * ld + DIM + d]);
        vv = to_f32<T>(base[(int64_t)n * ld + 2 * DIM + d]);
        kk = kk > 0.f ? kk : 0.f;
      }
      sk[r][d] = kk;
      sv[r][d] = vv;
    }
    __syncthreads();
    const int cnt = min(CH, n_end - nb);
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int pr = threadIdx.x + 256 * i;
      if (pr < PAIRS) {
        const int dv = pr / DIM, dk = pr - dv * DIM;
        float a = acc[i];
        if (dv < DIM) {
          for (int r = 0; r < cnt; ++r) a = fmaf(sv[r][dv], sk[r][dk], a);
        } else {
          for (int r = 0; r < cnt; ++r) a += sk[r][dk];
        }
        acc[i] = a;
      }
    }
    __syncthreads();
  }
  float* o = kv + (b * groups + g) * (int64_t)PAIRS;
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int pr = threadIdx.x + 256 * i;
    if (pr < PAIRS) {
      if (n_split == 1) o[pr] = acc[i];
      else atomicAdd(o + pr, acc[i]);
    }
  }
}


template <typename T> __device__ inline void load4(const T* p, float* v);
template <> __device__ inline void load4<bf16_t>(const bf16_t* p, float* v) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(u.x << 16);
  v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16);
  v[3] = __uint_as_float(u.y & 0xffff0000u);
}
template <> __device__ inline void load4<float>(const float* p, float* v) {
  const float4 a = *reinterpret_cast<const float4*>(p);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
}

// kv reduction with every accumulator in registers: block = (image, token split), thread =
// (token lane, group, part) where a part owns 4 of the DIM v-rows of its group's (DIM+1) x DIM
// matrix.  A token's row of `ms` is read by groups*DIM/4 adjacent threads, i.e. fully coalesced.
// The reduction is DETERMINISTIC: token lanes are summed in lane order through LDS, the splits are
// written as partials and summed in split order by mla_kv_sum_kernel (no floating-point atomics,
// so repeated runs give bit-identical results).
template <typename T, int DIM>
__global__ __launch_bounds__(256) void mla_kv_kernel(const T* __restrict__ ms, int ld, float* __restrict__ partial,
                                                     int N, int groups, int n_split) {
  constexpr int P = DIM / 4;
  constexpr int PAIRS = (DIM + 1) * DIM;
  extern __shared__ float skv[];  // [TL][groups][PAIRS]
  const int tpt = groups * P, TL = 256 / tpt;
  const int GP = groups * PAIRS;
  const int split = blockIdx.x % n_split;
  const int64_t b = blockIdx.x / n_split;
  const int tl = threadIdx.x / tpt, r = threadIdx.x - tl * tpt;
  const int g = r / P, part = r - g * P;
  const int per = (N + n_split - 1) / n_split;
  const int n_begin = split * per, n_end = min(N, n_begin + per);
  if (tl < TL) {
    float acc[4][DIM], ksum[DIM];
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      ksum[j] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i][j] = 0.f;
    }
    const T* base = ms + (b * N) * (int64_t)ld + g * 3 * DIM;
#pragma unroll 2
    for (int n = n_begin + tl; n < n_end; n += TL) {
      const T* row = base + (int64_t)n * ld;
      float k[DIM], v[4];
#pragma unroll
      for (int d0 = 0; d0 < DIM; d0 += VEC) Vec8<T>::load(row + DIM + d0, k + d0);
      load4<T>(row + 2 * DIM + 4 * part, v);
#pragma unroll
      for (int j = 0; j < DIM; ++j) {
        k[j] = k[j] > 0.f ? k[j] : 0.f;
        ksum[j] += k[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = fmaf(v[i], k[j], acc[i][j]);
      }
    }
    float* o = skv + (size_t)tl * GP + g * PAIRS;  // every element of this lane's slice has one writer
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < DIM; ++j) o[(4 * part + i) * DIM + j] = acc[i][j];
    if (part == 0) {
#pragma unroll
      for (int j = 0; j < DIM; ++j) o[DIM * DIM + j] = ksum[j];
    }
  }
  __syncthreads();
  float* out = partial + (b * n_split + split) * (int64_t)GP;
  for (int i = threadIdx.x; i < GP; i += 256) {
    float a = skv[i];
    for (int t = 1; t < TL; ++t) a += skv[(size_t)t * GP + i];
    out[i] = a;
  }
}

__global__ void mla_kv_sum_kernel(const float* __restrict__ partial, float* __restrict__ kv, int GP, int n_split,
                                  int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, element)
  if (i >= total) return;
  const int64_t b = i / GP;
  const int e = (int)(i - b * GP);
  const float* p = partial + b * n_split * (int64_t)GP + e;
  float a = p[0];
  for (int sp = 1; sp < n_split; ++sp) a += p[(int64_t)sp * GP];
  kv[i] = a;
}

// out = (kv . relu(q)) / den.  kv sits in LDS as [pair/4][group][4] so that the 16-lane groups of
// a ds_read_b128 (lanes = consecutive groups) touch distinct banks; every thread serves
// `tokens_per_thread` tokens of its group to amortise the LDS fill.
template <typename T, int DIM>
__global__ __launch_bounds__(256) void mla_apply_kernel(const T* __restrict__ ms, int ld,
                                                        const float* __restrict__ kv, T* __restrict__ out,
                                                        int ld_out, int N, int groups, int tokens_per_thread) {
  extern __shared__ float skv[];
  constexpr int PAIRS = (DIM + 1) * DIM;
  const int64_t b = blockIdx.y;
  for (int i = threadIdx.x; i < groups * PAIRS; i += blockDim.x) {
    const int g = i / PAIRS, idx = i - g * PAIRS;
    skv[((idx >> 2) * groups + g) * 4 + (idx & 3)] = kv[b * groups * (int64_t)PAIRS + i];
  }
  __syncthreads();
  const int rows_per_block = blockDim.x / groups;
  const int rl = threadIdx.x / groups, g = threadIdx.x - rl * groups;
  const float4* kv4 = reinterpret_cast<const float4*>(skv) + g;
  for (int it = 0; it < tokens_per_thread; ++it) {
    const int n = (blockIdx.x * tokens_per_thread + it) * rows_per_block + rl;
    if (n >= N) return;
    const T* qp = ms + (b * N + n) * (int64_t)ld + g * 3 * DIM;
    float q[DIM];
#pragma unroll
    for (int d0 = 0; d0 < DIM; d0 += VEC) {
      Vec8<T>::load(qp + d0, q + d0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) q[d0 + e] = q[d0 + e] > 0.f ? q[d0 + e] : 0.f;
    }
    float den = 0.f;
#pragma unroll
    for (int c = 0; c < DIM / 4; ++c) {
      const float4 w = kv4[(DIM * DIM / 4 + c) * groups];
      den = fmaf(w.x, q[4 * c], den);
      den = fmaf(w.y, q[4 * c + 1], den);
      den = fmaf(w.z, q[4 * c + 2], den);
      den = fmaf(w.w, q[4 * c + 3], den);
    }
    const float inv = 1.f / (den + 1e-15f);
    T* op = out + (b * N + n) * (int64_t)ld_out + g * DIM;
#pragma unroll
    for (int d0 = 0; d0 < DIM; d0 += VEC) {
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < DIM / 4; ++c) {
          const float4 w = kv4[((d0 + e) * (DIM / 4) + c) * groups];
          a = fmaf(w.x, q[4 * c], a);
          a = fmaf(w.y, q[4 * c + 1], a);
          a = fmaf(w.z, q[4 * c + 2], a);
          a = fmaf(w.w, q[4 * c + 3], a);
        }
        o[e] = a * inv;
      }
      Vec8<T>::store(op + d0, o);
    }
  }
}

// ------------------------------------------------------------------------------------
// ViT-H teacher helpers (model/vitdet.py)
// ------------------------------------------------------------------------------------
// PatchEmbed (vitdet.py:312-337): Conv2d(3, D, k=P, s=P, bias=False) as a GEMM over patch rows.
// img NCHW fp32 -> A [B*G*G][ldk] with k = (c*P + ky)*P + kx (the conv weight's own order), the
// ldk - 3*P*P padding columns are zeroed.
template <typename T>
__global__ void patchify_kernel(const float* __restrict__ img, T* __restrict__ a, int B, int S, int P, int G, int ldk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * G * G * ldk;
  if (i >= total) return;
  const int k = (int)(i % ldk);
  const int64_t m = i / ldk;
  const int px = (int)(m % G), py = (int)((m / G) % G);
  const int64_t b = m / ((int64_t)G * G);
  float v = 0.f;
  if (k < 3 * P * P) {
    const int c = k / (P * P), r = k - c * P * P;
    const int ky = r / P, kx = r - ky * P;
    v = img[((b * 3 + c) * S + (py * P + ky)) * (int64_t)S + px * P + kx];
  }
  a[i] = from_f32<T>(v);
}

// 2-D axial RoPE on q and k in place (vitdet.py:41-90,421-457): qkv rows [3][heads][64]; the pair
// (x[2i], x[2i+1]) of every head is rotated by angle table[pos][i] (cos | sin, fp32), where pos is
// the token's index inside its ws x ws attention window (the whole map for global blocks).
template <typename T>
__global__ void vit_rope_kernel(T* __restrict__ qkv, const float* __restrict__ cs, int64_t rows, int H, int W, int ws,
                                int heads) {
  const int half = 32;  // complex pairs per 64-wide head
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, q|k, head, pair)
  const int64_t total = rows * 2 * heads * half;
  if (i >= total) return;
  const int pr = (int)(i % half);
  const int h = (int)((i / half) % heads);
  const int which = (int)((i / ((int64_t)half * heads)) % 2);
  const int64_t row = i / ((int64_t)half * heads * 2);
  const int x = (int)(row % W), y = (int)((row / W) % H);
  const int pos = (y % ws) * ws + (x % ws);
  const float c = cs[((int64_t)pos * half + pr) * 2], s_ = cs[((int64_t)pos * half + pr) * 2 + 1];
  T* p = qkv + row * 3 * (int64_t)heads * 64 + (int64_t)which * heads * 64 + h * 64 + 2 * pr;
  const float a = to_f32<T>(p[0]), b = to_f32<T>(p[1]);
  p[0] = from_f32<T>(a * c - b * s_);
  p[1] = from_f32<T>(a * s_ + b * c);
}

// Softmax attention over ws x ws windows of an [B][H][W] token map (ws = H = W: global), head dim HD.
// q/k/v live in one row-major buffer (row stride ld; q at q_off + h*HD, k at k_off + h*HD, ...).
// One query per thread with q and the output accumulator in registers; keys and values stream
// through LDS in chunks of KC (fp32, broadcast reads), online softmax in fp32.
template <typename T, int HD, int KC>
__global__ __launch_bounds__(256) void attn_window_kernel(const T* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                          T* __restrict__ out, int ldo, int H, int W, int ws, int heads,
                                                          float scale) {
  __shared__ float sk[KC][HD];
  __shared__ float sv[KC][HD];
  const int N = ws * ws;
  const int nwx = W / ws, nwy = H / ws;
  const int h = blockIdx.y;
  const int win = blockIdx.z % (nwx * nwy);
  const int64_t b = blockIdx.z / (nwx * nwy);
  const int wy = win / nwx, wx = win - wy * nwx;
  auto row_of = [&](int i) -> int64_t {  // token i of this window -> row of the token map
    const int y = wy * ws + i / ws, x = wx * ws + i % ws;
    return (b * H + y) * (int64_t)W + x;
  };
  const int qi = blockIdx.x * 256 + threadIdx.x;
  const bool valid = qi < N;
  float q[HD], acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { q[d] = 0.f; acc[d] = 0.f; }
  if (valid) {
    const T* src = qkv + row_of(qi) * ld + q_off + h * HD;
#pragma unroll
    for (int c = 0; c < HD / VEC; ++c) Vec8<T>::load(src + c * VEC, q + c * VEC);
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] *= scale;
  }
  float mx = -INFINITY, sum = 0.f;
  for (int j0 = 0; j0 < N; j0 += KC) {
    __syncthreads();
    for (int i = threadIdx.x; i < KC * (HD / VEC); i += 256) {
      const int j = i / (HD / VEC), c = i - j * (HD / VEC);
      float kk[VEC], vv[VEC];
      if (j0 + j < N) {
        const T* src = qkv + row_of(j0 + j) * ld + h * HD + c * VEC;
        Vec8<T>::load(src + k_off, kk);
        Vec8<T>::load(src + v_off, vv);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { kk[e] = 0.f; vv[e] = 0.f; }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) { sk[j][c * VEC + e] = kk[e]; sv[j][c * VEC + e] = vv[e]; }
    }
    __syncthreads();
    const int jn = min(KC, N - j0);
    for (int j = 0; j < jn; ++j) {
      const float4* kr = reinterpret_cast<const float4*>(sk[j]);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 4; c += 2) {
        const float4 a = kr[c], bq = kr[c + 1];
        s0 = fmaf(q[4 * c], a.x, s0); s0 = fmaf(q[4 * c + 1], a.y, s0);
        s0 = fmaf(q[4 * c + 2], a.z, s0); s0 = fmaf(q[4 * c + 3], a.w, s0);
        s1 = fmaf(q[4 * c + 4], bq.x, s1); s1 = fmaf(q[4 * c + 5], bq.y, s1);
        s1 = fmaf(q[4 * c + 6], bq.z, s1); s1 = fmaf(q[4 * c + 7], bq.w, s1);
      }
      const float sc = s0 + s1;
      if (sc > mx) {
        const float f = __expf(mx - sc);
        sum *= f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] *= f;
        mx = sc;
      }
      const float p = __expf(sc - mx);
      sum += p;
      const float4* vr = reinterpret_cast<const float4*>(sv[j]);
#pragma unroll
      for (int c = 0; c < HD / 4; ++c) {
        const float4 v4 = vr[c];
        acc[4 * c] = fmaf(p, v4.x, acc[4 * c]);
        acc[4 * c + 1] = fmaf(p, v4.y, acc[4 * c + 1]);
        acc[4 * c + 2] = fmaf(p, v4.z, acc[4 * c + 2]);
        acc[4 * c + 3] = fmaf(p, v4.w, acc[4 * c + 3]);
      }
    }
  }
  if (!valid) return;
  const float inv = 1.f / sum;
  T* dst = out + row_of(qi) * ldo + h * HD;
#pragma unroll
  for (int c = 0; c < HD / VEC; ++c) {
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = acc[c * VEC + e] * inv;
    Vec8<T>::store(dst + c * VEC, o);
  }
}

// MFMA flash attention for the bf16 engine, head dim 64 (ViT-H: 576-token windows and the 5184-token global blocks, with 2-D
// axial RoPE) or 32 (PCS fusion encoder: 8 heads x 32 over the 5184 image tokens, plain sequences), Nk a multiple of 64.
// Workgroup = 128 queries (4 wavefronts x 32) of one (image, window, head); keys / values stream through LDS in tiles of 64.
//   S^T[key][query] = K Q^T      v_mfma_f32_32x32x16_bf16 with A = K rows (from LDS), B = Q (registers)
//   online softmax per query = per lane column (lane & 31), fp32, base 2; the two half-waves hold different keys of the same
//                     query and exchange their maxima with one v_permlane32_swap
//   O^T[d][query]  += V^T P      A = V^T rows, B = P: the C layout of S^T is exactly the B-operand layout once the k slots of a
//                     16-key step are read as keys {4g..4g+3, 8+4g..8+4g+3}, so P never leaves its registers
// What the round-2 kernels (one per head dim) spent their time on, and what this one does instead:
//   * RoPE was applied to K while staging, i.e. once per 128-query block (4.5 x per window, 40 x per global map), with 32
//     bytes of cos/sin per 16 bytes of K: K is now rotated IN PLACE by one vectorised pass (vit_rope_k_kernel, 85 MB
//     read + written per block at B = 8) and only Q is rotated here, once per query;
//   * V was transposed in registers on its way into LDS (~100 VALU operations per staging thread and tile): V now goes
//     to LDS row-major with 16-byte writes, as HD / 16 [64 keys][16 d] sub-tiles, and the V^T fragments of O^T += V^T P
//     are read with ds_read_b64_tr_b16 -- a 16-lane group reads a [4 keys][16 d] block and lane i receives column
//     i, which is exactly the "4 consecutive keys of one channel" group the P layout asks for;
//   * staging was synchronous (load -> LDS -> barrier -> compute -> barrier): the next tile's 16-byte loads are now
//     issued BEFORE the tile's compute into registers and written to the OTHER LDS buffer after it: one barrier per
//     tile, global latency under the MFMA / softmax phase;
//   * 128 queries (4 wavefronts) per workgroup as before, but at <= 168 VGPRs so that three workgroups share a CU with
//     three wavefronts on EVERY SIMD (a first cut with 6-wave / 192-query workgroups -- both ViT-H token counts are
//     multiples of 192 -- left two SIMDs of four half empty: a second workgroup's 2 + 2 + 1 + 1 waves did not fit beside
//     the first; 1.72 ms on the global map against 1.30 ms for this shape; capping the kernel at 128 VGPRs for four waves
//     per SIMD spills the staged tile to scratch inside the loop: 1.83 ms; 3-wave / 96-query workgroups, which leave no
//     idle query slot in a 576-token window, measured the same there and 9 % slower on the global map);
//   * softmax: the scale is folded into one FMA per score (exp2(s * c - m * c)); the accumulator is rescaled only when
//     some lane's running maximum actually grew (wave-uniform branch; bit-identical to always rescaling).
// ViT-H at B = 8, per block: windows 0.35 -> 0.20 ms (+ 0.03 ms K rotation), global 2.35 -> 1.30 ms (680 TFLOP/s).
__global__ void vit_rope_k_kernel(bf16_t* __restrict__ qkv, int ld, int k_off, const float* __restrict__ rope, int64_t rows, int H,
                                  int W, int ws, int heads) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, head, 8-channel slot)
  if (idx >= rows * heads * 8) return;
  const int slot = (int)(idx & 7);
  const int64_t rh = idx >> 3;
  const int h = (int)(rh % heads);
  const int64_t row = rh / heads;
  const int x = (int)(row % W), y = (int)((row / W) % H);
  const int pos = (y % ws) * ws + (x % ws);
  bf16_t* p = qkv + row * ld + k_off + h * 64 + slot * 8;
  const float4 c0 = *reinterpret_cast<const float4*>(rope + ((int64_t)pos * 32 + slot * 4) * 2);
  const float4 c1 = *reinterpret_cast<const float4*>(rope + ((int64_t)pos * 32 + slot * 4) * 2 + 4);
  const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  u32x4 v = *reinterpret_cast<const u32x4*>(p), r;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = __uint_as_float(v[i] << 16), b_ = __uint_as_float(v[i] & 0xffff0000u);
    r[i] = pack_bf16x2(a * cs[2 * i] - b_ * cs[2 * i + 1], a * cs[2 * i + 1] + b_ * cs[2 * i]);
  }
  *reinterpret_cast<u32x4*>(p) = r;
}

// Tokens: image b holds RQ rows in q / out and RK rows in kv; a workgroup's window starts wrow = (wy ws) W + wx ws rows into
// the image and token i of the window is row (i / ws) W + i % ws of it.  Plain sequences are the window W = ws = Nk, H = 1.
typedef short s16x4_v __attribute__((ext_vector_type(4)));
template <int HD>
__global__ __launch_bounds__(256, 3) void attn_mfma_kernel(const bf16_t* __restrict__ q, int ldq, int q_off,
                                                           const bf16_t* __restrict__ kv, int ldk, int k_off, int v_off,
                                                           bf16_t* __restrict__ out, int ldo, int Nq, int Nk, int64_t RQ, int64_t RK,
                                                           int W, int ws, int nwx, int nwin, float scale_log2e,
                                                           const float* __restrict__ rope_q) {
  constexpr int KT = 64, NT = 256, QB = 128;
  constexpr int SL = HD / 8;            // 16-byte slots per K row
  constexpr int NSUB = HD / 16;         // V sub-tiles [64 keys][16 d]
  constexpr int DB = HD / 32;           // 32-channel blocks of O^T
  constexpr int CHUNKS = KT * SL;       // 16-byte chunks per K (and per V) tile: 512 / 256
  constexpr int CH = (CHUNKS + NT - 1) / NT;
  constexpr int KSTEP = NT / SL;        // chunk i of a thread is chunk 0 moved KSTEP keys on: same slot, same sub-tile
  constexpr int KROW = HD * 2;          // bytes per K row in LDS; 16-byte slots XOR-swizzled per row pair / quad
  constexpr int KBYTES = KT * KROW;
  constexpr int VS = 64 * 32 + 128;     // V sub-tile (32-byte rows) + 128: the two sub-tiles a half-wave reads together then
  constexpr int VBYTES = NSUB * VS;     //   sit on different halves of the 256-byte bank row
  static_assert(HD == 64 || HD == 32, "head dim");
  static_assert(CHUNKS % NT == 0, "every thread stages the same number of chunks");
  __shared__ __attribute__((aligned(16))) char sK[2][KBYTES];
  __shared__ __attribute__((aligned(16))) char sV[2][VBYTES];
  const int h = blockIdx.y;
  const int win = blockIdx.z % nwin;
  const int64_t b = blockIdx.z / nwin;
  const int wy = win / nwx, wx = win - wy * nwx;
  const int64_t wrow = (int64_t)(wy * ws) * W + wx * ws;
  const int64_t row0q = b * RQ + wrow, row0k = b * RK + wrow;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int qi = blockIdx.x * QB + wave * 32 + l31;
  const bool valid = qi < Nq;
  const int qc = valid ? qi : Nq - 1;
  const int64_t qrow = row0q + (qc / ws) * (int64_t)W + qc % ws;
  auto kswz = [](int key, int slot) -> int { return HD == 64 ? (slot ^ ((key >> 1) & 7)) : (slot ^ ((key >> 2) & 3)); };
  // Q fragments (B operand): 8 channels d = 16 s + 8 g .. +7 of this lane's query, rotated here (2-D axial RoPE,
  // vitdet.py:68-90: pairs (x[2i], x[2i+1]), fp32 arithmetic, rounded back to bf16 like the reference)
  u32x4 qf[HD / 16];
  {
    const bf16_t* src = q + qrow * ldq + q_off + h * HD;
#pragma unroll
    for (int s_ = 0; s_ < HD / 16; ++s_) {
      qf[s_] = *reinterpret_cast<const u32x4*>(src + s_ * 16 + g * 8);
      if (rope_q) {
        const float* cs = rope_q + ((int64_t)qc * (HD / 2) + s_ * 8 + g * 4) * 2;
        const float4 c0 = *reinterpret_cast<const float4*>(cs), c1 = *reinterpret_cast<const float4*>(cs + 4);
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = __uint_as_float(qf[s_][i] << 16), b_ = __uint_as_float(qf[s_][i] & 0xffff0000u);
          qf[s_][i] = pack_bf16x2(a * cc[2 * i] - b_ * cc[2 * i + 1], a * cc[2 * i + 1] + b_ * cc[2 * i]);
        }
      }
    }
  }
  // ---- staging roles: CHUNKS K chunks and CHUNKS V chunks of 16 bytes per tile, chunk ids tid + NT i --------------------
  // K: key = id / SL, slot = id % SL.  V: 8 consecutive lanes write 4 keys x 32 bytes of ONE sub-tile (a conflict-free
  // 128-byte run): id -> c0 = id & 1, sub-tile = (id >> 3) % NSUB, key = 4 (id / (8 NSUB)) + ((id >> 1) & 3).
  // The token of a chunk is tracked as a 32-bit ELEMENT offset from the window's first kv row (uniform base pointer + VGPR
  // offset: no 64-bit or quarter-rate integer arithmetic in the loop) plus its column x in the window, which decides the wrap.
  unsigned koff[CH], voff[CH];
  int kx_[CH], vx_[CH];
  unsigned kdst[CH];
  const int slot = tid % SL, kk0 = tid / SL;
  const int c0 = tid & 1, sub = (tid >> 3) % NSUB, vk0 = 4 * (tid / (8 * NSUB)) + ((tid >> 1) & 3);
  const bf16_t* __restrict__ kvb = kv + row0k * ldk + h * HD;   // wave-uniform
  const unsigned vdst0 = (unsigned)(sub * VS + vk0 * 32 + c0 * 16);  // chunk i: + 32 KSTEP i
  const int dq = KT / ws, dr = KT - dq * ws;  // a tile ahead = dq window rows and dr columns
  const unsigned dtile = (unsigned)((dq * W + dr) * ldk), dwrap = (unsigned)((W - ws) * ldk);
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int kk = kk0 + KSTEP * i, vk = vk0 + KSTEP * i;
    kx_[i] = kk % ws;
    vx_[i] = vk % ws;
    koff[i] = (unsigned)(((kk / ws) * W + kx_[i]) * ldk + k_off + slot * 8);
    voff[i] = (unsigned)(((vk / ws) * W + vx_[i]) * ldk + v_off + sub * 16 + c0 * 8);
    kdst[i] = (unsigned)(kk * KROW + (kswz(kk, slot) << 4));
  }
  u32x4 kreg[CH], vreg[CH];
  auto issue = [&]() {  // the loads of the tile the tokens currently point at; then advance them one tile
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      kreg[i] = *reinterpret_cast<const u32x4*>(kvb + koff[i]);
      vreg[i] = *reinterpret_cast<const u32x4*>(kvb + voff[i]);
      kx_[i] += dr; koff[i] += dtile;
      if (kx_[i] >= ws) { kx_[i] -= ws; koff[i] += dwrap; }
      vx_[i] += dr; voff[i] += dtile;
      if (vx_[i] >= ws) { vx_[i] -= ws; voff[i] += dwrap; }
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      *reinterpret_cast<u32x4*>(sK[buf] + kdst[i]) = kreg[i];
      *reinterpret_cast<u32x4*>(sV[buf] + vdst0 + 32 * KSTEP * i) = vreg[i];
    }
  };
  // V^T fragment addressing (transpose read): 16-lane group (l31 >> 4, g) reads keys k0 .. k0 + 3 of sub-tile
  // 2 db + (l31 >> 4); lane i of the group supplies the address of row k0 + (i >> 2), bytes 8 (i & 3) .. + 7
  const unsigned vfrag = (unsigned)((l31 >> 4) * VS + (4 * g + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);

  f32x16_v o[DB];
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;  // m: running maximum of the RAW scores; lsum: this half-wave's share of the denominator

  issue();
  commit(0);
  __syncthreads();
  const int ntiles = Nk / KT;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1;
    const bool more = t + 1 < ntiles;
    if (more) issue();
    // ---- S^T = K Q^T for the two 32-key blocks --------------------------------------------
    f32x16_v sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
      const int key = kb * 32 + l31;
#pragma unroll
      for (int s_ = 0; s_ < HD / 16; ++s_) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(sK[buf] + key * KROW + (kswz(key, s_ * 2 + g) << 4));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, kf),
                                                           __builtin_bit_cast(bf16x8_v, qf[s_]), sacc[kb], 0, 0, 0);
      }
    }
    // ---- online softmax (base 2, raw-score maximum) -------------------------------------------
    float mt = sacc[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mt = fmaxf(mt, sacc[kb][r]);
    {  // the other half-wave holds the other 32 keys of the same query: one v_permlane32_swap instead of a trip through LDS
      const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mt), __float_as_uint(mt), false, false);
      mt = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    }
    if (__any(mt > m)) {  // wave-uniform: some query of this wave has a new maximum
      const float mn = fmaxf(m, mt);
      const float alpha = __builtin_amdgcn_exp2f((m - mn) * scale_log2e);
      m = mn;
      lsum *= alpha;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
    }
    const float nm = -m * scale_log2e;
    u32x4 pf[2][2];  // P as B-operand fragments: [key block][16-key step]
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(fmaf(sacc[kb][r], scale_log2e, nm));
        lsum += pv[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        pf[kb][s2].x = pack_bf16x2(pv[8 * s2 + 0], pv[8 * s2 + 1]);
        pf[kb][s2].y = pack_bf16x2(pv[8 * s2 + 2], pv[8 * s2 + 3]);
        pf[kb][s2].z = pack_bf16x2(pv[8 * s2 + 4], pv[8 * s2 + 5]);
        pf[kb][s2].w = pack_bf16x2(pv[8 * s2 + 6], pv[8 * s2 + 7]);
      }
    }
    // ---- O^T += V^T P ---------------------------------------------------------------------------
    {
      typedef __attribute__((address_space(3))) s16x4_v* lds_v4;
      const auto vb = (__attribute__((address_space(3))) char*)sV[buf] + vfrag;
#pragma unroll
      for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const int off = db * 2 * VS + (kb * 32 + 16 * s2) * 32;  // keys 16 s2 + 4g .. +3, then + 8
            const s16x4_v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vb + off));
            const s16x4_v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vb + off + 8 * 32));
            const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
            const u32x4 vf = {l2.x, l2.y, h2.x, h2.y};
            o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, vf),
                                                            __builtin_bit_cast(bf16x8_v, pf[kb][s2]), o[db], 0, 0, 0);
          }
    }
    if (more) commit(buf ^ 1);
    __syncthreads();
  }
  {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(lsum), __float_as_uint(lsum), false, false);
    lsum = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
  }
  if (!valid) return;
  const float inv = 1.f / lsum;
  bf16_t* dst = out + qrow * ldo + h * HD;
#pragma unroll
  for (int db = 0; db < DB; ++db)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {  // registers 4 q4 .. 4 q4 + 3 are channels db*32 + 8 q4 + 4 g + {0..3}
      const uint2 w_ = make_uint2(pack_bf16x2(o[db][4 * q4] * inv, o[db][4 * q4 + 1] * inv),
                                  pack_bf16x2(o[db][4 * q4 + 2] * inv, o[db][4 * q4 + 3] * inv));
      *reinterpret_cast<uint2*>(dst + db * 32 + 8 * q4 + 4 * g) = w_;
    }
}

// Few queries x many keys (PCS decoder image cross-attention: 201 x 5184 with the box-relative bias; geometry
// CLS: 1 x 5184): one block = 32 queries of one (image, head); its four waves take the 64-key tiles round robin
// (each with its own K / V^T staging area), then the partial (m, l, O) are merged through LDS.
// key_mask [B][Nk] (1 = ignore) or null; bias_y [B][heads][Nq][Hk], bias_x [B][heads][Nq][Wk] (Wk % 4 == 0) or null.
__global__ __launch_bounds__(256) void attn_mfma32_splitk_kernel(const bf16_t* __restrict__ q, int ldq, int q_off,
                                                                 const bf16_t* __restrict__ kv, int ldk, int k_off, int v_off,
                                                                 bf16_t* __restrict__ out, int ldo, int Nq, int Nk, int heads,
                                                                 const uint8_t* __restrict__ key_mask,
                                                                 const float* __restrict__ bias_y,
                                                                 const float* __restrict__ bias_x, int Hk, int Wk, int bias_q0,
                                                                 float scale_log2e) {
  constexpr int HD = 32, KT = 64, VP = 136, NWV = 4;
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) char sK_all[NWV][KT * 64];
  __shared__ __attribute__((aligned(16))) char sVt_all[NWV][HD * VP];
  __shared__ float s_m[NWV][32], s_l[NWV][32];
  __shared__ float s_o[NWV][HD][33];
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  char* sK = sK_all[wave];
  char* sVt = sVt_all[wave];
  const int qi = blockIdx.x * 32 + l31;
  const bool valid = qi < Nq;
  const int qc = valid ? qi : Nq - 1;
  u32x4 qf[2];
  {
    const bf16_t* src = q + (b * Nq + qc) * (int64_t)ldq + q_off + h * HD;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) qf[s_] = *reinterpret_cast<const u32x4*>(src + s_ * 16 + g * 8);
  }
  const bool biased = bias_y != nullptr && qc >= bias_q0;
  const float* by = bias_y ? bias_y + ((b * heads + h) * Nq + qc) * (int64_t)Hk : nullptr;
  const float* bx = bias_x ? bias_x + ((b * heads + h) * Nq + qc) * (int64_t)Wk : nullptr;
  const uint8_t* km = key_mask ? key_mask + b * Nk : nullptr;
  f32x16_v o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const bf16_t* kbase = kv + b * Nk * (int64_t)ldk + h * HD;
  const int ntiles = (Nk + KT - 1) / KT;
  for (int t = wave; t < ntiles; t += NWV) {   // wave-private tiles: LDS traffic of one wave is in order
    const int j0 = t * KT;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i, key = idx >> 2, slot = idx & 3;
      const int kj = min(j0 + key, Nk - 1);
      const u32x4 v = *reinterpret_cast<const u32x4*>(kbase + (int64_t)kj * ldk + k_off + slot * 8);
      *reinterpret_cast<u32x4*>(sK + key * 64 + ((slot ^ ((key >> 2) & 3)) << 4)) = v;
    }
    {
      const int dch = lane & 3, kq = lane >> 2;
      u32x4 u[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        u[i] = *reinterpret_cast<const u32x4*>(kbase + (int64_t)min(j0 + kq * 4 + i, Nk - 1) * ldk + v_off + dch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int w_ = e >> 1;
        uint32_t a0, a1;
        if (e & 1) {
          a0 = (u[0][w_] >> 16) | (u[1][w_] & 0xffff0000u);
          a1 = (u[2][w_] >> 16) | (u[3][w_] & 0xffff0000u);
        } else {
          a0 = (u[0][w_] & 0xffffu) | (u[1][w_] << 16);
          a1 = (u[2][w_] & 0xffffu) | (u[3][w_] << 16);
        }
        *reinterpret_cast<uint2*>(sVt + (dch * 8 + e) * VP + kq * 8) = make_uint2(a0, a1);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    f32x16_v sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
      const int key = kb * 32 + l31;
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(sK + key * 64 + (((s_ * 2 + g) ^ ((key >> 2) & 3)) << 4));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, kf),
                                                           __builtin_bit_cast(bf16x8_v, qf[s_]), sacc[kb], 0, 0, 0);
      }
    }
    const bool edge = km != nullptr || j0 + KT > Nk;
    float mt = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int jg = j0 + kb * 32 + 8 * q4 + 4 * g;  // keys jg .. jg+3 = accumulator rows 4*q4 .. 4*q4+3
        float add[4] = {0.f, 0.f, 0.f, 0.f};
        if (biased) {
          const int jc = min(jg, Nk - 4);
          const int ky = jc / Wk, kx = jc - ky * Wk;
          const float yv = by[ky];
          const float4 xv = *reinterpret_cast<const float4*>(bx + kx);
          add[0] = (yv + xv.x) * LOG2E; add[1] = (yv + xv.y) * LOG2E;
          add[2] = (yv + xv.z) * LOG2E; add[3] = (yv + xv.w) * LOG2E;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float sc = fmaf(sacc[kb][4 * q4 + e], scale_log2e, add[e]);
          if (edge) {
            const int j = jg + e;
            if (j >= Nk || (km && km[j])) sc = -INFINITY;
          }
          sacc[kb][4 * q4 + e] = sc;
          mt = fmaxf(mt, sc);
        }
      }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float mref = mn == -INFINITY ? 0.f : mn;   // a fully masked prefix keeps everything at zero
    const float alpha = __builtin_amdgcn_exp2f(m - mref);
    m = mn;
    lsum *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
    u32x4 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(sacc[kb][r] - mref);
        lsum += pv[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        pf[kb][s2].x = pack_bf16x2(pv[8 * s2 + 0], pv[8 * s2 + 1]);
        pf[kb][s2].y = pack_bf16x2(pv[8 * s2 + 2], pv[8 * s2 + 3]);
        pf[kb][s2].z = pack_bf16x2(pv[8 * s2 + 4], pv[8 * s2 + 5]);
        pf[kb][s2].w = pack_bf16x2(pv[8 * s2 + 6], pv[8 * s2 + 7]);
      }
    }
    const char* vrow = sVt + l31 * VP;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const int koff = (kb * 32 + 16 * s2 + 4 * g) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow + koff);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + koff + 16);
        const u32x4 vf = {lo.x, lo.y, hi.x, hi.y};
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, vf),
                                                    __builtin_bit_cast(bf16x8_v, pf[kb][s2]), o, 0, 0, 0);
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  lsum += __shfl_xor(lsum, 32, 64);
  if (g == 0) { s_m[wave][l31] = m; s_l[wave][l31] = lsum; }
#pragma unroll
  for (int r = 0; r < 16; ++r) s_o[wave][(r & 3) + 8 * (r >> 2) + 4 * g][l31] = o[r];
  __syncthreads();
  {  // merge: thread -> (query, 4 channels)
    const int qq = tid & 31, dg = tid >> 5;
    const int qo_ = blockIdx.x * 32 + qq;
    if (qo_ >= Nq) return;
    float M = -INFINITY;
#pragma unroll
    for (int w_ = 0; w_ < NWV; ++w_) M = fmaxf(M, s_m[w_][qq]);
    float L = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w_ = 0; w_ < NWV; ++w_) {
      const float mw = s_m[w_][qq];
      const float f = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - M);
      L = fmaf(s_l[w_][qq], f, L);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = fmaf(s_o[w_][dg * 4 + e][qq], f, acc[e]);
    }
    const float inv = 1.f / L;
    bf16_t* dst = out + (b * Nq + qo_) * (int64_t)ldo + h * HD + dg * 4;
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16x2(acc[0] * inv, acc[1] * inv), pack_bf16x2(acc[2] * inv, acc[3] * inv));
  }
}

// The same scheme for TinyViT's windows (bf16, head dim 32; tiny_vit.py:265-293,339-372): WS x WS windows over
// a map that is zero-padded to a multiple of WS before the attention's LayerNorm (padded positions
// carry the constant `pad_qkv`), additive bias[h][|dy|*WS+|dx|], qkv rows [heads][q32|k32|v32].
//
// One workgroup per (window, head); all keys of the window are staged once.  The softmax side of this kernel costs
// several times its MFMA time, so the layout is chosen to make the per-score work small: tokens sit in SLOTS
// slot = y * WP + x with the window row padded to WP = 16 (WS = 14) or 8 (WS = 7) columns -- 224 / 56 slots, i.e.
// exactly 7 / 2 MFMA tiles of 32 -- so that in the 32x32 accumulator layout (key = 8 (r >> 2) + (r & 3) + 4 g) a
// register's key ROW is a compile-time constant and its key COLUMN takes 8 (4) values per lane.  The relative-position
// bias of a score is then one LDS read at rowoff(|qy - ky|) + dxoff[c]: one v_sad_u32 per key row, one v_add per score;
// the surplus slots (x >= WS) read a column of -inf.  (The previous version kept the reference's y * WS + x order and
// spent ~15 integer operations per score on divisions by 14; it also padded 196 keys to 256.)
template <int WS, int NW>
__global__ __launch_bounds__(NW * 64) void attn_mfma32_win_kernel(const bf16_t* __restrict__ qkv, int ld,
                                                                  const bf16_t* __restrict__ pad_qkv,
                                                                  const float* __restrict__ bias, bf16_t* __restrict__ out,
                                                                  int ldo, int H, int W, int heads, int nwx, int nwy) {
  constexpr int HD = 32, WP = WS > 8 ? 16 : 8, NS = WS * WP, NKT = (NS + 31) / 32, NSP = NKT * 32, NT = NW * 64;
  constexpr int VP = NSP * 2 + 8;  // bytes per V^T row
  constexpr int NC = WP / 2;       // distinct key columns per lane
  static_assert(NW * 32 == NSP, "one wave per 32 query slots");
  constexpr float LOG2E = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) char sK[NSP * 64];   // [slot][32 d] bf16, 64-byte rows, chunks ^ (slot>>2)&3
  __shared__ __attribute__((aligned(16))) char sVt[HD * VP];  // [d][slot]
  __shared__ float sb[WS * WP];                                // [|dy|][|dx|] of this head x log2(e); columns >= WS: -inf
  const int h = blockIdx.y;
  const int win = blockIdx.x % (nwx * nwy);
  const int64_t b = blockIdx.x / (nwx * nwy);
  const int wy = win / nwx, wx = win - wy * nwx;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  // token (yy, xx) of the window -> its qkv row for head h (the constant row for zero-padded positions)
  auto tok = [&](int yy, int xx) -> const bf16_t* {
    const int y = wy * WS + yy, x = wx * WS + xx;
    return (y < H && x < W) ? qkv + ((b * H + y) * (int64_t)W + x) * ld + h * 96 : pad_qkv + h * 96;
  };
  for (int i = tid; i < WS * WP; i += NT) {
    const int dy = i / WP, dx = i % WP;
    sb[i] = dx < WS ? bias[h * (WS * WS) + dy * WS + dx] * LOG2E : -INFINITY;
  }
  for (int c = tid; c < NSP * 4; c += NT) {  // K: 4 chunks of 16 bytes per slot
    const int key = c >> 2, ch = c & 3;
    const int ky = key / WP, kx = key % WP;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (ky < WS && kx < WS) v = *reinterpret_cast<const u32x4*>(tok(ky, kx) + 32 + ch * 8);
    *reinterpret_cast<u32x4*>(sK + key * 64 + ((ch ^ ((key >> 2) & 3)) << 4)) = v;
  }
  for (int c = tid; c < NSP; c += NT) {  // V^T: (8 d) x (4 slots) patches; the 4 slots share a window row
    const int dch = c & 3, kq = c >> 2;
    const int ky = (kq * 4) / WP, kx0 = (kq * 4) % WP;
    u32x4 u[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      u[i] = (ky < WS && kx0 + i < WS) ? *reinterpret_cast<const u32x4*>(tok(ky, kx0 + i) + 64 + dch * 8) : z;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int w_ = e >> 1;
      uint32_t a0, a1;
      if (e & 1) {
        a0 = (u[0][w_] >> 16) | (u[1][w_] & 0xffff0000u);
        a1 = (u[2][w_] >> 16) | (u[3][w_] & 0xffff0000u);
      } else {
        a0 = (u[0][w_] & 0xffffu) | (u[1][w_] << 16);
        a1 = (u[2][w_] & 0xffffu) | (u[3][w_] << 16);
      }
      *reinterpret_cast<uint2*>(sVt + (dch * 8 + e) * VP + kq * 8) = make_uint2(a0, a1);
    }
  }
  // this lane's query slot
  const int qi = wave * 32 + l31;
  const int qy = qi / WP, qx = qi % WP;
  const bool qin = qy < WS && qx < WS;
  const bool valid = qin && wy * WS + qy < H && wx * WS + qx < W;
  const int qyc = qin ? qy : 0, qxc = qin ? qx : 0;
  u32x4 qf[2];
  {
    const bf16_t* src = tok(qyc, qxc);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) qf[s_] = *reinterpret_cast<const u32x4*>(src + s_ * 16 + g * 8);
  }
  // bias addressing: byte offset of this lane's |dx| column for each of its NC key columns, and qy scaled to a row pitch
  int dxoff[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int kx = (WP == 16 ? (c & 3) + 8 * (c >> 2) : c) + 4 * g;
    const int dx = qxc > kx ? qxc - kx : kx - qxc;
    dxoff[c] = (kx < WS ? dx : WS) * 4;
  }
  const unsigned qrow = (unsigned)qyc * (WP * 4);
  const char* sbb = reinterpret_cast<const char*>(sb);

  f32x16_v o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const float scale_log2e = 0.17677669529663687f * LOG2E;  // 32^-0.5 * log2(e)
  __syncthreads();

#pragma unroll
  for (int j0 = 0; j0 < NSP; j0 += 64) {
    const int nkb = j0 + 32 < NSP ? 2 : 1;
    f32x16_v sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb >= nkb) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
      const int key = j0 + kb * 32 + l31;
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(sK + key * 64 + (((s_ * 2 + g) ^ ((key >> 2) & 3)) << 4));
        sacc[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, kf),
                                                           __builtin_bit_cast(bf16x8_v, qf[s_]), sacc[kb], 0, 0, 0);
      }
    }
    float mt = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb >= nkb) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int slot = j0 + kb * 32 + (r & 3) + 8 * (r >> 2);  // + 4 g: same window row
        const int ky = slot / WP;
        float sc = -INFINITY;
        if (ky < WS) {  // compile time
          const int c = WP == 16 ? (r & 3) + 4 * ((r >> 2) & 1) : (r & 3);
          const unsigned rowoff = __usad(qrow, (unsigned)(ky * WP * 4), 0u);
          sc = fmaf(sacc[kb][r], scale_log2e, *reinterpret_cast<const float*>(sbb + rowoff + dxoff[c]));
        }
        sacc[kb][r] = sc;
        mt = fmaxf(mt, sc);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    lsum *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] *= alpha;
    const char* vrow = sVt + l31 * VP;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      if (kb >= nkb) break;
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv[r] = __builtin_amdgcn_exp2f(sacc[kb][r] - mn);
        lsum += pv[r];
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        u32x4 pf;
        pf.x = pack_bf16x2(pv[8 * s2 + 0], pv[8 * s2 + 1]);
        pf.y = pack_bf16x2(pv[8 * s2 + 2], pv[8 * s2 + 3]);
        pf.z = pack_bf16x2(pv[8 * s2 + 4], pv[8 * s2 + 5]);
        pf.w = pack_bf16x2(pv[8 * s2 + 6], pv[8 * s2 + 7]);
        const int koff = (j0 + kb * 32 + 16 * s2 + 4 * g) * 2;
        const uint2 lo = *reinterpret_cast<const uint2*>(vrow + koff);
        const uint2 hi = *reinterpret_cast<const uint2*>(vrow + koff + 16);
        const u32x4 vf = {lo.x, lo.y, hi.x, hi.y};
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, vf), __builtin_bit_cast(bf16x8_v, pf), o,
                                                    0, 0, 0);
      }
    }
  }
  lsum += __shfl_xor(lsum, 32, 64);
  if (!valid) return;
  const float inv = 1.f / lsum;
  bf16_t* dst = out + ((b * H + wy * WS + qy) * (int64_t)W + wx * WS + qx) * ldo + h * HD;
#pragma unroll
  for (int q4 = 0; q4 < 4; ++q4) {
    const uint2 w_ = make_uint2(pack_bf16x2(o[4 * q4] * inv, o[4 * q4 + 1] * inv),
                                pack_bf16x2(o[4 * q4 + 2] * inv, o[4 * q4 + 3] * inv));
    *reinterpret_cast<uint2*>(dst + 8 * q4 + 4 * g) = w_;
  }
}

// ------------------------------------------------------------------------------------
// Squeeze-Excite (timm SqueezeExcite as used by RepViT, repvit.py:136,150):
//   gate[b][c] = sigmoid(W2 . relu(W1 . mean_hw(x[b]) + b1) + b2);  x *= gate
// three small kernels: per-channel sums (atomics over pixel splits), the two tiny FCs (one
// workgroup per image), and the in-place scaling.  fp32 statistics regardless of T.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void se_pool_kernel(const T* __restrict__ x, int ld, float* __restrict__ partial,
                                                      int HW, int C, int splits) {
  extern __shared__ float red[];  // [lanes][C]; deterministic: lanes are summed in order, no atomics
  const int CG = C / VEC;
  const int b = blockIdx.x / splits, sp = blockIdx.x - b * splits;
  const int lanes = 256 / CG;  // pixel lanes (launcher guarantees CG <= 256)
  const int pl = threadIdx.x / CG, cg = threadIdx.x - pl * CG;
  if (pl < lanes) {
    const int per = (HW + splits - 1) / splits;
    const int p0 = sp * per, p1 = min(HW, p0 + per);
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int p = p0 + pl; p < p1; p += lanes) {
      float v[VEC];
      Vec8<T>::load(x + ((int64_t)b * HW + p) * ld + cg * VEC, v);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[pl * C + cg * VEC + e] = acc[e];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C; i += 256) {
    float a = red[i];
    for (int l = 1; l < lanes; ++l) a += red[l * C + i];
    partial[((int64_t)b * splits + sp) * C + i] = a;
  }
}

__global__ __launch_bounds__(256) void se_fc_kernel(const float* __restrict__ partial, int splits,
                                                    const float* __restrict__ w1, const float* __restrict__ b1,
                                                    const float* __restrict__ w2, const float* __restrict__ b2,
                                                    float* __restrict__ gate, int C, int R, float inv_hw) {
  extern __shared__ float sm[];  // mean[C] | hidden[R]
  float* mean = sm;
  float* hid = sm + C;
  const int b = blockIdx.x;
  for (int i = threadIdx.x; i < C; i += 256) {
    float a = 0.f;
    for (int sp = 0; sp < splits; ++sp) a += partial[((int64_t)b * splits + sp) * C + i];
    mean[i] = a * inv_hw;
  }
  __syncthreads();
  for (int r = threadIdx.x; r < R; r += 256) {
    float a = b1[r];
    for (int c = 0; c < C; ++c) a = fmaf(w1[(int64_t)r * C + c], mean[c], a);
    hid[r] = a > 0.f ? a : 0.f;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = b2[c];
    for (int r = 0; r < R; ++r) a = fmaf(w2[(int64_t)c * R + r], hid[r], a);
    gate[(int64_t)b * C + c] = 1.f / (1.f + expf(-a));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void se_scale_kernel(T* __restrict__ x, int ld, const float* __restrict__ gate,
                                                       int HW, int C, int64_t total) {
  const int CG = C / VEC;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (b, pixel, channel group)
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const int64_t bp = i / CG;
  const int64_t b = bp / HW;
  float v[VEC];
  T* px = x + bp * ld + cg * VEC;
  Vec8<T>::load(px, v);
  const float* g = gate + b * C + cg * VEC;
#pragma unroll
  for (int e = 0; e < VEC; ++e) v[e] *= g[e];
  Vec8<T>::store(px, v);
}

// ------------------------------------------------------------------------------------
// TinyViT window attention (tiny_vit.py:265-293,339-372): tokens [B][H][W][heads*(q32|k32|v32)]
// are partitioned into WS x WS windows (the map is zero-padded up to a multiple of WS BEFORE the
// attention's LayerNorm, so a padded position carries the constant qkv(LayerNorm(0)) = `pad_qkv`
// and takes part as a key/value); out = softmax(q k^T / sqrt(32) + bias[h][|dy|*WS+|dx|]) v.
// One (window, head) pair per thread group: K and V of the pair sit in LDS as fp32, every thread
// owns QB queries and runs an online softmax over the N = WS*WS keys (LDS reads are broadcasts).
// ------------------------------------------------------------------------------------
template <typename T, int WS, int QB, int PAIRS>
__global__ __launch_bounds__(PAIRS * ((WS * WS + QB - 1) / QB)) void window_attn_kernel(
    const T* __restrict__ qkv, int ld, const T* __restrict__ pad_qkv, const float* __restrict__ bias,
    T* __restrict__ out, int ldo, int H, int W, int heads, int nwx, int nwy, int total_pairs) {
  constexpr int N = WS * WS, HD = 32, TPP = (N + QB - 1) / QB;
  constexpr int PAIR_FLOATS = 2 * N * HD + N;
  extern __shared__ float smem[];
  const int pl = threadIdx.x / TPP, tl = threadIdx.x - pl * TPP;
  const int pair = blockIdx.x * PAIRS + pl;
  const bool pvalid = pair < total_pairs;
  float* sk = smem + pl * PAIR_FLOATS;
  float* sv = sk + N * HD;
  float* sb = sv + N * HD;
  int h = 0, wx = 0, wy = 0, b = 0;
  if (pvalid) {
    h = pair % heads;
    const int win = pair / heads;
    wx = win % nwx;
    wy = (win / nwx) % nwy;
    b = win / (nwx * nwy);
    for (int i = tl; i < N * (HD / VEC); i += TPP) {
      const int j = i / (HD / VEC), c = i - j * (HD / VEC);
      const int y = wy * WS + j / WS, x = wx * WS + j % WS;
      const T* src = (y < H && x < W) ? qkv + ((int64_t)(b * H + y) * W + x) * ld + h * 3 * HD : pad_qkv + h * 3 * HD;
      float kk[VEC], vv[VEC];
      Vec8<T>::load(src + HD + c * VEC, kk);
      Vec8<T>::load(src + 2 * HD + c * VEC, vv);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        sk[j * HD + c * VEC + e] = kk[e];
        sv[j * HD + c * VEC + e] = vv[e];
      }
    }
    for (int i = tl; i < N; i += TPP) sb[i] = bias[h * N + i];
  }
  __syncthreads();
  if (!pvalid) return;
  float q[QB][HD], acc[QB][HD], mx[QB], sum[QB];
  int qy[QB], qx[QB];
  bool qok[QB];
#pragma unroll
  for (int r = 0; r < QB; ++r) {
    const int qi = tl * QB + r;
    qy[r] = qi / WS;
    qx[r] = qi - qy[r] * WS;
    const int y = wy * WS + qy[r], x = wx * WS + qx[r];
    qok[r] = qi < N && y < H && x < W;
    mx[r] = -INFINITY;
    sum[r] = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) { acc[r][d] = 0.f; q[r][d] = 0.f; }
    if (qok[r]) {
      const T* src = qkv + ((int64_t)(b * H + y) * W + x) * ld + h * 3 * HD;
#pragma unroll
      for (int c = 0; c < HD / VEC; ++c) Vec8<T>::load(src + c * VEC, q[r] + c * VEC);
    }
  }
  const float scale = 0.17677669529663687f;  // 32^-0.5
  int jy = 0, jx = 0;
  for (int j = 0; j < N; ++j) {
    const float4* kr = reinterpret_cast<const float4*>(sk + j * HD);
    float s[QB];
#pragma unroll
    for (int r = 0; r < QB; ++r) s[r] = 0.f;
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const float4 k4 = kr[c];
#pragma unroll
      for (int r = 0; r < QB; ++r) {
        s[r] = fmaf(q[r][4 * c], k4.x, s[r]);
        s[r] = fmaf(q[r][4 * c + 1], k4.y, s[r]);
        s[r] = fmaf(q[r][4 * c + 2], k4.z, s[r]);
        s[r] = fmaf(q[r][4 * c + 3], k4.w, s[r]);
      }
    }
    float p[QB];
#pragma unroll
    for (int r = 0; r < QB; ++r) {
      const int dy = qy[r] > jy ? qy[r] - jy : jy - qy[r], dx = qx[r] > jx ? qx[r] - jx : jx - qx[r];
      const float sc = s[r] * scale + sb[(dy * WS + dx) % N];
      if (sc > mx[r]) {  // new running maximum: rescale what has been accumulated
        const float f = __expf(mx[r] - sc);
        sum[r] *= f;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[r][d] *= f;
        mx[r] = sc;
      }
      p[r] = __expf(sc - mx[r]);
      sum[r] += p[r];
    }
    const float4* vr = reinterpret_cast<const float4*>(sv + j * HD);
#pragma unroll
    for (int c = 0; c < HD / 4; ++c) {
      const float4 v4 = vr[c];
#pragma unroll
      for (int r = 0; r < QB; ++r) {
        acc[r][4 * c] = fmaf(p[r], v4.x, acc[r][4 * c]);
        acc[r][4 * c + 1] = fmaf(p[r], v4.y, acc[r][4 * c + 1]);
        acc[r][4 * c + 2] = fmaf(p[r], v4.z, acc[r][4 * c + 2]);
        acc[r][4 * c + 3] = fmaf(p[r], v4.w, acc[r][4 * c + 3]);
      }
    }
    if (++jx == WS) { jx = 0; ++jy; }
  }
#pragma unroll
  for (int r = 0; r < QB; ++r) {
    if (!qok[r]) continue;
    const int y = wy * WS + qy[r], x = wx * WS + qx[r];
    const float inv = 1.f / sum[r];
    T* dst = out + ((int64_t)(b * H + y) * W + x) * ldo + h * HD;
#pragma unroll
    for (int c = 0; c < HD / VEC; ++c) {
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e) o[e] = acc[r][c * VEC + e] * inv;
      Vec8<T>::store(dst + c * VEC, o);
    }
  }
}

// ------------------------------------------------------------------------------------
// bilinear resize, align_corners=False (model_builder.py:779-786)
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void resize_bilinear_kernel(const T* __restrict__ in, T* __restrict__ out, int B, int IH,
                                       int IW, int OH, int OW, int C) {
  const int CG = C / VEC;
  const int64_t total = (int64_t)B * OH * OW * CG;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int cg = (int)(idx % CG);
  const int64_t pix = idx / CG;
  const int ox = (int)(pix % OW);
  const int oy = (int)((pix / OW) % OH);
  const int64_t b = pix / ((int64_t)OW * OH);
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  float fy = ((float)oy + 0.5f) * sy - 0.5f;
  float fx = ((float)ox + 0.5f) * sx - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const T* base = in + b * IH * (int64_t)IW * C + cg * VEC;
  float a[VEC], bb[VEC], c[VEC], d[VEC], o[VEC];
  Vec8<T>::load(base + ((int64_t)y0 * IW + x0) * C, a);
  Vec8<T>::load(base + ((int64_t)y0 * IW + x1) * C, bb);
  Vec8<T>::load(base + ((int64_t)y1 * IW + x0) * C, c);
  Vec8<T>::load(base + ((int64_t)y1 * IW + x1) * C, d);
#pragma unroll
  for (int e = 0; e < VEC; ++e)
    o[e] = hy * (hx * a[e] + lx * bb[e]) + ly * (hx * c[e] + lx * d[e]);
  Vec8<T>::store(out + pix * C + cg * VEC, o);
}

// ------------------------------------------------------------------------------------
// Bilinear resize AFTER the first layer of a neck level (the "commuted" neck front end).
// The student head ends with F.interpolate(32 -> 72, bilinear) (model_builder.py:779-786) and every neck level starts with
// a per-pixel linear map of the resized tensor: a ConvTranspose2d(k2, s2) or a 1x1 conv (necks.py:42-92).  A per-pixel
// channel map commutes with a per-channel spatial interpolation whose weights sum to one (bias included), so the GEMM runs
// on the 32 x 32 map (5 x fewer rows) and THIS kernel interpolates its output: in [B][IH][IW][taps * C] (tap-major channel
// blocks as the ConvT GEMM lays them out, taps = 4, or taps = 1 for the 1x1) -> out [B][s*OH + 2P][s*OW + 2P][C] with
// s = 2 for taps = 4 (ConvT pixel shuffle: output pixel (2y + dy, 2x + dx) takes tap dy*2 + dx), + bias, activation,
// optionally inside a 1-pixel zero border (P).  Same interpolation arithmetic as resize_bilinear_kernel.  HBM-write bound.
// A thread owns one SOURCE cell (cy, cx) -- the four corner vectors are loaded once -- and writes every output pixel of the
// 72-grid whose interpolation footprint starts in that cell (2-3 per axis for 32 -> 72): the naive one-thread-per-output
// form re-reads its four corners per output (4 x the write traffic out of L2; 2.45 ms per step for the six launches).
// The per-axis maps (first source index and fraction of every output index; first output index of every source cell) are
// built once per workgroup in LDS with ATen's float arithmetic; a thread's own index arithmetic is 32-bit shifts
// (grid.y = image x source row).  Rows are blended first (hy a + ly c, hy b + ly d), then columns: 2 FMAs per output
// element instead of 7 operations -- the kernel has as much arithmetic as memory time at 8 channels per thread.
__device__ __forceinline__ float bilinear_srcf(int o, float scale) {  // source coordinate of output o (align_corners=False)
  const float f = ((float)o + 0.5f) * scale - 0.5f;
  return f < 0.f ? 0.f : f;
}
template <typename T, bool ROWS_FIRST>
__global__ __launch_bounds__(256) void resize_shuffle_kernel(const T* __restrict__ in, const float* __restrict__ bias,
                                                             T* __restrict__ out, int IH, int IW, int OH, int OW, int C,
                                                             int taps, int act, int P, int cg_shift, int tap_shift) {
  extern __shared__ __attribute__((aligned(16))) char rs_smem[];
  int* ys0 = reinterpret_cast<int*>(rs_smem);            // [OH] first source row of output row
  float* yl = reinterpret_cast<float*>(ys0 + OH);        // [OH] its fraction
  int* xs0 = reinterpret_cast<int*>(yl + OH);            // [OW]
  float* xl = reinterpret_cast<float*>(xs0 + OW);        // [OW]
  int* yfirst = reinterpret_cast<int*>(xl + OW);         // [IH] first output row whose footprint starts in the cell, or -1
  int* xfirst = yfirst + IH;                             // [IW]
  const int tid = threadIdx.x;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  for (int i = tid; i < IH + IW; i += 256) yfirst[i] = -1;
  __syncthreads();
  for (int o = tid; o < OH + OW; o += 256) {
    const bool isy = o < OH;
    const int oo = isy ? o : o - OH;
    const float sc = isy ? sy : sx;
    const float f = bilinear_srcf(oo, sc);
    const int s0 = (int)f;
    (isy ? ys0 : xs0)[oo] = s0;
    (isy ? yl : xl)[oo] = f - (float)s0;
    if (oo == 0 || (int)bilinear_srcf(oo - 1, sc) != s0) (isy ? yfirst : xfirst)[s0] = oo;
  }
  __syncthreads();
  const int CG = 1 << cg_shift;
  const unsigned r = blockIdx.x * 256u + (unsigned)tid;
  const int cg = (int)(r & (unsigned)(CG - 1));
  const int tap = (int)((r >> cg_shift) & (unsigned)(taps - 1));
  const int cx = (int)(r >> (cg_shift + tap_shift));
  if (cx >= IW) return;
  const int b = blockIdx.y / IH, cy = blockIdx.y - b * IH;
  const int oy0 = yfirst[cy], ox0 = xfirst[cx];
  if (oy0 < 0 || ox0 < 0) return;  // owns no output
  const int s = taps == 4 ? 2 : 1;
  const int FW = s * OW, FH = s * OH;
  const int y1 = cy + (cy < IH - 1 ? 1 : 0), x1 = cx + (cx < IW - 1 ? 1 : 0);
  const int CI = taps * C;
  const T* base = in + (int64_t)b * IH * IW * CI + tap * C + cg * VEC;
  float a[VEC], bb[VEC], c[VEC], d[VEC], bv[VEC];
  Vec8<T>::load(base + ((int64_t)cy * IW + cx) * CI, a);
  Vec8<T>::load(base + ((int64_t)cy * IW + x1) * CI, bb);
  Vec8<T>::load(base + ((int64_t)y1 * IW + cx) * CI, c);
  Vec8<T>::load(base + ((int64_t)y1 * IW + x1) * CI, d);
#pragma unroll
  for (int e = 0; e < VEC; ++e) bv[e] = bias ? bias[cg * VEC + e] : 0.f;
  const int dy = tap >> 1, dx = tap & 1;
  for (int oy = oy0; oy < OH && ys0[oy] == cy; ++oy) {
    const float ly = yl[oy], hy = 1.f - ly;
    float t[VEC], u[VEC];
    if constexpr (ROWS_FIRST) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        t[e] = fmaf(hy, a[e], ly * c[e]);
        u[e] = fmaf(hy, bb[e], ly * d[e]);
      }
    }
    const int Y = s == 2 ? 2 * oy + dy : oy;
    T* orow = out + (((int64_t)b * (FH + 2 * P) + Y + P) * (int64_t)(FW + 2 * P) + P) * C + cg * VEC;
    for (int ox = ox0; ox < OW && xs0[ox] == cx; ++ox) {
      const float lx = xl[ox], hx = 1.f - lx;
      const int X = s == 2 ? 2 * ox + dx : ox;
      float o[VEC];
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        o[e] = ROWS_FIRST ? fmaf(hx, t[e], fmaf(lx, u[e], bv[e]))
                          : (hy * (hx * a[e] + lx * bb[e]) + ly * (hx * c[e] + lx * d[e])) + bv[e];  // ATen's order
      act_apply_n<VEC>(o, act);
      Vec8<T>::store(orow + (int64_t)X * C, o);
    }
  }
}

// The same kernel without the per-workgroup LDS tables (used for the 1x1 level and as the general fallback; the ConvT levels
// take the row-persistent form below): bilinear_srcf is monotone, so the first output index whose source index reaches the
// cell is at most two steps above the estimate (cell + 0.5) / scale - 1.5.  ATen's operation order, as the table form.
template <typename T>
__global__ __launch_bounds__(256) void resize_shuffle_direct_kernel(const T* __restrict__ in, const float* __restrict__ bias,
                                                                    T* __restrict__ out, int IH, int IW, int OH, int OW, int C,
                                                                    int taps, int act, int P, int cg_shift, int tap_shift, int abl) {
  const int CG = 1 << cg_shift;
  const unsigned r = blockIdx.x * 256u + (unsigned)threadIdx.x;
  const int cg = (int)(r & (unsigned)(CG - 1));
  const int tap = (int)((r >> cg_shift) & (unsigned)(taps - 1));
  const int cx = (int)(r >> (cg_shift + tap_shift));
  if (cx >= IW) return;
  const int b = blockIdx.y / IH, cy = blockIdx.y - b * IH;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  int oy0 = (int)(((float)cy + 0.5f) * ((float)OH / (float)IH) - 1.5f);
  int ox0 = (int)(((float)cx + 0.5f) * ((float)OW / (float)IW) - 1.5f);
  oy0 = oy0 < 0 ? 0 : oy0;
  ox0 = ox0 < 0 ? 0 : ox0;
  while (oy0 < OH && (int)bilinear_srcf(oy0, sy) < cy) ++oy0;
  while (ox0 < OW && (int)bilinear_srcf(ox0, sx) < cx) ++ox0;
  if (oy0 >= OH || ox0 >= OW || (int)bilinear_srcf(oy0, sy) != cy || (int)bilinear_srcf(ox0, sx) != cx) return;  // owns no output
  // the columns of this cell: at most MAXO per axis (down-scaling cells own 0 or 1, up-scaling by r owns <= ceil(r) + 1)
  constexpr int MAXO = 4;
  float lxs[MAXO];
  int nx = 0;
#pragma unroll
  for (int i = 0; i < MAXO; ++i) {
    const float f = bilinear_srcf(ox0 + i, sx);
    const bool in_cell = ox0 + i < OW && (int)f == cx && nx == i;
    lxs[i] = f - (float)cx;
    nx += in_cell ? 1 : 0;
  }
  const int s = taps == 4 ? 2 : 1;
  const int FW = s * OW, FH = s * OH;
  const int y1 = cy + (cy < IH - 1 ? 1 : 0), x1 = cx + (cx < IW - 1 ? 1 : 0);
  const int CI = taps * C;
  const T* base = in + (int64_t)b * IH * IW * CI + tap * C + cg * VEC;
  float a[VEC], bb[VEC], c[VEC], d[VEC];
#ifdef ESAM3_DEV
  if (abl & 1) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) { a[e] = (float)(cx + e); bb[e] = (float)(cy - e); c[e] = (float)(tap * e); d[e] = (float)(cg + e); }
  } else
#endif
  {
    Vec8<T>::load(base + ((int64_t)cy * IW + cx) * CI, a);
    Vec8<T>::load(base + ((int64_t)cy * IW + x1) * CI, bb);
    Vec8<T>::load(base + ((int64_t)y1 * IW + cx) * CI, c);
    Vec8<T>::load(base + ((int64_t)y1 * IW + x1) * CI, d);
  }
  float bv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) bv[e] = bias ? bias[cg * VEC + e] : 0.f;
  const int dy = tap >> 1, dx = tap & 1;
  for (int oy = oy0; oy < OH; ++oy) {
    const float fy = bilinear_srcf(oy, sy);
    if ((int)fy != cy) break;
    const float ly = fy - (float)cy, hy = 1.f - ly;
    const int Y = s == 2 ? 2 * oy + dy : oy;
    T* orow = out + (((int64_t)b * (FH + 2 * P) + Y + P) * (int64_t)(FW + 2 * P) + P) * C + cg * VEC;
#pragma unroll
    for (int i = 0; i < MAXO; ++i) {
      if (i < nx) {
        const float lx = lxs[i], hx = 1.f - lx;
        const int X = s == 2 ? 2 * (ox0 + i) + dx : ox0 + i;
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (hy * (hx * a[e] + lx * bb[e]) + ly * (hx * c[e] + lx * d[e])) + bv[e];  // ATen's order
        act_apply_n<VEC>(o, act);
#ifdef ESAM3_DEV
        if ((abl & 2) && o[0] != 12345.678f) continue;             // ablation: no stores (the arithmetic stays live)
        if ((abl & 4) && (i != 0 || oy != oy0)) continue;          // ablation: one store per thread
#endif
        Vec8<T>::store(orow + (int64_t)X * C, o);
      }
    }
  }
}

// Row-persistent, wave-uniform form for the ConvT levels (taps = 4, C = 256 or 512).  Measured with loads and stores ablated
// (profiles/r04/resize_shuffle_abl.txt) the per-thread forms spend 0.146 ms of the level-0 launch on VALU issue alone -- every
// lane recomputes the cell's output range, fractions and 64-bit addresses -- and a plain fill writes the same bytes in 0.105 ms.
// Here one WAVE owns (image, source row cy, 512 consecutive channels of the tap-major input row, a run of source columns) and
// walks the columns: the cell's output rows / columns, fractions and row base addresses are wave-uniform (scalar registers,
// scalar branches), the right-hand corners of one cell are the left-hand corners of the next (half the loads) and the column
// after that is requested one cell ahead, so a wave never waits for a load it has just issued.  The per-axis maps (first
// output index and count of every source cell, fraction of every output index) are built by the launcher with ATen's float
// arithmetic and travel in the kernel arguments; lane i keeps column i's entries, the walk reads them with v_readlane.
// With C = 512 the wave is one tap; with C = 256 it is the tap pair (dy, 0 / 1): the column parity is the lane's upper bit.
struct RsAxisTables {
  static constexpr int MAX_IN = 64, MAX_OUT = 192;
  int first[2][MAX_IN];    // [0] rows, [1] columns: first output index whose interpolation footprint starts in the cell
  int count[2][MAX_IN];    // how many consecutive output indices do
  float frac[2][MAX_OUT];  // fraction of every output index
};
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
template <typename T, int L2CG, int ACT>
__global__ __launch_bounds__(256) void resize_shuffle_row_kernel(const T* __restrict__ in, const float* __restrict__ bias,
                                                                 T* __restrict__ out, int nrows, int IH, int IW, int OH, int OW, int P,
                                                                 int split, const RsAxisTables tab, int abl) {
  constexpr int CG = 1 << L2CG, C = CG * VEC, UPC = 4 * CG / 64;  // wave units per cell
  constexpr int MAXO = 4;
#ifdef ESAM3_DEV
  int lin = 0;   // ablation 8: the wave's stores go to one private contiguous region instead of the pixel-shuffled rows
#endif
  const int lane = threadIdx.x & 63;
  const int unit = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * 4u + (threadIdx.x >> 6)));
  const int tg = unit % UPC, r_ = unit / UPC;
  const int seg = r_ % split, row = r_ / split;
  if (row >= nrows) return;
  const int b = row / IH, cy = row - b * IH;
  const int per = (IW + split - 1) / split, cxs = seg * per, cxe = cxs + per < IW ? cxs + per : IW;
  const int oy0 = tab.first[0][cy], ny = tab.count[0][cy];
  if (ny == 0 || cxs >= cxe) return;
  float lys[MAXO];
#pragma unroll
  for (int j = 0; j < MAXO; ++j) lys[j] = tab.frac[0][oy0 + j < OH ? oy0 + j : OH - 1];
  // lane i: the column tables of source column i
  const int lc = lane < IW ? lane : IW - 1;
  const int xf_l = tab.first[1][lc], xn_l = tab.count[1][lc];
  float xfr_l[MAXO];
#pragma unroll
  for (int k = 0; k < MAXO; ++k) xfr_l[k] = tab.frac[1][xf_l + k < OW ? xf_l + k : OW - 1];
  const int FW = 2 * OW, FH = 2 * OH;
  const int y1 = cy + (cy < IH - 1 ? 1 : 0);
  constexpr int CI = 4 * C;
  const int dy = CG == 64 ? tg >> 1 : tg;
  const int dx_lane = CG == 64 ? 0 : lane >> L2CG, dx_wave = CG == 64 ? tg & 1 : 0;
  const int cg = lane & (CG - 1);
  const T* rowT = in + ((int64_t)(b * IH + cy) * IW) * CI + tg * 512 + lane * VEC;
  const T* rowB = in + ((int64_t)(b * IH + y1) * IW) * CI + tg * 512 + lane * VEC;
  f32x2_v bv[VEC / 2];
#pragma unroll
  for (int e = 0; e < VEC / 2; ++e)
    bv[e] = bias ? f32x2_v{bias[cg * VEC + 2 * e], bias[cg * VEC + 2 * e + 1]} : f32x2_v{0.f, 0.f};
  const int ooff = dx_lane * C + cg * VEC;   // the lane's element offset inside an output row segment
  T* const obase = out + (((int64_t)b * (FH + 2 * P) + dy + P) * (int64_t)(FW + 2 * P) + P + dx_wave) * C + ooff;
  const int64_t orow_pitch = (int64_t)(FW + 2 * P) * C;
  float La[VEC], Lc[VEC], Ra[VEC], Rc[VEC];
  Vec8<T>::load(rowT + (int64_t)cxs * CI, La);
  Vec8<T>::load(rowB + (int64_t)cxs * CI, Lc);
  {
    const int x1 = cxs + 1 < IW ? cxs + 1 : IW - 1;
    Vec8<T>::load(rowT + (int64_t)x1 * CI, Ra);
    Vec8<T>::load(rowB + (int64_t)x1 * CI, Rc);
  }
  for (int cx = cxs; cx < cxe; ++cx) {
    float Na[VEC], Nc[VEC];
    {
      const int x2 = cx + 2 < IW ? cx + 2 : IW - 1;   // requested now, consumed as the next cell's right-hand corners
      Vec8<T>::load(rowT + (int64_t)x2 * CI, Na);
      Vec8<T>::load(rowB + (int64_t)x2 * CI, Nc);
    }
    const int ox0 = __builtin_amdgcn_readlane(xf_l, cx), nx = __builtin_amdgcn_readlane(xn_l, cx);
    float lxs[MAXO];
#pragma unroll
    for (int k = 0; k < MAXO; ++k) lxs[k] = readlane_f(xfr_l[k], cx);
    if (nx != 0) {
      for (int j = 0; j < ny; ++j) {
        const float ly = j == 0 ? lys[0] : j == 1 ? lys[1] : j == 2 ? lys[2] : lys[3], hy = 1.f - ly;
        f32x2_v t[VEC / 2], u[VEC / 2];
#pragma unroll
        for (int e = 0; e < VEC / 2; ++e) {
          t[e] = __builtin_elementwise_fma((f32x2_v)(hy), f32x2_v{La[2 * e], La[2 * e + 1]}, (f32x2_v)(ly) * f32x2_v{Lc[2 * e], Lc[2 * e + 1]});
          u[e] = __builtin_elementwise_fma((f32x2_v)(hy), f32x2_v{Ra[2 * e], Ra[2 * e + 1]}, (f32x2_v)(ly) * f32x2_v{Rc[2 * e], Rc[2 * e + 1]});
        }
        T* orow = obase + (int64_t)(2 * (oy0 + j)) * orow_pitch + (int64_t)(2 * ox0) * C;
#pragma unroll
        for (int i = 0; i < MAXO; ++i) {
          if (i < nx) {
            const float lx = lxs[i], hx = 1.f - lx;
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC / 2; ++e) {
              const f32x2_v v = __builtin_elementwise_fma((f32x2_v)(hx), t[e], __builtin_elementwise_fma((f32x2_v)(lx), u[e], bv[e]));
              o[2 * e] = v.x;
              o[2 * e + 1] = v.y;
            }
            act_apply_n<VEC>(o, ACT);
#ifdef ESAM3_DEV
            if ((abl & 2) && o[0] + o[1] + o[2] + o[3] + o[4] + o[5] + o[6] + o[7] != 12345.678f) continue;  // ablation: no stores
            if (abl & 8) {   // ablation: the same bytes, each wave into its own contiguous run (wrapped into the tensor)
              const int64_t npx = (int64_t)(nrows / IH) * (FH + 2 * P) * (FW + 2 * P) * C / 512;
              Vec8<T>::store(out + (((int64_t)unit * 216 + (lin++)) % npx) * 512 + lane * VEC, o);
              continue;
            }
#endif
            Vec8<T>::store(orow + 2 * i * C, o);
          }
        }
      }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { La[e] = Ra[e]; Lc[e] = Rc[e]; Ra[e] = Na[e]; Rc[e] = Nc[e]; }
  }
}
// host side of the tables: area_pixel_compute_source_index's float arithmetic (ATen UpSample.h, align_corners = False)
static bool rs_build_tables(RsAxisTables& t, int IH, int IW, int OH, int OW) {
  if (IH > RsAxisTables::MAX_IN || IW > RsAxisTables::MAX_IN || OH > RsAxisTables::MAX_OUT || OW > RsAxisTables::MAX_OUT) return false;
  const int in[2] = {IH, IW}, on[2] = {OH, OW};
  for (int ax = 0; ax < 2; ++ax) {
    for (int c = 0; c < RsAxisTables::MAX_IN; ++c) t.first[ax][c] = t.count[ax][c] = 0;
    for (int o = 0; o < RsAxisTables::MAX_OUT; ++o) t.frac[ax][o] = 0.f;
    const float scale = (float)in[ax] / (float)on[ax];
    int prev = -1;
    for (int o = 0; o < on[ax]; ++o) {
      volatile float f = ((float)o + 0.5f) * scale;   // two roundings, as the CPU ATen kernel does
      f = f - 0.5f;
      if (f < 0.f) f = 0.f;
      int s0 = (int)f;
      if (s0 > in[ax] - 1) s0 = in[ax] - 1;
      t.frac[ax][o] = f - (float)s0;
      if (s0 != prev) { t.first[ax][s0] = o; prev = s0; }
      if (++t.count[ax][s0] > 4) return false;   // the kernel unrolls four columns per cell
    }
  }
  return true;
}

// ------------------------------------------------------------------------------------
// LayerNorm over the last dim (nn.LayerNorm, transformer.py:136-146; LayerNorm2d in NHWC,
// sam/common.py:27-39), optional residual add before and activation after.
// One wavefront per row; C <= 64 * MAXPL.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void layernorm_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                 T* __restrict__ out, int64_t rows, int C, float eps, int act) {
  constexpr int MAXPL = 16;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const T* xp = x + row * C;
  const T* rp = res ? res + row * C : nullptr;
  float v[MAXPL];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + 64 * i;
    float t = 0.f;
    if (c < C) {
      t = to_f32<T>(xp[c]);
      if (rp) t += to_f32<T>(rp[c]);
    }
    v[i] = t;
    sum += t;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float mean = sum / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + 64 * i;
    const float d = c < C ? v[i] - mean : 0.f;
    sq += d * d;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  const float rstd = 1.f / sqrtf(sq / (float)C + eps);
  T* op = out + row * C;
#pragma unroll
  for (int i = 0; i < MAXPL; ++i) {
    const int c = lane + 64 * i;
    if (c < C) {
      float y = (v[i] - mean) * rstd * gamma[c] + beta[c];
      op[c] = from_f32<T>(act_apply(y, act));
    }
  }
}

// Vectorised variant for C % 8 == 0: LPR (a power of two) lanes share a row, a lane owns up to NCH 8-channel chunks
// (16-byte loads and stores; the scalar kernel above moves 2 bytes per lane per load) and keeps gamma / beta of its
// chunks in registers while the wave walks `iters` groups of 64/LPR rows.
template <typename TI, typename TO, int NCH>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const TI* __restrict__ x, const TI* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            TO* __restrict__ out, int64_t rows, int C, float eps, int act,
                                                            int lpr, int iters) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (lpr - 1), rw = lane / lpr, rpw = 64 / lpr;  // lane within the row, row within the wave's group
  const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nchunk = C >> 3;
  float gm[NCH][8], bt[NCH][8];
  bool ok[NCH];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int ci = sub + lpr * j;
    ok[j] = ci < nchunk;
    // 16-byte loads (round 6: element by element these were 16 single-dword loads per chunk -- as many load instructions as the wave's whole
    // walk over its rows; same values)
    float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, b0 = g0, b1 = g0;
    if (ok[j]) {
      g0 = *reinterpret_cast<const float4*>(gamma + ci * 8);
      g1 = *reinterpret_cast<const float4*>(gamma + ci * 8 + 4);
      b0 = *reinterpret_cast<const float4*>(beta + ci * 8);
      b1 = *reinterpret_cast<const float4*>(beta + ci * 8 + 4);
    }
    gm[j][0] = g0.x; gm[j][1] = g0.y; gm[j][2] = g0.z; gm[j][3] = g0.w; gm[j][4] = g1.x; gm[j][5] = g1.y; gm[j][6] = g1.z; gm[j][7] = g1.w;
    bt[j][0] = b0.x; bt[j][1] = b0.y; bt[j][2] = b0.z; bt[j][3] = b0.w; bt[j][4] = b1.x; bt[j][5] = b1.y; bt[j][6] = b1.z; bt[j][7] = b1.w;
  }
  const float invC = 1.f / (float)C;
  for (int it = 0; it < iters; ++it) {
    const int64_t row = (wave_id * iters + it) * rpw + rw;
    const bool rok = row < rows;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int64_t off = row * C + (int64_t)(sub + lpr * j) * 8;
      if (rok && ok[j]) {
        Vec8<TI>::load(x + off, v[j]);
        if (res) {
          float r[8];
          Vec8<TI>::load(res + off, r);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[j][e] += r[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[j][e] = 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) sum += v[j][e];
    }
    for (int o = lpr >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum * invC;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      if (ok[j]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[j][e] - mean;
          sq += d * d;
        }
      }
    for (int o = lpr >> 1; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.f / sqrtf(sq * invC + eps);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (rok && ok[j]) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = (v[j][e] - mean) * rstd * gm[j][e] + bt[j][e];
        act_apply_n<8>(y, act);
        Vec8<TO>::store(out + row * C + (int64_t)(sub + lpr * j) * 8, y);
      }
    }
  }
}

template <typename T>
__global__ void cast_to_f32_kernel(const T* __restrict__ in, float* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = to_f32<T>(in[i]);
}
template <typename T>
__global__ void cast_from_f32_kernel(const float* __restrict__ in, T* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    out[i] = from_f32<T>(in[i]);
}

template <typename T>
__global__ void nhwc_to_nchw_f32_kernel(const T* __restrict__ in, float* __restrict__ out, int HW,
                                        int C) {
  // 32x32 LDS tile transpose per (b): in [HW][C] -> out [C][HW]
  __shared__ float tile[32][33];
  const int64_t b = blockIdx.z;
  const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty in [0,8)
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, c = c0 + tx;
    tile[r][tx] = (p < HW && c < C) ? to_f32<T>(in[(b * HW + p) * (int64_t)C + c]) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, p = p0 + tx;
    if (c < C && p < HW) out[(b * C + c) * (int64_t)HW + p] = tile[tx][r];
  }
}

__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int64_t HW,
                                     int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, pixel)
  if (i >= total) return;
  const int64_t b = i / HW, p = i - b * HW;
  const uint8_t* s = in + i * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float x = (float)s[c] / 255.0f;
    out[(b * 3 + c) * HW + p] = (x - 0.5f) / 0.5f;
  }
}

// P1: uint8 HWC image of any size -> antialiased bilinear resize to (OH, OW) -> round to uint8
// -> x/255 -> (x-0.5)/0.5, fp32 CHW.  The reference runs torchvision v2.Resize on a uint8 device
// tensor (sam3_image_processor.py:24-31,57-58), which resizes in fp32 with the triangle filter of
// torch's upsample_bilinear2d_aa (support = max(scale,1), taps normalised to sum 1), rounds half
// to even and casts back to uint8.  One thread per output pixel; tap weights are recomputed per
// thread (a 1024->1008 resize has 3x3 taps).
template <int PX>  // bytes per input pixel: 3 (RGB) or 4 (RGBX, the in-memory layout of a PIL "RGB" image; byte 3 is ignored)
__global__ __launch_bounds__(256) void resize_aa_u8_kernel(const uint8_t* __restrict__ in, int H, int W,
                                                           float* __restrict__ out, int OH, int OW) {
#pragma clang fp contract(off)  // separate multiply and add like the upstream kernel: results land on
                                // .5 rounding ties often enough (uint8 inputs) for an fma to show
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (ox >= OW) return;
  in += (int64_t)blockIdx.z * H * W * PX;     // image of the batch (equal sizes)
  out += (int64_t)blockIdx.z * 3 * OH * OW;
  const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
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
    const uint8_t* row = in + ((int64_t)(y0 + jy) * W + x0) * PX;
    float r[3] = {0.f, 0.f, 0.f};
    for (int jx = 0; jx < nx; ++jx) {
      float wx = aa_tap(jx, x0, xm, inv_x);
      if (tx != 0.f) wx /= tx;
#pragma unroll
      for (int c = 0; c < 3; ++c) r[c] += (float)row[jx * PX + c] * wx;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += r[c] * wy;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float q = fminf(fmaxf(rintf(acc[c]), 0.f), 255.f);  // back to uint8 (round half to even)
    const float x = q / 255.0f;
    out[((int64_t)c * OH + oy) * OW + ox] = (x - 0.5f) / 0.5f;
  }
}

// border pixels of [B][Hp][Wp][C]: rows 0 and Hp-1, columns 0 and Wp-1; 16 bytes per thread
__global__ void zero_border_kernel(char* __restrict__ x, int B, int Hp, int Wp, int row_bytes) {
  const int per_img = 2 * Wp + 2 * (Hp - 2);  // border pixels per image
  const int chunks = row_bytes / 16;
  const int64_t total = (int64_t)B * per_img * chunks;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)(i % chunks);
  const int64_t pi = i / chunks;
  const int k = (int)(pi % per_img);
  const int64_t b = pi / per_img;
  int y, xx;
  if (k < Wp) { y = 0; xx = k; }
  else if (k < 2 * Wp) { y = Hp - 1; xx = k - Wp; }
  else { const int r = k - 2 * Wp; y = 1 + (r >> 1); xx = (r & 1) ? Wp - 1 : 0; }
  uint4* dst = reinterpret_cast<uint4*>(x + ((b * Hp + y) * (int64_t)Wp + xx) * row_bytes) + ch;
  *dst = make_uint4(0, 0, 0, 0);
}

inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

}  // namespace

int esam3_launch_zero_border(int dtype, void* x, int B, int Hp, int Wp, int C, hipStream_t s) {
  const int row_bytes = C * (dtype == 0 ? 4 : 2);
  if (row_bytes % 16) { esam3_set_error("zero_border: C=%d not 16-byte aligned", C); return -1; }
  const int64_t total = (int64_t)B * (2 * Wp + 2 * (Hp - 2)) * (row_bytes / 16);
  hipLaunchKernelGGL(zero_border_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, (char*)x, B, Hp, Wp, row_bytes);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_preprocess_u8(const uint8_t* in, float* out, int B, int H, int W, hipStream_t s) {
  const int64_t HW = (int64_t)H * W, total = HW * B;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, in, out, HW, total);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_resize_aa_u8(const uint8_t* in, int B, int H, int W, float* out, int OH, int OW, hipStream_t s, int pixel_bytes) {
  if (pixel_bytes == 4)
    hipLaunchKernelGGL(resize_aa_u8_kernel<4>, dim3(blocks_for(OW, 256), (unsigned)OH, (unsigned)B), dim3(256), 0, s, in, H, W, out,
                       OH, OW);
  else
    hipLaunchKernelGGL(resize_aa_u8_kernel<3>, dim3(blocks_for(OW, 256), (unsigned)OH, (unsigned)B), dim3(256), 0, s, in, H, W, out,
                       OH, OW);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

#define DISPATCH_T(dtype, ...)                  \
  do {                                          \
    if ((dtype) == 0) { using T = float; __VA_ARGS__; } \
    else { using T = bf16_t; __VA_ARGS__; }     \
  } while (0)

int esam3_launch_stem(int dtype, const float* img, const float* w, const float* bias, void* out,
                      int B, int H, int W, int Cout, int act, hipStream_t s) {
  if (Cout > 64 || Cout % VEC) { esam3_set_error("stem: Cout=%d unsupported", Cout); return -1; }
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  static const bool stem_valu = esam3_dev_flag("ESAM3_STEM_VALU") != 0;  // A/B timing
  if (dtype == 1 && !stem_valu && (Cout == 16 || Cout == 32 || Cout == 48 || Cout == 64) && !(((uintptr_t)out) & 7)) {
    const int tiles_x = (OW + 15) / 16, tiles_y = (OH + 15) / 16;
    const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)B);
    if ((int64_t)3 * H * W < ((int64_t)1 << 31) && !esam3_dev_flag("ESAM3_STEM_OLD")) {   // round 6: persistent workgroups, three per CU
      const unsigned ntiles = (unsigned)(tiles_x * tiles_y) * (unsigned)B;
      unsigned g = 256u * 3u;
      if (g > ntiles) g = ntiles;
#define ESAM3_STEM_MFMA_P(CO) hipLaunchKernelGGL(stem_mfma_p_kernel<CO>, dim3(g), dim3(256), 0, s, img, w, bias, (bf16_t*)out, H, W, OH, OW, tiles_x, tiles_x * tiles_y, ntiles, act)
      if (Cout == 16) ESAM3_STEM_MFMA_P(16);
      else if (Cout == 32) ESAM3_STEM_MFMA_P(32);
      else if (Cout == 48) ESAM3_STEM_MFMA_P(48);
      else ESAM3_STEM_MFMA_P(64);
#undef ESAM3_STEM_MFMA_P
      HIP_CHECK_RET(hipGetLastError());
      return 0;
    }
#define ESAM3_STEM_MFMA(CO) hipLaunchKernelGGL(stem_mfma_kernel<CO>, grid, dim3(256), 0, s, img, w, bias, (bf16_t*)out, H, W, OH, OW, tiles_x, act)
    if (Cout == 16) ESAM3_STEM_MFMA(16);
    else if (Cout == 32) ESAM3_STEM_MFMA(32);
    else if (Cout == 48) ESAM3_STEM_MFMA(48);
    else ESAM3_STEM_MFMA(64);
#undef ESAM3_STEM_MFMA
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  const unsigned gx = blocks_for((int64_t)OW * (Cout / VEC), 256), gy = (unsigned)(B * OH);
  DISPATCH_T(dtype, hipLaunchKernelGGL(stem_kernel<T>, dim3(gx * gy), dim3(256), 0, s, img, w, bias, (T*)out, B,
                                       H, W, Cout, act, gx, gy));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_stem_dsconv(int dtype, const float* img, const float* w0, const float* b0, const float* wd, const float* bd,
                             const void* wp, int ldw, const float* bp, void* out, int B, int H, int W, hipStream_t s, int variant) {
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const int tiles_x = (OW + 15) / 16, tiles_y = (OH + 15) / 16;
  const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)B);
  static const bool valu = esam3_dev_flag("ESAM3_STEM_VALU") != 0;  // A/B timing
  if (dtype == 1 && !valu && ldw % 4 == 0 && !(((uintptr_t)wp) & 7) && variant != 1 && (int64_t)3 * H * W < ((int64_t)1 << 31) &&
      !esam3_dev_flag("ESAM3_STEM_OLD")) {
    // round 6: persistent workgroups (operands built once, next tile's image halo prefetched); dev builds: ESAM3_STEM_WGS per CU
    const unsigned ntiles = (unsigned)(tiles_x * tiles_y) * (unsigned)B;
    const int per_cu = esam3_dev_flag("ESAM3_STEM_WGS") > 0 ? esam3_dev_flag("ESAM3_STEM_WGS") : 3;
    unsigned g = 256u * (unsigned)per_cu;
    if (g > ntiles) g = ntiles;
    hipLaunchKernelGGL(stem_dsconv_mfma_p_kernel, dim3(g), dim3(256), 0, s, img, w0, b0, wd, bd, (const bf16_t*)wp, ldw, bp,
                       (bf16_t*)out, H, W, OH, OW, tiles_x, tiles_x * tiles_y, ntiles);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  if (dtype == 1 && !valu && ldw % 4 == 0 && !(((uintptr_t)wp) & 7)) {
    hipLaunchKernelGGL(stem_dsconv_mfma_kernel, grid, dim3(256), 0, s, img, w0, b0, wd, bd, (const bf16_t*)wp, ldw, bp, (bf16_t*)out,
                       H, W, OH, OW, tiles_x);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(stem_dsconv_kernel<T>, grid, dim3(256), 0, s, img, w0, b0, wd, bd, (const T*)wp, ldw, bp,
                                       (T*)out, H, W, OH, OW, tiles_x));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_dwconv(int dtype, const void* in, int ld_in, const float* w, const float* bias,
                        void* out, int ld_out, int B, int H, int W, int C, int ksize, int stride,
                        int act, hipStream_t s) {
  if (C % VEC || (ksize != 3 && ksize != 5) || (stride != 1 && stride != 2)) {
    esam3_set_error("dwconv: unsupported C=%d k=%d s=%d", C, ksize, stride);
    return -1;
  }
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
#define ESAM3_DW(KS, ST, NP)                                                                        \
  do {                                                                                              \
    const unsigned gx = blocks_for((int64_t)((OW + NP - 1) / NP) * (C / VEC), 256);                 \
    const unsigned gy = (unsigned)(B * OH);                                                         \
    DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv_kernel<T, KS, ST, NP>), dim3(gx * gy), dim3(256), 0, s, \
                                         (const T*)in, ld_in, w, bias, (T*)out, ld_out, H, W, C, OH, OW, \
                                         act, gx, gy));                                             \
  } while (0)
#define ESAM3_DW3(ST, NP, R)                                                                        \
  do {                                                                                              \
    const unsigned gx = blocks_for((int64_t)((OW + NP - 1) / NP) * (C / VEC), 256);                 \
    const int strips = (OH + R - 1) / R;                                                            \
    const unsigned gy = (unsigned)(B * strips);                                                     \
    DISPATCH_T(dtype, hipLaunchKernelGGL((dwconv3_strip_kernel<T, ST, NP, R>), dim3(gx * gy), dim3(256), 0, \
                                         s, (const T*)in, ld_in, w, bias, (T*)out, ld_out, H, W, C, OH,  \
                                         OW, act, gx, gy, strips));                                 \
  } while (0)
  static const int no_mfma = esam3_dev_flag("ESAM3_DW_NOMFMA", 0);  // dev A/B
  if (dtype == 1 && stride == 1 && C % 64 == 0 && !no_mfma && (ld_in * 2) % 16 == 0 && (ld_out * 2) % 8 == 0 &&
      !(((uintptr_t)in) & 15) && !(((uintptr_t)out) & 7) && (!bias || !(((uintptr_t)bias) & 15)) && H * W >= 256) {
    static const int jm = esam3_dev_flag("ESAM3_DW_JM", 1);  // dev: 8 or 16 tile rows
    const int J = jm == 2 ? 2 : 1;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 8 * J - 1) / (8 * J);
    const dim3 grid((unsigned)(tiles_x * tiles_y), (unsigned)(C / 64), (unsigned)B);
    const size_t lds = (size_t)(8 * J + ksize - 1) * (16 + ksize - 1) * 192;
#define ESAM3_DWM(KS_, J_)                                                                                              \
  do {                                                                                                                  \
    auto kern = dwconv_mfma_kernel<KS_, J_>;                                                                            \
    if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return -1;                                \
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)in, ld_in, w, bias, (bf16_t*)out, ld_out, H, W, C, act, tiles_x); \
  } while (0)
    if (ksize == 3) { if (J == 2) ESAM3_DWM(3, 2); else ESAM3_DWM(3, 1); }
    else { if (J == 2) ESAM3_DWM(5, 2); else ESAM3_DWM(5, 1); }
#undef ESAM3_DWM
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  static const int no_tiled = esam3_dev_flag("ESAM3_DW_NOTILED", 0);  // dev A/B
  const int esz_ = dtype == 0 ? 4 : 2;
  if (stride == 1 && C % 64 == 0 && !no_tiled && (ld_in * esz_) % 16 == 0 && (ld_out * esz_) % 16 == 0 &&
      !(((uintptr_t)in) & 15) && !(((uintptr_t)out) & 15) && H * W >= 256) {
    static const int j3 = esam3_dev_flag("ESAM3_DW_J3", 2);  // dev: rows of 8 per tile, k = 3 / k = 5
    static const int j5 = esam3_dev_flag("ESAM3_DW_J5", 2);
    const int J = (ksize == 3 ? j3 : j5) == 1 ? 1 : 2;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 8 * J - 1) / (8 * J);
    const int64_t total = (int64_t)tiles_x * tiles_y * (C / 64) * B;
    static int n_cu = 0;
    if (!n_cu) {
      int dev = 0;
      hipDeviceProp_t prop;
      HIP_CHECK_RET(hipGetDevice(&dev));
      HIP_CHECK_RET(hipGetDeviceProperties(&prop, dev));
      n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const dim3 grid((unsigned)(total < 2 * (int64_t)n_cu ? total : 2 * (int64_t)n_cu));  // persistent, two workgroups per CU
    const size_t pitch_b = 64 * (size_t)esz_ + 32;
    const size_t lds = (size_t)(8 * J + ksize - 1) * (16 + ksize - 1) * pitch_b + sizeof(float) * (size_t)(ksize * ksize + 1) * 64;
#define ESAM3_DWT(KS_, J_)                                                                                                   \
  do {                                                                                                                   \
    if (dtype == 0) {                                                                                                    \
      auto kern = dwconv_tiled_kernel<float, KS_, J_>;                                                                    \
      if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return -1;                               \
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const float*)in, ld_in, w, bias, (float*)out, ld_out, H, W, C, act, tiles_x, tiles_y, B); \
    } else {                                                                                                             \
      auto kern = dwconv_tiled_kernel<bf16_t, KS_, J_>;                                                                   \
      if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024)) return -1;                               \
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, (const bf16_t*)in, ld_in, w, bias, (bf16_t*)out, ld_out, H, W, C, act, tiles_x, tiles_y, B); \
    }                                                                                                                    \
  } while (0)
    if (ksize == 3) { if (J == 2) ESAM3_DWT(3, 2); else ESAM3_DWT(3, 1); }
    else { if (J == 2) ESAM3_DWT(5, 2); else ESAM3_DWT(5, 1); }
#undef ESAM3_DWT
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  static const int no_strip = esam3_dev_flag("ESAM3_DW_NOSTRIP", 0);
  if (ksize == 3 && stride == 1 && !no_strip) ESAM3_DW3(1, 2, 8);
  else if (ksize == 3 && !no_strip) ESAM3_DW3(2, 2, 4);
  else if (ksize == 3 && stride == 1) ESAM3_DW(3, 1, 2);
  else if (ksize == 3) ESAM3_DW(3, 2, 2);
  else if (stride == 1) ESAM3_DW(5, 1, 2);
  else ESAM3_DW(5, 2, 1);
#undef ESAM3_DW
#undef ESAM3_DW3
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_grouped_pw(int dtype, const void* in, int ld_in, const float* w, void* out,
                            int ld_out, int64_t rows, int C, int gs, hipStream_t s) {
  if (C % VEC || gs % VEC || C % gs) { esam3_set_error("grouped_pw: C=%d gs=%d", C, gs); return -1; }
  const int64_t total = rows * (C / VEC);
  DISPATCH_T(dtype, hipLaunchKernelGGL(grouped_pw_kernel<T>, dim3(blocks_for(total, 256)), dim3(256),
                                       0, s, (const T*)in, ld_in, w, (T*)out, ld_out, rows, C, gs));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// token splits of the kv reduction for (B, N, threads-per-token)
static int mla_splits(int B, int N, int tpt) {
  const int TL = 256 / tpt;
  int n_split = 1;
  while ((int64_t)B * n_split < 1024 && N / (n_split * 2) >= 16 * TL) n_split *= 2;
  return n_split;
}

// fp32 scratch needed by esam3_launch_lite_mla: final kv [B][groups][PAIRS] + per-split partials
int64_t esam3_lite_mla_scratch_floats(int B, int N, int groups, int dim) {
  const int64_t gp = (int64_t)groups * (dim + 1) * dim;
  const int tpt = groups * (dim / 4);
  const int n_split = tpt <= 256 ? mla_splits(B, N, tpt) : 1;
  return (int64_t)B * gp * (1 + n_split);
}

template <typename T, int DIM>
static int lite_mla_t(const void* ms, int ld, void* out, int ld_out, float* kv, int B, int N,
                      int groups, hipStream_t s) {
  constexpr int PAIRS = (DIM + 1) * DIM;
  if (groups > 256) { esam3_set_error("lite_mla: groups=%d too large", groups); return -1; }
  const size_t lds = sizeof(float) * (size_t)groups * PAIRS;
  const int tpt = groups * (DIM / 4);
  const size_t lds_kv = tpt <= 256 ? lds * (256 / tpt) : 0;
  if (tpt <= 256 && lds_kv <= 160 * 1024 - 1024) {
    const int n_split = mla_splits(B, N, tpt);
    float* partial = kv + (size_t)B * groups * PAIRS;
    auto kern = mla_kv_kernel<T, DIM>;
    if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(kern), 160 * 1024 - 1024)) return -1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * n_split)), dim3(256), lds_kv, s, (const T*)ms, ld, partial, N, groups,
                       n_split);
    const int64_t total = (int64_t)B * groups * PAIRS;
    hipLaunchKernelGGL(mla_kv_sum_kernel, dim3(blocks_for(total, 256)), dim3(256), 0, s, partial, kv, groups * PAIRS,
                       n_split, total);
  } else {  // unusual head counts: generic kernel (global fp32 atomics, order-dependent rounding)
    int n_split = 1;
    while ((int64_t)B * groups * n_split < 512 && N / (n_split * 2) >= 256) n_split *= 2;
    if (n_split > 1)
      HIP_CHECK_RET(hipMemsetAsync(kv, 0, sizeof(float) * (size_t)B * groups * PAIRS, s));
    hipLaunchKernelGGL((mla_kv_generic_kernel<T, DIM>), dim3((unsigned)(B * groups * n_split)), dim3(256),
                       0, s, (const T*)ms, ld, kv, N, groups, n_split);
  }
  const int rpb = 256 / groups;
  const int threads = rpb * groups;
  int tptok = 8;
  while (tptok > 1 && (int64_t)B * ((N + rpb * tptok - 1) / (rpb * tptok)) < 1024) tptok /= 2;
  hipLaunchKernelGGL((mla_apply_kernel<T, DIM>), dim3((unsigned)((N + rpb * tptok - 1) / (rpb * tptok)), (unsigned)B),
                     dim3(threads), lds, s, (const T*)ms, ld, kv, (T*)out, ld_out, N, groups, tptok);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_lite_mla(int dtype, const void* ms, int ld, void* out, int ld_out, float* kv, int B,
                          int N, int groups, int dim, hipStream_t s) {
  if (dim == 16) {
    return dtype == 0 ? lite_mla_t<float, 16>(ms, ld, out, ld_out, kv, B, N, groups, s)
                      : lite_mla_t<bf16_t, 16>(ms, ld, out, ld_out, kv, B, N, groups, s);
  } else if (dim == 32) {
    return dtype == 0 ? lite_mla_t<float, 32>(ms, ld, out, ld_out, kv, B, N, groups, s)
                      : lite_mla_t<bf16_t, 32>(ms, ld, out, ld_out, kv, B, N, groups, s);
  }
  esam3_set_error("lite_mla: dim=%d unsupported", dim);
  return -1;
}

template <typename T, int WS, int QB, int PAIRS>
static int launch_window_attn(const void* qkv, int ld, const void* pad_qkv, const float* bias, void* out, int ldo, int B,
                              int H, int W, int heads, hipStream_t s) {
  constexpr int N = WS * WS, TPP = (N + QB - 1) / QB;
  constexpr size_t lds = sizeof(float) * PAIRS * (2 * N * 32 + N);
  auto kern = window_attn_kernel<T, WS, QB, PAIRS>;
  if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds)) return -1;
  const int nwx = (W + WS - 1) / WS, nwy = (H + WS - 1) / WS;
  const int total = B * nwx * nwy * heads;
  hipLaunchKernelGGL(kern, dim3((unsigned)((total + PAIRS - 1) / PAIRS)), dim3(PAIRS * TPP), lds, s, (const T*)qkv, ld,
                     (const T*)pad_qkv, bias, (T*)out, ldo, H, W, heads, nwx, nwy, total);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// plain (unmasked, unbiased) softmax attention, heads x 32, bf16, Nk % 64 == 0: the MFMA kernel; returns 1 if
// the shape is not eligible (the caller then uses the generic fp32 core)
int esam3_launch_attn_mfma32(const void* q, int ldq, int q_off, const void* kv, int ldk, int k_off, int v_off, void* out,
                             int ldo, int B, int Nq, int Nk, int heads, hipStream_t s) {
  static const bool no_mfma = esam3_dev_flag("ESAM3_ATTN_VALU") != 0;
  if (no_mfma || Nk % 64 || Nq < 64 || ldq % 8 || ldk % 8 || q_off % 8 || k_off % 8 || v_off % 8 || ldo % 4) return 1;
  dim3 grid((unsigned)((Nq + 127) / 128), (unsigned)heads, (unsigned)B);  // plain sequences: one "window" W = ws = Nk
  hipLaunchKernelGGL(attn_mfma_kernel<32>, grid, dim3(256), 0, s, (const bf16_t*)q, ldq, q_off, (const bf16_t*)kv, ldk, k_off, v_off,
                     (bf16_t*)out, ldo, Nq, Nk, (int64_t)Nq, (int64_t)Nk, Nk, Nk, 1, 1, 0.17677669529663687f * 1.4426950408889634f,
                     (const float*)nullptr);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_attn_mfma32_splitk(const void* q, int ldq, int q_off, const void* kv, int ldk, int k_off, int v_off,
                                    void* out, int ldo, int B, int Nq, int Nk, int heads, const uint8_t* key_mask,
                                    const float* bias_y, const float* bias_x, int Hk, int Wk, int bias_q0, hipStream_t s) {
  static const bool no_mfma = esam3_dev_flag("ESAM3_ATTN_VALU") != 0;
  if (no_mfma || Nk < 4 || ldq % 8 || ldk % 8 || q_off % 8 || k_off % 8 || v_off % 8 || ldo % 4) return 1;
  if (bias_y && (Wk % 4 || Nk != Hk * Wk || Nk % 4)) return 1;
  dim3 grid((unsigned)((Nq + 31) / 32), (unsigned)heads, (unsigned)B);
  hipLaunchKernelGGL(attn_mfma32_splitk_kernel, grid, dim3(256), 0, s, (const bf16_t*)q, ldq, q_off, (const bf16_t*)kv, ldk,
                     k_off, v_off, (bf16_t*)out, ldo, Nq, Nk, heads, key_mask, bias_y, bias_x, Hk, Wk, bias_q0,
                     0.17677669529663687f * 1.4426950408889634f);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_window_attn(int dtype, const void* qkv, int ld, const void* pad_qkv, const float* bias, void* out,
                             int ldo, int B, int H, int W, int heads, int ws, hipStream_t s) {
  static const bool no_mfma = esam3_dev_flag("ESAM3_ATTN_VALU") != 0;
  if (dtype == 1 && !no_mfma && ld % 8 == 0 && ldo % 4 == 0 && (ws == 7 || ws == 14)) {
    const int nwx = (W + ws - 1) / ws, nwy = (H + ws - 1) / ws;
    const unsigned gz = (unsigned)(B * nwx * nwy);
    if (ws == 7)
      hipLaunchKernelGGL((attn_mfma32_win_kernel<7, 2>), dim3(gz, (unsigned)heads), dim3(128), 0, s, (const bf16_t*)qkv, ld,
                         (const bf16_t*)pad_qkv, bias, (bf16_t*)out, ldo, H, W, heads, nwx, nwy);
    else
      hipLaunchKernelGGL((attn_mfma32_win_kernel<14, 7>), dim3(gz, (unsigned)heads), dim3(448), 0, s, (const bf16_t*)qkv,
                         ld, (const bf16_t*)pad_qkv, bias, (bf16_t*)out, ldo, H, W, heads, nwx, nwy);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  if (ws == 7)
    return dtype == 0 ? launch_window_attn<float, 7, 2, 4>(qkv, ld, pad_qkv, bias, out, ldo, B, H, W, heads, s)
                      : launch_window_attn<bf16_t, 7, 2, 4>(qkv, ld, pad_qkv, bias, out, ldo, B, H, W, heads, s);
  if (ws == 14)
    return dtype == 0 ? launch_window_attn<float, 14, 2, 1>(qkv, ld, pad_qkv, bias, out, ldo, B, H, W, heads, s)
                      : launch_window_attn<bf16_t, 14, 2, 1>(qkv, ld, pad_qkv, bias, out, ldo, B, H, W, heads, s);
  esam3_set_error("window_attn: window size %d unsupported", ws);
  return -1;
}

static int se_splits(int B, int HW) {
  int splits = 1;
  while (B * splits < 1024 && HW / (splits * 2) >= 64) splits *= 2;
  return splits;
}
// fp32 scratch of esam3_launch_squeeze_excite: per-split channel sums [B][splits][C]
int64_t esam3_squeeze_excite_scratch_floats(int B, int HW, int C) { return (int64_t)B * se_splits(B, HW) * C; }

int esam3_launch_patchify(int dtype, const float* img, void* a, int B, int S, int P, int ldk, hipStream_t s) {
  const int G = S / P;
  const int64_t total = (int64_t)B * G * G * ldk;
  DISPATCH_T(dtype, hipLaunchKernelGGL(patchify_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, img, (T*)a, B, S, P,
                                       G, ldk));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_vit_rope(int dtype, void* qkv, const float* cos_sin, int64_t rows, int H, int W, int ws, int heads,
                          hipStream_t s) {
  const int64_t total = rows * 2 * heads * 32;
  DISPATCH_T(dtype, hipLaunchKernelGGL(vit_rope_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (T*)qkv, cos_sin,
                                       rows, H, W, ws, heads));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_attn_window(int dtype, void* qkv, int ld, int q_off, int k_off, int v_off, void* out, int ldo, int B,
                             int H, int W, int ws, int heads, int hd, const float* rope, hipStream_t s) {
  if (hd != 64 || H % ws || W % ws) { esam3_set_error("attn_window: hd=%d ws=%d H=%d W=%d unsupported", hd, ws, H, W); return -1; }
  const int N = ws * ws;
  static const bool no_mfma = esam3_dev_flag("ESAM3_ATTN_VALU") != 0;
  if (dtype == 1 && N % 64 == 0 && !no_mfma && ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && ldo % 4 == 0) {
    const int64_t rows = (int64_t)B * H * W;
    if (rope)  // K rotated in place, once; Q is rotated by the attention kernel as it loads it
      hipLaunchKernelGGL(vit_rope_k_kernel, dim3(blocks_for(rows * heads * 8, 256)), dim3(256), 0, s, (bf16_t*)qkv, ld, k_off, rope,
                         rows, H, W, ws, heads);
    const int nwx = W / ws, nwin = nwx * (H / ws);
    dim3 grid(blocks_for(N, 128), (unsigned)heads, (unsigned)(B * nwin));
    hipLaunchKernelGGL(attn_mfma_kernel<64>, grid, dim3(256), 0, s, (const bf16_t*)qkv, ld, q_off, (const bf16_t*)qkv, ld, k_off, v_off,
                       (bf16_t*)out, ldo, N, N, (int64_t)H * W, (int64_t)H * W, W, ws, nwx, nwin, 0.125f * 1.4426950408889634f, rope);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  if (rope) {  // the VALU kernel reads rotated q / k: rotate in place first (needs the ViT row layout)
    if (q_off != 0 || k_off != heads * hd || ld != 3 * heads * hd) { esam3_set_error("attn_window: rope needs the [3][heads][hd] row layout"); return -1; }
    if (esam3_launch_vit_rope(dtype, qkv, rope, (int64_t)B * H * W, H, W, ws, heads, s)) return -1;
  }
  dim3 grid(blocks_for(N, 256), (unsigned)heads, (unsigned)(B * (H / ws) * (W / ws)));
  DISPATCH_T(dtype, hipLaunchKernelGGL((attn_window_kernel<T, 64, 32>), grid, dim3(256), 0, s, (const T*)qkv, ld, q_off,
                                       k_off, v_off, (T*)out, ldo, H, W, ws, heads, 0.125f));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_squeeze_excite(int dtype, void* x, int ld, float* sums, float* gate, const float* w1,
                                const float* b1, const float* w2, const float* b2, int B, int HW, int C, int R,
                                hipStream_t s) {
  if (C % VEC || C / VEC > 256 || C + R > 12288) { esam3_set_error("squeeze_excite: C=%d R=%d", C, R); return -1; }
  const int splits = se_splits(B, HW);
  const int lanes = 256 / (C / VEC);
  DISPATCH_T(dtype, hipLaunchKernelGGL(se_pool_kernel<T>, dim3((unsigned)(B * splits)), dim3(256),
                                       sizeof(float) * (size_t)lanes * C, s, (const T*)x, ld, sums, HW, C, splits));
  hipLaunchKernelGGL(se_fc_kernel, dim3((unsigned)B), dim3(256), sizeof(float) * (C + R), s, sums, splits, w1, b1, w2, b2,
                     gate, C, R, 1.0f / (float)HW);
  const int64_t total = (int64_t)B * HW * (C / VEC);
  DISPATCH_T(dtype, hipLaunchKernelGGL(se_scale_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (T*)x, ld,
                                       gate, HW, C, total));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_resize_bilinear(int dtype, const void* in, void* out, int B, int IH, int IW, int OH,
                                 int OW, int C, hipStream_t s) {
  if (C % VEC) { esam3_set_error("resize: C=%d", C); return -1; }
  const int64_t total = (int64_t)B * OH * OW * (C / VEC);
  DISPATCH_T(dtype, hipLaunchKernelGGL(resize_bilinear_kernel<T>, dim3(blocks_for(total, 256)),
                                       dim3(256), 0, s, (const T*)in, (T*)out, B, IH, IW, OH, OW, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// host-only: the per-axis maps the row-persistent resize_shuffle kernel receives in its kernel arguments (test hook, no device work)
int esam3_resize_axis_tables_host(int in_size, int out_size, int* first, int* count, float* frac) {
  RsAxisTables t;
  if (in_size < 1 || out_size < 1 || !first || !count || !frac || !rs_build_tables(t, in_size, in_size, out_size, out_size)) {
    esam3_set_error("resize_axis_tables: %d -> %d is outside the row kernel's range (<= %d source, <= %d output indices, <= 4 outputs per cell)",
                    in_size, out_size, RsAxisTables::MAX_IN, RsAxisTables::MAX_OUT);
    return -1;
  }
  for (int c = 0; c < in_size; ++c) { first[c] = t.first[0][c]; count[c] = t.count[0][c]; }
  for (int o = 0; o < out_size; ++o) frac[o] = t.frac[0][o];
  return 0;
}

int esam3_launch_resize_shuffle(int dtype, const void* in, const float* bias, void* out, int B, int IH, int IW, int OH, int OW,
                                int C, int taps, int act, int out_pad, hipStream_t s) {
  const int CG = C / VEC;
  if (C % VEC || (taps != 1 && taps != 4) || (CG & (CG - 1))) {
    esam3_set_error("resize_shuffle: C=%d taps=%d (C / 8 must be a power of two)", C, taps);
    return -1;
  }
  int cg_shift = 0;
  while ((1 << cg_shift) < CG) ++cg_shift;
  const int tap_shift = taps == 4 ? 2 : 0;
  const size_t lds = sizeof(int) * 2 * ((size_t)OH + OW) + sizeof(int) * ((size_t)IH + IW);
  if (lds > 48 * 1024 || (int64_t)B * IH > 65535 || IH < 1 || IW < 1 || OH < 1 || OW < 1) {
    esam3_set_error("resize_shuffle: %dx%d -> %dx%d, B=%d out of range", IH, IW, OH, OW, B);
    return -1;
  }
  // one thread per source cell, tap and 8-channel group; grid.y = image x source row
  const dim3 grid((unsigned)(((int64_t)IW * taps * CG + 255) / 256), (unsigned)(B * IH));
  static const bool rows_first = esam3_dev_flag("ESAM3_RS_ROWS_FIRST") != 0;  // A/B: 2 FMAs per output element
  static const bool lds_tables = esam3_dev_flag("ESAM3_RS_LDS_TABLES") != 0;  // A/B: the table form
  // up-scaling by r: a cell owns at most ceil(r) + 1 outputs per axis; the direct form unrolls 4 columns
  const bool direct_ok = (int64_t)OW <= 3 * (int64_t)IW;
  static const bool no_wave = esam3_dev_flag("ESAM3_RS_NO_WAVE") != 0;        // A/B: the per-thread direct form
  RsAxisTables tab;
  // bf16 engine only: the row form blends rows first (fp32 rounding differs from ATen's order, far below a bf16 output step);
  // the fp32 engine keeps ATen's operation order to the letter
  if (dtype == 1 && direct_ok && !lds_tables && !rows_first && !no_wave && taps == 4 && (C == 512 || C == 256) &&
      (act == ACT_NONE || act == ACT_GELU) && rs_build_tables(tab, IH, IW, OH, OW)) {
    // one wave per (image, source row, 512-channel unit, column run): about 4 waves per SIMD over the whole chip
    const int upc = 4 * C / 512;
    int split = 1;
    while ((int64_t)B * IH * upc * split < 4096 && split * 2 <= IW) split *= 2;
    static const int split_env = esam3_dev_flag("ESAM3_RS_SPLIT");
    if (split_env > 0) split = split_env;
    const int64_t units = (int64_t)B * IH * upc * split;
    const dim3 wgrid((unsigned)((units + 3) / 4));
#define ESAM3_RS_ROW(L2, A)                                                                                                              \
  DISPATCH_T(dtype, hipLaunchKernelGGL((resize_shuffle_row_kernel<T, L2, A>), wgrid, dim3(256), 0, s, (const T*)in, bias, (T*)out, B * IH, \
                                       IH, IW, OH, OW, out_pad ? 1 : 0, split, tab, esam3_dev_flag("ESAM3_RS_ABL")))
    if (C == 512 && act == ACT_GELU) ESAM3_RS_ROW(6, ACT_GELU);
    else if (C == 512) ESAM3_RS_ROW(6, ACT_NONE);
    else if (act == ACT_GELU) ESAM3_RS_ROW(5, ACT_GELU);
    else ESAM3_RS_ROW(5, ACT_NONE);
#undef ESAM3_RS_ROW
  } else if (direct_ok && !lds_tables && !rows_first)
    DISPATCH_T(dtype, hipLaunchKernelGGL((resize_shuffle_direct_kernel<T>), grid, dim3(256), 0, s, (const T*)in, bias, (T*)out, IH,
                                         IW, OH, OW, C, taps, act, out_pad ? 1 : 0, cg_shift, tap_shift, esam3_dev_flag("ESAM3_RS_ABL")));
  else if (rows_first)
    DISPATCH_T(dtype, hipLaunchKernelGGL((resize_shuffle_kernel<T, true>), grid, dim3(256), lds, s, (const T*)in, bias, (T*)out, IH,
                                         IW, OH, OW, C, taps, act, out_pad ? 1 : 0, cg_shift, tap_shift));
  else
    DISPATCH_T(dtype, hipLaunchKernelGGL((resize_shuffle_kernel<T, false>), grid, dim3(256), lds, s, (const T*)in, bias, (T*)out, IH,
                                         IW, OH, OW, C, taps, act, out_pad ? 1 : 0, cg_shift, tap_shift));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// in_dtype / out_dtype: 0 f32, 1 bf16 (mixed = the fp32 residual stream of a bf16 engine: fp32 rows in, bf16 rows out)
int esam3_launch_layernorm_io(int in_dtype, int out_dtype, const void* x, const void* res, const float* gamma, const float* beta,
                              void* out, int64_t rows, int C, float eps, int act, hipStream_t s) {
  const int esz = in_dtype == 0 ? 4 : 2, osz = out_dtype == 0 ? 4 : 2;
  const bool aligned = !(((uintptr_t)x) & 15) && !(((uintptr_t)out) & 15) && !(res && (((uintptr_t)res) & 15)) && !(((uintptr_t)gamma) & 15) &&
                       !(((uintptr_t)beta) & 15) && (C * esz) % 16 == 0 &&
                       (C * osz) % 16 == 0;
  static const int no_vec = esam3_dev_flag("ESAM3_NO_LNVEC", 0);  // A/B, bisecting: 1 all, else that C
  const bool mixed = in_dtype != out_dtype;
  if (C % 8 == 0 && C <= 2048 && aligned && rows > 0 && (mixed || !((no_vec == 1 || no_vec == C) && C <= 1024))) {
    const int nchunk = C / 8;
    int lpr = 1;
    while (lpr < nchunk && lpr < 64) lpr <<= 1;
    const int nch = (nchunk + lpr - 1) / lpr;  // <= 4
    const int rpw = 64 / lpr;
    const int64_t groups = (rows + rpw - 1) / rpw;
    // enough waves to fill the chip several times over before a wave starts walking more than one group
    int iters = (int)((groups + 16383) / 16384);
    if (iters > 8) iters = 8;
    const int64_t waves = (groups + iters - 1) / iters;
    const dim3 grid((unsigned)((waves + 3) / 4));
#define ESAM3_LN_IO(TI_, TO_, NCH_)                                                                                            \
  hipLaunchKernelGGL((layernorm_vec_kernel<TI_, TO_, NCH_>), grid, dim3(256), 0, s, (const TI_*)x, (const TI_*)res, gamma, beta, \
                     (TO_*)out, rows, C, eps, act, lpr, iters)
#define ESAM3_LN(NCH_)                                                                       \
  do {                                                                                       \
    if (in_dtype == 0 && out_dtype == 0) ESAM3_LN_IO(float, float, NCH_);                    \
    else if (in_dtype == 1 && out_dtype == 1) ESAM3_LN_IO(bf16_t, bf16_t, NCH_);             \
    else if (in_dtype == 0) ESAM3_LN_IO(float, bf16_t, NCH_);                                \
    else ESAM3_LN_IO(bf16_t, float, NCH_);                                                   \
  } while (0)
    if (nch == 1) { ESAM3_LN(1); } else if (nch == 2) { ESAM3_LN(2); } else if (nch == 3) { ESAM3_LN(3); } else { ESAM3_LN(4); }
#undef ESAM3_LN
#undef ESAM3_LN_IO
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  if (mixed) { esam3_set_error("layernorm: mixed in/out dtypes need C %% 8 == 0, C <= 2048 and 16-byte aligned rows (C=%d)", C); return -1; }
  const int dtype = in_dtype;
  if (C > 1024) { esam3_set_error("layernorm: C=%d > 1024", C); return -1; }
  DISPATCH_T(dtype, hipLaunchKernelGGL(layernorm_kernel<T>, dim3(blocks_for(rows, 4)), dim3(256), 0, s,
                                       (const T*)x, (const T*)res, gamma, beta, (T*)out, rows, C, eps,
                                       act));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_layernorm(int dtype, const void* x, const void* res, const float* gamma, const float* beta, void* out, int64_t rows,
                           int C, float eps, int act, hipStream_t s) {
  return esam3_launch_layernorm_io(dtype, dtype, x, res, gamma, beta, out, rows, C, eps, act, s);
}

int esam3_launch_cast_to_f32(int dtype, const void* in, float* out, int64_t n, hipStream_t s) {
  const unsigned g = (unsigned)min((int64_t)4096, (n + 255) / 256);
  if (n == 0) return 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_to_f32_kernel<T>, dim3(g), dim3(256), 0, s, (const T*)in,
                                       out, n));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_cast_from_f32(int dtype, const float* in, void* out, int64_t n, hipStream_t s) {
  const unsigned g = (unsigned)min((int64_t)4096, (n + 255) / 256);
  if (n == 0) return 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_from_f32_kernel<T>, dim3(g), dim3(256), 0, s, in, (T*)out,
                                       n));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_nhwc_to_nchw_f32(int dtype, const void* in, float* out, int B, int H, int W, int C,
                                  hipStream_t s) {
  const int HW = H * W;
  dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
  DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel<T>, grid, dim3(256), 0, s,
                                       (const T*)in, out, HW, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
