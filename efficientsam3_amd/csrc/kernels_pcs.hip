// Kernels of the PCS text-grounding detector (fusion encoder / DETR decoder / segmentation head glue).
// The heavy lifting is the shared implicit-GEMM, LayerNorm and attention code; what lives here is the
// generic 8-head x 32 attention core with key-padding mask and the decoder's separable box-relative
// position bias, the box / sine-embedding arithmetic, GroupNorm, and small layout helpers.
#include "kernels.h"

namespace {

constexpr int VEC = 8;
inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }


// ------------------------------------------------------------------------------------
// nn.MultiheadAttention core (after the input projections): heads x 32, softmax in fp32.
//   q rows [B*Nq][ldq] (+q_off), k / v rows [B*Nk][ldk] (+k_off / +v_off), head h at +h*32;
//   key_mask [B][Nk] (1 = ignore) or null; optional separable additive bias for image keys j = y*Wk + x:
//   bias_y [B][heads][Nq][Hk] + bias_x [B][heads][Nq][Wk] (decoder.py:333-415; rows with q < bias_q0 get none).
// One query per thread; keys / values stream through LDS in chunks of KC (broadcast reads).
// ------------------------------------------------------------------------------------
template <typename T, int KC>
__global__ __launch_bounds__(128) void mha_core_kernel(const T* __restrict__ q, int ldq, int q_off, const T* __restrict__ kv,
                                                       int ldk, int k_off, int v_off, T* __restrict__ out, int ldo,
                                                       int Nq, int Nk, int heads, const uint8_t* __restrict__ key_mask,
                                                       const float* __restrict__ bias_y, const float* __restrict__ bias_x,
                                                       int Hk, int Wk, int bias_q0) {
  constexpr int HD = 32;
  __shared__ float sk[KC][HD];
  __shared__ float sv[KC][HD];
  __shared__ uint8_t sm[KC];
  const int h = blockIdx.y;
  const int64_t b = blockIdx.z;
  const int qi = blockIdx.x * 128 + threadIdx.x;
  const bool valid = qi < Nq;
  float qr[HD], acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) { qr[d] = 0.f; acc[d] = 0.f; }
  if (valid) {
    const T* src = q + (b * Nq + qi) * (int64_t)ldq + q_off + h * HD;
#pragma unroll
    for (int c = 0; c < HD / VEC; ++c) Vec8<T>::load(src + c * VEC, qr + c * VEC);
#pragma unroll
    for (int d = 0; d < HD; ++d) qr[d] *= 0.17677669529663687f;  // 32^-0.5
  }
  const bool biased = bias_y != nullptr && valid && qi >= bias_q0;
  const float* by = biased ? bias_y + ((b * heads + h) * Nq + qi) * (int64_t)Hk : nullptr;
  const float* bx = biased ? bias_x + ((b * heads + h) * Nq + qi) * (int64_t)Wk : nullptr;
  float mx = -INFINITY, sum = 0.f;
  for (int j0 = 0; j0 < Nk; j0 += KC) {
    __syncthreads();
    for (int i = threadIdx.x; i < KC * (HD / VEC); i += 128) {
      const int j = i / (HD / VEC), c = i - j * (HD / VEC);
      float kk[VEC], vv[VEC];
      if (j0 + j < Nk) {
        const T* src = kv + (b * Nk + j0 + j) * (int64_t)ldk + h * HD + c * VEC;
        Vec8<T>::load(src + k_off, kk);
        Vec8<T>::load(src + v_off, vv);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { kk[e] = 0.f; vv[e] = 0.f; }
      }
#pragma unroll
      for (int e = 0; e < VEC; ++e) { sk[j][c * VEC + e] = kk[e]; sv[j][c * VEC + e] = vv[e]; }
    }
    for (int j = threadIdx.x; j < KC; j += 128) sm[j] = (j0 + j < Nk) ? (key_mask ? key_mask[b * Nk + j0 + j] : 0) : 1;
    __syncthreads();
    const int jn = min(KC, Nk - j0);
    for (int j = 0; j < jn; ++j) {
      if (sm[j]) continue;  // padded key: weight exp(-inf) = 0
      const float4* kr = reinterpret_cast<const float4*>(sk[j]);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int c = 0; c < HD / 4; c += 2) {
        const float4 a = kr[c], bq = kr[c + 1];
        s0 = fmaf(qr[4 * c], a.x, s0); s0 = fmaf(qr[4 * c + 1], a.y, s0);
        s0 = fmaf(qr[4 * c + 2], a.z, s0); s0 = fmaf(qr[4 * c + 3], a.w, s0);
        s1 = fmaf(qr[4 * c + 4], bq.x, s1); s1 = fmaf(qr[4 * c + 5], bq.y, s1);
        s1 = fmaf(qr[4 * c + 6], bq.z, s1); s1 = fmaf(qr[4 * c + 7], bq.w, s1);
      }
      float sc = s0 + s1;
      if (biased) {
        const int kj = j0 + j, ky = kj / Wk, kx = kj - ky * Wk;
        sc += by[ky] + bx[kx];
      }
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
  T* dst = out + (b * Nq + qi) * (int64_t)ldo + h * HD;
#pragma unroll
  for (int c = 0; c < HD / VEC; ++c) {
    float o[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = acc[c * VEC + e] * inv;
    Vec8<T>::store(dst + c * VEC, o);
  }
}

// text prompt tokens: fp32 [S][B][C] (esam3_encode_text layout) -> T [B][S+extra][C] rows 0..S-1; the mask
// [B][S] is widened to [B][S+extra] with the extra (geometry CLS) slots valid
template <typename T>
__global__ void pcs_prompt_kernel(const float* __restrict__ lang, const uint8_t* __restrict__ lmask, T* __restrict__ prompt,
                                  uint8_t* __restrict__ pmask, int B, int S, int extra, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int St = S + extra;
  if (i < (int64_t)B * St) {
    const int b = (int)(i / St), s_ = (int)(i % St);
    pmask[i] = s_ < S ? lmask[b * S + s_] : 0;
  }
  if (i >= (int64_t)B * S * C) return;
  const int c = (int)(i % C);
  const int64_t bs = i / C;
  const int s_ = (int)(bs % S), b = (int)(bs / S);
  prompt[((int64_t)b * St + s_) * C + c] = from_f32<T>(lang[((int64_t)s_ * B + b) * C + c]);
}

// rows [B][n_src][C] -> rows dst_row0.. of [B][n_dst][C]
template <typename T>
__global__ void copy_rows_kernel(const T* __restrict__ src, int n_src, T* __restrict__ dst, int n_dst, int dst_row0, int B,
                                 int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * n_src * C) return;
  const int c = (int)(i % C);
  const int64_t br = i / C;
  const int r = (int)(br % n_src), b = (int)(br / n_src);
  dst[((int64_t)b * n_dst + dst_row0 + r) * C + c] = src[i];
}

// fp32 [n][C] parameter rows broadcast to every image: dst [B][n_dst][C] rows dst_row0..dst_row0+n-1
template <typename T>
__global__ void bcast_rows_kernel(const float* __restrict__ src, int n, T* __restrict__ dst, int n_dst, int dst_row0, int B,
                                  int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * n * C) return;
  const int c = (int)(i % C);
  const int64_t br = i / C;
  const int r = (int)(br % n), b = (int)(br / n);
  dst[((int64_t)b * n_dst + dst_row0 + r) * C + c] = from_f32<T>(src[(int64_t)r * C + c]);
}

// gen_sineembed_for_position (model_misc.py:238-275) of cxcywh boxes -> [rows][512] = (y, x, w, h) x 128;
// rows flagged `skip_row0` (the presence token of every image) get zeros.
template <typename T>
__global__ void box_sine_kernel(const float* __restrict__ boxes, T* __restrict__ out, int64_t rows, int rows_per_img) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, coord 0..3, pair 0..63)
  if (i >= rows * 256) return;
  const int pr = (int)(i % 64), which = (int)((i / 64) % 4);
  const int64_t row = i / 256;
  const int src_c = which == 0 ? 1 : (which == 1 ? 0 : which);  // output order y, x, w, h
  T* dst = out + row * 512 + which * 128 + 2 * pr;
  if (row % rows_per_img == 0) { dst[0] = from_f32<T>(0.f); dst[1] = from_f32<T>(0.f); return; }
  const float v = boxes[row * 4 + src_c] * 6.283185307179586f;
  const float dim_t = powf(10000.f, (float)(2 * pr) / 128.f);  // 10000^(2*floor(k/2)/128), k = 2pr, 2pr+1
  const float a = v / dim_t;
  dst[0] = from_f32<T>(sinf(a));
  dst[1] = from_f32<T>(cosf(a));
}

// SequenceGeometryEncoder token assembly (geometry_encoders.py:600-695,790-815): one workgroup per (token slot,
// image), one thread per channel.  Slot order per image: its valid points, its valid boxes, the CLS token, padding.
//   x0 [B][Lg][256]      direct projections + all Linear biases + label embedding (or cls_embed); 0 for padding
//   a_samp [B][Lg][256]  bilinear grid_sample of the pre-normed image features at the point (zero padding,
//                        align_corners=False);  a_encp [B][Lg][256] = [sine(x) | sine(y)]
//   a_roi [B][Lg][49][256]  roi_align 7x7 (torchvision defaults: scale 1, adaptive sampling, aligned=False) of the
//                        box in feature pixels;  a_encb [B][Lg][264] = [sine(cy) | sine(cx) | h | w | 0...]
// The four A matrices are multiplied by the pool / pos-enc projection weights (GEMMs without bias) and summed
// into x0 by the caller; rows of the other kinds stay zero.  gmask [B][ld_mask] (+mask_off) and gmask_dense
// [B][Lg]: 1 = padding.
template <typename T>
__global__ __launch_bounds__(256) void geo_tokens_kernel(
    const float* __restrict__ points, const int32_t* __restrict__ plabels, const uint8_t* __restrict__ pmask, int Np,
    const float* __restrict__ boxes, const int32_t* __restrict__ blabels, const uint8_t* __restrict__ bmask, int Nb,
    const T* __restrict__ imgn, int H, int W, const float* __restrict__ w_pd, const float* __restrict__ b_pt,
    const float* __restrict__ w_bd, const float* __restrict__ b_bx, const float* __restrict__ label_embed,
    const float* __restrict__ cls, T* __restrict__ x0, T* __restrict__ a_samp, T* __restrict__ a_encp,
    T* __restrict__ a_roi, T* __restrict__ a_encb, uint8_t* __restrict__ gmask, int ld_mask, int mask_off,
    uint8_t* __restrict__ gmask_dense) {
  constexpr int C = 256;
  const int t = blockIdx.x, b = blockIdx.y, c = threadIdx.x, Lg = Np + Nb + 1;
  int np = 0, nb = 0;
  for (int i = 0; i < Np; ++i) np += pmask ? (pmask[b * Np + i] == 0) : 1;
  for (int i = 0; i < Nb; ++i) nb += bmask ? (bmask[b * Nb + i] == 0) : 1;
  const int64_t row = (int64_t)b * Lg + t;
  float xv = 0.f, sv = 0.f, ev = 0.f;
  const bool is_pt = t < np, is_box = !is_pt && t < np + nb, is_cls = t == np + nb;
  if (c == 0) {
    gmask[(int64_t)b * ld_mask + mask_off + t] = t > np + nb;
    gmask_dense[(int64_t)b * Lg + t] = t > np + nb;
  }
  const T* fimg = imgn + (int64_t)b * H * W * C;
  const float dim_t = powf(10000.f, (float)(2 * ((c & 127) >> 1)) / 128.f);
  if (is_pt) {
    const float px = points[((int64_t)b * Np + t) * 2], py = points[((int64_t)b * Np + t) * 2 + 1];
    xv = fmaf(w_pd[c * 2], px, fmaf(w_pd[c * 2 + 1], py, b_pt[c])) + label_embed[plabels[b * Np + t] * C + c];
    const float ix = px * (float)W - 0.5f, iy = py * (float)H - 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0i = (int)fx0, y0i = (int)fy0;
    const float lx = ix - fx0, ly = iy - fy0;
    auto tap = [&](int yy, int xx) -> float {
      return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? to_f32<T>(fimg[((int64_t)yy * W + xx) * C + c]) : 0.f;
    };
    sv = tap(y0i, x0i) * (1.f - ly) * (1.f - lx) + tap(y0i, x0i + 1) * (1.f - ly) * lx + tap(y0i + 1, x0i) * ly * (1.f - lx) +
         tap(y0i + 1, x0i + 1) * ly * lx;
    const float v = (c < 128 ? px : py) * 6.283185307179586f / dim_t;
    ev = (c & 1) ? cosf(v) : sinf(v);
  } else if (is_box) {
    const int j = t - np;
    const float* bx = boxes + ((int64_t)b * Nb + j) * 4;
    const float cx = bx[0], cy = bx[1], bw = bx[2], bh = bx[3];
    xv = fmaf(w_bd[c * 4], cx, fmaf(w_bd[c * 4 + 1], cy, fmaf(w_bd[c * 4 + 2], bw, fmaf(w_bd[c * 4 + 3], bh, b_bx[c])))) +
         label_embed[blabels[b * Nb + j] * C + c];
    const float v = (c < 128 ? cy : cx) * 6.283185307179586f / dim_t;
    a_encb[row * 264 + c] = from_f32<T>((c & 1) ? cosf(v) : sinf(v));
    if (c < 8) a_encb[row * 264 + 256 + c] = from_f32<T>(c == 0 ? bh : (c == 1 ? bw : 0.f));
    // roi_align: box corners in feature pixels, sides clamped to >= 1, ceil(side / 7) samples per bin and axis
    const float x1 = (cx - 0.5f * bw) * (float)W, y1 = (cy - 0.5f * bh) * (float)H;
    const float x2 = (cx + 0.5f * bw) * (float)W, y2 = (cy + 0.5f * bh) * (float)H;
    const float rw = fmaxf(x2 - x1, 1.f), rh = fmaxf(y2 - y1, 1.f);
    const float bin_h = rh / 7.f, bin_w = rw / 7.f;
    const int gh = (int)ceilf(rh / 7.f), gw = (int)ceilf(rw / 7.f);
    const float inv = 1.f / (float)max(gh * gw, 1);
    for (int phh = 0; phh < 7; ++phh)
      for (int pww = 0; pww < 7; ++pww) {
        float acc = 0.f;
        for (int iy = 0; iy < gh; ++iy) {
          float y = y1 + (float)phh * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
          if (y < -1.f || y > (float)H) continue;
          y = fmaxf(y, 0.f);
          int yl = (int)y, yh;
          if (yl >= H - 1) { yl = yh = H - 1; y = (float)yl; } else yh = yl + 1;
          const float ly = y - (float)yl, hy = 1.f - ly;
          for (int ixx = 0; ixx < gw; ++ixx) {
            float x = x1 + (float)pww * bin_w + ((float)ixx + 0.5f) * bin_w / (float)gw;
            if (x < -1.f || x > (float)W) continue;
            x = fmaxf(x, 0.f);
            int xl = (int)x, xh;
            if (xl >= W - 1) { xl = xh = W - 1; x = (float)xl; } else xh = xl + 1;
            const float lx = x - (float)xl, hx = 1.f - lx;
            acc += hy * hx * to_f32<T>(fimg[((int64_t)yl * W + xl) * C + c]) + hy * lx * to_f32<T>(fimg[((int64_t)yl * W + xh) * C + c]) +
                   ly * hx * to_f32<T>(fimg[((int64_t)yh * W + xl) * C + c]) + ly * lx * to_f32<T>(fimg[((int64_t)yh * W + xh) * C + c]);
          }
        }
        a_roi[(row * 49 + phh * 7 + pww) * C + c] = from_f32<T>(acc * inv);
      }
  } else if (is_cls) {
    xv = cls[c];
  }
  x0[row * C + c] = from_f32<T>(xv);
  a_samp[row * C + c] = from_f32<T>(sv);
  a_encp[row * C + c] = from_f32<T>(ev);
  if (!is_box) {
    a_encb[row * 264 + c] = from_f32<T>(0.f);
    if (c < 8) a_encb[row * 264 + 256 + c] = from_f32<T>(0.f);
    for (int k = 0; k < 49; ++k) a_roi[(row * 49 + k) * C + c] = from_f32<T>(0.f);
  }
}

// boxRPB "log" features -> MLP(2 -> 256 -> heads) (decoder.py:333-415): out_y [B][heads][Q][H], out_x [B][heads][Q][W] fp32
// (Q = nq_img query rows per image).  w1 [256][2], b1 [256], w2 [heads][256], b2 [heads]; heads <= 8.
// A thread owns FOUR consecutive coordinates of one (query row, axis) and walks the 256 hidden units once: per unit three 16-byte
// LDS reads (w1 pair + b1, the unit's 8 head weights) feed 4 x (2 + 8) FMAs.  (The first version -- one coordinate per thread,
// 11 scalar LDS reads per 11 FMAs -- was LDS-issue-bound: 0.33 ms per decoder layer at B = 8 for 1.3 GFLOP.)  Same summation order
// per output as before.
__global__ __launch_bounds__(256) void rpb_mlp_kernel(const float* __restrict__ boxes, const float* __restrict__ w1x,
                                                      const float* __restrict__ b1x, const float* __restrict__ w2x,
                                                      const float* __restrict__ b2x, const float* __restrict__ w1y,
                                                      const float* __restrict__ b1y, const float* __restrict__ w2y,
                                                      const float* __restrict__ b2y, float* __restrict__ out_y,
                                                      float* __restrict__ out_x, int64_t nq_total, int nq_img, int H, int W, int heads) {
  // per axis: wk [256] float4 {w1[k][0], w1[k][1], b1[k], 0} | w2t [256][8] (unit-major, heads padded to 8) | b2 [8]
  extern __shared__ __attribute__((aligned(16))) float sw[];
  constexpr int PER = 256 * 4 + 256 * 8 + 8;
  for (int i = threadIdx.x; i < 2 * PER; i += 256) {
    const int ax = i / PER, k = i - ax * PER;
    const float* w1 = ax ? w1y : w1x; const float* b1 = ax ? b1y : b1x;
    const float* w2 = ax ? w2y : w2x; const float* b2 = ax ? b2y : b2x;
    float v = 0.f;
    if (k < 1024) {
      const int u = k >> 2, c = k & 3;
      v = c < 2 ? w1[2 * u + c] : (c == 2 ? b1[u] : 0.f);
    } else if (k < 1024 + 2048) {
      const int u = (k - 1024) >> 3, hh = (k - 1024) & 7;
      v = hh < heads ? w2[hh * 256 + u] : 0.f;
    } else {
      const int hh = k - 3072;
      v = hh < heads ? b2[hh] : 0.f;
    }
    sw[i] = v;
  }
  __syncthreads();
  const int cW = (W + 3) >> 2, cH = (H + 3) >> 2, CL = cW + cH;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // (query row, axis, group of 4 coordinates)
  if (i >= nq_total * CL) return;
  const int t = (int)(i % CL);
  const int64_t qrow = i / CL;
  const int ax = t < cW ? 0 : 1;            // 0: x (W coordinates), 1: y (H coordinates)
  const int c0 = (ax ? t - cW : t) * 4;
  const int len = ax ? H : W;
  const float cx = boxes[qrow * 4 + 0], cy = boxes[qrow * 4 + 1], bw = boxes[qrow * 4 + 2], bh = boxes[qrow * 4 + 3];
  const float lo = ax ? cy - 0.5f * bh : cx - 0.5f * bw, hi = ax ? cy + 0.5f * bh : cx + 0.5f * bw;
  float d0[4], d1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float coord = (float)(c0 + j) / (float)len;
    float d[2] = {coord - lo, coord - hi};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float v = d[k] * 8.f;
      const float sgn = v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f);
      d[k] = sgn * log2f(fabsf(v) + 1.0f) / 3.0f;  // / log2(8)
    }
    d0[j] = d[0];
    d1[j] = d[1];
  }
  const float4* wk = reinterpret_cast<const float4*>(sw + ax * PER);
  const float4* w2t = reinterpret_cast<const float4*>(sw + ax * PER + 1024);
  const float* b2 = sw + ax * PER + 3072;
  float o[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) o[j][hh] = b2[hh];
  for (int k = 0; k < 256; ++k) {
    const float4 w = wk[k], wa = w2t[2 * k], wb = w2t[2 * k + 1];
    const float w2v[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float hdn = fmaf(w.x, d0[j], fmaf(w.y, d1[j], w.z));
      hdn = hdn > 0.f ? hdn : 0.f;
#pragma unroll
      for (int hh = 0; hh < 8; ++hh) o[j][hh] = fmaf(w2v[hh], hdn, o[j][hh]);
    }
  }
  const int64_t bimg = qrow / nq_img, qi = qrow - bimg * nq_img;
  float* dst = (ax ? out_y : out_x) + ((bimg * heads) * nq_img + qi) * len + c0;
  for (int hh = 0; hh < heads; ++hh)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (c0 + j < len) dst[(int64_t)hh * nq_img * len + j] = o[j][hh];
}

// box refinement (decoder.py:560-581): ref <- sigmoid(delta + inverse_sigmoid(ref)), delta rows [rows][ld] (T)
template <typename T>
__global__ void box_refine_kernel(const T* __restrict__ delta, int ld, float* __restrict__ ref, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (row, coordinate)
  if (i >= n) return;
  const int64_t row = i / 4;
  const int c = (int)(i % 4);
  float x = ref[i];
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float inv = logf(fmaxf(x, 1e-3f) / fmaxf(1.f - x, 1e-3f));
  const float v = to_f32<T>(delta[row * ld + c]) + inv;
  ref[i] = 1.f / (1.f + expf(-v));
}

// DotProductScoring tail (model_misc.py:52-91): masked mean of the prompt rows, then (after the two
// projections done by GEMMs) score[b][q] = clamp(<hs_proj[b][q], prompt_proj[b]> / 16, -12, 12)
template <typename T>
__global__ void masked_mean_kernel(const T* __restrict__ x, const uint8_t* __restrict__ mask, T* __restrict__ out, int S,
                                   int C) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float a = 0.f, n = 0.f;
    for (int s_ = 0; s_ < S; ++s_) {
      if (mask[b * S + s_]) continue;
      a += to_f32<T>(x[((int64_t)b * S + s_) * C + c]);
      n += 1.f;
    }
    out[(int64_t)b * C + c] = from_f32<T>(a / fmaxf(n, 1.f));
  }
}
template <typename T>
__global__ void dot_score_kernel(const T* __restrict__ hs, int rows_per_img, int row0, int nq, const T* __restrict__ pp,
                                 float* __restrict__ out, int B, int C, float scale, float clampv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (b, q)
  if (i >= B * nq) return;
  const int b = i / nq, q_ = i - b * nq;
  const T* h = hs + ((int64_t)b * rows_per_img + row0 + q_) * C;
  const T* p = pp + (int64_t)b * C;
  float a = 0.f;
  for (int c = 0; c < C; ++c) a = fmaf(to_f32<T>(h[c]), to_f32<T>(p[c]), a);
  a *= scale;
  out[i] = fminf(fmaxf(a, -clampv), clampv);
}

// PixelDecoder step (maskformer_segmentation.py:206-216): out = fine + nearest_upsample_x2(coarse), written
// into the interior of a zero-bordered buffer [B][2h+2][2w+2][C] for the 3x3 conv that follows
template <typename T>
__global__ void upsample_add_kernel(const T* __restrict__ fine, const T* __restrict__ coarse, T* __restrict__ out_padded,
                                    int B, int h, int w, int C) {
  const int CG = C / VEC;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int H2 = 2 * h, W2 = 2 * w;
  if (i >= (int64_t)B * H2 * W2 * CG) return;
  const int cg = (int)(i % CG);
  const int64_t pix = i / CG;
  const int x = (int)(pix % W2), y = (int)((pix / W2) % H2);
  const int64_t b = pix / ((int64_t)W2 * H2);
  float a[VEC], c_[VEC];
  Vec8<T>::load(fine + pix * C + cg * VEC, a);
  Vec8<T>::load(coarse + ((b * h + y / 2) * (int64_t)w + x / 2) * C + cg * VEC, c_);
#pragma unroll
  for (int e = 0; e < VEC; ++e) a[e] += c_[e];
  Vec8<T>::store(out_padded + ((b * (H2 + 2) + y + 1) * (int64_t)(W2 + 2) + x + 1) * C + cg * VEC, a);
}

// GroupNorm(groups, C) + ReLU over [B][HW][C] (maskformer_segmentation.py:186,216): deterministic two-level
// statistics (per-split partial sums, summed in order), then normalise in place.
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, float* __restrict__ partial, int HW, int C,
                                                       int groups, int splits) {
  __shared__ float rs[256], rq[256];
  const int sp = blockIdx.x % splits, g = (blockIdx.x / splits) % groups;
  const int64_t b = blockIdx.x / (splits * groups);
  const int gc = C / groups, cgs = gc / VEC;  // 8-channel chunks per group
  const int per = (HW + splits - 1) / splits;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  float s_ = 0.f, q_ = 0.f;
  for (int64_t i = threadIdx.x; i < (int64_t)(p1 - p0) * cgs; i += 256) {
    const int p = p0 + (int)(i / cgs), cc = (int)(i % cgs);
    float v[VEC];
    Vec8<T>::load(x + (b * HW + p) * (int64_t)C + g * gc + cc * VEC, v);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { s_ += v[e]; q_ = fmaf(v[e], v[e], q_); }
  }
  rs[threadIdx.x] = s_; rq[threadIdx.x] = q_;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rq[threadIdx.x] += rq[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[((b * groups + g) * splits + sp) * 2] = rs[0];
    partial[((b * groups + g) * splits + sp) * 2 + 1] = rq[0];
  }
}
template <typename T>
__global__ void gn_apply_relu_kernel(T* __restrict__ x, const float* __restrict__ partial, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, int HW, int C, int groups, int splits, float eps,
                                     int64_t total) {
  const int CG = C / VEC;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int cg = (int)(i % CG);
  const int64_t pix = i / CG;
  const int64_t b = pix / HW;
  const int gc = C / groups, g = cg * VEC / gc;
  float s_ = 0.f, q_ = 0.f;
  for (int sp = 0; sp < splits; ++sp) {
    s_ += partial[((b * groups + g) * splits + sp) * 2];
    q_ += partial[((b * groups + g) * splits + sp) * 2 + 1];
  }
  const float n = (float)HW * (float)gc;
  const float mean = s_ / n;
  const float var = fmaxf(q_ / n - mean * mean, 0.f);
  const float inv = 1.f / sqrtf(var + eps);
  float v[VEC];
  T* p = x + pix * C + cg * VEC;
  Vec8<T>::load(p, v);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float y = (v[e] - mean) * inv * gamma[cg * VEC + e] + beta[cg * VEC + e];
    v[e] = y > 0.f ? y : 0.f;
  }
  Vec8<T>::store(p, v);
}

// Round 6: the statistics of a (image, group) are finalised ONCE (gn_finalize_kernel: the same sequential sum over the splits, the same mean /
// variance / rsqrt expressions as gn_apply_relu_kernel above, which every thread used to repeat -- 128 global loads per 8 output channels:
// 472 us per launch at 288^2 x 256, B = 8, against 136 us of memory time); gn_apply_relu_fin_kernel then reads two floats per thread.  grid.y =
// image: no 64-bit divisions.  Per-element arithmetic unchanged: bit-identical (dev builds keep the old kernel behind ESAM3_GN_OLD).
__global__ void gn_finalize_kernel(const float* __restrict__ partial, float* __restrict__ stats, int BG, int splits, float n, float eps) {
  const int bg = blockIdx.x * blockDim.x + threadIdx.x;
  if (bg >= BG) return;
  float s_ = 0.f, q_ = 0.f;
  for (int sp = 0; sp < splits; ++sp) {
    s_ += partial[((int64_t)bg * splits + sp) * 2];
    q_ += partial[((int64_t)bg * splits + sp) * 2 + 1];
  }
  const float mean = s_ / n;
  const float var = fmaxf(q_ / n - mean * mean, 0.f);
  const float inv = 1.f / sqrtf(var + eps);
  stats[2 * bg] = mean;
  stats[2 * bg + 1] = inv;
}
template <typename T>
__global__ void gn_apply_relu_fin_kernel(T* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, int HW, int C, int groups) {
  const int CG = C / VEC;
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;   // (pixel, channel group) of image blockIdx.y
  if (i >= (unsigned)HW * (unsigned)CG) return;
  const int cg = (int)(i % (unsigned)CG);
  const unsigned pix = i / (unsigned)CG;
  const int b = blockIdx.y;
  const int gc = C / groups, g = cg * VEC / gc;
  const float mean = stats[2 * (b * groups + g)], inv = stats[2 * (b * groups + g) + 1];
  float v[VEC];
  T* p = x + ((int64_t)b * HW + pix) * C + cg * VEC;
  Vec8<T>::load(p, v);
#pragma unroll
  for (int e = 0; e < VEC; ++e) {
    const float y = (v[e] - mean) * inv * gamma[cg * VEC + e] + beta[cg * VEC + e];
    v[e] = y > 0.f ? y : 0.f;
  }
  Vec8<T>::store(p, v);
}

}  // namespace

#define DISPATCH_T(dtype, ...)                  \
  do {                                          \
    if ((dtype) == 0) { using T = float; __VA_ARGS__; } \
    else { using T = bf16_t; __VA_ARGS__; }     \
  } while (0)

int esam3_launch_mha_core(int dtype, const void* q, int ldq, int q_off, const void* kv, int ldk, int k_off, int v_off,
                          void* out, int ldo, int B, int Nq, int Nk, int heads, const uint8_t* key_mask,
                          const float* bias_y, const float* bias_x, int Hk, int Wk, int bias_q0, hipStream_t s) {
  if (heads > 8 && bias_y) { esam3_set_error("mha_core: heads=%d with bias", heads); return -1; }
  dim3 grid(blocks_for(Nq, 128), (unsigned)heads, (unsigned)B);
  DISPATCH_T(dtype, hipLaunchKernelGGL((mha_core_kernel<T, 64>), grid, dim3(128), 0, s, (const T*)q, ldq, q_off,
                                       (const T*)kv, ldk, k_off, v_off, (T*)out, ldo, Nq, Nk, heads, key_mask, bias_y, bias_x,
                                       Hk, Wk, bias_q0));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_pcs_prompt(int dtype, const float* lang, const uint8_t* lmask, void* prompt, uint8_t* pmask, int B, int S,
                            int extra, int C, hipStream_t s) {
  const int64_t total = (int64_t)B * (S + extra) * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(pcs_prompt_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, lang, lmask,
                                       (T*)prompt, pmask, B, S, extra, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_copy_rows(int dtype, const void* src, int n_src, void* dst, int n_dst, int dst_row0, int B, int C,
                           hipStream_t s) {
  const int64_t total = (int64_t)B * n_src * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(copy_rows_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const T*)src,
                                       n_src, (T*)dst, n_dst, dst_row0, B, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_bcast_rows(int dtype, const float* src, int n, void* dst, int n_dst, int dst_row0, int B, int C,
                            hipStream_t s) {
  const int64_t total = (int64_t)B * n * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(bcast_rows_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, src, n, (T*)dst,
                                       n_dst, dst_row0, B, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_box_sine(int dtype, const float* boxes, void* out, int64_t rows, int rows_per_img, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(box_sine_kernel<T>, dim3(blocks_for(rows * 256, 256)), dim3(256), 0, s, boxes,
                                       (T*)out, rows, rows_per_img));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_geo_tokens(int dtype, const float* points, const int32_t* plabels, const uint8_t* pmask, int Np,
                            const float* boxes, const int32_t* blabels, const uint8_t* bmask, int Nb, const void* imgn, int H,
                            int W, const float* w_pd, const float* b_pt, const float* w_bd, const float* b_bx,
                            const float* label_embed, const float* cls, void* x0, void* a_samp, void* a_encp, void* a_roi,
                            void* a_encb, uint8_t* gmask, int ld_mask, int mask_off, uint8_t* gmask_dense, int B, hipStream_t s) {
  dim3 grid((unsigned)(Np + Nb + 1), (unsigned)B);
  DISPATCH_T(dtype, hipLaunchKernelGGL(geo_tokens_kernel<T>, grid, dim3(256), 0, s, points, plabels, pmask, Np, boxes, blabels,
                                       bmask, Nb, (const T*)imgn, H, W, w_pd, b_pt, w_bd, b_bx, label_embed, cls, (T*)x0,
                                       (T*)a_samp, (T*)a_encp, (T*)a_roi, (T*)a_encb, gmask, ld_mask, mask_off, gmask_dense));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_rpb_mlp(const float* boxes, const float* const* wx, const float* const* wy, float* out_y, float* out_x,
                         int64_t nq_total, int nq_img, int H, int W, int heads, hipStream_t s) {
  if (heads < 1 || heads > 8) { esam3_set_error("rpb_mlp: heads=%d (<= 8 supported)", heads); return -1; }
  const size_t lds = sizeof(float) * 2 * (size_t)(256 * 4 + 256 * 8 + 8);
  hipLaunchKernelGGL(rpb_mlp_kernel, dim3(blocks_for(nq_total * ((W + 3) / 4 + (H + 3) / 4), 256)), dim3(256), lds, s, boxes, wx[0], wx[1],
                     wx[2], wx[3], wy[0], wy[1], wy[2], wy[3], out_y, out_x, nq_total, nq_img, H, W, heads);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_box_refine(int dtype, const void* delta, int ld, float* ref, int64_t rows, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(box_refine_kernel<T>, dim3(blocks_for(rows * 4, 256)), dim3(256), 0, s,
                                       (const T*)delta, ld, ref, rows * 4));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_masked_mean(int dtype, const void* x, const uint8_t* mask, void* out, int B, int S, int C, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(masked_mean_kernel<T>, dim3((unsigned)B), dim3(256), 0, s, (const T*)x, mask, (T*)out,
                                       S, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_dot_score(int dtype, const void* hs, int rows_per_img, int row0, int nq, const void* pp, float* out, int B,
                           int C, float scale, float clampv, hipStream_t s) {
  DISPATCH_T(dtype, hipLaunchKernelGGL(dot_score_kernel<T>, dim3(blocks_for((int64_t)B * nq, 128)), dim3(128), 0, s,
                                       (const T*)hs, rows_per_img, row0, nq, (const T*)pp, out, B, C, scale, clampv));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_upsample_add(int dtype, const void* fine, const void* coarse, void* out_padded, int B, int h, int w, int C,
                              hipStream_t s) {
  const int64_t total = (int64_t)B * 4 * h * w * (C / VEC);
  DISPATCH_T(dtype, hipLaunchKernelGGL(upsample_add_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const T*)fine,
                                       (const T*)coarse, (T*)out_padded, B, h, w, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int64_t esam3_groupnorm_scratch_floats(int B, int groups) { return (int64_t)B * groups * 64 * 2 + (int64_t)B * groups * 2; }   // 64 splits of (sum, sum of squares) + the finalised (mean, 1 / std)
int esam3_launch_groupnorm_relu(int dtype, void* x, float* partial, const float* gamma, const float* beta, int B, int HW,
                                int C, int groups, float eps, hipStream_t s) {
  if (C % groups || (C / groups) % VEC) { esam3_set_error("groupnorm: C=%d groups=%d", C, groups); return -1; }
  const int splits = 64;
  DISPATCH_T(dtype, hipLaunchKernelGGL(gn_stats_kernel<T>, dim3((unsigned)(B * groups * splits)), dim3(256), 0, s,
                                       (const T*)x, partial, HW, C, groups, splits));
  const int64_t total = (int64_t)B * HW * (C / VEC);
  if (!esam3_dev_flag("ESAM3_GN_OLD") && B <= 65535 && (int64_t)HW * (C / VEC) < ((int64_t)1 << 31)) {
    float* stats = partial + (int64_t)B * groups * splits * 2;   // behind the partials (esam3_groupnorm_scratch_floats)
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)((B * groups + 63) / 64)), dim3(64), 0, s, partial, stats, B * groups, splits,
                       (float)HW * (float)(C / groups), eps);
    DISPATCH_T(dtype, hipLaunchKernelGGL(gn_apply_relu_fin_kernel<T>, dim3(blocks_for((int64_t)HW * (C / VEC), 256), (unsigned)B), dim3(256), 0, s,
                                         (T*)x, stats, gamma, beta, HW, C, groups));
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(gn_apply_relu_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (T*)x, partial,
                                       gamma, beta, HW, C, groups, splits, eps, total));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
