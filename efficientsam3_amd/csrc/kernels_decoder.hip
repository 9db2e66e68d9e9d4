// Prompt-encoder / two-way mask-decoder / post-processing kernels
// (sam/prompt_encoder.py, sam/transformer.py, sam/mask_decoder.py, model/utils/sam1_utils.py).
#include "gemm_common.h"
#include "kernels.h"

namespace {

inline unsigned blocks_for(int64_t n, int bs) { return (unsigned)((n + bs - 1) / bs); }

// ------------------------------------------------------------------------------------
// out[bp][p][c] = in[src_img[bp]][p][c] + cbias[c] (+ dense[bp][p][c])
// (sam3_image.py:618-620 no_mem_embed; mask_decoder.py:199-206 repeat_interleave + dense)
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void gather_add_kernel(const T* __restrict__ in, const int* __restrict__ src_img,
                                  const float* __restrict__ cbias, const T* __restrict__ dense,
                                  T* __restrict__ out, int Bp, int64_t P, int C) {
  const int64_t total = (int64_t)Bp * P * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t bp = i / (P * C);
    const int64_t pc = i - bp * P * C;
    float v = to_f32<T>(in[(int64_t)src_img[bp] * P * C + pc]) + cbias[c];
    if (dense) v += to_f32<T>(dense[i]);
    out[i] = from_f32<T>(v);
  }
}
constexpr int VEC = 8;
// the same on 8-element vectors (C % 8 == 0, 16-byte aligned rows): grid.y = prompt, a thread owns 8 channels of one token
template <typename T>
__global__ __launch_bounds__(256) void gather_add_vec_kernel(const T* __restrict__ in, const int* __restrict__ src_img,
                                                             const float* __restrict__ cbias, const T* __restrict__ dense,
                                                             T* __restrict__ out, int64_t PC8, int C8) {
  const int64_t bp = blockIdx.y;
  const T* src = in + (int64_t)src_img[bp] * PC8 * VEC;
  const T* dn = dense ? dense + bp * PC8 * VEC : nullptr;
  T* dst = out + bp * PC8 * VEC;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < PC8; i += (int64_t)gridDim.x * 256) {
    const int c8 = (int)(i % C8);
    float v[VEC], d[VEC];
    Vec8<T>::load(src + i * VEC, v);
    const float4 b0 = *reinterpret_cast<const float4*>(cbias + c8 * VEC), b1 = *reinterpret_cast<const float4*>(cbias + c8 * VEC + 4);
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
    if (dn) {
      Vec8<T>::load(dn + i * VEC, d);
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] += d[e];
    }
    Vec8<T>::store(dst + i * VEC, v);
  }
}

// ------------------------------------------------------------------------------------
// Softmax attention with few queries and many keys (token -> image cross attention and
// token self attention; transformer.py:226-264).  grid = (B*heads, ceil(Nq/16));
// block = 16 queries x 16 key-lanes; each thread runs an online softmax over its strided
// subset of keys, then the 16 key-lanes of a query merge with wavefront shuffles.
// ------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ void attn_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                            const T* __restrict__ v, int ldv, T* __restrict__ o, int ldo, int Nq, int Nk,
                            int heads) {
  const int h = blockIdx.x % heads;
  const int64_t b = blockIdx.x / heads;
  const int qi = blockIdx.y * 16 + (threadIdx.x >> 4);
  const int kl = threadIdx.x & 15;
  const bool q_ok = qi < Nq;
  float qv[HD];
  const float scale = rsqrtf((float)HD);
  if (q_ok) {
    const T* qp = q + (b * Nq + qi) * (int64_t)ldq + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] = to_f32<T>(qp[d]) * scale;
  } else {
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] = 0.f;
  }
  float m = -3.0e38f, l = 0.f;
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  for (int j = kl; j < Nk; j += 16) {
    const T* kp = k + (b * Nk + j) * (int64_t)ldk + h * HD;
    const T* vp = v + (b * Nk + j) * (int64_t)ldv + h * HD;
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(qv[d], to_f32<T>(kp[d]), s);
    const float mn = fmaxf(m, s);
    const float alpha = __expf(m - mn);
    const float pexp = __expf(s - mn);
    l = l * alpha + pexp;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = acc[d] * alpha + pexp * to_f32<T>(vp[d]);
    m = mn;
  }
  // merge the 16 key-lanes (xor 1,2,4,8 stays inside the query's 16-lane group)
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) {
    const float m2 = __shfl_xor(m, off);
    const float l2 = __shfl_xor(l, off);
    const float mn = fmaxf(m, m2);
    const float a1 = __expf(m - mn), a2 = __expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      const float x2 = __shfl_xor(acc[d], off);
      acc[d] = acc[d] * a1 + x2 * a2;
    }
    m = mn;
  }
  if (q_ok && kl == 0) {
    T* op = o + (b * Nq + qi) * (int64_t)ldo + h * HD;
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = from_f32<T>(acc[d] * inv);
  }
}

// ------------------------------------------------------------------------------------
// Softmax attention with many queries and few keys (image -> token, Nk <= 64):
// thread = (query, head), head fastest, K/V of the batch item staged in LDS as fp32.
// ------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ void attn_fewkeys_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                                    const T* __restrict__ v, int ldv, T* __restrict__ o, int ldo,
                                    int Nq, int Nk, int heads) {
  extern __shared__ float skv[];  // K [Nk][heads*HD] then V [Nk][heads*HD]
  const int64_t b = blockIdx.y;
  const int D = heads * HD;
  float* sk = skv;
  float* sv = skv + Nk * D;
  for (int i = threadIdx.x; i < Nk * D; i += blockDim.x) {
    const int j = i / D, d = i - j * D;
    sk[i] = to_f32<T>(k[(b * Nk + j) * (int64_t)ldk + d]);
    sv[i] = to_f32<T>(v[(b * Nk + j) * (int64_t)ldv + d]);
  }
  __syncthreads();
  const int qpb = blockDim.x / heads;
  const int ql = threadIdx.x / heads, h = threadIdx.x - ql * heads;
  const int qi = blockIdx.x * qpb + ql;
  if (qi >= Nq || ql >= qpb) return;
  const float scale = rsqrtf((float)HD);
  const T* qp = q + (b * Nq + qi) * (int64_t)ldq + h * HD;
  float qv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) qv[d] = to_f32<T>(qp[d]) * scale;
  float m = -3.0e38f;
  for (int j = 0; j < Nk; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(qv[d], sk[j * D + h * HD + d], s);
    m = fmaxf(m, s);
  }
  float l = 0.f, acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  for (int j = 0; j < Nk; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s = fmaf(qv[d], sk[j * D + h * HD + d], s);
    const float pexp = __expf(s - m);
    l += pexp;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = fmaf(pexp, sv[j * D + h * HD + d], acc[d]);
  }
  const float inv = 1.f / l;
  T* op = o + (b * Nq + qi) * (int64_t)ldo + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) op[d] = from_f32<T>(acc[d] * inv);
}

// Variant for head dim 16, <= NKMAX keys, 16-byte aligned rows: a thread still owns one (query, head), but its 16 query
// values arrive in two 16-byte loads (the 8 heads of a query = 256 contiguous bytes over 8 lanes), K / V come out of
// LDS as float4, the scores stay in registers between the max and the exp-sum pass, and a thread walks QPT queries so
// that a workgroup stages K / V once per QPT * 32 queries.
template <typename T, int NKMAX, int QPT>
__global__ __launch_bounds__(256) void attn_fewkeys16_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                                                            const T* __restrict__ v, int ldv, T* __restrict__ o, int ldo,
                                                            int Nq, int Nk, int heads) {
  constexpr int HD = 16;
  extern __shared__ float skv[];  // K [Nk][heads*HD] then V [Nk][heads*HD]
  const int64_t b = blockIdx.y;
  const int D = heads * HD;
  float* sk = skv;
  float* sv = skv + Nk * D;
  for (int i = threadIdx.x; i < Nk * D; i += blockDim.x) {
    const int j = i / D, d = i - j * D;
    sk[i] = to_f32<T>(k[(b * Nk + j) * (int64_t)ldk + d]);
    sv[i] = to_f32<T>(v[(b * Nk + j) * (int64_t)ldv + d]);
  }
  __syncthreads();
  const int qpb = blockDim.x / heads;
  const int ql = threadIdx.x / heads, h = threadIdx.x - ql * heads;
  const float scale = rsqrtf((float)HD);
  const float4* k4 = reinterpret_cast<const float4*>(sk + h * HD);
  const float4* v4 = reinterpret_cast<const float4*>(sv + h * HD);
  const int d4 = D / 4;
#pragma unroll 1
  for (int t = 0; t < QPT; ++t) {
    const int qi = (blockIdx.x * QPT + t) * qpb + ql;
    if (qi >= Nq) return;
    const T* qp = q + (b * Nq + qi) * (int64_t)ldq + h * HD;
    float qv[HD];
    Vec8<T>::load(qp, qv);
    Vec8<T>::load(qp + 8, qv + 8);
#pragma unroll
    for (int d = 0; d < HD; ++d) qv[d] *= scale;
    float sc[NKMAX];
    float m = -3.0e38f;
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) {
      float a = -3.0e38f;
      if (j < Nk) {
        a = 0.f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 kk = k4[j * d4 + c];
          a = fmaf(qv[4 * c], kk.x, a);
          a = fmaf(qv[4 * c + 1], kk.y, a);
          a = fmaf(qv[4 * c + 2], kk.z, a);
          a = fmaf(qv[4 * c + 3], kk.w, a);
        }
      }
      sc[j] = a;
      m = fmaxf(m, a);
    }
    float l = 0.f, acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
#pragma unroll
    for (int j = 0; j < NKMAX; ++j) {
      if (j < Nk) {
        const float pexp = __expf(sc[j] - m);
        l += pexp;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 vv = v4[j * d4 + c];
          acc[4 * c] = fmaf(pexp, vv.x, acc[4 * c]);
          acc[4 * c + 1] = fmaf(pexp, vv.y, acc[4 * c + 1]);
          acc[4 * c + 2] = fmaf(pexp, vv.z, acc[4 * c + 2]);
          acc[4 * c + 3] = fmaf(pexp, vv.w, acc[4 * c + 3]);
        }
      }
    }
    const float inv = 1.f / l;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] *= inv;
    T* op = o + (b * Nq + qi) * (int64_t)ldo + h * HD;
    Vec8<T>::store(op, acc);
    Vec8<T>::store(op + 8, acc + 8);
  }
}

// ------------------------------------------------------------------------------------
// tokens = [obj_score | iou | mask x4 | sparse prompt embeddings]
// PositionEmbeddingRandom on (coords + 0.5) / img_size, label embeddings
// (prompt_encoder.py:74-118,214-243; mask_decoder.py:177-197).  fp32 math.
// ------------------------------------------------------------------------------------
template <typename T>
__global__ void build_tokens_kernel(const float* __restrict__ out_tokens, const float* __restrict__ coords,
                                    const int* __restrict__ labels, const float* __restrict__ gauss,
                                    const float* __restrict__ point_emb,
                                    const float* __restrict__ not_a_point, T* __restrict__ tokens, int Bp,
                                    int Np, int pad, float img_size) {
  const int Tn = 6 + Np + (pad ? 1 : 0);
  const int64_t total = (int64_t)Bp * Tn * 256;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i & 255);
  const int t = (int)((i >> 8) % Tn);
  const int64_t bp = (i >> 8) / Tn;
  float val;
  if (t < 6) {
    val = out_tokens[t * 256 + c];
  } else {
    const int pi = t - 6;
    int lab = -1;
    float x = 0.f, y = 0.f;
    if (pi < Np) {
      lab = labels[bp * Np + pi];
      x = coords[(bp * Np + pi) * 2 + 0];
      y = coords[(bp * Np + pi) * 2 + 1];
    }
    if (lab == -1) {
      val = not_a_point[c];
    } else {
      const float cx = 2.f * ((x + 0.5f) / img_size) - 1.f;
      const float cy = 2.f * ((y + 0.5f) / img_size) - 1.f;
      const int f = c & 127;
      const float ang = 6.283185307179586f * (cx * gauss[f] + cy * gauss[128 + f]);
      val = (c < 128) ? sinf(ang) : cosf(ang);
      if (lab >= 0 && lab < 4) val += point_emb[lab * 256 + c];
    }
  }
  tokens[i] = from_f32<T>(val);
}

// ------------------------------------------------------------------------------------
// MobileCLIP-S0 text encoder helpers (backbones/mobile_clip.py).  The sequence is 16..77 tokens
// long, so these are latency-sized kernels: rows = B*S tokens of 512 channels.
// ------------------------------------------------------------------------------------
// forward_embedding (mobile_clip.py:815-823): x[b][s] = table[token] + pos[s]; also the fp32
// [S][B][D] copy the reference returns as language_embeds (text_encoder_student.py:58).
template <typename T>
__global__ void text_embed_kernel(const int64_t* __restrict__ tokens, const float* __restrict__ table,
                                  const float* __restrict__ pos, T* __restrict__ x, float* __restrict__ embeds_sbd,
                                  int B, int S, int D, int vocab) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * S * D) return;
  const int d = (int)(i % D);
  const int64_t bs = i / D;
  const int s_ = (int)(bs % S), b = (int)(bs / S);
  int64_t tok = tokens[bs];
  tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);
  const float v = table[tok * D + d] + pos[(int64_t)s_ * D + d];
  x[i] = from_f32<T>(v);
  if (embeds_sbd) embeds_sbd[((int64_t)s_ * B + b) * D + d] = v;
}

// depthwise 1 x KW conv along the sequence with zero padding KW/2 (RepMixer / ConvFFN,
// mobile_clip.py:499-640 after folding their BatchNorm branches): w [KW][D] fp32, bias [D] or null
template <typename T>
__global__ void seq_dwconv_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  T* __restrict__ out, int B, int S, int D, int KW) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * S * D) return;
  const int d = (int)(i % D);
  const int64_t bs = i / D;
  const int s_ = (int)(bs % S);
  const int64_t b = bs / S;
  float acc = bias ? bias[d] : 0.f;
  for (int t = 0; t < KW; ++t) {
    const int sj = s_ + t - KW / 2;
    if (sj >= 0 && sj < S) acc = fmaf(w[t * D + d], to_f32<T>(x[(b * S + sj) * D + d]), acc);
  }
  out[i] = from_f32<T>(acc);
}

// MultiHeadAttention (mobile_clip.py:354-425): qkv rows [3][heads][64]; softmax in fp32 over all
// S keys (no padding mask is passed by the student encoder), or over keys <= query when the model masks
// causally (MobileCLIP-B, mobile_clip.py:826-832).  One wavefront per (b, head, query): lane = channel of the
// 64-wide head.
template <typename T>
__global__ __launch_bounds__(64) void text_attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, int S, int heads,
                                                       int causal) {
  const int i = blockIdx.x % S;
  const int h = (blockIdx.x / S) % heads;
  const int64_t b = blockIdx.x / (S * heads);
  const int D = heads * 64, lane = threadIdx.x;
  const T* base = qkv + b * S * 3 * (int64_t)D;
  const float q = to_f32<T>(base[(int64_t)i * 3 * D + h * 64 + lane]) * 0.125f;  // 64^-0.5
  float m = -INFINITY, l = 0.f, acc = 0.f;
  const int jn = causal ? i + 1 : S;
  for (int j = 0; j < jn; ++j) {
    float sc = q * to_f32<T>(base[(int64_t)j * 3 * D + D + h * 64 + lane]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sc += __shfl_xor(sc, o, 64);
    const float v = to_f32<T>(base[(int64_t)j * 3 * D + 2 * D + h * 64 + lane]);
    const float mn = fmaxf(m, sc);
    const float f = __expf(m - mn), p = __expf(sc - mn);
    l = l * f + p;
    acc = acc * f + p * v;
    m = mn;
  }
  out[(b * S + i) * (int64_t)D + h * 64 + lane] = from_f32<T>(acc / l);
}

// [B][S][C] T -> [S][B][C] fp32 (language_features layout, text_encoder_student.py:58)
template <typename T>
__global__ void bsc_to_sbc_f32_kernel(const T* __restrict__ x, float* __restrict__ out, int B, int S, int C) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * S * C) return;
  const int c = (int)(i % C);
  const int64_t bs = i / C;
  const int s_ = (int)(bs % S), b = (int)(bs / S);
  out[((int64_t)s_ * B + b) * C + c] = to_f32<T>(x[i]);
}

// ------------------------------------------------------------------------------------
// PromptEncoder._embed_masks (sam/prompt_encoder.py:51-59,131-134): mask_downscaling =
// Conv k2s2 1->4, LayerNorm2d, GELU, Conv k2s2 4->16, LayerNorm2d, GELU, Conv 1x1 16->256 on a
// [Bp,1,288,288] fp32 mask -> dense prompt embedding [Bp][72*72][256].  A workgroup serves 64
// output pixels: 64 threads reduce their 4x4 input patch to the 16-vector, then all 256 threads
// (one per output channel, its 16 weights in registers) expand it.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return gelu_fast(x); }

template <typename T>
__global__ __launch_bounds__(256) void mask_embed_kernel(const float* __restrict__ mask, const float* __restrict__ w0,
                                                         const float* __restrict__ b0, const float* __restrict__ g1,
                                                         const float* __restrict__ be1, const float* __restrict__ w3,
                                                         const float* __restrict__ b3, const float* __restrict__ g4,
                                                         const float* __restrict__ be4, const float* __restrict__ w6,
                                                         const float* __restrict__ b6, T* __restrict__ out, int IN, int E) {
  __shared__ float h[64][17];
  const int64_t bp = blockIdx.y;
  const int P = E * E;
  const int p0 = blockIdx.x * 64;
  if (threadIdx.x < 64) {
    const int p = p0 + threadIdx.x;
    float v16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v16[i] = 0.f;
    if (p < P) {
      const int Y = p / E, X = p - Y * E;
      const float* m = mask + (bp * IN + 4 * Y) * (int64_t)IN + 4 * X;
      float c1[2][2][4];  // conv1 + LN + GELU at the 2x2 positions feeding this pixel
#pragma unroll
      for (int qy = 0; qy < 2; ++qy)
#pragma unroll
        for (int qx = 0; qx < 2; ++qx) {
          float a[4], mean = 0.f;
#pragma unroll
          for (int co = 0; co < 4; ++co) {
            float acc = b0[co];
#pragma unroll
            for (int ky = 0; ky < 2; ++ky)
#pragma unroll
              for (int kx = 0; kx < 2; ++kx)
                acc = fmaf(w0[co * 4 + ky * 2 + kx], m[(2 * qy + ky) * IN + 2 * qx + kx], acc);
            a[co] = acc;
            mean += acc;
          }
          mean *= 0.25f;
          float var = 0.f;
#pragma unroll
          for (int co = 0; co < 4; ++co) var += (a[co] - mean) * (a[co] - mean);
          const float inv = 1.f / sqrtf(var * 0.25f + 1e-6f);
#pragma unroll
          for (int co = 0; co < 4; ++co) c1[qy][qx][co] = gelu_erf(g1[co] * ((a[co] - mean) * inv) + be1[co]);
        }
      float mean = 0.f;
#pragma unroll
      for (int co = 0; co < 16; ++co) {
        float acc = b3[co];
#pragma unroll
        for (int ci = 0; ci < 4; ++ci)
#pragma unroll
          for (int ky = 0; ky < 2; ++ky)
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) acc = fmaf(w3[((co * 4 + ci) * 2 + ky) * 2 + kx], c1[ky][kx][ci], acc);
        v16[co] = acc;
        mean += acc;
      }
      mean *= (1.f / 16.f);
      float var = 0.f;
#pragma unroll
      for (int co = 0; co < 16; ++co) var += (v16[co] - mean) * (v16[co] - mean);
      const float inv = 1.f / sqrtf(var * (1.f / 16.f) + 1e-6f);
#pragma unroll
      for (int co = 0; co < 16; ++co) v16[co] = gelu_erf(g4[co] * ((v16[co] - mean) * inv) + be4[co]);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) h[threadIdx.x][i] = v16[i];
  }
  __syncthreads();
  const int c = threadIdx.x;
  float w[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) w[k] = w6[c * 16 + k];
  const float bias = b6[c];
  const int n = min(64, P - p0);
  for (int j = 0; j < n; ++j) {
    float acc = bias;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(w[k], h[j][k], acc);
    out[(bp * P + p0 + j) * 256 + c] = from_f32<T>(acc);
  }
}

// ------------------------------------------------------------------------------------
// Mask head tail in one kernel (mask_decoder.py:213-231): output_upscaling.3 = ConvTranspose2d(64 -> 32, k2 s2) of the
// 144^2 map + the 288^2 high-res feature (feat_s0), GELU, then masks[bp][k] = sum_c hyper[bp][k][c] * up[bp][c].
// The 288^2 x 32-channel tensor (170 MB at 32 prompts) is never written: a wave takes 32 consecutive pixels of the 144^2
// map, runs the four taps as 32 x 32 x 64 MFMA products, adds bias + feat_s0, applies GELU and dots its 16 channels per
// tap with the four hypernetwork vectors in registers; the two half-waves (channel halves) are summed by a lane swap.
// bf16 only (the f32 mode keeps the separate kernels).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upscale_mask_kernel(const bf16_t* __restrict__ u1, const bf16_t* __restrict__ wt, int kp,
                                                          const float* __restrict__ bias, const bf16_t* __restrict__ feat,
                                                          const int* __restrict__ img_of, const bf16_t* __restrict__ hyper, int ld_h,
                                                          float* __restrict__ masks, int S /*144*/, int frags_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, g = lane >> 5;
  const int bp = blockIdx.y;
  const int P = S * S, OS = 2 * S;
  const int64_t P4 = (int64_t)OS * OS;
  const int img = img_of ? img_of[bp] : bp;
  // weights: tap j = rows j*32 .. j*32+31 of wt [128][kp]; this lane's fragment rows are channel l31
  u32x4 fw[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      fw[j][ks] = *reinterpret_cast<const u32x4*>(wt + (int64_t)(j * 32 + l31) * kp + (2 * ks + g) * 8);
  // this lane's 16 channels are 8q + 4g + e: bias and the four hypernetwork vectors at those channels
  float bs[16], hw[4][16];
  // round 6: 4 + 16 vector loads (the lane's channels 8q + 4g .. + 3 are contiguous) where the element-wise form compiled to 16 dword and 64
  // two-byte loads per wave -- for three 32-pixel fragments of work; same values
  const bool vec_ok = (ld_h & 3) == 0 && !(((uintptr_t)hyper) & 7) && !(((uintptr_t)bias) & 15);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c0 = 8 * q + 4 * g;
    if (vec_ok) {
      float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (bias) b4 = *reinterpret_cast<const float4*>(bias + c0);
      bs[4 * q] = b4.x; bs[4 * q + 1] = b4.y; bs[4 * q + 2] = b4.z; bs[4 * q + 3] = b4.w;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint2 u = *reinterpret_cast<const uint2*>(hyper + ((int64_t)bp * 4 + k) * ld_h + c0);
        hw[k][4 * q] = __uint_as_float(u.x << 16); hw[k][4 * q + 1] = __uint_as_float(u.x & 0xffff0000u);
        hw[k][4 * q + 2] = __uint_as_float(u.y << 16); hw[k][4 * q + 3] = __uint_as_float(u.y & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        bs[4 * q + e] = bias ? bias[c0 + e] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) hw[k][4 * q + e] = to_f32<bf16_t>(hyper[((int64_t)bp * 4 + k) * ld_h + c0 + e]);
      }
    }
  }
  const bf16_t* ub = u1 + (int64_t)bp * P * 64;
  const bf16_t* fb = feat + (int64_t)img * P4 * 32;
  float* mb = masks + (int64_t)bp * 4 * P4;
  const int frag0 = (blockIdx.x * 4 + wave) * frags_per_wave;
  for (int f = 0; f < frags_per_wave; ++f) {
    const int pix0 = (frag0 + f) * 32;
    if (pix0 >= P) break;  // wave-uniform
    int pix = pix0 + l31;
    const bool pok = pix < P;
    if (!pok) pix = P - 1;
    const int y = pix / S, x = pix - y * S;
    u32x4 fa[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fa[ks] = *reinterpret_cast<const u32x4*>(ub + (int64_t)pix * 64 + (2 * ks + g) * 8);
    float part[4][4];  // [tap][mask]
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x16_v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) MmaOps<bf16_t>::mma(fw[j][ks], fa[ks], acc);
      const int oy = 2 * y + (j >> 1), ox = 2 * x + (j & 1);
      const bf16_t* rp = fb + ((int64_t)oy * OS + ox) * 32 + 4 * g;
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint2 u = *reinterpret_cast<const uint2*>(rp + 8 * q);
        const float r[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                            __uint_as_float(u.y & 0xffff0000u)};
        // GELU on PAIRS (round 6): gelu_fast(x) evaluates the packed form on {x, x} and keeps one half -- twice the VALU work of this
        // VALU-bound kernel's largest term; the same function per element, bit-identical masks
        float ge[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) ge[e] = acc[4 * q + e] + bs[4 * q + e] + r[e];
        gelu_fast_n<4>(ge);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // the separate kernels stored this value in bf16 before the product read it back
          const float v = to_f32<bf16_t>(from_f32<bf16_t>(ge[e]));
#pragma unroll
          for (int k = 0; k < 4; ++k) s4[k] = fmaf(v, hw[k][4 * q + e], s4[k]);
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) part[j][k] = s4[k];
    }
    // sum the two channel halves: after the swap a lane holds (its own value of one tap row, the other half's value of it)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // taps 0,1 (output row 2y) end up complete in lanes 0-31, taps 2,3 (row 2y+1) in lanes 32-63
      float keep0 = g == 0 ? part[0][k] : part[2][k], keep1 = g == 0 ? part[1][k] : part[3][k];
      const float give0 = g == 0 ? part[2][k] : part[0][k], give1 = g == 0 ? part[3][k] : part[1][k];
      keep0 += __shfl_xor(give0, 32);
      keep1 += __shfl_xor(give1, 32);
      if (pok) {
        float* mp = mb + (int64_t)k * P4 + (int64_t)(2 * y + g) * OS + 2 * x;
        *reinterpret_cast<float2*>(mp) = make_float2(keep0, keep1);
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// masks[bp][k][p] = sum_c hyper[bp][k][c] * up[bp][p][c]   (mask_decoder.py:230-231)
// ------------------------------------------------------------------------------------
template <typename T, int C>
__global__ void mask_product_kernel(const T* __restrict__ hyper, int ld_h, const T* __restrict__ up,
                                    float* __restrict__ masks, int64_t P) {
  __shared__ float sh[4 * C];
  const int64_t bp = blockIdx.y;
  for (int i = threadIdx.x; i < 4 * C; i += blockDim.x)
    sh[i] = to_f32<T>(hyper[(bp * 4 + i / C) * (int64_t)ld_h + (i % C)]);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const T* u = up + (bp * P + p) * C;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
  for (int c = 0; c < C; ++c) {
    const float x = to_f32<T>(u[c]);
    a0 = fmaf(sh[c], x, a0);
    a1 = fmaf(sh[C + c], x, a1);
    a2 = fmaf(sh[2 * C + c], x, a2);
    a3 = fmaf(sh[3 * C + c], x, a3);
  }
  float* m = masks + bp * 4 * P + p;
  m[0] = a0; m[P] = a1; m[2 * P] = a2; m[3 * P] = a3;
}

// stability counters for mask 0: cnt[2*bp] = #(m > delta), cnt[2*bp+1] = #(m > -delta)
__global__ __launch_bounds__(1024) void stability_count_kernel(const float* __restrict__ all_masks, int* __restrict__ cnt,
                                       int64_t P, float delta) {
  const int64_t bp = blockIdx.y;
  const float* m = all_masks + bp * 4 * P;
  int ci = 0, cu = 0;
  // round 6: 16-byte loads and 32-bit indices (the scalar form with 64-bit loop arithmetic read 10.6 MB in 46 us); counts are integers: same result
  if (P < ((int64_t)1 << 31) && ((uintptr_t)m & 15) == 0) {
    const unsigned n4 = (unsigned)(P >> 2), stride = gridDim.x * blockDim.x;
    const float4* m4 = reinterpret_cast<const float4*>(m);
    for (unsigned q = blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
      const float4 x = m4[q];
      ci += (x.x > delta) + (x.y > delta) + (x.z > delta) + (x.w > delta);
      cu += (x.x > -delta) + (x.y > -delta) + (x.z > -delta) + (x.w > -delta);
    }
    for (unsigned q = 4 * n4 + blockIdx.x * blockDim.x + threadIdx.x; q < (unsigned)P; q += stride) {
      const float x = m[q];
      ci += x > delta;
      cu += x > -delta;
    }
  } else {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P;
         p += (int64_t)gridDim.x * blockDim.x) {
      const float x = m[p];
      ci += x > delta;
      cu += x > -delta;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { ci += __shfl_xor(ci, o); cu += __shfl_xor(cu, o); }
  // one atomic pair per WORKGROUP (round 6): 32 x 4 waves adding to the same two words per prompt serialised in L2 -- the launch took 44 us for
  // 10.6 MB with the loads already vectorised
  __shared__ int red[2][16];
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][w] = ci; red[1][w] = cu; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = 0, b = 0;
    for (int i = 0; i < nw; ++i) { a += red[0][i]; b += red[1][i]; }
    atomicAdd(cnt + 2 * bp, a);
    atomicAdd(cnt + 2 * bp + 1, b);
  }
}

template <typename T>
__global__ void select_masks_kernel(const float* __restrict__ all_masks, const T* __restrict__ all_iou,
                                    int ld_iou, float* __restrict__ out_masks, float* __restrict__ out_iou,
                                    const int* __restrict__ cnt, int64_t P, int multimask, float thresh) {
  const int64_t bp = blockIdx.y;
  const float* src = all_masks + bp * 4 * P;
  float iou[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) iou[i] = to_f32<T>(all_iou[bp * ld_iou + i]);
  if (multimask) {
    float* dst = out_masks + bp * 3 * P;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < 3 * P;
         p += (int64_t)gridDim.x * blockDim.x)
      dst[p] = src[P + p];
    if (blockIdx.x == 0 && threadIdx.x < 3) out_iou[bp * 3 + threadIdx.x] = iou[1 + threadIdx.x];
  } else {
    const float ai = (float)cnt[2 * bp], au = (float)cnt[2 * bp + 1];
    const float stab = au > 0.f ? ai / au : 1.f;
    int sel = 0;
    if (!(stab >= thresh)) {  // fall back to the best of masks 1..3 (first max, torch.argmax)
      sel = 1;
      if (iou[2] > iou[sel]) sel = 2;
      if (iou[3] > iou[sel]) sel = 3;
    }
    float* dst = out_masks + bp * P;
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P;
         p += (int64_t)gridDim.x * blockDim.x)
      dst[p] = src[sel * P + p];
    if (blockIdx.x == 0 && threadIdx.x == 0) out_iou[bp] = iou[sel];
  }
}

// ------------------------------------------------------------------------------------
// Hole filling (sam1_utils.py:77-104 + perflib/connected_components.py): 8-connected
// components of background pixels (score <= thr); components with area <= max_area are
// set to thr + 10.  One workgroup per mask; union-find with atomicMin on int32 labels
// living in global memory (324 KB per 288x288 mask -> L2 resident).  Only component
// *areas* are consumed, so label values need not match any other implementation.
// ------------------------------------------------------------------------------------
// find with path halving: a visited node is re-pointed at its grandparent with atomicMin (parents only ever decrease, an ancestor is
// always a valid parent, and atomicMin cannot undo a concurrent re-hooking to something smaller).  Without it the walks of the one
// huge background component -- most pixels of a mask -- stayed as long as cc_merge's hooks had left them.
__device__ inline int uf_find(int* lab, int x) {
  int r = x;
  while (true) {
    const int p = __hip_atomic_load(lab + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == r) break;
    const int gp = __hip_atomic_load(lab + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (gp != p) atomicMin(lab + r, gp);
    r = gp;
  }
  return r;
}
__device__ inline void uf_union(int* lab, int a, int b) {
  while (true) {
    a = uf_find(lab, a);
    b = uf_find(lab, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }  // a > b: hook a under b
    const int old = atomicMin(lab + a, b);
    if (old == a) return;
    a = old;  // somebody else re-hooked a; retry with its new parent
  }
}

// Four grid-wide passes (the kernel boundaries are the global barriers):
//   cc_init  : label = first pixel of the horizontal background run (inside the wavefront's
//              64-pixel segment), so most pixels start already merged;
//   cc_merge : lock-free unions with the W / N / NW / NE neighbours, skipping the ones that
//              are implied by the neighbour's own unions;
//   cc_count : per-root areas, one atomicAdd per (wavefront, run of equal roots);
//   cc_apply : write thr+10 into components of area <= max_area.
__global__ void cc_init_kernel(const float* __restrict__ in, int* __restrict__ lab, int* __restrict__ area,
                               int W, int HW, int64_t total, float thr) {
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = gi < total;
  const int lane = threadIdx.x & 63;
  const int i = valid ? (int)(gi % HW) : 0;
  const bool bg = valid && in[gi] <= thr;
  const unsigned long long m = __ballot(bg);
  if (!valid) return;
  int v = -1;
  if (bg) {
    const unsigned long long below = ~m & ((1ull << lane) - 1ull);
    int start = below ? 64 - __clzll(below) : 0;  // first lane of this run inside the wave
    const int x = i % W;
    if (lane - x > start) start = lane - x;       // runs do not cross the row start
    v = i - (lane - start);
  }
  lab[gi] = v;
  area[gi] = 0;
}

__global__ void cc_merge_kernel(const float* __restrict__ in, int* __restrict__ labels, int W, int H, int HW,
                                int64_t total, float thr) {
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total) return;
  const int64_t n = gi / HW;
  const int i = (int)(gi - n * HW);
  const float* src = in + n * HW;
  if (!(src[i] <= thr)) return;
  int* lab = labels + n * HW;
  const int y = i / W, x = i - y * W;
  const bool w_bg = x > 0 && src[i - 1] <= thr;
  // cc_init merged runs only inside one 64-pixel wavefront segment: stitch the seams
  if (w_bg && (gi & 63) == 0) uf_union(lab, i, i - 1);
  if (y > 0) {
    const bool n_bg = src[i - W] <= thr;
    const bool nw_bg = x > 0 && src[i - W - 1] <= thr;
    const bool ne_bg = x < W - 1 && src[i - W + 1] <= thr;
    if (n_bg) {
      if (!(w_bg && nw_bg)) uf_union(lab, i, i - W);
    } else {
      if (nw_bg && !w_bg) uf_union(lab, i, i - W - 1);
      if (ne_bg) uf_union(lab, i, i - W + 1);
    }
  }
}

__global__ void cc_count_kernel(int* __restrict__ labels, int* __restrict__ areas, int HW, int64_t total,
                                int sat) {
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = gi < total;
  const int lane = threadIdx.x & 63;
  int64_t key = -1;  // global index of the root, -1 = not a background pixel
  if (valid) {
    const int64_t n = gi / HW;
    const int i = (int)(gi - n * HW);
    int* lab = labels + n * HW;
    const int v = __hip_atomic_load(lab + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v >= 0) {
      // cc_init pointed every pixel of a horizontal run (inside this wavefront's 64-pixel segment) at the run's first pixel, so most
      // labels name a pixel that another lane of this wave holds.  A pixel and its label are in one set: a lane whose label points
      // into the wave takes the root THAT lane found instead of walking itself; a lane that is its own label, whose label lies
      // outside the wave, or whose source did not walk, walks itself.
      const bool walk = v == i || i - v > lane;
      int r = walk ? uf_find(lab, i) : 0;
      const int src = walk ? lane : lane - (i - v);
      const int r_src = __shfl(r, src), src_walked = __shfl(walk ? 1 : 0, src);
      if (!walk) r = src_walked ? r_src : uf_find(lab, i);
      __hip_atomic_store(lab + i, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // path compression
      key = n * HW + r;
    }
  }
  // run-length aggregate equal keys of consecutive lanes
  const int64_t prev = __shfl_up(key, 1);
  const bool leader = key >= 0 && (lane == 0 || prev != key);
  const bool brk = lane == 0 || prev != key;  // any change of key starts a new run
  const unsigned long long bm = __ballot(brk);
  if (leader) {
    const unsigned long long above = lane == 63 ? 0ull : (bm >> (lane + 1));
    const int len = above ? __ffsll((long long)above) : 64 - lane;
    // only "area <= max_area" is consumed: once a component is known to exceed `sat`, further
    // (heavily contended) atomics on its root are skipped -- the count stays > max_area
    if (__hip_atomic_load(areas + key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= sat)
      atomicAdd(areas + key, len);
  }
}

__global__ void cc_apply_kernel(const float* __restrict__ in, float* __restrict__ out,
                                const int* __restrict__ labels, const int* __restrict__ areas, int HW,
                                int64_t total, float thr, float max_area) {
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total) return;
  float v = in[gi];
  if (v <= thr) {
    const int64_t n = gi / HW;
    const int r = labels[gi];  // compressed to the root by cc_count
    if ((float)areas[n * HW + r] <= max_area) v = thr + 10.f;
  }
  out[gi] = v;
}

// ------------------------------------------------------------------------------------
// Tiled hole filling (round 5; SURVEY.md 2.2 asked for labelling per tile in LDS with a seam merge; the reference's own GPU
// algorithm is the block-local + merge scheme of perflib/triton/connected_components.py:203-394; plan checked against scipy by
// tools/cc_tiled_prototype.py).  The grid-wide union-find above spends its time on the ONE huge background component of a
// mask: every walk and every count atomically converges on its root in L2 (cc_merge + cc_count = 0.28 ms of a 1 ms decode).
// Here a workgroup owns one full-width STRIP of TH rows (TH x W <= 9216 pixels: 288 x 32) and does the same union-find on LDS
// words; strips only meet at horizontal seams:
//   cc_tile : labels = first pixel of the horizontal run (wave-local), unions with W (at 64-lane seams) / N / NW / NE inside the
//             strip, flatten, per-root areas (run-length aggregated LDS atomics, saturating) -> labels[pixel] = mask-local index
//             of the strip-local root, areas[pixel] = local area at roots, 0 elsewhere;
//   cc_seam : the first row of every strip but the first: the same N / NW / NE rule across the seam, as unions of strip-local
//             roots in global memory -- a few hundred components per mask instead of 83 k pixels;
//   cc_sum  : every strip-local root that is no longer a global root adds its area to its global root's (saturating);
//   cc_apply_tiled : pixel -> strip root -> global root (read-only walk) -> fill when the merged area <= max_area.
// Integer work, order-independent results (areas are sums; only `area <= max_area` is consumed): bit-exact against scipy.
// ------------------------------------------------------------------------------------
constexpr int CC_TPX = 9216;   // pixels per strip: two int arrays in LDS = 72 KB, two workgroups per CU

__device__ inline int uf_find_ro(const int* lab, int x) {   // no compression: chains are pixel -> strip root -> a few hooks
  while (true) {
    const int p = __hip_atomic_load(lab + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    x = p;
  }
}

__global__ __launch_bounds__(1024) void cc_tile_kernel(const float* __restrict__ in, int* __restrict__ labels, int* __restrict__ areas,
                                                        int W, int H, int TH, int strips, float thr, int sat) {
  extern __shared__ int cc_lds[];
  int* lab = cc_lds;             // [CC_TPX] parent pointers (strip-local pixel index), -1 = foreground
  int* area = cc_lds + CC_TPX;   // [CC_TPX]
  const int tid = threadIdx.x, lane = tid & 63;
  const int n = blockIdx.x / strips, st = blockIdx.x - n * strips;
  const int y0 = st * TH, rows = min(TH, H - y0), npx = rows * W;
  const int64_t base = (int64_t)n * H * W + (int64_t)y0 * W;
  const float* src = in + base;
  const int iters = (npx + 1023) / 1024;
  // ---- a: run-start labels (every lane of a wave takes part in the ballot) ----
  for (int k = 0; k < iters; ++k) {
    const int i = k * 1024 + tid;
    const bool valid = i < npx;
    const bool bg = valid && src[i] <= thr;
    const unsigned long long m = __ballot(bg);
    if (valid) {
      int v = -1;
      if (bg) {
        const unsigned long long below = ~m & ((1ull << lane) - 1ull);
        int start = below ? 64 - __clzll(below) : 0;   // first lane of this run inside the wave's 64-pixel segment
        const int x = i % W;
        if (lane - x > start) start = lane - x;          // runs do not cross the row start
        v = i - (lane - start);
      }
      lab[i] = v;
      area[i] = 0;
    }
  }
  __syncthreads();
  // ---- b: unions inside the strip (cc_merge_kernel's rule) ----
  for (int k = 0; k < iters; ++k) {
    const int i = k * 1024 + tid;
    if (i >= npx || lab[i] < 0) continue;
    const int y = i / W, x = i - y * W;
    const bool w_bg = x > 0 && lab[i - 1] >= 0;
    if (w_bg && (i & 63) == 0) uf_union(lab, i, i - 1);   // runs were merged inside one 64-lane segment only
    if (y > 0) {
      const bool n_bg = lab[i - W] >= 0;
      const bool nw_bg = x > 0 && lab[i - W - 1] >= 0;
      const bool ne_bg = x < W - 1 && lab[i - W + 1] >= 0;
      if (n_bg) {
        if (!(w_bg && nw_bg)) uf_union(lab, i, i - W);
      } else {
        if (nw_bg && !w_bg) uf_union(lab, i, i - W - 1);
        if (ne_bg) uf_union(lab, i, i - W + 1);
      }
    }
  }
  __syncthreads();
  // ---- c: flatten + areas (one LDS atomic per run of equal roots inside a wave) ----
  for (int k = 0; k < iters; ++k) {
    const int i = k * 1024 + tid;
    int key = -1;
    if (i < npx && lab[i] >= 0) {
      key = uf_find(lab, i);
      lab[i] = key;   // a root is always a valid parent: plain store
    }
    const int prev = __shfl_up(key, 1);
    const bool brk = lane == 0 || prev != key;
    const unsigned long long bm = __ballot(brk);
    if (key >= 0 && brk) {
      const unsigned long long above = lane == 63 ? 0ull : (bm >> (lane + 1));
      const int len = above ? __ffsll((long long)above) : 64 - lane;
      if (area[key] <= sat) atomicAdd(area + key, len);   // only `area <= max_area` is consumed
    }
  }
  __syncthreads();
  // ---- d: strip-local roots as mask-local pixel indices; areas at roots, 0 elsewhere ----
  const int off = y0 * W;
  for (int i = tid; i < npx; i += 1024) {
    const int l = lab[i];
    labels[base + i] = l < 0 ? -1 : off + l;
    areas[base + i] = l == i ? area[i] : 0;
  }
}

__global__ void cc_seam_kernel(int* __restrict__ labels, int W, int H, int TH, int strips, int n_masks) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int per = (strips - 1) * W;
  if (t >= n_masks * per) return;
  const int n = t / per, r = t - n * per;
  const int st = 1 + r / W, x = r - (st - 1) * W;
  int* lab = labels + (int64_t)n * H * W;
  const int i = st * TH * W + x;   // first row of strip st; the row above belongs to strip st - 1
  if (lab[i] < 0) return;           // (lab >= 0 <=> background: unions only ever write non-negative parents)
  const bool w_bg = x > 0 && __hip_atomic_load(lab + i - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
  const bool n_bg = __hip_atomic_load(lab + i - W, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
  const bool nw_bg = x > 0 && __hip_atomic_load(lab + i - W - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
  const bool ne_bg = x < W - 1 && __hip_atomic_load(lab + i - W + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0;
  if (n_bg) {
    if (!(w_bg && nw_bg)) uf_union(lab, i, i - W);
  } else {
    if (nw_bg && !w_bg) uf_union(lab, i, i - W - 1);
    if (ne_bg) uf_union(lab, i, i - W + 1);
  }
}

__global__ void cc_sum_kernel(int* __restrict__ labels, int* __restrict__ areas, int HW, int64_t total, int sat) {
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total) return;
  const int a = areas[gi];   // > 0 exactly at the strip-local roots (cc_tile); nobody adds to a root that is not a global root
  if (a <= 0) return;
  const int64_t n = gi / HW;
  const int i = (int)(gi - n * HW);
  const int r = uf_find_ro(labels + n * HW, i);
  if (r == i) return;
  int* dst = areas + n * HW + r;
  if (__hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= sat) atomicAdd(dst, a);
}

__global__ void cc_apply_tiled_kernel(const float* __restrict__ in, float* __restrict__ out, const int* __restrict__ labels,
                                      const int* __restrict__ areas, int HW, int64_t total, float thr, float max_area) {
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= total) return;
  float v = in[gi];
  if (v <= thr) {
    const int64_t n = gi / HW;
    const int* lab = labels + n * HW;
    int r = labels[gi];
    while (true) {   // kernel boundary = all unions done: plain loads
      const int p = lab[r];
      if (p == r) break;
      r = p;
    }
    if ((float)areas[n * HW + r] <= max_area) v = thr + 10.f;
  }
  out[gi] = v;
}

// bilinear upsample (align_corners=False) of fp32 masks, optional > thr -> u8
// The bilinear sample of both upsampling kernels, with the contractions written out (left to the compiler, the two kernels were fused
// differently and disagreed in the last bit of one result in ~1e5): source coordinate = (o + 0.5) scale - 0.5 as one fma, then
// hy (hx a + lx b) + ly (hx c + lx d) as fma(hy, fma(hx, a, lx b), ly fma(hx, c, lx d)).
__device__ __forceinline__ float up_src_coord(int o, float scale) {
  const float f = fmaf((float)o + 0.5f, scale, -0.5f);
  return f < 0.f ? 0.f : f;
}
__device__ __forceinline__ float up_bilerp(float hy, float ly, float hx, float lx, float a, float b, float c, float d) {
  return fmaf(hy, fmaf(hx, a, lx * b), ly * fmaf(hx, c, lx * d));
}
__global__ void upsample_masks_px_kernel(const float* __restrict__ in, float* __restrict__ out_f32,
                                      uint8_t* __restrict__ out_u8, int IH, int IW, int OH, int OW,
                                      float thr) {
  const int64_t n = blockIdx.y;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)OH * OW) return;
  const int ox = (int)(idx % OW), oy = (int)(idx / OW);
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const float fy = up_src_coord(oy, sy), fx = up_src_coord(ox, sx);
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < IH - 1 ? 1 : 0), x1 = x0 + (x0 < IW - 1 ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* s = in + n * IH * (int64_t)IW;
  const float v = up_bilerp(hy, ly, hx, lx, s[y0 * IW + x0], s[y0 * IW + x1], s[y1 * IW + x0], s[y1 * IW + x1]);
  if (out_f32) out_f32[n * OH * (int64_t)OW + idx] = v;
  if (out_u8) out_u8[n * OH * (int64_t)OW + idx] = v > thr ? 1 : 0;
}

// Round 6: the same interpolation (expression for expression: bit-identical outputs, dev builds keep the per-pixel kernel above behind
// ESAM3_UPSAMPLE_OLD), one thread = 8 consecutive pixels of an output row.  The per-pixel form spent its time on two 64-bit divisions per pixel
// and on byte stores (64 bytes per wave instruction): 86.6 us for 32 masks of 1008 x 1008 = 0.37 TB/s of output.  Here grid.y = output row and
// grid.z = mask (no divisions), the row's source rows / fractions are computed once per thread, and a thread writes its 8 mask bytes (and its
// 8 floats) with one (two) wide stores when the row length allows it.
__global__ __launch_bounds__(128) void upsample_masks_kernel(const float* __restrict__ in, float* __restrict__ out_f32,
                                                             uint8_t* __restrict__ out_u8, int IH, int IW, int OH, int OW, float thr) {
  extern __shared__ __attribute__((aligned(16))) float up_rows[];   // the two source rows of this output row: [2][IW]
  const int64_t n = blockIdx.z;
  const int oy = blockIdx.y;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const float fy = up_src_coord(oy, sy);
  const int y0 = (int)fy;
  const int y1 = y0 + (y0 < IH - 1 ? 1 : 0);
  const float ly = fy - (float)y0;
  const float hy = 1.f - ly;
  const float* s = in + n * IH * (int64_t)IW;
  for (int i = threadIdx.x; i < 2 * IW; i += blockDim.x) up_rows[i] = i < IW ? s[y0 * IW + i] : s[y1 * IW + (i - IW)];
  __syncthreads();
  for (int ox0 = threadIdx.x * 8; ox0 < OW; ox0 += blockDim.x * 8) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ox = ox0 + e < OW ? ox0 + e : OW - 1;
    const float fx = up_src_coord(ox, sx);
    const int x0 = (int)fx;
    const int x1 = x0 + (x0 < IW - 1 ? 1 : 0);
    const float lx = fx - (float)x0;
    const float hx = 1.f - lx;
    v[e] = up_bilerp(hy, ly, hx, lx, up_rows[x0], up_rows[x1], up_rows[IW + x0], up_rows[IW + x1]);
  }
  const int64_t o = n * OH * (int64_t)OW + (int64_t)oy * OW + ox0;
  const bool full = ox0 + 8 <= OW;
  if (out_f32) {
    if (full && (OW & 3) == 0 && ((uintptr_t)out_f32 & 15) == 0) {
      *reinterpret_cast<float4*>(out_f32 + o) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(out_f32 + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (ox0 + e < OW) out_f32[o + e] = v[e];
    }
  }
  if (out_u8) {
    if (full && (OW & 7) == 0 && ((uintptr_t)out_u8 & 7) == 0) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        lo |= (v[e] > thr ? 1u : 0u) << (8 * e);
        hi |= (v[4 + e] > thr ? 1u : 0u) << (8 * e);
      }
      *reinterpret_cast<uint2*>(out_u8 + o) = make_uint2(lo, hi);
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (ox0 + e < OW) out_u8[o + e] = v[e] > thr ? 1 : 0;
    }
  }
  }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    o[i] = from_f32<T>(to_f32<T>(a[i]) + to_f32<T>(b[i]));
}
// o (bf16) = a (f32) + b (f32): the GEMM-side copy of an fp32 token stream plus its fp32 positional tokens
__global__ void add_f32_to_bf16_kernel(const float* __restrict__ a, const float* __restrict__ b, bf16_t* __restrict__ o, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    o[i] = f32_to_bf16(a[i] + (b ? b[i] : 0.f));
}
template <typename T>
__global__ void strided_to_f32_kernel(const T* __restrict__ in, int ld, float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = to_f32<T>(in[i * ld]);
}

__global__ void clamp_kernel(float* x, int64_t n, float lo, float hi) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    x[i] = fminf(fmaxf(x[i], lo), hi);
}

}  // namespace

#define DISPATCH_T(dtype, ...)                  \
  do {                                          \
    if ((dtype) == 0) { using T = float; __VA_ARGS__; } \
    else { using T = bf16_t; __VA_ARGS__; }     \
  } while (0)

int esam3_launch_gather_add(int dtype, const void* in, const int* src_img, const float* cbias,
                            const void* dense, void* out, int Bp, int64_t P, int C, hipStream_t s) {
  const int64_t total = (int64_t)Bp * P * C;
  if (C % VEC == 0 && Bp <= 65535 && !((uintptr_t)in & 15) && !((uintptr_t)out & 15) && !((uintptr_t)dense & 15) && !((uintptr_t)cbias & 15)) {
    const int64_t pc8 = P * C / VEC;
    const dim3 grid((unsigned)min((int64_t)256, (pc8 + 255) / 256), (unsigned)Bp);
    DISPATCH_T(dtype, hipLaunchKernelGGL(gather_add_vec_kernel<T>, grid, dim3(256), 0, s, (const T*)in, src_img, cbias, (const T*)dense,
                                         (T*)out, pc8, C / VEC));
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  const unsigned g = (unsigned)min((int64_t)8192, (total + 255) / 256);
  DISPATCH_T(dtype, hipLaunchKernelGGL(gather_add_kernel<T>, dim3(g), dim3(256), 0, s, (const T*)in,
                                       src_img, cbias, (const T*)dense, (T*)out, Bp, P, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

namespace {
// ------------------------------------------------------------------------------------
// Token -> image cross attention of the two-way decoder (sam/transformer.py:165-170,226-264): <= 16 query tokens,
// 8 heads x 16, thousands of keys.  The image-side K / V (Nk x 128 per prompt) are the only real traffic:
//   * a workgroup owns one key chunk of one prompt; K and V tiles of TK keys go HBM -> LDS with 16-byte coalesced loads
//     (whole 256-byte rows), once;
//   * thread = (key parity s, token t, head h): its 16 query values sit in registers, the K / V slices of head h are
//     LDS reads that the 16 tokens of a head share (broadcast, conflict-free: 8 heads = 8 distinct 32-byte slots);
//   * fp32 online softmax per thread; the per-chunk (max, sum, acc[16]) partials are merged by attn_t2i_merge_kernel
//     in a fixed order (bit-reproducible).
// ------------------------------------------------------------------------------------
constexpr int T2I_HEADS = 8, T2I_HD = 16, T2I_D = 128, T2I_MAXQ = 16, T2I_PART = 18;  // partial = m, l, acc[16]

template <typename T> __device__ __forceinline__ void load_head16(const T* p, float (&f)[16]);
template <> __device__ __forceinline__ void load_head16<bf16_t>(const bf16_t* p, float (&f)[16]) {
  const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 8);
  const uint32_t u[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f[2 * i] = __uint_as_float(u[i] << 16);
    f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void load_head16<float>(const float* p, float (&f)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 a = *reinterpret_cast<const float4*>(p + 4 * i);
    f[4 * i] = a.x; f[4 * i + 1] = a.y; f[4 * i + 2] = a.z; f[4 * i + 3] = a.w;
  }
}

template <typename T, int TK>
__global__ __launch_bounds__(256) void attn_t2i_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                       float* __restrict__ part, int Nq, int Nk, int chunk_keys, int nchunks) {
  constexpr int EPC = 16 / (int)sizeof(T);          // elements per 16-byte load
  constexpr int SLOTS = T2I_D / EPC;                // 16-byte slots per row
  __shared__ __attribute__((aligned(16))) T sk[TK * T2I_D];
  __shared__ __attribute__((aligned(16))) T sv[TK * T2I_D];
  const int chunk = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int tid = threadIdx.x;
  const int sidx = tid >> 7, t = (tid >> 3) & 15, h = tid & 7;
  const bool active = t < Nq;
  float qv[T2I_HD];
  {
    const float scale = 0.25f;  // 1 / sqrt(16)
    const T* qp = q + (b * Nq + (active ? t : 0)) * (int64_t)T2I_D + h * T2I_HD;
#pragma unroll
    for (int d = 0; d < T2I_HD; ++d) qv[d] = active ? to_f32<T>(qp[d]) * scale : 0.f;
  }
  float m = -3.0e38f, l = 0.f, acc[T2I_HD];
#pragma unroll
  for (int d = 0; d < T2I_HD; ++d) acc[d] = 0.f;
  const int k0 = chunk * chunk_keys;
  const int k1 = min(Nk, k0 + chunk_keys);
  const T* kb = k + b * (int64_t)Nk * T2I_D;
  const T* vb = v + b * (int64_t)Nk * T2I_D;
  // software pipeline: the global loads of tile i+1 are in flight (in registers) while tile i is consumed from LDS
  constexpr int LPT = TK * SLOTS / 256;  // 16-byte loads per thread per tile and operand
  uint4 rk[LPT], rv[LPT];
  auto fetch = [&](int j0) {
    const int nkeys = min(TK, k1 - j0);
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int i = tid + u * 256;
      const int r = i / SLOTS, sl = i - r * SLOTS;
      rk[u] = make_uint4(0u, 0u, 0u, 0u);
      rv[u] = rk[u];
      if (r < nkeys) {
        rk[u] = *reinterpret_cast<const uint4*>(kb + (int64_t)(j0 + r) * T2I_D + sl * EPC);
        rv[u] = *reinterpret_cast<const uint4*>(vb + (int64_t)(j0 + r) * T2I_D + sl * EPC);
      }
    }
  };
  if (k0 < k1) fetch(k0);
  for (int j0 = k0; j0 < k1; j0 += TK) {
    const int nkeys = min(TK, k1 - j0);
#pragma unroll
    for (int u = 0; u < LPT; ++u) {
      const int i = tid + u * 256;
      *reinterpret_cast<uint4*>(sk + i * EPC) = rk[u];  // row-major [TK][128]: slot i of the tile
      *reinterpret_cast<uint4*>(sv + i * EPC) = rv[u];
    }
    __syncthreads();
    if (j0 + TK < k1) fetch(j0 + TK);
    if (active) {
      for (int r = sidx; r < nkeys; r += 2) {
        float kf[T2I_HD], vf[T2I_HD];
        load_head16<T>(sk + r * T2I_D + h * T2I_HD, kf);  // two 16-byte (bf16) / four (f32) LDS reads
        load_head16<T>(sv + r * T2I_D + h * T2I_HD, vf);
        float sc = 0.f;
#pragma unroll
        for (int d = 0; d < T2I_HD; ++d) sc = fmaf(qv[d], kf[d], sc);
        if (sc > m) {  // new running maximum (rare after the first keys): rescale what has been accumulated
          const float alpha = __expf(m - sc);
          l *= alpha;
#pragma unroll
          for (int d = 0; d < T2I_HD; ++d) acc[d] *= alpha;
          m = sc;
        }
        const float pe = __expf(sc - m);
        l += pe;
#pragma unroll
        for (int d = 0; d < T2I_HD; ++d) acc[d] = fmaf(pe, vf[d], acc[d]);
      }
    }
    __syncthreads();
  }
  if (active) {
    float* pp = part + ((((b * nchunks + chunk) * 2 + sidx) * T2I_MAXQ + t) * T2I_HEADS + h) * T2I_PART;
    pp[0] = m;
    pp[1] = l;
#pragma unroll
    for (int d = 0; d < T2I_HD; ++d) pp[2 + d] = acc[d];
  }
}

template <typename T>
__global__ void attn_t2i_merge_kernel(const float* __restrict__ part, T* __restrict__ o, int B, int Nq, int nparts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (b, t, h)
  if (i >= B * Nq * T2I_HEADS) return;
  const int h = i % T2I_HEADS, t = (i / T2I_HEADS) % Nq;
  const int64_t b = i / (T2I_HEADS * Nq);
  float m = -3.0e38f, l = 0.f, acc[T2I_HD];
#pragma unroll
  for (int d = 0; d < T2I_HD; ++d) acc[d] = 0.f;
  for (int pi = 0; pi < nparts; ++pi) {  // fixed order
    const float* pp = part + (((b * nparts + pi) * T2I_MAXQ + t) * T2I_HEADS + h) * T2I_PART;
    const float m2 = pp[0], l2 = pp[1];
    if (l2 == 0.f) continue;  // a part that saw no key
    const float mn = fmaxf(m, m2);
    const float a1 = __expf(m - mn), a2 = __expf(m2 - mn);
    l = l * a1 + l2 * a2;
#pragma unroll
    for (int d = 0; d < T2I_HD; ++d) acc[d] = acc[d] * a1 + pp[2 + d] * a2;
    m = mn;
  }
  T* op = o + (b * Nq + t) * (int64_t)T2I_D + h * T2I_HD;
  const float inv = 1.f / l;
#pragma unroll
  for (int d = 0; d < T2I_HD; ++d) op[d] = from_f32<T>(acc[d] * inv);
}

}  // namespace

// chunks of keys per prompt for the token -> image kernel, and the fp32 scratch it needs
static int t2i_chunks(int B, int Nk) {
  int n = (2 * 256 + B - 1) / B;  // about two workgroups per CU
  n = n < 1 ? 1 : (n > 32 ? 32 : n);
  const int max_by_keys = (Nk + 127) / 128;
  return n < max_by_keys ? n : max_by_keys;
}
int64_t esam3_attn_scratch_floats(int B, int Nq, int Nk, int heads, int hd) {
  if (heads != T2I_HEADS || hd != T2I_HD || Nq > T2I_MAXQ || Nk < 1024) return 0;
  return (int64_t)B * t2i_chunks(B, Nk) * 2 * T2I_MAXQ * T2I_HEADS * T2I_PART;
}

// merge of per-chunk (max, sum, acc[16]) partials [B][nparts][16 tokens][8 heads][18] -> o [B][Nq][128]
int esam3_launch_attn_t2i_merge(int dtype, const float* parts, void* o, int B, int Nq, int nparts, hipStream_t s) {
  const int total = B * Nq * T2I_HEADS;
  DISPATCH_T(dtype, hipLaunchKernelGGL((attn_t2i_merge_kernel<T>), dim3(blocks_for(total, 128)), dim3(128), 0, s, parts, (T*)o, B, Nq, nparts));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_attn(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv,
                      void* o, int ldo, int B, int Nq, int Nk, int heads, int hd, float* scratch, hipStream_t s) {
  if (scratch && esam3_attn_scratch_floats(B, Nq, Nk, heads, hd) > 0 && ldq == T2I_D && ldk == T2I_D && ldv == T2I_D &&
      ldo == T2I_D && !((uintptr_t)k & 15) && !((uintptr_t)v & 15)) {
    const int nch = t2i_chunks(B, Nk);
    int chunk_keys = (Nk + nch - 1) / nch;
    chunk_keys = (chunk_keys + 63) / 64 * 64;
    dim3 grid((unsigned)nch, (unsigned)B);
    if (dtype == 0)
      hipLaunchKernelGGL((attn_t2i_kernel<float, 32>), grid, dim3(256), 0, s, (const float*)q, (const float*)k, (const float*)v,
                         scratch, Nq, Nk, chunk_keys, nch);
    else
      hipLaunchKernelGGL((attn_t2i_kernel<bf16_t, 64>), grid, dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                         (const bf16_t*)v, scratch, Nq, Nk, chunk_keys, nch);
    HIP_CHECK_RET(hipGetLastError());
    const int total = B * Nq * T2I_HEADS;
    DISPATCH_T(dtype, hipLaunchKernelGGL((attn_t2i_merge_kernel<T>), dim3(blocks_for(total, 128)), dim3(128), 0, s, scratch,
                                         (T*)o, B, Nq, nch * 2));
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  dim3 grid((unsigned)(B * heads), (unsigned)((Nq + 15) / 16));
  if (hd == 16) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((attn_kernel<T, 16>), grid, dim3(256), 0, s, (const T*)q, ldq,
                                         (const T*)k, ldk, (const T*)v, ldv, (T*)o, ldo, Nq, Nk, heads));
  } else if (hd == 32) {
    DISPATCH_T(dtype, hipLaunchKernelGGL((attn_kernel<T, 32>), grid, dim3(256), 0, s, (const T*)q, ldq,
                                         (const T*)k, ldk, (const T*)v, ldv, (T*)o, ldo, Nq, Nk, heads));
  } else {
    esam3_set_error("attn: head dim %d unsupported", hd);
    return -1;
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_attn_fewkeys(int dtype, const void* q, int ldq, const void* k, int ldk, const void* v,
                              int ldv, void* o, int ldo, int B, int Nq, int Nk, int heads, int hd,
                              hipStream_t s) {
  if (hd != 16 || Nk > 64 || 256 % heads) {
    esam3_set_error("attn_fewkeys: hd=%d Nk=%d heads=%d unsupported", hd, Nk, heads);
    return -1;
  }
  const int qpb = 256 / heads;
  const size_t lds = sizeof(float) * 2 * (size_t)Nk * heads * hd;
  const int esz = dtype == 0 ? 4 : 2;
  auto al = [&](const void* p_, int ld) { return !(((uintptr_t)p_) & 15) && (ld * esz) % 16 == 0; };
  static const bool no_fk16 = esam3_dev_flag("ESAM3_NO_FEWKEYS16") != 0;  // A/B, bisecting
  if (!no_fk16 && Nk <= 16 && al(q, ldq) && al(o, ldo) && (heads * hd) % 4 == 0) {
    constexpr int QPT = 4;
    dim3 grid4((unsigned)((Nq + qpb * QPT - 1) / (qpb * QPT)), (unsigned)B);
    if (Nk <= 8) {
      DISPATCH_T(dtype, hipLaunchKernelGGL((attn_fewkeys16_kernel<T, 8, QPT>), grid4, dim3(256), lds, s, (const T*)q, ldq,
                                           (const T*)k, ldk, (const T*)v, ldv, (T*)o, ldo, Nq, Nk, heads));
    } else {
      DISPATCH_T(dtype, hipLaunchKernelGGL((attn_fewkeys16_kernel<T, 16, QPT>), grid4, dim3(256), lds, s, (const T*)q, ldq,
                                           (const T*)k, ldk, (const T*)v, ldv, (T*)o, ldo, Nq, Nk, heads));
    }
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  dim3 grid((unsigned)((Nq + qpb - 1) / qpb), (unsigned)B);
  DISPATCH_T(dtype, hipLaunchKernelGGL((attn_fewkeys_kernel<T, 16>), grid, dim3(256), lds, s,
                                       (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv, (T*)o, ldo,
                                       Nq, Nk, heads));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_build_tokens(int dtype, const float* out_tokens, const float* coords, const int* labels,
                              const float* gauss, const float* point_emb, const float* not_a_point,
                              void* tokens, int Bp, int Np, int pad, float img_size, hipStream_t s) {
  const int Tn = 6 + Np + (pad ? 1 : 0);
  const int64_t total = (int64_t)Bp * Tn * 256;
  DISPATCH_T(dtype, hipLaunchKernelGGL(build_tokens_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0,
                                       s, out_tokens, coords, labels, gauss, point_emb, not_a_point,
                                       (T*)tokens, Bp, Np, pad, img_size));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_text_embed(int dtype, const int64_t* tokens, const float* table, const float* pos, void* x,
                            float* embeds_sbd, int B, int S, int D, int vocab, hipStream_t s) {
  const int64_t total = (int64_t)B * S * D;
  DISPATCH_T(dtype, hipLaunchKernelGGL(text_embed_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, tokens, table,
                                       pos, (T*)x, embeds_sbd, B, S, D, vocab));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_seq_dwconv(int dtype, const void* x, const float* w, const float* bias, void* out, int B, int S, int D,
                            int KW, hipStream_t s) {
  const int64_t total = (int64_t)B * S * D;
  DISPATCH_T(dtype, hipLaunchKernelGGL(seq_dwconv_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s, (const T*)x, w,
                                       bias, (T*)out, B, S, D, KW));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_text_attn(int dtype, const void* qkv, void* out, int B, int S, int heads, int hd, int causal, hipStream_t s) {
  if (hd != 64) { esam3_set_error("text_attn: head dim %d unsupported", hd); return -1; }
  DISPATCH_T(dtype, hipLaunchKernelGGL(text_attn_kernel<T>, dim3((unsigned)(B * heads * S)), dim3(64), 0, s, (const T*)qkv,
                                       (T*)out, S, heads, causal));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_bsc_to_sbc_f32(int dtype, const void* x, float* out, int B, int S, int C, hipStream_t s) {
  const int64_t total = (int64_t)B * S * C;
  DISPATCH_T(dtype, hipLaunchKernelGGL(bsc_to_sbc_f32_kernel<T>, dim3(blocks_for(total, 256)), dim3(256), 0, s,
                                       (const T*)x, out, B, S, C));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_mask_embed(int dtype, const float* mask, const float* const* w, void* out, int Bp, int in_size,
                            int emb_size, hipStream_t s) {
  if (in_size != 4 * emb_size) { esam3_set_error("mask_embed: %d != 4*%d", in_size, emb_size); return -1; }
  dim3 grid(blocks_for((int64_t)emb_size * emb_size, 64), (unsigned)Bp);
  DISPATCH_T(dtype, hipLaunchKernelGGL(mask_embed_kernel<T>, grid, dim3(256), 0, s, mask, w[0], w[1], w[2], w[3], w[4],
                                       w[5], w[6], w[7], w[8], w[9], (T*)out, in_size, emb_size));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_mask_product(int dtype, const void* hyper, int ld_h, const void* up, float* masks,
                              int Bp, int64_t P, int C, hipStream_t s) {
  if (C != 32) { esam3_set_error("mask_product: C=%d unsupported", C); return -1; }
  dim3 grid(blocks_for(P, 256), (unsigned)Bp);
  DISPATCH_T(dtype, hipLaunchKernelGGL((mask_product_kernel<T, 32>), grid, dim3(256), 0, s,
                                       (const T*)hyper, ld_h, (const T*)up, masks, P));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_upscale_mask(const void* u1, const void* wt, int kp, const float* bias, const void* feat, const int* img_of,
                              const void* hyper, int ld_h, float* masks, int Bp, int S, hipStream_t s) {
  if (kp % 8 || (((uintptr_t)u1) & 15) || (((uintptr_t)wt) & 15) || (((uintptr_t)feat) & 7) || (((uintptr_t)masks) & 7)) {
    esam3_set_error("upscale_mask: misaligned operands");
    return -1;
  }
  const int frags = (S * S + 31) / 32;
  const int fpw = 3;  // fragments per wave: the weights / hypernetwork vectors are loaded once per wave
  dim3 grid((unsigned)((frags + 4 * fpw - 1) / (4 * fpw)), (unsigned)Bp);
  hipLaunchKernelGGL(upscale_mask_kernel, grid, dim3(256), 0, s, (const bf16_t*)u1, (const bf16_t*)wt, kp, bias, (const bf16_t*)feat,
                     img_of, (const bf16_t*)hyper, ld_h, masks, S, fpw);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_select_masks(int dtype, const float* all_masks, const void* all_iou, int ld_iou,
                              float* out_masks, float* out_iou, int* counters, int Bp, int64_t P,
                              int multimask, float delta, float thresh, hipStream_t s) {
  if (!multimask) {
    HIP_CHECK_RET(hipMemsetAsync(counters, 0, sizeof(int) * 2 * (size_t)Bp, s));
    hipLaunchKernelGGL(stability_count_kernel, dim3(8, (unsigned)Bp), dim3(1024), 0, s, all_masks,
                       counters, P, delta);
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL(select_masks_kernel<T>, dim3(64, (unsigned)Bp), dim3(256), 0, s,
                                       all_masks, (const T*)all_iou, ld_iou, out_masks, out_iou, counters,
                                       P, multimask, thresh));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_fill_holes(const float* in, float* out, int* labels, int* areas, int n, int H, int W,
                            float thr, float max_area, hipStream_t s) {
  if (n <= 0) return 0;
  const int HW = H * W;
  const int64_t total = (int64_t)n * HW;
  const dim3 grid(blocks_for(total, 256)), blk(256);
  if (W <= CC_TPX && (int64_t)n * H < (1ll << 30) && !esam3_dev_flag("ESAM3_CC_OLD")) {   // strips in LDS + seam merge
    const int TH = CC_TPX / W < H ? CC_TPX / W : H, strips = (H + TH - 1) / TH;
    const int lds = 2 * CC_TPX * (int)sizeof(int);
    if (esam3_allow_dyn_lds((const void*)cc_tile_kernel, lds)) return -1;   // memoised per (device, kernel): every launch asks, as the other kernels do
    hipLaunchKernelGGL(cc_tile_kernel, dim3((unsigned)(n * strips)), dim3(1024), lds, s, in, labels, areas, W, H, TH, strips, thr, (int)max_area);
    if (strips > 1) {
      const int nt = n * (strips - 1) * W;
      hipLaunchKernelGGL(cc_seam_kernel, dim3((unsigned)((nt + 255) / 256)), blk, 0, s, labels, W, H, TH, strips, n);
      hipLaunchKernelGGL(cc_sum_kernel, grid, blk, 0, s, labels, areas, HW, total, (int)max_area);
    }
    hipLaunchKernelGGL(cc_apply_tiled_kernel, grid, blk, 0, s, in, out, labels, areas, HW, total, thr, max_area);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(cc_init_kernel, grid, blk, 0, s, in, labels, areas, W, HW, total, thr);
  hipLaunchKernelGGL(cc_merge_kernel, grid, blk, 0, s, in, labels, W, H, HW, total, thr);
  hipLaunchKernelGGL(cc_count_kernel, grid, blk, 0, s, labels, areas, HW, total, (int)max_area);
  hipLaunchKernelGGL(cc_apply_kernel, grid, blk, 0, s, in, out, labels, areas, HW, total, thr, max_area);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_upsample_masks(const float* in, float* out_f32, uint8_t* out_u8, int n, int IH, int IW,
                                int OH, int OW, float thr, hipStream_t s) {
  if (n <= 0) return 0;
  if (esam3_dev_flag("ESAM3_UPSAMPLE_OLD") || OH > 65535 || n > 65535 || IW > 4096) {   // A/B (dev builds) / grid limits: one thread per pixel
    dim3 grid(blocks_for((int64_t)OH * OW, 256), (unsigned)n);
    hipLaunchKernelGGL(upsample_masks_px_kernel, grid, dim3(256), 0, s, in, out_f32, out_u8, IH, IW, OH, OW, thr);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  dim3 grid(1, (unsigned)OH, (unsigned)n);   // one workgroup per output row: its two source rows in LDS, 8 pixels per thread and step
  hipLaunchKernelGGL(upsample_masks_kernel, grid, dim3(128), (size_t)2 * IW * sizeof(float), s, in, out_f32, out_u8, IH, IW, OH, OW,
                     thr);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_clamp(float* x, int64_t n, float lo, float hi, hipStream_t s) {
  if (n <= 0) return 0;
  const unsigned g = (unsigned)min((int64_t)4096, (n + 255) / 256);
  hipLaunchKernelGGL(clamp_kernel, dim3(g), dim3(256), 0, s, x, n, lo, hi);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_add_f32_to_bf16(const float* a, const float* b, void* out, int64_t n, hipStream_t s) {
  int64_t g = (n + 255) / 256;
  if (g > 4096) g = 4096;
  if (g < 1) g = 1;
  hipLaunchKernelGGL(add_f32_to_bf16_kernel, dim3((unsigned)g), dim3(256), 0, s, a, b, (bf16_t*)out, n);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_add(int dtype, const void* a, const void* b, void* out, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  const unsigned g = (unsigned)min((int64_t)4096, (n + 255) / 256);
  DISPATCH_T(dtype, hipLaunchKernelGGL(add_kernel<T>, dim3(g), dim3(256), 0, s, (const T*)a, (const T*)b,
                                       (T*)out, n));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_launch_strided_to_f32(int dtype, const void* in, int ld, float* out, int64_t n, hipStream_t s) {
  if (n <= 0) return 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL(strided_to_f32_kernel<T>, dim3(blocks_for(n, 256)), dim3(256), 0, s,
                                       (const T*)in, ld, out, n));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
