// Stage-1 training path, the TinyViT students (stage1/model.py:397-406 -> TinyViTAdapter over
// sam3/backbones/tiny_vit.py under model.train(), stage1/train_image_encoder_stage1.py:165-226).  What a TinyViTBlock needs beyond the
// convolution / BatchNorm / activation kernels of kernels_train.hip:
//   esam3_ln_train_forward / _backward     nn.LayerNorm over the channels of [M][C] rows (Attention.norm, Mlp.norm: tiny_vit.py:201,236):
//                                          y, saved mean / rstd; dx, dgamma, dbeta (per-row statistics on one wavefront, the two column
//                                          sums over fixed row partitions: deterministic)
//   esam3_win_attn_train_forward / _backward   Attention.forward without its Linear layers (tiny_vit.py:271-293): per window and head
//                                          softmax(q k^T scale + bias) v on [windows][N][heads * 96] qkv rows (per head q | k | v of 32
//                                          channels), N = 49 | 196 tokens; backward gives d(qkv) and dS = the gradient of the bias-added
//                                          logits per window (the bias gradient is its sum over the windows, esam3_colsum)
//   esam3_attn_bias_gather_sum             d(attention_biases)[h][o] = sum of dBias[h][i][j] over the (i, j) with
//                                          attention_bias_idxs[i][j] = o (tiny_vit.py:240-254), fixed order
// All arithmetic fp32; activations / gradients fp32 or bf16.  One workgroup per (window, head): q / dO rows live in registers, the other
// operand's rows are read from LDS as broadcasts (every lane the same address), so there are no bank conflicts by construction.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/esam3.h"
#include "esam3_common.h"
#include "kernels.h"

namespace {

template <int DT> struct VElem;  // 0 f32, 1 bf16
template <> struct VElem<0> {
  using type = float;
  static __device__ inline void load8(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ inline void store8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct VElem<1> {
  using type = uint16_t;
  static __device__ inline void load8(const uint16_t* p, float* v) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ inline void store8(uint16_t* p, const float* v) {
    uint4 o;
    o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = o;
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- LayerNorm -----------------------------------------------------------------------------------------------------------------------------
// one wavefront per row; a lane holds the 8-channel groups lane, lane + 64 (C <= 1024).  Two-pass statistics (mean, then the centred
// second moment), biased variance, as torch.nn.functional.layer_norm.
constexpr int LN_G = 2;
template <int DT>
__global__ __launch_bounds__(256) void ln_forward_kernel(const typename VElem<DT>::type* __restrict__ x, typename VElem<DT>::type* __restrict__ y,
                                                         int64_t M, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, C8 = C / 8;
  const float inv_c = 1.f / (float)C;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    float v[LN_G][8];
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < LN_G; ++g) {
      const int cg = lane + 64 * g;
      if (cg < C8) {
        VElem<DT>::load8(x + row * C + cg * 8, v[g]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += v[g][e];
      }
    }
    const float mean = wave_sum(s) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int g = 0; g < LN_G; ++g)
      if (lane + 64 * g < C8) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = v[g][e] - mean;
          q += d * d;
        }
      }
    const float rstd = rsqrtf(wave_sum(q) * inv_c + eps);
#pragma unroll
    for (int g = 0; g < LN_G; ++g) {
      const int cg = lane + 64 * g;
      if (cg < C8) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[g][e] - mean) * rstd * gamma[cg * 8 + e] + beta[cg * 8 + e];
        VElem<DT>::store8(y + row * C + cg * 8, o);
      }
    }
    if (lane == 0) {
      mean_out[row] = mean;
      rstd_out[row] = rstd;
    }
  }
}

// dx = rstd (g - mean_c(g) - xhat mean_c(g xhat)), g = dy gamma; dgamma = sum_rows dy xhat, dbeta = sum_rows dy.
// A wave walks the rows (wave index + k * waves of the grid): the set of rows a lane sums is a function of (M, grid) only.
// partial [block][2][C]; ln_reduce_kernel adds the blocks in a fixed order.
constexpr int LN_BLOCKS_MAX = 512;
template <int DT>
__global__ __launch_bounds__(256) void ln_backward_kernel(const typename VElem<DT>::type* __restrict__ x, const typename VElem<DT>::type* __restrict__ dy,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean_in,
                                                          const float* __restrict__ rstd_in, typename VElem<DT>::type* __restrict__ dx, int64_t M,
                                                          int C, float* __restrict__ partial) {
  __shared__ float red[4][2][LN_G * 64 * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, C8 = C / 8;
  const float inv_c = 1.f / (float)C;
  float ag[LN_G][8], ab[LN_G][8];
#pragma unroll
  for (int g = 0; g < LN_G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) ag[g][e] = ab[g][e] = 0.f;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < M; row += (int64_t)gridDim.x * 4) {
    const float mean = mean_in[row], rstd = rstd_in[row];
    float xh[LN_G][8], gg[LN_G][8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int g = 0; g < LN_G; ++g) {
      const int cg = lane + 64 * g;
      if (cg < C8) {
        float xv[8], dv[8];
        VElem<DT>::load8(x + row * C + cg * 8, xv);
        VElem<DT>::load8(dy + row * C + cg * 8, dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          xh[g][e] = (xv[e] - mean) * rstd;
          gg[g][e] = dv[e] * gamma[cg * 8 + e];
          s1 += gg[g][e];
          s2 += gg[g][e] * xh[g][e];
          ag[g][e] += dv[e] * xh[g][e];
          ab[g][e] += dv[e];
        }
      }
    }
    const float a = wave_sum(s1) * inv_c, b = wave_sum(s2) * inv_c;
#pragma unroll
    for (int g = 0; g < LN_G; ++g) {
      const int cg = lane + 64 * g;
      if (cg < C8) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rstd * (gg[g][e] - a - xh[g][e] * b);
        VElem<DT>::store8(dx + row * C + cg * 8, o);
      }
    }
  }
#pragma unroll
  for (int g = 0; g < LN_G; ++g)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[wave][0][(g * 64 + lane) * 8 + e] = ag[g][e];
      red[wave][1][(g * 64 + lane) * 8 + e] = ab[g][e];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * C; i += 256) {
    const int which = i / C, c = i - which * C;
    const int cg = c / 8, e = c & 7, g = cg / 64, ln = cg - g * 64;
    const int idx = (g * 64 + ln) * 8 + e;
    partial[(int64_t)blockIdx.x * 2 * C + i] = (red[0][which][idx] + red[1][which][idx]) + (red[2][which][idx] + red[3][which][idx]);
  }
}
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ partial, int blocks, int n, float* __restrict__ out_a, float* __restrict__ out_b,
                                                        int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = 0;
  for (; b + 3 < blocks; b += 4) {
    s0 += partial[(int64_t)b * n + i];
    s1 += partial[(int64_t)(b + 1) * n + i];
    s2 += partial[(int64_t)(b + 2) * n + i];
    s3 += partial[(int64_t)(b + 3) * n + i];
  }
  for (; b < blocks; ++b) s0 += partial[(int64_t)b * n + i];
  const float t = (s0 + s1) + (s2 + s3);
  if (i < C) out_a[i] = t;
  else out_b[i - C] = t;
}
int ln_blocks(int64_t M) {
  const int64_t want = (M + 3) / 4;
  return (int)(want < LN_BLOCKS_MAX ? want : LN_BLOCKS_MAX);
}

// ---- window attention, head dim 32 ---------------------------------------------------------------------------------------------------------
constexpr int HD = 32;

// stage rows [N][32] of one (window, head) operand into LDS as fp32: `src` points at the operand's first channel of token 0, rows `ld` apart
template <int DT>
__device__ __forceinline__ void stage_rows(const typename VElem<DT>::type* __restrict__ src, int ld, int N, float* __restrict__ dst) {
  for (int c = threadIdx.x; c < N * 4; c += blockDim.x) {
    const int r = c >> 2, part = c & 3;
    float v[8];
    VElem<DT>::load8(src + (int64_t)r * ld + part * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) dst[r * HD + part * 8 + e] = v[e];
  }
}
template <int DT>
__device__ __forceinline__ void load_row(const typename VElem<DT>::type* __restrict__ src, float (&v)[HD]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) VElem<DT>::load8(src + p * 8, &v[p * 8]);
}
template <int DT>
__device__ __forceinline__ void store_row(typename VElem<DT>::type* __restrict__ dst, const float (&v)[HD]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) VElem<DT>::store8(dst + p * 8, &v[p * 8]);
}
__device__ __forceinline__ float dot32(const float (&a)[HD], const float* __restrict__ b) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int c = 0; c < HD; c += 4) {
    const float4 w = *reinterpret_cast<const float4*>(b + c);
    s0 += a[c] * w.x; s1 += a[c + 1] * w.y; s2 += a[c + 2] * w.z; s3 += a[c + 3] * w.w;
  }
  return (s0 + s1) + (s2 + s3);
}

// forward: thread i owns query row i; K, V of the (window, head) in LDS; online softmax over the N keys.
// The bias table is symmetric (offsets are |p1 - p2|, tiny_vit.py:245): row i is read as column i, coalesced over the threads.
template <int DT>
__global__ __launch_bounds__(256) void win_attn_forward_kernel(const typename VElem<DT>::type* __restrict__ qkv, const float* __restrict__ bias,
                                                               typename VElem<DT>::type* __restrict__ out, float* __restrict__ lse, int N, int heads,
                                                               float scale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sk = lds;
  float* sv = lds + N * HD;
  const int w = blockIdx.x, h = blockIdx.y, ld = heads * 3 * HD;
  const typename VElem<DT>::type* base = qkv + (int64_t)w * N * ld + h * 3 * HD;
  stage_rows<DT>(base + HD, ld, N, sk);
  stage_rows<DT>(base + 2 * HD, ld, N, sv);
  __syncthreads();
  const int i = threadIdx.x;
  if (i >= N) return;
  float q[HD], o[HD];
  load_row<DT>(base + (int64_t)i * ld, q);
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    q[c] *= scale;
    o[c] = 0.f;
  }
  const float* brow = bias + (int64_t)h * N * N + i;   // bias[h][j][i] = bias[h][i][j]
  float m = -3.0e38f, l = 0.f;
  for (int j = 0; j < N; ++j) {
    const float s = dot32(q, sk + j * HD) + brow[(int64_t)j * N];
    const float mn = fmaxf(m, s);
    const float corr = __expf(m - mn), p = __expf(s - mn);
    l = l * corr + p;
    const float* vj = sv + j * HD;
#pragma unroll
    for (int c = 0; c < HD; ++c) o[c] = o[c] * corr + p * vj[c];
    m = mn;
  }
  const float inv = 1.f / l;
#pragma unroll
  for (int c = 0; c < HD; ++c) o[c] *= inv;
  store_row<DT>(out + ((int64_t)w * N + i) * (heads * HD) + h * HD, o);
  lse[((int64_t)w * heads + h) * N + i] = m + __logf(l);
}

// backward.  P_ij = exp(s_ij - lse_i); dP_ij = dO_i . v_j; D_i = dO_i . O_i (= sum_j P_ij dP_ij); dS_ij = P_ij (dP_ij - D_i);
//   phase 0 (blockIdx.z = 0, thread = query row i, K / V rows broadcast from LDS):   dq_i = scale sum_j dS_ij k_j
//   phase 1 (blockIdx.z = 1, thread = key row j, Q / dO rows broadcast from LDS):    dk_j = scale sum_i dS_ij q_i,  dv_j = sum_i P_ij dO_i,
//                                                                                    dS written [window][head][i][j] (coalesced over j)
// The two phases are separate workgroups: each stages only the two operands it broadcasts (2 N 32 floats: three workgroups fit a CU where
// one with all four did not -- the single-workgroup form spent 15 ms of a 54 ms TinyViT-11M step, profiles/r05/stage1_step_tiny_vit_11m_kernel_stats.csv).
template <int DT>
__global__ __launch_bounds__(256) void win_attn_backward_kernel(const typename VElem<DT>::type* __restrict__ qkv, const float* __restrict__ bias,
                                                                const typename VElem<DT>::type* __restrict__ out, const float* __restrict__ lse,
                                                                const typename VElem<DT>::type* __restrict__ dout,
                                                                typename VElem<DT>::type* __restrict__ dqkv, float* __restrict__ ds_out, int N, int heads,
                                                                float scale) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* sa = lds;                 // phase 0: K      phase 1: Q
  float* sb = sa + N * HD;         // phase 0: V      phase 1: dO
  float* slse = sb + N * HD;
  float* sd = slse + N;
  const int w = blockIdx.x, h = blockIdx.y, phase = blockIdx.z, ld = heads * 3 * HD, ldo = heads * HD;
  const typename VElem<DT>::type* base = qkv + (int64_t)w * N * ld + h * 3 * HD;
  const typename VElem<DT>::type* dobase = dout + (int64_t)w * N * ldo + h * HD;
  if (phase == 0) {
    stage_rows<DT>(base + HD, ld, N, sa);
    stage_rows<DT>(base + 2 * HD, ld, N, sb);
  } else {
    stage_rows<DT>(base, ld, N, sa);
    stage_rows<DT>(dobase, ldo, N, sb);
  }
  const int t = threadIdx.x;
  float mine[HD], other[HD];      // phase 0: q_i (scaled), dO_i; phase 1: k_j (scaled), v_j
  if (t < N) {
    load_row<DT>(dobase + (int64_t)t * ldo, other);
    load_row<DT>(out + ((int64_t)w * N + t) * ldo + h * HD, mine);
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < HD; ++c) d += other[c] * mine[c];
    sd[t] = d;
    slse[t] = lse[((int64_t)w * heads + h) * N + t];
  }
  __syncthreads();
  if (t >= N) return;
  const float* bcol = bias + (int64_t)h * N * N + t;     // bias[h][.][t]: symmetric table, coalesced over the threads
  if (phase == 0) {
    load_row<DT>(base + (int64_t)t * ld, mine);           // q_i; `other` still holds dO_i
#pragma unroll
    for (int c = 0; c < HD; ++c) mine[c] *= scale;
    float dq[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dq[c] = 0.f;
    const float lse_i = slse[t], d_i = sd[t];
    for (int j = 0; j < N; ++j) {
      const float* kj = sa + j * HD;
      const float s = dot32(mine, kj) + bcol[(int64_t)j * N];
      const float p = __expf(s - lse_i);
      const float dsv = p * (dot32(other, sb + j * HD) - d_i) * scale;
#pragma unroll
      for (int c = 0; c < HD; ++c) dq[c] += dsv * kj[c];
    }
    store_row<DT>(dqkv + ((int64_t)w * N + t) * ld + h * 3 * HD, dq);
  } else {
    load_row<DT>(base + (int64_t)t * ld + HD, mine);        // k_j
    load_row<DT>(base + (int64_t)t * ld + 2 * HD, other);   // v_j
#pragma unroll
    for (int c = 0; c < HD; ++c) mine[c] *= scale;
    float dk[HD], dv[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) dk[c] = dv[c] = 0.f;
    float* dsrow = ds_out + ((int64_t)w * heads + h) * N * N + t;
    for (int i = 0; i < N; ++i) {
      const float* qi = sa + i * HD;
      const float* doi = sb + i * HD;
      const float s = dot32(mine, qi) + bcol[(int64_t)i * N];     // scale q_i . k_j + bias[h][i][j]
      const float p = __expf(s - slse[i]);
      const float dsv = p * (dot32(other, doi) - sd[i]);
      dsrow[(int64_t)i * N] = dsv;
      const float dss = dsv * scale;
#pragma unroll
      for (int c = 0; c < HD; ++c) {
        dv[c] += p * doi[c];
        dk[c] += dss * qi[c];
      }
    }
    store_row<DT>(dqkv + ((int64_t)w * N + t) * ld + h * 3 * HD + HD, dk);
    store_row<DT>(dqkv + ((int64_t)w * N + t) * ld + h * 3 * HD + 2 * HD, dv);
  }
}

// ---- round 6: the same two operations on the matrix unit (bf16 activations) ---------------------------------------------------------------
// The kernels above spend ~320 fp32 multiply-adds per (query, key) pair on the VALU: 37 ms of a 102 ms TinyViT-11M batch-32 step
// (profiles/r06/r06_kernel_stats_stage1_step_tiny_vit_11m_b32.csv).  Under autocast the reference runs these products in bf16 with fp32
// accumulation (tiny_vit.py:283-291 inside torch.autocast); so do these: v_mfma_f32_32x32x16_bf16 on 32 x 32 tiles of the N x N logits.
// Every product is written in ONE form, C[m][n] = sum_k X[m][k] Y[n][k] with k contiguous in both operands (a lane's 8 k-values are one
// 16-byte LDS read), which asks for
//   - q, k, v, dO staged row-major [token][32] AND (q, k, dO; v in the forward) transposed [32][token] in LDS;
//   - P and dS tiles rounded to bf16 and passed through a per-wave LDS scratch, written in the orientation the next product reads.
// One workgroup per (window, head), NW = 4 or 8 waves (8 when the 7 tiles of a 196-token window would otherwise take two rounds of four waves and
// the LDS still fits); N is padded to 32-token tiles with zero rows (their P / dS are forced to 0).
// Backward: phase B gives a wave the key tiles J = wave, wave + 4, .. (dK_J, dV_J accumulated over all query tiles, dS written for the bias
// gradient), phase A the query tiles (dQ_I over all key tiles); logits and probabilities are recomputed in both (4 of the 14 MFMAs per tile pair).
typedef float wa_f32x16 __attribute__((ext_vector_type(16)));
constexpr int WA_P = 40;   // halves per staged row of 32 (80 bytes)

__device__ __forceinline__ bf16x8_v wa_op(const uint16_t* base, int row0, int pitch, int k0, int lane) {
  return *reinterpret_cast<const bf16x8_v*>(base + (row0 + (lane & 31)) * pitch + k0 + 8 * (lane >> 5));
}
// acc += X[rows x0 ..][k] Y[rows y0 ..][k] over 32 k-values starting at kx0 / ky0
__device__ __forceinline__ wa_f32x16 wa_nt(wa_f32x16 acc, const uint16_t* X, int x0, int xp, int kx0, const uint16_t* Y, int y0, int yp, int ky0, int lane) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa_op(X, x0, xp, kx0 + 16 * s, lane), wa_op(Y, y0, yp, ky0 + 16 * s, lane), acc, 0, 0, 0);
  return acc;
}
__device__ __forceinline__ int wa_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }   // accumulator register -> tile row
__device__ __forceinline__ wa_f32x16 wa_zero() {
  wa_f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}
// rows [0, NP) of a [N][ld] bf16 matrix (32 channels from `src`) -> natural [NP][WA_P] and (if T) transposed [32][TP]; zeros past N
__device__ __forceinline__ void wa_stage(const uint16_t* __restrict__ src, int ld, int N, int NP, uint16_t* nat, uint16_t* T, int TP) {
  for (int c = threadIdx.x; c < NP * 4; c += blockDim.x) {
    const int r = c >> 2, part = c & 3;
    uint4 q = make_uint4(0u, 0u, 0u, 0u);
    if (r < N) q = *reinterpret_cast<const uint4*>(src + (int64_t)r * ld + part * 8);
    if (nat) *reinterpret_cast<uint4*>(nat + r * WA_P + part * 8) = q;
    if (T) {
      const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        T[(part * 8 + 2 * e) * TP + r] = (uint16_t)(w[e] & 0xffffu);
        T[(part * 8 + 2 * e + 1) * TP + r] = (uint16_t)(w[e] >> 16);
      }
    }
  }
}

template <bool TAB, int NW>
__global__ __launch_bounds__(64 * NW) void win_attn_forward_mfma_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ bias,
                                                                    uint16_t* __restrict__ out, float* __restrict__ lse, int N, int heads, float scale,
                                                                    const float* __restrict__ tab, int ws) {
  // `tab` [heads][ws ws] (N = ws ws): the attention_biases parameter itself (tiny_vit.py:240-254: offset (|dy|, |dx|) has index |dy| ws + |dx|);
  // the head's row is staged in LDS and indexed per pair -- with the gathered [N][N] table in global memory every tile step waited ~1.4 us for
  // its 16 bias values with one wave per SIMD to hide it (the first MFMA form was no faster than the VALU kernels: profiles/r06/wattn_mfma_first.txt)
  extern __shared__ __attribute__((aligned(16))) uint16_t wsm[];
  constexpr int NTMAX = 8;
  const int NP = (N + 31) & ~31, NT = NP >> 5, TP = NP + 8;
  uint16_t* sQ = wsm;
  uint16_t* sK = sQ + NP * WA_P;
  uint16_t* sVT = sK + NP * WA_P;          // [32][TP]
  uint16_t* scr = sVT + 32 * TP;           // [NW waves][32][WA_P]
  float* sl = reinterpret_cast<float*>(scr + NW * 32 * WA_P);   // [NW waves][32]
  float* stab = sl + NW * 32;                                   // [256]
  uint16_t* spos = reinterpret_cast<uint16_t*>(stab + 256);     // [NP]: (row << 8) | column of the token in its window
  const int w = blockIdx.x, h = blockIdx.y, ld = heads * 3 * HD, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint16_t* base = qkv + (int64_t)w * N * ld + h * 3 * HD;
  if constexpr (TAB) {
    for (int t = threadIdx.x; t < 256; t += 64 * NW) stab[t] = t < N ? tab[(int64_t)h * N + t] : 0.f;
    for (int t = threadIdx.x; t < NP; t += 64 * NW) spos[t] = t < N ? (uint16_t)(((t / ws) << 8) | (t % ws)) : (uint16_t)0;
  }
  wa_stage(base, ld, N, NP, sQ, nullptr, TP);
  wa_stage(base + HD, ld, N, NP, sK, nullptr, TP);
  wa_stage(base + 2 * HD, ld, N, NP, nullptr, sVT, TP);
  __syncthreads();
  uint16_t* myscr = scr + wave * 32 * WA_P;
  float* mysl = sl + wave * 32;
  const float* bh = bias + (int64_t)h * N * N;
  auto bias_at = [&](int a, int b) -> float {   // a, b < N; the table is symmetric
    if constexpr (TAB) {
      const int pa = spos[a], pb = spos[b];
      const int dr = (pa >> 8) - (pb >> 8), dc = (pa & 255) - (pb & 255);
      return stab[(dr < 0 ? -dr : dr) * ws + (dc < 0 ? -dc : dc)];
    } else {
      return bh[(int64_t)a * N + b];
    }
  };
  for (int I = wave; I < NT; I += NW) {
    const int i = I * 32 + (lane & 31);        // this lane's query (the tiles are S^T: column = query, rows = keys)
    wa_f32x16 st[NTMAX];
    float mx = -3.0e38f;
#pragma unroll
    for (int J = 0; J < NTMAX; ++J) {
      if (J < NT) {
        st[J] = wa_nt(wa_zero(), sK, J * 32, WA_P, 0, sQ, I * 32, WA_P, 0, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = J * 32 + wa_row(r, lane);
          const bool ok = j < N && i < N;
          const float sv = ok ? st[J][r] * scale + bias_at(j, i) : -3.0e38f;   // symmetric table: [j][i] = [i][j], coalesced over i
          st[J][r] = sv;
          mx = fmaxf(mx, sv);
        }
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float l = 0.f;
#pragma unroll
    for (int J = 0; J < NTMAX; ++J)
      if (J < NT) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = st[J][r] > -1.0e38f ? __expf(st[J][r] - mx) : 0.f;
          st[J][r] = pv;
          l += pv;
        }
      }
    l += __shfl_xor(l, 32);
    wa_f32x16 o = wa_zero();
#pragma unroll
    for (int J = 0; J < NTMAX; ++J)
      if (J < NT) {
        // P tile natural [query][key]: this lane's query row, keys 8 q4 + 4 g + (0..3): one 8-byte write per q4
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const uint32_t lo = pack_bf16x2(st[J][q4 * 4], st[J][q4 * 4 + 1]), hi = pack_bf16x2(st[J][q4 * 4 + 2], st[J][q4 * 4 + 3]);
          *reinterpret_cast<uint2*>(myscr + (lane & 31) * WA_P + 8 * q4 + 4 * (lane >> 5)) = make_uint2(lo, hi);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        o = wa_nt(o, myscr, 0, WA_P, 0, sVT, 0, TP, J * 32, lane);   // O[i][c] += sum_j P[i][j] V[j][c]
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // the reads are done before the next tile overwrites the scratch
      }
    if (lane < 32) mysl[lane] = l;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // o: column = channel (lane & 31), rows = queries of the tile
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int il = wa_row(r, lane), ig = I * 32 + il;
      if (ig < N) out[((int64_t)w * N + ig) * (heads * HD) + h * HD + (lane & 31)] = f32_to_bf16(o[r] / mysl[il]);
    }
    if (lane < 32 && i < N) lse[((int64_t)w * heads + h) * N + i] = mx + __logf(l);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

template <bool TAB, int NW>
__global__ __launch_bounds__(64 * NW) void win_attn_backward_mfma_kernel(const uint16_t* __restrict__ qkv, const float* __restrict__ bias,
                                                                     const uint16_t* __restrict__ out, const float* __restrict__ lse,
                                                                     const uint16_t* __restrict__ dout, uint16_t* __restrict__ dqkv,
                                                                     float* __restrict__ ds_out, int N, int heads, float scale,
                                                                     const float* __restrict__ tab, int ws) {
  extern __shared__ __attribute__((aligned(16))) uint16_t wsm[];
  const int NP = (N + 31) & ~31, NT = NP >> 5, TP = NP + 8;
  uint16_t* sQ = wsm;
  uint16_t* sK = sQ + NP * WA_P;
  uint16_t* sV = sK + NP * WA_P;
  uint16_t* sdO = sV + NP * WA_P;
  uint16_t* sQT = sdO + NP * WA_P;         // [32][TP]
  uint16_t* sKT = sQT + 32 * TP;
  uint16_t* sdOT = sKT + 32 * TP;
  uint16_t* scr = sdOT + 32 * TP;          // [NW waves][2][32][WA_P]
  float* slse = reinterpret_cast<float*>(scr + NW * 2 * 32 * WA_P);   // [NP]
  float* sd = slse + NP;                                              // [NP]
  float* stab = sd + NP;                                              // [256]
  uint16_t* spos = reinterpret_cast<uint16_t*>(stab + 256);           // [NP]
  const int w = blockIdx.x, h = blockIdx.y, ld = heads * 3 * HD, ldo = heads * HD, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if constexpr (TAB) {
    for (int t = threadIdx.x; t < 256; t += 64 * NW) stab[t] = t < N ? tab[(int64_t)h * N + t] : 0.f;
    for (int t = threadIdx.x; t < NP; t += 64 * NW) spos[t] = t < N ? (uint16_t)(((t / ws) << 8) | (t % ws)) : (uint16_t)0;
  }
  const uint16_t* base = qkv + (int64_t)w * N * ld + h * 3 * HD;
  const uint16_t* dobase = dout + (int64_t)w * N * ldo + h * HD;
  wa_stage(base, ld, N, NP, sQ, sQT, TP);
  wa_stage(base + HD, ld, N, NP, sK, sKT, TP);
  wa_stage(base + 2 * HD, ld, N, NP, sV, nullptr, TP);
  wa_stage(dobase, ldo, N, NP, sdO, sdOT, TP);
  for (int t = threadIdx.x; t < NP; t += 64 * NW) {
    float d = 0.f, ls = 0.f;
    if (t < N) {
      float a[HD], b[HD];
      load_row<1>(dobase + (int64_t)t * ldo, a);
      load_row<1>(out + ((int64_t)w * N + t) * ldo + h * HD, b);
#pragma unroll
      for (int c = 0; c < HD; ++c) d += a[c] * b[c];
      ls = lse[((int64_t)w * heads + h) * N + t];
    }
    sd[t] = d;
    slse[t] = ls;
  }
  __syncthreads();
  uint16_t* s0 = scr + wave * 2 * 32 * WA_P;
  uint16_t* s1 = s0 + 32 * WA_P;
  const float* bh = bias + (int64_t)h * N * N;
  const int jl = lane & 31;
  auto bias_at = [&](int a, int b) -> float {   // a, b < N
    if constexpr (TAB) {
      const int pa = spos[a], pb = spos[b];
      const int dr = (pa >> 8) - (pb >> 8), dc = (pa & 255) - (pb & 255);
      return stab[(dr < 0 ? -dr : dr) * ws + (dc < 0 ? -dc : dc)];
    } else {
      return bh[(int64_t)a * N + b];
    }
  };
  // ---- phase B: dK_J, dV_J, dS ----
  for (int J = wave; J < NT; J += NW) {
    wa_f32x16 dk = wa_zero(), dv = wa_zero();
    const int j = J * 32 + jl;
    for (int I = 0; I < NT; ++I) {
      const wa_f32x16 S = wa_nt(wa_zero(), sQ, I * 32, WA_P, 0, sK, J * 32, WA_P, 0, lane);      // rows = queries, column = this lane's key
      const wa_f32x16 dP = wa_nt(wa_zero(), sdO, I * 32, WA_P, 0, sV, J * 32, WA_P, 0, lane);
      float pv[16], dsv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = I * 32 + wa_row(r, lane);
        const bool ok = i < N && j < N;
        const float p = ok ? __expf(S[r] * scale + bias_at(i, j) - slse[i]) : 0.f;
        const float dsr = p * (dP[r] - sd[i]);
        if (ok) ds_out[(((int64_t)w * heads + h) * N + i) * N + j] = dsr;
        pv[r] = p;
        dsv[r] = dsr * scale;
      }
      // transposed tiles [key][query]: this lane's key row, queries 8 q4 + 4 g + (0..3)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int off = jl * WA_P + 8 * q4 + 4 * (lane >> 5);
        *reinterpret_cast<uint2*>(s0 + off) = make_uint2(pack_bf16x2(pv[q4 * 4], pv[q4 * 4 + 1]), pack_bf16x2(pv[q4 * 4 + 2], pv[q4 * 4 + 3]));
        *reinterpret_cast<uint2*>(s1 + off) = make_uint2(pack_bf16x2(dsv[q4 * 4], dsv[q4 * 4 + 1]), pack_bf16x2(dsv[q4 * 4 + 2], dsv[q4 * 4 + 3]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dv = wa_nt(dv, s0, 0, WA_P, 0, sdOT, 0, TP, I * 32, lane);   // dV[j][c] += sum_i P[i][j] dO[i][c]
      dk = wa_nt(dk, s1, 0, WA_P, 0, sQT, 0, TP, I * 32, lane);    // dK[j][c] += scale sum_i dS[i][j] q[i][c]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jg = J * 32 + wa_row(r, lane);
      if (jg < N) {
        uint16_t* o = dqkv + ((int64_t)w * N + jg) * ld + h * 3 * HD + (lane & 31);
        o[HD] = f32_to_bf16(dk[r]);
        o[2 * HD] = f32_to_bf16(dv[r]);
      }
    }
  }
  // ---- phase A: dQ_I ----
  for (int I = wave; I < NT; I += NW) {
    wa_f32x16 dq = wa_zero();
    for (int J = 0; J < NT; ++J) {
      const int j = J * 32 + jl;
      const wa_f32x16 S = wa_nt(wa_zero(), sQ, I * 32, WA_P, 0, sK, J * 32, WA_P, 0, lane);
      const wa_f32x16 dP = wa_nt(wa_zero(), sdO, I * 32, WA_P, 0, sV, J * 32, WA_P, 0, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int il = wa_row(r, lane), i = I * 32 + il;
        const bool ok = i < N && j < N;
        const float p = ok ? __expf(S[r] * scale + bias_at(i, j) - slse[i]) : 0.f;
        s0[il * WA_P + jl] = f32_to_bf16(p * (dP[r] - sd[i]) * scale);   // natural tile [query][key]
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      dq = wa_nt(dq, s0, 0, WA_P, 0, sKT, 0, TP, J * 32, lane);    // dQ[i][c] += scale sum_j dS[i][j] k[j][c]
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ig = I * 32 + wa_row(r, lane);
      if (ig < N) dqkv[((int64_t)w * N + ig) * ld + h * 3 * HD + (lane & 31)] = f32_to_bf16(dq[r]);
    }
  }
}
static int wa_fwd_lds(int N, int NW) { const int NP = (N + 31) & ~31; return (2 * NP * WA_P + 32 * (NP + 8) + NW * 32 * WA_P) * 2 + NW * 32 * 4 + 1024 + 2 * NP; }
static int wa_bwd_lds(int N, int NW) { const int NP = (N + 31) & ~31; return (4 * NP * WA_P + 3 * 32 * (NP + 8) + NW * 2 * 32 * WA_P) * 2 + 2 * NP * 4 + 1024 + 2 * NP; }
static int wa_waves(int N, bool bwd, bool tab) {   // 8 waves when the window has more than four tiles and the LDS still fits 160 KB
  const int tiles = (N + 31) / 32;
  if (tiles <= 2 && !esam3_dev_flag("ESAM3_WATTN_4W")) return 2;   // a 49-token window is two tiles: two of four waves would idle
  if (tiles <= 4 || !tab || esam3_dev_flag("ESAM3_WATTN_4W")) return 4;   // (the form reading the gathered table spills at 256 VGPRs)
  return (bwd ? wa_bwd_lds(N, 8) : wa_fwd_lds(N, 8)) <= 160 * 1024 ? 8 : 4;
}

// ---- window partition / reverse (tiny_vit.py:350-374) as ONE copy each way (round 6) -----------------------------------------------------------
// partition: x [B][H][W][C] -> windows [B (PH / ws) (PW / ws)][ws ws][C] with PH, PW = H, W rounded up to the window side and zeros in the padding;
// reverse: the inverse gather, dropping the padding.  torch did each in two copies (F.pad then the transposed reshape; the transposed reshape then
// the slice): 73 copy kernels, 3.1 ms of an 82 ms TinyViT-11M batch-32 step.  A thread moves 16 bytes; pure data movement.
template <bool REVERSE>
__global__ __launch_bounds__(256) void window_permute_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int B, int H, int W, int C16, int ws,
                                                             int NWY, int NWX) {
  const int64_t total = REVERSE ? (int64_t)B * H * W * C16 : (int64_t)B * NWY * NWX * ws * ws * C16;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C16);
    int64_t r = i / C16;
    if constexpr (REVERSE) {   // i indexes the image: (b, y, x)
      const int x = (int)(r % W);
      r /= W;
      const int y = (int)(r % H), b = (int)(r / H);
      const int64_t win = ((int64_t)b * NWY + y / ws) * NWX + x / ws;
      dst[i] = src[((win * ws + y % ws) * ws + x % ws) * C16 + c];
    } else {                   // i indexes the windows: (b, wy, wx, iy, ix)
      const int ix = (int)(r % ws);
      r /= ws;
      const int iy = (int)(r % ws);
      r /= ws;
      const int wx = (int)(r % NWX);
      r /= NWX;
      const int wy = (int)(r % NWY), b = (int)(r / NWY);
      const int y = wy * ws + iy, x = wx * ws + ix;
      dst[i] = (y < H && x < W) ? src[(((int64_t)b * H + y) * W + x) * C16 + c] : make_uint4(0u, 0u, 0u, 0u);
    }
  }
}

// out[h][o] = sum over the items of offset o (CSR: start[o] .. start[o + 1]) of full[h][item]: one wavefront per (head, offset), lane l adds
// the items l, l + 64, ... in list order, the 64 lane sums are added by the butterfly of wave_sum -- a fixed order (deterministic)
__global__ __launch_bounds__(256) void bias_gather_sum_kernel(const float* __restrict__ full, const int* __restrict__ start, const int* __restrict__ items,
                                                              float* __restrict__ out, int heads, int NN, int n_off) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= heads * n_off) return;
  const int h = i / n_off, o = i - h * n_off;
  float s = 0.f;
  for (int k = start[o] + lane; k < start[o + 1]; k += 64) s += full[(int64_t)h * NN + items[k]];
  s = wave_sum(s);
  if (lane == 0) out[i] = s;
}

int bad(const char* what) {
  esam3_set_error("%s: bad argument", what);
  return -1;
}

}  // namespace

extern "C" {

int esam3_ln_train_forward(int dtype, const void* x, void* y, int64_t M, int C, const float* gamma, const float* beta, float eps, float* mean,
                           float* rstd, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !y || !gamma || !beta || !mean || !rstd || M <= 0 || C <= 0 || C % 8 || C > LN_G * 64 * 8)
    return bad("esam3_ln_train_forward (fp32 / bf16; C a multiple of 8, at most 1024)");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((M + 3) / 4 < 65535 ? (M + 3) / 4 : 65535);
  if (dtype == 0) hipLaunchKernelGGL(ln_forward_kernel<0>, dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, M, C, gamma, beta, eps, mean, rstd);
  else hipLaunchKernelGGL(ln_forward_kernel<1>, dim3(grid), dim3(256), 0, s, (const uint16_t*)x, (uint16_t*)y, M, C, gamma, beta, eps, mean, rstd);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int64_t esam3_ln_train_workspace(int C) { return C > 0 ? (int64_t)sizeof(float) * LN_BLOCKS_MAX * 2 * C : 0; }

int esam3_ln_train_backward(int dtype, const void* x, const void* dy, const float* gamma, const float* mean, const float* rstd, void* dx,
                            float* dgamma, float* dbeta, int64_t M, int C, void* workspace, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !dy || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace || M <= 0 || C <= 0 || C % 8 ||
      C > LN_G * 64 * 8)
    return bad("esam3_ln_train_backward (fp32 / bf16; C a multiple of 8, at most 1024)");
  hipStream_t s = (hipStream_t)stream;
  const int blocks = ln_blocks(M);
  float* partial = (float*)workspace;
  if (dtype == 0)
    hipLaunchKernelGGL(ln_backward_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, (const float*)dy, gamma, mean, rstd, (float*)dx, M, C,
                       partial);
  else
    hipLaunchKernelGGL(ln_backward_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, (const uint16_t*)x, (const uint16_t*)dy, gamma, mean, rstd,
                       (uint16_t*)dx, M, C, partial);
  hipLaunchKernelGGL(ln_reduce_kernel, dim3((unsigned)((2 * C + 255) / 256)), dim3(256), 0, s, partial, blocks, 2 * C, dgamma, dbeta, C);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

static int win_attn_forward_impl(int dtype, const void* qkv, const float* bias, const float* tab, int ws, void* out, float* lse, int windows, int N,
                                 int heads, float scale, void* stream);
int esam3_win_attn_train_forward(int dtype, const void* qkv, const float* bias, void* out, float* lse, int windows, int N, int heads, float scale,
                                 void* stream) {
  return win_attn_forward_impl(dtype, qkv, bias, nullptr, 0, out, lse, windows, N, heads, scale, stream);
}
// the same with the attention_biases parameter itself beside the gathered table: tab [heads][ws ws], N = ws ws, bias[h][i][j] = tab[h][|dy| ws + |dx|]
// (tiny_vit.py:240-254).  The bf16 kernels index `tab` from LDS instead of reading `bias` from global memory; fp32 (validation mode) reads `bias`.
int esam3_win_attn_train_forward_tab(int dtype, const void* qkv, const float* bias, const float* tab, int ws, void* out, float* lse, int windows,
                                     int heads, float scale, void* stream) {
  if (!tab || ws <= 0 || ws > 16) return bad("esam3_win_attn_train_forward_tab (window side 1 .. 16)");
  return win_attn_forward_impl(dtype, qkv, bias, tab, ws, out, lse, windows, ws * ws, heads, scale, stream);
}
static int win_attn_forward_impl(int dtype, const void* qkv, const float* bias, const float* tab, int ws, void* out, float* lse, int windows, int N,
                                 int heads, float scale, void* stream) {
  if ((dtype != 0 && dtype != 1) || !qkv || !bias || !out || !lse || windows <= 0 || N <= 0 || N > 256 || heads <= 0 || heads > 65535)
    return bad("esam3_win_attn_train_forward (fp32 / bf16; at most 256 tokens per window; head dim 32)");
  hipStream_t s = (hipStream_t)stream;
  const int threads = (N + 63) / 64 * 64, lds = 2 * N * HD * (int)sizeof(float);
  if (dtype == 0) {
    if (esam3_allow_dyn_lds((const void*)win_attn_forward_kernel<0>, 2 * 256 * HD * (int)sizeof(float))) return -1;
    hipLaunchKernelGGL(win_attn_forward_kernel<0>, dim3((unsigned)windows, (unsigned)heads), dim3(threads), lds, s, (const float*)qkv, bias, (float*)out, lse, N,
                       heads, scale);
  } else if (!esam3_dev_flag("ESAM3_WATTN_OLD")) {   // round 6: on the matrix unit
    const int nw = wa_waves(N, false, tab != nullptr), lds_b = wa_fwd_lds(N, nw);
#define ESAM3_WA_FWD(TAB_, NW_)                                                                                                          \
  do {                                                                                                                                   \
    if (esam3_allow_dyn_lds((const void*)win_attn_forward_mfma_kernel<TAB_, NW_>, 160 * 1024)) return -1;                                \
    hipLaunchKernelGGL((win_attn_forward_mfma_kernel<TAB_, NW_>), dim3((unsigned)windows, (unsigned)heads), dim3(64 * NW_), lds_b, s,   \
                       (const uint16_t*)qkv, bias, (uint16_t*)out, lse, N, heads, scale, tab, ws);                                       \
  } while (0)
    if (tab && nw == 8) ESAM3_WA_FWD(true, 8);
    else if (tab && nw == 2) ESAM3_WA_FWD(true, 2);
    else if (tab) ESAM3_WA_FWD(true, 4);
    else if (nw == 2) ESAM3_WA_FWD(false, 2);
    else ESAM3_WA_FWD(false, 4);
#undef ESAM3_WA_FWD
  } else {
    if (esam3_allow_dyn_lds((const void*)win_attn_forward_kernel<1>, 2 * 256 * HD * (int)sizeof(float))) return -1;
    hipLaunchKernelGGL(win_attn_forward_kernel<1>, dim3((unsigned)windows, (unsigned)heads), dim3(threads), lds, s, (const uint16_t*)qkv, bias, (uint16_t*)out,
                       lse, N, heads, scale);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

static int win_attn_backward_impl(int dtype, const void* qkv, const float* bias, const float* tab, int ws, const void* out, const float* lse,
                                  const void* dout, void* dqkv, float* ds, int windows, int N, int heads, float scale, void* stream);
int esam3_win_attn_train_backward(int dtype, const void* qkv, const float* bias, const void* out, const float* lse, const void* dout, void* dqkv,
                                  float* ds, int windows, int N, int heads, float scale, void* stream) {
  return win_attn_backward_impl(dtype, qkv, bias, nullptr, 0, out, lse, dout, dqkv, ds, windows, N, heads, scale, stream);
}
int esam3_win_attn_train_backward_tab(int dtype, const void* qkv, const float* bias, const float* tab, int ws, const void* out, const float* lse,
                                      const void* dout, void* dqkv, float* ds, int windows, int heads, float scale, void* stream) {
  if (!tab || ws <= 0 || ws > 16) return bad("esam3_win_attn_train_backward_tab (window side 1 .. 16)");
  return win_attn_backward_impl(dtype, qkv, bias, tab, ws, out, lse, dout, dqkv, ds, windows, ws * ws, heads, scale, stream);
}
static int win_attn_backward_impl(int dtype, const void* qkv, const float* bias, const float* tab, int ws, const void* out, const float* lse,
                                  const void* dout, void* dqkv, float* ds, int windows, int N, int heads, float scale, void* stream) {
  if ((dtype != 0 && dtype != 1) || !qkv || !bias || !out || !lse || !dout || !dqkv || !ds || windows <= 0 || N <= 0 || N > 256 || heads <= 0 ||
      heads > 65535)
    return bad("esam3_win_attn_train_backward (fp32 / bf16; at most 256 tokens per window; head dim 32)");
  hipStream_t s = (hipStream_t)stream;
  const int threads = (N + 63) / 64 * 64, lds = (2 * N * HD + 2 * N) * (int)sizeof(float);
  const int lds_max = (2 * 256 * HD + 2 * 256) * (int)sizeof(float);
  if (dtype == 0) {
    if (esam3_allow_dyn_lds((const void*)win_attn_backward_kernel<0>, lds_max)) return -1;
    hipLaunchKernelGGL(win_attn_backward_kernel<0>, dim3((unsigned)windows, (unsigned)heads, 2), dim3(threads), lds, s, (const float*)qkv, bias, (const float*)out,
                       lse, (const float*)dout, (float*)dqkv, ds, N, heads, scale);
  } else if (!esam3_dev_flag("ESAM3_WATTN_OLD")) {   // round 6: on the matrix unit
    const int nw = wa_waves(N, true, tab != nullptr), lds_b = wa_bwd_lds(N, nw);
#define ESAM3_WA_BWD(TAB_, NW_)                                                                                                          \
  do {                                                                                                                                   \
    if (esam3_allow_dyn_lds((const void*)win_attn_backward_mfma_kernel<TAB_, NW_>, 160 * 1024)) return -1;                               \
    hipLaunchKernelGGL((win_attn_backward_mfma_kernel<TAB_, NW_>), dim3((unsigned)windows, (unsigned)heads), dim3(64 * NW_), lds_b, s,  \
                       (const uint16_t*)qkv, bias, (const uint16_t*)out, lse, (const uint16_t*)dout, (uint16_t*)dqkv, ds, N, heads,      \
                       scale, tab, ws);                                                                                                  \
  } while (0)
    if (tab && nw == 8) ESAM3_WA_BWD(true, 8);
    else if (tab && nw == 2) ESAM3_WA_BWD(true, 2);
    else if (tab) ESAM3_WA_BWD(true, 4);
    else if (nw == 2) ESAM3_WA_BWD(false, 2);
    else ESAM3_WA_BWD(false, 4);
#undef ESAM3_WA_BWD
  } else {
    if (esam3_allow_dyn_lds((const void*)win_attn_backward_kernel<1>, lds_max)) return -1;
    hipLaunchKernelGGL(win_attn_backward_kernel<1>, dim3((unsigned)windows, (unsigned)heads, 2), dim3(threads), lds, s, (const uint16_t*)qkv, bias,
                       (const uint16_t*)out, lse, (const uint16_t*)dout, (uint16_t*)dqkv, ds, N, heads, scale);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// x_dev [B][H][W][C] <-> windows_dev [B ceil(H / ws) ceil(W / ws)][ws ws][C] (dtype 0 fp32 / 1 bf16; C * element size a multiple of 16 bytes);
// reverse 0: partition (zeros in the padding), 1: the inverse
int esam3_window_partition(int dtype, const void* src, void* dst, int B, int H, int W, int C, int ws, int reverse, void* stream) {
  const int esz = dtype == 0 ? 4 : 2;
  if ((dtype != 0 && dtype != 1) || !src || !dst || B <= 0 || H <= 0 || W <= 0 || C <= 0 || ws <= 0 || (C * esz) % 16 || (((uintptr_t)src) & 15) ||
      (((uintptr_t)dst) & 15))
    return bad("esam3_window_partition (rows of a multiple of 16 bytes, 16-byte aligned tensors)");
  const int C16 = C * esz / 16, NWY = (H + ws - 1) / ws, NWX = (W + ws - 1) / ws;
  const int64_t total = reverse ? (int64_t)B * H * W * C16 : (int64_t)B * NWY * NWX * ws * ws * C16;
  const unsigned grid = (unsigned)(total / 256 + 1 < 16384 ? total / 256 + 1 : 16384);
  if (reverse)
    hipLaunchKernelGGL(window_permute_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, B, H, W, C16, ws, NWY, NWX);
  else
    hipLaunchKernelGGL(window_permute_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, B, H, W, C16, ws, NWY, NWX);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_attn_bias_gather_sum(const float* full, const int* start, const int* items, float* out, int heads, int NN, int n_off, void* stream) {
  if (!full || !start || !items || !out || heads <= 0 || NN <= 0 || n_off <= 0) return bad("esam3_attn_bias_gather_sum");
  const int n = heads * n_off;
  hipLaunchKernelGGL(bias_gather_sum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, full, start, items, out, heads, NN, n_off);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // extern "C"
