// Fused kernels of the two-way transformer's IMAGE side (bf16 engine), round 4.
//
//   i2t_block_kernel   "image attends to the tokens" (sam/transformer.py:177-182) for every image token in one pass:
//                        q   = (x + pe) Wq^T + bq                        256 -> 128      (pe Wq^T is a precomputed [P][128] table)
//                        o_h = softmax_t(q_h . k_{t,h} / 4) v_{t,h}      8 heads x 16, T <= 16 prompt tokens
//                        y   = o Wo^T + bo + x                           128 -> 256
//                        x'  = LayerNorm(y)                              norm4
//                      The layer-by-layer path ran q_proj GEMM -> attn_fewkeys -> out_proj GEMM -> layernorm: four launches that
//                      read / wrote the 5184 x 256 stream of every prompt four times (0.2 ms per layer at 32 prompts, each launch
//                      at 0.3 of the HBM roof because K = 256 / 128 leaves a 256 x 256 tile GEMM no loop to amortise its prologue).
//                      Every step is local to a pixel row, so here a wave owns 16 pixel rows and chains the four steps on the
//                      matrix cores in the TRANSPOSED form (channels = M, pixels = N of v_mfma_f32_16x16x16_bf16):
//                        Q^T = Wq X^T  ->  S^T_h = K_h Q_h^T  ->  O^T_h = V_h^T P_h^T  ->  Y^T = Wo O^T
//                      The C layout of one product (lane = pixel column, 4 consecutive rows) IS the B-operand layout of the next
//                      (lane = column, 4 consecutive k), so q, the probabilities and o never leave the registers, the softmax
//                      denominator is one more product with an all-ones A operand, and the residual x is the B fragment the first
//                      product already holds.  Wq / Wo live in LDS for the whole (persistent) workgroup; the LayerNorm statistics
//                      are 64 in-lane values + two cross-row shuffles.  HBM traffic: the stream once in, once out.
//
// Rounding points follow the layer-by-layer path (q and o are rounded to bf16 where that path stored them); the probabilities
// enter the P.V product as bf16 (as the reference's own bf16 autocast does), the pre-LayerNorm sum stays fp32.
#include "gemm_common.h"
#include "kernels.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int I2T_WQ_PITCH = 528;   // bytes per Wq row in LDS: 512 + 16 (row m, lane group g -> bank 4 m + 2 g: conflict free)
constexpr int I2T_WO_PITCH = 272;   // 256 + 16
constexpr int I2T_LDS = 128 * I2T_WQ_PITCH + 256 * I2T_WO_PITCH + (128 + 256 * 3) * 4;

struct I2tParams {
  const bf16_t* x;      // [Bp][P][256] image tokens (keys)
  bf16_t* out;          // may alias x
  const bf16_t* wq;     // [128][kpq]
  const bf16_t* wo;     // [256][kpo]
  const bf16_t* peq;    // [P][128]  pe . Wq^T
  const float *bq, *bo, *gamma, *beta;
  const bf16_t *tk, *tv;  // [Bp][T][ld*]: k / v projections of the prompt tokens
  int kpq, kpo, ldk, ldv, Bp, P, T;
  float eps;
};

__device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
  const uint2 u = make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
  return __builtin_bit_cast(s16x4, u);
}
__device__ __forceinline__ float bf16_bits_to_f32(short s) { return __builtin_bit_cast(float, ((uint32_t)(uint16_t)s) << 16); }
__device__ __forceinline__ float round_bf16(float v) { return bf16_to_f32(f32_to_bf16(v)); }

__global__ __launch_bounds__(512) void i2t_block_kernel(const I2tParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sWq = smem;
  char* sWo = smem + 128 * I2T_WQ_PITCH;
  float* sBq = reinterpret_cast<float*>(sWo + 256 * I2T_WO_PITCH);
  float* sBo = sBq + 128;
  float* sG = sBo + 256;
  float* sBt = sG + 256;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // ---- weights -> LDS, once per workgroup ------------------------------------------------------------------------------
  for (int i = tid; i < 128 * 32; i += 512) {
    const int row = i >> 5, c = i & 31;
    *reinterpret_cast<uint4*>(sWq + row * I2T_WQ_PITCH + c * 16) = *reinterpret_cast<const uint4*>(p.wq + (int64_t)row * p.kpq + c * 8);
  }
  for (int i = tid; i < 256 * 16; i += 512) {
    const int row = i >> 4, c = i & 15;
    *reinterpret_cast<uint4*>(sWo + row * I2T_WO_PITCH + c * 16) = *reinterpret_cast<const uint4*>(p.wo + (int64_t)row * p.kpo + c * 8);
  }
  if (tid < 128) sBq[tid] = p.bq[tid];
  if (tid < 256) { sBo[tid] = p.bo[tid]; sG[tid] = p.gamma[tid]; sBt[tid] = p.beta[tid]; }
  __syncthreads();

  const int m = lane & 15, g = lane >> 4;
  const int gpp = p.P >> 4;                               // 16-row groups per prompt
  const int64_t G = (int64_t)p.Bp * gpp;
  const int64_t NW = (int64_t)gridDim.x * 8, w = (int64_t)blockIdx.x * 8 + wave;
  const int64_t g_begin = w * G / NW, g_end = (w + 1) * G / NW;   // a contiguous run of groups: mostly one prompt
  const s16x4 ones = {(short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80};
  const unsigned aq = (unsigned)(m * I2T_WQ_PITCH + g * 8), ao = (unsigned)(m * I2T_WO_PITCH + g * 8);
  int cur_b = -1;
  s16x4 kA[8], vA[8];
  for (int64_t gg = g_begin; gg < g_end; ++gg) {
    const int b = (int)(gg / gpp), p0 = (int)(gg - (int64_t)b * gpp) * 16;
    if (b != cur_b) {   // wave-uniform: the token-side operands of this prompt
      cur_b = b;
      const int tr = m < p.T ? m : p.T - 1;
      const bf16_t* kr = p.tk + ((int64_t)b * p.T + tr) * p.ldk + 4 * g;
#pragma unroll
      for (int h = 0; h < 8; ++h) kA[h] = *reinterpret_cast<const s16x4*>(kr + 16 * h);   // A = K_h: row t = m, k = dims 4 g ..
#pragma unroll
      for (int h = 0; h < 8; ++h) {                                                    // A = V_h^T: row d = m, k = t = 4 g ..
        short e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int t = 4 * g + j;
          e[j] = t < p.T ? __builtin_bit_cast(short, p.tv[((int64_t)b * p.T + t) * p.ldv + 16 * h + m]) : (short)0;
        }
        vA[h] = s16x4{e[0], e[1], e[2], e[3]};
      }
    }
    const int64_t row = (int64_t)b * p.P + p0 + m;
    const bf16_t* xr = p.x + row * 256 + 4 * g;
    s16x4 xf[16];   // B = X^T: column = pixel m, k = channels 16 kb + 4 g ..  (also the residual, in the C layout of Y^T)
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) xf[kb] = *reinterpret_cast<const s16x4*>(xr + 16 * kb);
    const bf16_t* per = p.peq + (int64_t)(p0 + m) * 128 + 4 * g;
    s16x4 pef[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) pef[h] = *reinterpret_cast<const s16x4*>(per + 16 * h);

    s16x4 ob[8];   // o (bf16) per head: the B operand of the output projection
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      // ---- Q_h^T = Wq[16 h .. 16 h + 16) X^T  (+ bias + pe) -------------------------------------------------------------
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const s16x4 a = *reinterpret_cast<const s16x4*>(sWq + aq + h * 16 * I2T_WQ_PITCH + kb * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, xf[kb], acc, 0, 0, 0);
      }
      float q[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)   // rounded to bf16 where the layer-by-layer path stored q, then the 1 / sqrt(16) scale (exact)
        q[i] = round_bf16(acc[i] + sBq[16 * h + 4 * g + i] + bf16_bits_to_f32(pef[h][i])) * 0.25f;
      const s16x4 qb = pack4(q[0], q[1], q[2], q[3]);
      // ---- S^T = K_h Q_h^T: rows t = 4 g + i, column = pixel; softmax over t ---------------------------------------------
      f32x4 s = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kA[h], qb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      float mx = -3.0e38f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (4 * g + i >= p.T) s[i] = -3.0e38f;
        mx = fmaxf(mx, s[i]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = 4 * g + i < p.T ? __expf(s[i] - mx) : 0.f;
      const s16x4 pb = pack4(e[0], e[1], e[2], e[3]);
      // ---- O^T = V_h^T P^T and the denominator ones . P^T (every row of it is sum_t p) ------------------------------------
      const f32x4 o = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vA[h], pb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const f32x4 den = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ones, pb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const float inv = 1.f / den[0];
      ob[h] = pack4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
      __builtin_amdgcn_sched_barrier(0);   // one head at a time: hoisting the next heads' LDS reads spills
    }
    // ---- Y^T = Wo O^T + bo + x, LayerNorm over the 256 channels of a pixel (= 64 values in the lane x 4 lane groups) --------
    float y[16][4];
    float sum = 0.f;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        const s16x4 a = *reinterpret_cast<const s16x4*>(sWo + ao + mb * 16 * I2T_WO_PITCH + h * 32);
        acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, ob[h], acc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        y[mb][i] = acc[i] + sBo[16 * mb + 4 * g + i] + bf16_bits_to_f32(xf[mb][i]);
        sum += y[mb][i];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    const float mean = sum * (1.f / 256.f);
    float sq = 0.f;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float d = y[mb][i] - mean;
        sq = fmaf(d, d, sq);
      }
    sq += __shfl_xor(sq, 16);
    sq += __shfl_xor(sq, 32);
    const float rstd = rsqrtf(sq * (1.f / 256.f) + p.eps);
    bf16_t* orow = p.out + row * 256 + 4 * g;
#pragma unroll
    for (int mb = 0; mb < 16; ++mb) {
      float r[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = (y[mb][i] - mean) * rstd * sG[16 * mb + 4 * g + i] + sBt[16 * mb + 4 * g + i];
      *reinterpret_cast<s16x4*>(orow + 16 * mb) = pack4(r[0], r[1], r[2], r[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
//   t2i_mfma_kernel    "tokens attend to the image" (sam/transformer.py:165-170): <= 16 query tokens, 8 heads x 16, thousands of
//                      keys.  A workgroup owns one key chunk of one prompt, wave h owns head h.  K / V tiles of 64 keys go
//                      HBM -> LDS as whole rows (the only real traffic); per tile and head
//                        S^T[key, t] = K_h Q_h^T           4 products (A = K rows out of LDS, B = the scaled queries, registers)
//                        online softmax over the keys      in-lane over 16 values + two cross-row shuffles per 64 keys
//                        O^T[d, t]  += V_h^T P             4 products (A = ds_read_b64_tr_b16 of the V tile, B = P = the C layout
//                                                          of S^T as it stands: lane = token column in both)
//                      so the running maximum, the rescale factor and the accumulator of a token live in the same lane.  The
//                      per-chunk (max, sum, acc[16]) partials are merged by attn_t2i_merge_kernel in a fixed order
//                      (bit-reproducible), as for the VALU kernel this replaces (0.087 ms per launch at 32 prompts, 43 us of it
//                      pure FMA issue: 5184 keys x 128 (token, head) pairs x 32 multiply-adds per thread).
constexpr int T2I_KP = 272, T2I_VP = 288, T2I_TK = 64;   // LDS row pitches (bytes): K rows 4 m + 2 g, V rows 8 r + 2 c banks

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
__device__ __forceinline__ s16x4 lds_tr16(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p)); }

struct T2iParams {
  const bf16_t *q, *k, *v;
  float* part;
  int ldq, ldk, ldv, Nq, Nk, chunk_keys, nchunks;
};

__global__ __launch_bounds__(512) void t2i_mfma_kernel(const T2iParams p) {
  __shared__ __attribute__((aligned(16))) char sK[2][T2I_TK * T2I_KP];
  __shared__ __attribute__((aligned(16))) char sV[2][T2I_TK * T2I_VP];
  const int tid = threadIdx.x, lane = tid & 63, h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int chunk = blockIdx.x;
  const int64_t b = blockIdx.y;
  const int k0 = chunk * p.chunk_keys, k1 = min(p.Nk, k0 + p.chunk_keys);
  const bf16_t* kb = p.k + b * (int64_t)p.Nk * p.ldk;
  const bf16_t* vb = p.v + b * (int64_t)p.Nk * p.ldv;
  // B = Q_h^T scaled by 1 / sqrt(16) (exact in bf16): column t = m, k = dims 4 g ..
  s16x4 qB = {0, 0, 0, 0};
  if (m < p.Nq) {
    const s16x4 raw = *reinterpret_cast<const s16x4*>(p.q + (b * p.Nq + m) * (int64_t)p.ldq + 16 * h + 4 * g);
    qB = pack4(bf16_bits_to_f32(raw[0]) * 0.25f, bf16_bits_to_f32(raw[1]) * 0.25f, bf16_bits_to_f32(raw[2]) * 0.25f,
               bf16_bits_to_f32(raw[3]) * 0.25f);
  }
  // tile staging: 64 rows x 256 bytes per operand = 1024 16-byte pieces, two per thread and operand
  uint4 rk[2], rv[2];
  auto fetch = [&](int j0) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * 512, r = i >> 4, sl = i & 15;
      rk[u] = make_uint4(0u, 0u, 0u, 0u);
      rv[u] = rk[u];
      if (j0 + r < k1) {
        rk[u] = *reinterpret_cast<const uint4*>(kb + (int64_t)(j0 + r) * p.ldk + sl * 8);
        rv[u] = *reinterpret_cast<const uint4*>(vb + (int64_t)(j0 + r) * p.ldv + sl * 8);
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * 512, r = i >> 4, sl = i & 15;
      *reinterpret_cast<uint4*>(&sK[buf][r * T2I_KP + sl * 16]) = rk[u];
      *reinterpret_cast<uint4*>(&sV[buf][r * T2I_VP + sl * 16]) = rv[u];
    }
  };
  float mx = -3.0e38f, l = 0.f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};   // O^T: rows d = 4 g + i, column t = m
  if (k0 < k1) {
    fetch(k0);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int j0 = k0; j0 < k1; j0 += T2I_TK, buf ^= 1) {
    const bool more = j0 + T2I_TK < k1;
    if (more) fetch(j0 + T2I_TK);
    f32x4 s[4];
    float tmax = -3.0e38f;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      const s16x4 a = *reinterpret_cast<const s16x4*>(&sK[buf][(kg * 16 + m) * T2I_KP + 32 * h + 8 * g]);   // row key, k = dims
      s[kg] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, qB, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (j0 + kg * 16 + 4 * g + i >= k1) s[kg][i] = -3.0e38f;
        tmax = fmaxf(tmax, s[kg][i]);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float mn = fmaxf(mx, tmax);
    const float alpha = __expf(mx - mn);
    mx = mn;
    l *= alpha;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] *= alpha;
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
      float e[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) e[i] = __expf(s[kg][i] - mn);   // a masked key: exp(-3e38 - mn) = 0
      const s16x4 pb = pack4(e[0], e[1], e[2], e[3]);
#pragma unroll
      for (int i = 0; i < 4; ++i) l += bf16_bits_to_f32(pb[i]);   // the denominator sums what the numerator multiplies
      // A = V_h^T: row d = m, k = keys kg * 16 + 4 g ..: the transposing read of the row-major tile
      const s16x4 va = lds_tr16(&sV[buf][(kg * 16 + 4 * g + (m >> 2)) * T2I_VP + 32 * h + (m & 3) * 8]);
      acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(va, pb, acc, 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
  l += __shfl_xor(l, 16);
  l += __shfl_xor(l, 32);
  if (m < p.Nq) {   // partial = m, l, acc[16]   (attn_t2i_merge_kernel's layout)
    float* pp = p.part + ((((b * p.nchunks + chunk)) * 16 + m) * 8 + h) * 18;
    if (g == 0) { pp[0] = mx; pp[1] = l; }
#pragma unroll
    for (int i = 0; i < 4; ++i) pp[2 + 4 * g + i] = acc[i];
  }
}

}  // namespace

bool esam3_i2t_fused_ok(int dtype, int P, int T, int heads, int hd, int C) {
  return dtype == 1 && heads == 8 && hd == 16 && C == 256 && P % 16 == 0 && T >= 1 && T <= 16;
}

int esam3_launch_i2t_fused(const void* x, void* out, const void* wq, int kpq, const float* bq, const void* peq, const void* wo, int kpo,
                           const float* bo, const float* gamma, const float* beta, float eps, const void* tk, int ldk, const void* tv,
                           int ldv, int Bp, int P, int T, hipStream_t s) {
  if (!esam3_i2t_fused_ok(1, P, T, 8, 16, 256) || kpq < 256 || kpo < 128 || (kpq % 8) || (kpo % 8) || (ldk % 4) || Bp < 1) {
    esam3_set_error("i2t_fused: unsupported shape (P=%d T=%d kpq=%d kpo=%d)", P, T, kpq, kpo);
    return -1;
  }
  static const int ok = esam3_allow_dyn_lds((const void*)i2t_block_kernel, I2T_LDS);
  if (ok) return -1;
  I2tParams p;
  p.x = (const bf16_t*)x; p.out = (bf16_t*)out; p.wq = (const bf16_t*)wq; p.wo = (const bf16_t*)wo; p.peq = (const bf16_t*)peq;
  p.bq = bq; p.bo = bo; p.gamma = gamma; p.beta = beta; p.tk = (const bf16_t*)tk; p.tv = (const bf16_t*)tv;
  p.kpq = kpq; p.kpo = kpo; p.ldk = ldk; p.ldv = ldv; p.Bp = Bp; p.P = P; p.T = T; p.eps = eps;
  const int64_t G = (int64_t)Bp * (P / 16);
  int grid = (int)((G + 7) / 8);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(i2t_block_kernel, dim3(grid), dim3(512), I2T_LDS, s, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---- token -> image attention on the matrix cores --------------------------------------------------------------------------
static int t2i_mfma_chunks(int Bp, int Nk, int* chunk_keys) {
  int nch = (768 + Bp - 1) / Bp;                 // about three workgroups per CU over the whole launch
  if (nch < 1) nch = 1;
  int ck = (Nk + nch - 1) / nch;
  ck = (ck + T2I_TK - 1) / T2I_TK * T2I_TK;
  *chunk_keys = ck;
  return (Nk + ck - 1) / ck;
}
bool esam3_t2i_mfma_ok(int dtype, int Nq, int Nk, int heads, int hd) {
  return dtype == 1 && heads == 8 && hd == 16 && Nq >= 1 && Nq <= 16 && Nk >= 1;
}
int64_t esam3_t2i_mfma_scratch_floats(int Bp, int Nq, int Nk) {
  int ck;
  return (int64_t)Bp * t2i_mfma_chunks(Bp, Nk, &ck) * 16 * 8 * 18;
}
int esam3_launch_t2i_mfma(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, void* o, int Bp, int Nq, int Nk,
                          float* scratch, hipStream_t s) {
  if (!esam3_t2i_mfma_ok(1, Nq, Nk, 8, 16) || (ldq % 4) || (ldk % 8) || (ldv % 8) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || !scratch) {
    esam3_set_error("t2i_mfma: unsupported shape (Nq=%d Nk=%d ldq=%d ldk=%d ldv=%d)", Nq, Nk, ldq, ldk, ldv);
    return -1;
  }
  T2iParams p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.part = scratch;
  p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.Nq = Nq; p.Nk = Nk;
  p.nchunks = t2i_mfma_chunks(Bp, Nk, &p.chunk_keys);
  hipLaunchKernelGGL(t2i_mfma_kernel, dim3((unsigned)p.nchunks, (unsigned)Bp), dim3(512), 0, s, p);
  HIP_CHECK_RET(hipGetLastError());
  return esam3_launch_attn_t2i_merge(1, scratch, o, Bp, Nq, p.nchunks, s);
}
