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
  s16x4 kA[8], vA[8], xf[16], xn[16];
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
    // B = X^T: column = pixel m, k = channels 16 kb + 4 g ..  (also the residual, in the C layout of Y^T); the first group's rows are
    // loaded here, every later group's were requested during the previous group's output projection
    if (gg == g_begin) {
      const bf16_t* xr = p.x + row * 256 + 4 * g;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) xf[kb] = *reinterpret_cast<const s16x4*>(xr + 16 * kb);
    }
    const bf16_t* per = p.peq + (int64_t)(p0 + m) * 128 + 4 * g;
    s16x4 pef[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) pef[h] = *reinterpret_cast<const s16x4*>(per + 16 * h);

    s16x4 ob[8];   // o (bf16) per head: the B operand of the output projection
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      // ---- Q_h^T = Wq[16 h .. 16 h + 16) X^T  (+ bias + pe) -------------------------------------------------------------
      // four partial sums: sixteen products chained on ONE accumulator run at the matrix pipe's latency, not its rate
      f32x4 part[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
        const s16x4 a = *reinterpret_cast<const s16x4*>(sWq + aq + h * 16 * I2T_WQ_PITCH + kb * 32);
        part[kb & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, xf[kb], part[kb & 3], 0, 0, 0);
      }
      const f32x4 acc = (part[0] + part[1]) + (part[2] + part[3]);
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
      if (h & 1) __builtin_amdgcn_sched_barrier(0);   // two heads at a time (their chains interleave); hoisting more LDS reads spills
    }
    // ---- Y^T = Wo O^T + bo + x, LayerNorm over the 256 channels of a pixel (= 64 values in the lane x 4 lane groups) --------
    float y[16][4];
    float sum = 0.f;
    if (gg + 1 < g_end) {   // the next group's rows (contiguous: the groups of a wave are consecutive 16-row blocks of the stream)
      const bf16_t* xr = p.x + (row + 16) * 256 + 4 * g;
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) xn[kb] = *reinterpret_cast<const s16x4*>(xr + 16 * kb);
    }
#pragma unroll
    for (int mq = 0; mq < 4; ++mq) {   // four output blocks at a time: four independent accumulation chains
      f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int h = 0; h < 8; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const s16x4 a = *reinterpret_cast<const s16x4*>(sWo + ao + (4 * mq + j) * 16 * I2T_WO_PITCH + h * 32);
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, ob[h], acc[j], 0, 0, 0);
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int mb = 4 * mq + j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          y[mb][i] = acc[j][i] + sBo[16 * mb + 4 * g + i] + bf16_bits_to_f32(xf[mb][i]);
          sum += y[mb][i];
        }
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
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) xf[kb] = xn[kb];
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

// ---------------------------------------------------------------------------------------------------------------------------
//   rowlin256_kernel   out[r] = x[r] W^T + bias + table[r mod P]  for 256 -> 256 channels over hundreds of thousands of rows: the merged
//                      [k | v] projection of the image tokens.  As a 256 x 256 x 64 tile GEMM this shape has a K loop of four steps:
//                      every tile pays the pipeline fill and a 128 KB epilogue for 33 MFLOP, and the launch ran at 0.3 of the HBM
//                      roof (0.07 ms for 170 MB).  Here the 128 KB of weights stay in LDS for the persistent workgroup, a wave owns
//                      32 rows: their fragments come straight from global memory in the B layout of v_mfma_f32_32x32x16_bf16
//                      (Y^T = W X^T, channels = M), the eight 32-channel blocks accumulate side by side (eight independent chains),
//                      and the C layout leaves 4 consecutive channels of a row per lane for the bias / table add and the store.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bf16x8_v ld_a16(const bf16_t* p) { return __builtin_bit_cast(bf16x8_v, *reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ bf16x8_v ld_b16(const char* p) { return __builtin_bit_cast(bf16x8_v, *reinterpret_cast<const uint4*>(p)); }

constexpr int RL_PITCH = 528;   // bytes per weight row in LDS (512 + 16: the 16-lane groups of a ds_read_b128 hit 64 distinct banks)
constexpr int RL_LDS = 256 * RL_PITCH + 256 * 4;

struct RowLinParams {
  const bf16_t *x, *w, *table;
  const float* bias;
  bf16_t* out;
  int64_t rows;
  int kp, P, abl;
};

__global__ __launch_bounds__(512) void rowlin256_kernel(const RowLinParams p) {
  extern __shared__ __attribute__((aligned(16))) char rl_smem[];
  char* sW = rl_smem;
  float* sB = reinterpret_cast<float*>(rl_smem + 256 * RL_PITCH);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 256 * 32; i += 512) {
    const int row = i >> 5, c = i & 31;
    *reinterpret_cast<uint4*>(sW + row * RL_PITCH + c * 16) = *reinterpret_cast<const uint4*>(p.w + (int64_t)row * p.kp + c * 8);
  }
  if (tid < 256) sB[tid] = p.bias ? p.bias[tid] : 0.f;
  __syncthreads();
  const int m = lane & 31, hk = lane >> 5;
  const int64_t G = p.rows >> 5, NW = (int64_t)gridDim.x * 8, w = (int64_t)blockIdx.x * 8 + wave;
  const int64_t g_begin = w * G / NW, g_end = (w + 1) * G / NW;
  const unsigned aw = (unsigned)(m * RL_PITCH + hk * 16);
  for (int64_t gg = g_begin; gg < g_end; ++gg) {
    const int64_t row = gg * 32 + m;
    const bf16_t* xr = p.x + row * 256 + 8 * hk;
    bf16x8_v xf[16];   // B = X^T: column = row m, k = 16 kb + 8 hk ..
#ifdef ESAM3_DEV
    if (p.abl & 2) {   // ablation: no row loads
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) xf[kb] = __builtin_bit_cast(bf16x8_v, make_uint4((unsigned)(row + kb), 0x3f803f80u, (unsigned)lane, 0x3f003f00u));
    } else
#endif
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) xf[kb] = ld_a16(xr + 16 * kb);
    f32x16 acc[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[mb][i] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
      for (int mb = 0; mb < 8; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_b16(sW + aw + mb * 32 * RL_PITCH + kb * 32), xf[kb], acc[mb], 0, 0, 0);
      if ((kb & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the weight reads four k-steps deep at most
    }
    // C layout: value 4 j + i of block mb = channel 32 mb + 8 j + 4 hk + i of row m
    const int prow = p.table ? (int)(row % p.P) : 0;
    const bf16_t* tr = p.table ? p.table + (int64_t)prow * 256 + 4 * hk : nullptr;
    bf16_t* orow = p.out + row * 256 + 4 * hk;
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c0 = 32 * mb + 8 * j;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = acc[mb][4 * j + i] + sB[c0 + 4 * hk + i];
        if (tr) {
          const s16x4 t4 = *reinterpret_cast<const s16x4*>(tr + c0);
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] += bf16_bits_to_f32(t4[i]);
        }
#ifdef ESAM3_DEV
        if ((p.abl & 1) && v[0] + v[1] + v[2] + v[3] != 12345.678f) continue;   // ablation: no stores
#endif
        *reinterpret_cast<s16x4*>(orow + c0) = pack4(v[0], v[1], v[2], v[3]);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// TOKEN side of the two-way transformer and the output heads (sam/transformer.py:143-182, mask_decoder.py:224-242) in five
// launches instead of fifty-seven.  The token stream of a prompt is T <= 16 rows x 256 channels: every Linear on it is
// launch-bound as a GEMM of its own (7-9 us per launch, 0.5 ms per step at 32 prompts).  Here one workgroup owns one prompt
// and keeps its fp32 stream, the positional tokens and every intermediate in LDS; a Linear is the transposed product
// Y^T[N x 16] = W[N x K] X^T[K x 16] on v_mfma_f32_16x16x32_bf16 with the packed weight rows read straight from L2 as the A
// operand (16 bytes per lane) and X^T out of LDS as the B operand; a wave takes 16-channel output blocks round-robin and its
// C layout (4 consecutive channels of one token per lane) goes back to LDS as one 8-byte store per block.  The stages are
// cut where the image side has to run:
//   tok_a   self-attention block + norm1, then q of the token -> image attention            (per layer)
//   tok_b   token -> image out_proj + norm2, MLP + norm3, k / v of the image -> token attention (+ q of the final attention)
//   tok_d   final out_proj + norm, hypernetwork MLPs, IoU head, object-score head
// Rounding points are those of the layer-by-layer path: Linear outputs that were bf16 tensors there are rounded to bf16 here,
// the stream and its LayerNorms stay fp32, `queries` (the bf16 copy that feeds the GEMMs) is bf16(stream).
constexpr int TOK_XP = 528;           // bytes per row of a [16][256] bf16 operand in LDS (256 * 2 + 16)
constexpr int TOK_HP = 2048 * 2 + 16; // MLP hidden
constexpr int TOK_AP = 128 * 2 + 16;  // [16][128] operand (attention output of the image-side kernels)

struct TokLin { const bf16_t* w; const float* b; int ldw; };


// One 16-channel output block of a Linear: where its weight rows, its bias and its X operand are, how many of its channels exist
struct TokBlk {
  const bf16_t* w;     // packed weight row of channel 0 of the block
  const float* bias;   // bias of channel 0 of the block, or null
  const char* sx;      // X operand (row-major bf16 in LDS)
  int ldw, nvalid;     // weight row pitch (elements); channels of the block that exist (bias reads are guarded by it)
};
// Y^T block by block over a LIST of blocks (several Linears that share K can run as one list): blk(i) describes block i,
// epi(i, acc) receives its channels 4 g + j (j = 0..3) of token m = lane & 15, bias included.  A wave takes blocks wave, wave + 8,
// ...; the weight rows are the A operand straight from L2, 16 bytes per lane and k-step, fetched ONE CHUNK (8 k-steps) AHEAD of
// the products that consume them -- the first version waited for every chunk it had just requested and ran at the latency of
// one L2 round trip per 16 x 256 block (0.093 ms for the MLP stage at 32 prompts instead of the 0.015 ms its 2.2 MB of weights
// take through one CU's 64-byte / clock L1).
template <int K, typename Blk, typename Epi>
__device__ __forceinline__ void tok_linear_list(int nblocks, Blk blk, int xp, int wave, int lane, Epi epi) {
  constexpr int KB = K / 32, CH = KB < 8 ? KB : 8, NCH = KB / CH;
  static_assert(KB % CH == 0, "K must be a multiple of 256 or at most 256");
  const int m = lane & 15, g = lane >> 4;
  const int nmine = nblocks > wave ? (nblocks - wave + 7) / 8 : 0;
  const int total = nmine * NCH;
  if (total == 0) return;
  bf16x8_v bufA[CH], bufB[CH];
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](bf16x8_v (&buf)[CH], int it) {
    const TokBlk d = blk(wave + 8 * (it / NCH));
    const bf16_t* wr = d.w + (int64_t)m * d.ldw + 8 * g + 32 * CH * (it % NCH);
#pragma unroll
    for (int j = 0; j < CH; ++j) buf[j] = ld_a16(wr + 32 * j);
  };
  auto compute = [&](const bf16x8_v (&buf)[CH], int it) {
    const int bi = wave + 8 * (it / NCH), c = it % NCH;
    const TokBlk d = blk(bi);
    if (c == 0) acc = f32x4{0.f, 0.f, 0.f, 0.f};
    const char* xr = d.sx + m * xp + (32 * CH * c + 8 * g) * 2;
#pragma unroll
    for (int j = 0; j < CH; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(buf[j], ld_b16(xr + 64 * j), acc, 0, 0, 0);
    if (c == NCH - 1) {
      f32x4 r = acc;
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] += (d.bias && 4 * g + i < d.nvalid) ? d.bias[4 * g + i] : 0.f;
      epi(bi, r);
    }
  };
  fetch(bufA, 0);
#pragma unroll 1
  for (int it = 0; it < total; it += 2) {
    if (it + 1 < total) fetch(bufB, it + 1);
    compute(bufA, it);
    if (it + 2 < total) fetch(bufA, it + 2);
    if (it + 1 < total) compute(bufB, it + 1);
  }
}
// a single Linear as a list: block nb = channels [16 nb, 16 nb + 16)
template <int K, typename Epi>
__device__ __forceinline__ void tok_linear(const TokLin L, int N, const char* sX, int xp, int wave, int lane, Epi epi) {
  tok_linear_list<K>((N + 15) / 16, [&](int nb) {
    return TokBlk{L.w + (int64_t)16 * nb * L.ldw, L.b ? L.b + 16 * nb : nullptr, sX, L.ldw, N - 16 * nb};
  }, xp, wave, lane, epi);
}
// store a C-layout block as bf16 into a row-major LDS operand
__device__ __forceinline__ void st_block_bf16(char* sY, int yp, int nb, int lane, const f32x4& v, bool relu = false) {
  const int m = lane & 15, g = lane >> 4;
  f32x4 r = v;
  if (relu) {
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = fmaxf(r[i], 0.f);
  }
  *reinterpret_cast<s16x4*>(sY + m * yp + (16 * nb + 4 * g) * 2) = pack4(r[0], r[1], r[2], r[3]);
}
// LayerNorm of the 16 x 256 fp32 stream in LDS, in place (rows >= T hold zeros and stay finite); 512 threads, 2 rows per wave
__device__ __forceinline__ void tok_layernorm(float* sS, const float* gamma, const float* beta, float eps, int wave, int lane) {
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    float* row = sS + (2 * wave + rr) * 256;
    const float4 v = *reinterpret_cast<const float4*>(row + 4 * lane);
    float s = v.x + v.y + v.z + v.w;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) s += __shfl_xor(s, off);
    const float mean = s * (1.f / 256.f);
    const float d0 = v.x - mean, d1 = v.y - mean, d2 = v.z - mean, d3 = v.w - mean;
    float q = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) q += __shfl_xor(q, off);
    const float rstd = rsqrtf(q * (1.f / 256.f) + eps);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * lane), bt = *reinterpret_cast<const float4*>(beta + 4 * lane);
    *reinterpret_cast<float4*>(row + 4 * lane) = make_float4(d0 * rstd * gm.x + bt.x, d1 * rstd * gm.y + bt.y, d2 * rstd * gm.z + bt.z,
                                                             d3 * rstd * gm.w + bt.w);
  }
}
// bf16 operand rows from the stream: X = bf16(S (+ T)), rows >= T zero; thread = (row, 8 channels)
__device__ __forceinline__ void tok_make_x(char* sX, const float* sS, const float* sT, int T, int tid) {
  const int r = tid >> 5, c = (tid & 31) * 8;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = r < T ? sS[r * 256 + c + e] + (sT ? sT[r * 256 + c + e] : 0.f) : 0.f;
  *reinterpret_cast<uint4*>(sX + r * TOK_XP + c * 2) =
      make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
}
__device__ __forceinline__ void tok_load_stream(float* sS, const float* g, int T, int tid) {   // [T][256] fp32 -> LDS, rows >= T zero
  for (int i = tid; i < 16 * 64; i += 512) {
    const int r = i >> 6;
    *reinterpret_cast<float4*>(sS + 4 * i) = r < T ? *reinterpret_cast<const float4*>(g + 4 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__device__ __forceinline__ void tok_store_stream(float* g, const float* sS, int T, int tid) {
  for (int i = tid; i < T * 64; i += 512) *reinterpret_cast<float4*>(g + 4 * i) = *reinterpret_cast<const float4*>(sS + 4 * i);
}
// [T][128] bf16 rows (global, contiguous) -> a row-major LDS operand, rows >= T zero
__device__ __forceinline__ void tok_load_a128(char* sA, const bf16_t* g, int T, int tid) {
  if (tid < 256) {
    const int r = tid >> 4, sl = tid & 15;
    *reinterpret_cast<uint4*>(sA + r * TOK_AP + sl * 16) =
        r < T ? *reinterpret_cast<const uint4*>(g + r * 128 + sl * 8) : make_uint4(0u, 0u, 0u, 0u);
  }
}
// store a C-layout block of an N = 128 projection to global [T][128] bf16
__device__ __forceinline__ void st_block_g128(bf16_t* gy, int T, int nb, int lane, const f32x4& v) {
  const int m = lane & 15, g = lane >> 4;
  if (m < T) *reinterpret_cast<s16x4*>(gy + m * 128 + 16 * nb + 4 * g) = pack4(v[0], v[1], v[2], v[3]);
}

struct TokAParams {
  float* q32;          // [Bp][T][256] fp32 stream, in / out
  const float* t32;    // [Bp][T][256] positional tokens
  bf16_t* tq;          // out: q of the token -> image attention [Bp][T][128]
  TokLin sq, sk, sv, so, xq;
  const float *g1, *b1;
  int T, first;        // first layer: q / k take no positional tokens and the attention output REPLACES the stream
  float eps;
};

__global__ __launch_bounds__(512) void tok_a_kernel(const TokAParams p) {
  __shared__ __attribute__((aligned(16))) float sS[16 * 256], sT[16 * 256];
  __shared__ __attribute__((aligned(16))) char sXa[16 * TOK_XP], sXb[16 * TOK_XP], sQ[16 * TOK_XP], sK[16 * TOK_XP], sV[16 * TOK_XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int T = p.T;
  tok_load_stream(sS, p.q32 + b * T * 256, T, tid);
  tok_load_stream(sT, p.t32 + b * T * 256, T, tid);
  __syncthreads();
  tok_make_x(sXa, sS, p.first ? nullptr : sT, T, tid);
  tok_make_x(sXb, sS, nullptr, T, tid);
  __syncthreads();
  tok_linear_list<256>(48, [&](int i) {   // q | k | v: 3 x 16 blocks in one list
    const int which = i >> 4, nb = i & 15;
    const TokLin& L = which == 0 ? p.sq : (which == 1 ? p.sk : p.sv);
    return TokBlk{L.w + (int64_t)16 * nb * L.ldw, L.b ? L.b + 16 * nb : nullptr, which == 2 ? sXb : sXa, L.ldw, 16};
  }, TOK_XP, wave, lane, [&](int i, const f32x4& a) {
    const int which = i >> 4;
    st_block_bf16(which == 0 ? sQ : (which == 1 ? sK : sV), TOK_XP, i & 15, lane, a);
  });
  __syncthreads();
  // self-attention, 8 heads x 32: thread = (token t, head h, dim quarter dq); fp32 from the bf16 rows, as attn_kernel
  {
    const int t = tid >> 5, h = (tid >> 2) & 7, dq = tid & 3;
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (t < T) {
      const float scale = rsqrtf(32.f);
      float sc[16], mx = -3.0e38f;
      const bf16_t* qr = reinterpret_cast<const bf16_t*>(sQ + t * TOK_XP) + 32 * h;
      for (int j = 0; j < T; ++j) {
        const bf16_t* kr = reinterpret_cast<const bf16_t*>(sK + j * TOK_XP) + 32 * h;
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) a = fmaf(bf16_to_f32(qr[d]) * scale, bf16_to_f32(kr[d]), a);
        sc[j] = a;
        mx = fmaxf(mx, a);
      }
      float l = 0.f;
      for (int j = 0; j < T; ++j) {
        const float e = __expf(sc[j] - mx);
        l += e;
        const bf16_t* vr = reinterpret_cast<const bf16_t*>(sV + j * TOK_XP) + 32 * h + 8 * dq;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = fmaf(e, bf16_to_f32(vr[d]), o[d]);
      }
      const float inv = 1.f / l;
#pragma unroll
      for (int d = 0; d < 8; ++d) o[d] *= inv;
    }
    // sXa's readers (the q / k projections) finished before the barrier above: the attention output reuses it
    *reinterpret_cast<uint4*>(sXa + t * TOK_XP + (32 * h + 8 * dq) * 2) =
        make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
  }
  __syncthreads();
  tok_linear<256>(p.so, 256, sXa, TOK_XP, wave, lane, [&](int nb, const f32x4& a) {
    const int m = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* s = sS + m * 256 + 16 * nb + 4 * g + i;
      *s = m < T ? (p.first ? 0.f : *s) + a[i] : 0.f;
    }
  });
  __syncthreads();
  tok_layernorm(sS, p.g1, p.b1, p.eps, wave, lane);
  __syncthreads();
  tok_make_x(sXa, sS, sT, T, tid);
  tok_store_stream(p.q32 + b * T * 256, sS, T, tid);
  __syncthreads();
  tok_linear<256>(p.xq, 128, sXa, TOK_XP, wave, lane, [&](int nb, const f32x4& a) { st_block_g128(p.tq + b * T * 128, T, nb, lane, a); });
}

constexpr int TOK_B_SPLIT = 8;   // workgroups per prompt in tok_b: 256 hidden channels of the MLP each

struct TokBParams {
  float* q32;
  const float* t32;
  const bf16_t* ta;    // [Bp][T][128] output of the token -> image attention
  bf16_t *tk, *tv;     // out: k / v of the image -> token attention [Bp][T][128]
  bf16_t* tq;          // out (last layer): q of the final token -> image attention, or null
  TokLin xo, l1, l2, ik, iv, fq;
  const float *g2, *b2, *g3, *b3;
  float* part;         // scratch [Bp][TOK_B_SPLIT][16][256] fp32: the slices' shares of mlp.lin2's output
  float* mid;          // scratch [Bp][16][256] fp32: the stream after norm2 (written by slice 0 of tok_b1, read by tok_b2)
  int T;
  float eps;
};

// tok_b1, grid (prompt, slice).  The MLP's 2 MB of weights through ONE compute unit's L1 (64 bytes / clock) is 15 us at best and
// was 80 us in practice, so a prompt is spread over TOK_B_SPLIT workgroups: each recomputes the cheap prefix (out_proj + norm2,
// 64 KB of weights), takes 256 of the 2048 hidden channels through lin1 AND through the matching K slice of lin2, and leaves its
// share of lin2's output in `part`.  tok_b2, grid (prompt): adds the shares in slice order (bit-reproducible: no floating-point
// atomics), applies norm3 and projects k / v for the image -> token attention.  (A single launch in which the last slice to
// arrive finishes the prompt was slower than the one-workgroup form: the device-scope release fence in front of the arrival
// counter writes back the XCD's whole L2 -- 256 times per launch.)
__global__ __launch_bounds__(512) void tok_b1_kernel(const TokBParams p) {
  __shared__ __attribute__((aligned(16))) float sS[16 * 256];
  __shared__ __attribute__((aligned(16))) char sXb[16 * TOK_XP], sH[16 * TOK_XP], sA[16 * TOK_AP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int64_t b = blockIdx.x;
  const int sl = blockIdx.y;
  const int T = p.T;
  tok_load_stream(sS, p.q32 + b * T * 256, T, tid);
  tok_load_a128(sA, p.ta + b * T * 128, T, tid);
  __syncthreads();
  tok_linear<128>(p.xo, 256, sA, TOK_AP, wave, lane, [&](int nb, const f32x4& a) {
    if (m < T) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sS[m * 256 + 16 * nb + 4 * g + i] += a[i];
    }
  });
  __syncthreads();
  tok_layernorm(sS, p.g2, p.b2, p.eps, wave, lane);
  __syncthreads();
  tok_make_x(sXb, sS, nullptr, T, tid);
  if (sl == 0) tok_store_stream(p.mid + b * 16 * 256, sS, T, tid);
  __syncthreads();
  // hidden channels [256 sl, 256 sl + 256) = relu(lin1), then their share of lin2 (the K slice at column 256 sl of its weight)
  tok_linear_list<256>(16, [&](int i) {
    const int nb = 16 * sl + i;
    return TokBlk{p.l1.w + (int64_t)16 * nb * p.l1.ldw, p.l1.b + 16 * nb, sXb, p.l1.ldw, 16};
  }, TOK_XP, wave, lane, [&](int i, const f32x4& a) { st_block_bf16(sH, TOK_XP, i & 15, lane, a, true); });
  __syncthreads();
  float* mine = p.part + ((b * TOK_B_SPLIT + sl) * 16) * 256;
  tok_linear_list<256>(16, [&](int nb) {
    return TokBlk{p.l2.w + (int64_t)16 * nb * p.l2.ldw + 256 * sl, nullptr, sH, p.l2.ldw, 16};
  }, TOK_XP, wave, lane, [&](int nb, const f32x4& a) {
    *reinterpret_cast<float4*>(mine + m * 256 + 16 * nb + 4 * g) = make_float4(a[0], a[1], a[2], a[3]);
  });
}

__global__ __launch_bounds__(512) void tok_b2_kernel(const TokBParams p) {
  __shared__ __attribute__((aligned(16))) float sS[16 * 256], sT[16 * 256];
  __shared__ __attribute__((aligned(16))) char sXa[16 * TOK_XP], sXb[16 * TOK_XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t b = blockIdx.x;
  const int T = p.T;
  tok_load_stream(sT, p.t32 + b * T * 256, T, tid);
  for (int i = tid; i < 16 * 64; i += 512) {   // stream = norm2 output + lin2 bias + the shares in slice order; rows >= T zero
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < T * 64) {
      const float4 base = *reinterpret_cast<const float4*>(p.mid + b * 16 * 256 + 4 * i);
      const float4 bb = *reinterpret_cast<const float4*>(p.l2.b + 4 * (i & 63));
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < TOK_B_SPLIT; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(p.part + ((b * TOK_B_SPLIT + k) * 16) * 256 + 4 * i);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      out = make_float4(base.x + bb.x + sum.x, base.y + bb.y + sum.y, base.z + bb.z + sum.z, base.w + bb.w + sum.w);
    }
    *reinterpret_cast<float4*>(sS + 4 * i) = out;
  }
  __syncthreads();
  tok_layernorm(sS, p.g3, p.b3, p.eps, wave, lane);
  __syncthreads();
  tok_make_x(sXa, sS, sT, T, tid);
  tok_make_x(sXb, sS, nullptr, T, tid);
  tok_store_stream(p.q32 + b * T * 256, sS, T, tid);
  __syncthreads();
  tok_linear_list<256>(p.tq ? 24 : 16, [&](int i) {   // k | v (| q of the final attention): 8 blocks each
    const int which = i >> 3, nb = i & 7;
    const TokLin& L = which == 0 ? p.ik : (which == 1 ? p.iv : p.fq);
    return TokBlk{L.w + (int64_t)16 * nb * L.ldw, L.b ? L.b + 16 * nb : nullptr, which == 1 ? sXb : sXa, L.ldw, 16};
  }, TOK_XP, wave, lane, [&](int i, const f32x4& a) {
    const int which = i >> 3;
    st_block_g128((which == 0 ? p.tk : (which == 1 ? p.tv : p.tq)) + b * T * 128, T, i & 7, lane, a);
  });
}

struct TokDParams {
  const float* q32;    // in: stream before the final attention's residual
  const bf16_t* ta;
  bf16_t* hs;          // out: bf16(hs) [Bp][T][256] (the layer-by-layer path's `queries`), or null
  TokLin xo;
  const float *gf, *bf;
  TokLin mlp[6][3];    // 0-3 hypernetwork MLPs (mask tokens 2 + i), 4 IoU head (token 1), 5 object-score head (token 0)
  bf16_t* hyper;       // [Bp][4][32]
  float* iou;          // [Bp][8] fp32, sigmoid applied, 4 used
  bf16_t* obj;         // [Bp][8], 1 used
  int T;
  float eps;
};

// grid (prompt, MLP r): the six heads are independent after the final attention's out_proj + norm, which every workgroup of a
// prompt recomputes (64 KB of weights) instead of waiting for a sibling -- 192 workgroups of 0.3 MB each instead of 32 of 1.7 MB
__global__ __launch_bounds__(512) void tok_d_kernel(const TokDParams p) {
  __shared__ __attribute__((aligned(16))) float sS[16 * 256];
  __shared__ __attribute__((aligned(16))) char sX[16 * TOK_XP], sY[16 * TOK_XP], sA[16 * TOK_AP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int64_t b = blockIdx.x;
  const int r = blockIdx.y;                                   // 0-3 hypernetwork MLPs, 4 IoU head, 5 object-score head
  const int tok = r < 4 ? 2 + r : (r == 4 ? 1 : 0);           // the token it reads (mask_decoder.py:224-242)
  const int T = p.T;
  tok_load_stream(sS, p.q32 + b * T * 256, T, tid);
  tok_load_a128(sA, p.ta + b * T * 128, T, tid);
  __syncthreads();
  tok_linear<128>(p.xo, 256, sA, TOK_AP, wave, lane, [&](int nb, const f32x4& a) {
    if (m < T) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sS[m * 256 + 16 * nb + 4 * g + i] += a[i];
    }
  });
  __syncthreads();
  tok_layernorm(sS, p.gf, p.bf, p.eps, wave, lane);
  __syncthreads();
  tok_make_x(sX, sS, nullptr, T, tid);   // = bf16(hs)
  __syncthreads();
  if (p.hs && r == 0)
    for (int i = tid; i < T * 32; i += 512) {
      const int rr = i >> 5, c = (i & 31) * 8;
      *reinterpret_cast<uint4*>(p.hs + (b * T + rr) * 256 + c) = *reinterpret_cast<const uint4*>(sX + rr * TOK_XP + c * 2);
    }
  // the 3-layer MLP on ONE token: every product runs over all 16 columns of the operand (the matrix cores are idle anyway) and
  // only the column of the token is kept -- it becomes row 0 of the next layer's operand
  for (int i = tid; i < 16 * TOK_XP / 16; i += 512) reinterpret_cast<uint4*>(sY)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  tok_linear<256>(p.mlp[r][0], 256, sX, TOK_XP, wave, lane, [&](int nb, const f32x4& a) {
    if (m == tok) *reinterpret_cast<s16x4*>(sY + (16 * nb + 4 * g) * 2) = pack4(fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(a[2], 0.f), fmaxf(a[3], 0.f));
  });
  __syncthreads();
  tok_linear<256>(p.mlp[r][1], 256, sY, TOK_XP, wave, lane, [&](int nb, const f32x4& a) {
    if (m == 0) *reinterpret_cast<s16x4*>(sX + (16 * nb + 4 * g) * 2) = pack4(fmaxf(a[0], 0.f), fmaxf(a[1], 0.f), fmaxf(a[2], 0.f), fmaxf(a[3], 0.f));
  });
  __syncthreads();
  // layer 2: 32 hypernetwork outputs of a mask token (bf16), 4 IoU scores (fp32 accumulators + sigmoid, never rounded to bf16:
  // they pick the single-mask fallback by argmax, mask_decoder.py:236-242), 1 object score
  tok_linear<256>(p.mlp[r][2], r < 4 ? 32 : (r == 4 ? 4 : 1), sX, TOK_XP, wave, lane, [&](int nb, const f32x4& a) {
    if (m != 0) return;
    if (r < 4) {
      *reinterpret_cast<s16x4*>(p.hyper + (b * 4 + r) * 32 + 16 * nb + 4 * g) = pack4(a[0], a[1], a[2], a[3]);
    } else if (r == 4) {
      if (g == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) p.iou[b * 8 + j] = 1.f / (1.f + expf(-a[j]));
      }
    } else if (g == 0) {
      p.obj[b * 8] = f32_to_bf16(a[0]);
    }
  });
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
  if (esam3_allow_dyn_lds((const void*)i2t_block_kernel, I2T_LDS)) return -1;   // memoised per (device, kernel): asked on every launch
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
  if (!esam3_t2i_mfma_ok(1, Nq, Nk, 8, 16) || Bp < 1 || Bp > 65535 || (ldq % 4) || (ldk % 8) || (ldv % 8) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || !scratch) {
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

// ---- token side ------------------------------------------------------------------------------------------------------------
bool esam3_tok_fused_ok(int dtype, int T) { return dtype == 1 && T >= 6 && T <= 16; }

static TokLin tl(const esam3_tok_lin& l) { return TokLin{(const bf16_t*)l.w, l.bias, l.ldw}; }

int esam3_launch_tok_a(float* q32, const float* t32, void* tq, const esam3_tok_lin lin[5], const float* g1, const float* b1, float eps,
                       int Bp, int T, int first, hipStream_t s) {
  if (!esam3_tok_fused_ok(1, T)) { esam3_set_error("tok_a: T = %d unsupported", T); return -1; }
  TokAParams p;
  p.q32 = q32; p.t32 = t32; p.tq = (bf16_t*)tq;
  p.sq = tl(lin[0]); p.sk = tl(lin[1]); p.sv = tl(lin[2]); p.so = tl(lin[3]); p.xq = tl(lin[4]);
  p.g1 = g1; p.b1 = b1; p.T = T; p.first = first; p.eps = eps;
  hipLaunchKernelGGL(tok_a_kernel, dim3((unsigned)Bp), dim3(512), 0, s, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int64_t esam3_tok_b_scratch_bytes(int Bp) { return (int64_t)Bp * (TOK_B_SPLIT + 1) * 16 * 256 * 4; }
int esam3_launch_tok_b(float* q32, const float* t32, const void* ta, void* tk, void* tv, void* tq_final, const esam3_tok_lin lin[6],
                       const float* g2, const float* b2, const float* g3, const float* b3, float eps, void* scratch, int Bp, int T,
                       hipStream_t s) {
  if (!esam3_tok_fused_ok(1, T) || !scratch) { esam3_set_error("tok_b: T = %d unsupported", T); return -1; }
  TokBParams p;
  p.q32 = q32; p.t32 = t32; p.ta = (const bf16_t*)ta; p.tk = (bf16_t*)tk; p.tv = (bf16_t*)tv; p.tq = (bf16_t*)tq_final;
  p.xo = tl(lin[0]); p.l1 = tl(lin[1]); p.l2 = tl(lin[2]); p.ik = tl(lin[3]); p.iv = tl(lin[4]); p.fq = tl(lin[5]);
  p.g2 = g2; p.b2 = b2; p.g3 = g3; p.b3 = b3; p.T = T; p.eps = eps;
  p.part = (float*)scratch;
  p.mid = p.part + (int64_t)Bp * TOK_B_SPLIT * 16 * 256;
  hipLaunchKernelGGL(tok_b1_kernel, dim3((unsigned)Bp, (unsigned)TOK_B_SPLIT), dim3(512), 0, s, p);
  HIP_CHECK_RET(hipGetLastError());
  hipLaunchKernelGGL(tok_b2_kernel, dim3((unsigned)Bp), dim3(512), 0, s, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int esam3_launch_tok_d(float* q32, const void* ta, void* hs, const esam3_tok_lin& xo, const float* gf, const float* bf, float eps,
                       const esam3_tok_lin mlp[18], void* hyper, float* iou, void* obj, int Bp, int T, hipStream_t s) {
  if (!esam3_tok_fused_ok(1, T) || Bp < 1) { esam3_set_error("tok_d: T = %d, Bp = %d unsupported", T, Bp); return -1; }
  TokDParams p;
  p.q32 = q32; p.ta = (const bf16_t*)ta; p.hs = (bf16_t*)hs; p.xo = tl(xo); p.gf = gf; p.bf = bf;
  for (int r = 0; r < 6; ++r)
    for (int l = 0; l < 3; ++l) p.mlp[r][l] = tl(mlp[r * 3 + l]);
  p.hyper = (bf16_t*)hyper; p.iou = iou; p.obj = (bf16_t*)obj; p.T = T; p.eps = eps;
  hipLaunchKernelGGL(tok_d_kernel, dim3((unsigned)Bp, 6u), dim3(512), 0, s, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---- 256 -> 256 row-wise linear with a position table -------------------------------------------------------------------------
bool esam3_rowlin256_ok(int dtype, int64_t rows, int N, int K, int P) {
  return dtype == 1 && N == 256 && K == 256 && rows > 0 && rows % 32 == 0 && (P <= 0 || P % 32 == 0);
}
int esam3_launch_rowlin256(const void* x, const void* w, int kp, const float* bias, const void* table, int P, void* out, int64_t rows,
                           hipStream_t s) {
  if (!esam3_rowlin256_ok(1, rows, 256, 256, table ? P : 0) || kp < 256 || (kp % 8)) {
    esam3_set_error("rowlin256: unsupported shape (rows=%lld kp=%d P=%d)", (long long)rows, kp, P);
    return -1;
  }
  if (esam3_allow_dyn_lds((const void*)rowlin256_kernel, RL_LDS)) return -1;   // memoised per (device, kernel): asked on every launch
  RowLinParams p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.table = (const bf16_t*)table; p.bias = bias; p.out = (bf16_t*)out; p.rows = rows;
  p.kp = kp; p.P = P; p.abl = esam3_dev_flag("ESAM3_RL_ABL");
  const int64_t G = rows / 32;
  int grid = (int)((G + 7) / 8);
  if (grid > 256) grid = 256;
  hipLaunchKernelGGL(rowlin256_kernel, dim3(grid), dim3(512), RL_LDS, s, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
