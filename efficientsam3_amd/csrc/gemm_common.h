// Pieces shared by the implicit-GEMM kernels (gemm_conv.hip): MFMA wrappers, the LDS
// swizzle and the fused epilogue (bias / activation / residual / ConvT pixel-shuffle /
// zero-border padded output) for one 8-channel chunk of one output row.
#pragma once
#include "esam3_common.h"

// 16-byte register value as a first-class vector (an array of struct uint4 is not reliably
// promoted out of scratch by the compiler).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct MmaOps;
template <> struct MmaOps<bf16_t> {
  // one 16-byte fragment = 8 bf16 along K -> one v_mfma_f32_32x32x16_bf16
  static __device__ __forceinline__ void mma(const u32x4& w, const u32x4& a, f32x16_v& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, w),
                                                  __builtin_bit_cast(bf16x8_v, a), acc, 0, 0, 0);
  }
};
template <> struct MmaOps<float> {
  // one 16-byte fragment = 4 f32 along K -> four v_mfma_f32_32x32x2_f32 (exact f32 FMA chain)
  static __device__ __forceinline__ void mma(const u32x4& w, const u32x4& a, f32x16_v& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
  }
};

// LDS tiles are [rows][128 B]; the 16-byte slot index is XOR-ed with (row>>1)&7 so that the
// 16-lane ds_read_b128 groups and the 8-lane ds_write_b128 groups touch distinct banks.
__device__ __forceinline__ int swz(int row, int slot) { return (slot ^ ((row >> 1) & 7)) << 4; }

// Epilogue for output row m, channels [n, n+8): v holds the raw accumulators.
template <typename T>
__device__ __forceinline__ void epilogue_chunk(const GemmParams& p, int64_t m64, int n, float (&v)[8]) {
  const unsigned m = (unsigned)m64;  // the launcher guarantees M < 2^31: 32-bit divisions only
  constexpr int OC = 8;
  T* __restrict__ gO = reinterpret_cast<T*>(p.out);
  const T* __restrict__ gR = reinterpret_cast<const T*>(p.res);
  const int HW = p.H * p.W;
  const int P = p.out_pad;  // 0 or 1: output written inside a zero border
  int64_t o_off, r_off;
  int bias_n = n;
  if (p.out_mode == OUT_CONVT2X2) {
    const int tap = n / p.convt_cout, co = n - tap * p.convt_cout;
    const unsigned b = m / (unsigned)HW;
    const unsigned rem = m - b * (unsigned)HW;
    const unsigned h = rem / (unsigned)p.W, w = rem - h * (unsigned)p.W;
    const unsigned oh = 2 * h + (tap >> 1), ow = 2 * w + (tap & 1);
    const int OHp = 2 * p.H + 2 * P, OWp = 2 * p.W + 2 * P;
    o_off = ((int64_t)(b * (unsigned)OHp + oh + P) * OWp + ow + P) * p.ldc + co;
    const unsigned rb = p.res_bidx ? (unsigned)p.res_bidx[b] : b;
    r_off = ((int64_t)rb * 4 * HW + (int64_t)oh * (2 * p.W) + ow) * p.ldr + co;
    bias_n = co;
  } else {
    if (P) {
      const unsigned b = m / (unsigned)HW;
      const unsigned rem = m - b * (unsigned)HW;
      const unsigned h = rem / (unsigned)p.W, w = rem - h * (unsigned)p.W;
      o_off = ((int64_t)(b * (unsigned)(p.H + 2) + h + 1) * (p.W + 2) + w + 1) * p.ldc + n;
    } else {
      o_off = (int64_t)m * p.ldc + n;
    }
    unsigned rrow = m;
    if (p.res_mod > 0) rrow = m % (unsigned)p.res_mod;
    else if (p.res_bidx) {
      const unsigned b = m / (unsigned)HW;
      rrow = (unsigned)p.res_bidx[b] * (unsigned)HW + (m - b * (unsigned)HW);
    }
    r_off = (int64_t)rrow * p.ldr + n;
  }
  const int valid = (p.N - n) < OC ? (p.N - n) : OC;
  float r[OC];
#pragma unroll
  for (int e = 0; e < OC; ++e) {
    r[e] = (gR && e < valid) ? to_f32<T>(gR[r_off + e]) : 0.f;
    if (p.bias && e < valid) v[e] += p.bias[bias_n + e];
    if (!p.res_after_act) v[e] += r[e];
  }
  act_apply_n<OC>(v, p.act);
  if (p.res_after_act) {
#pragma unroll
    for (int e = 0; e < OC; ++e) v[e] += r[e];
  }
  const bool vec = (valid == OC) && (((uintptr_t)(gO + o_off)) & 15) == 0;
  if (vec) {
    if constexpr (sizeof(T) == 2) {
      u32x4 o;
      o.x = pack_bf16x2(v[0], v[1]);
      o.y = pack_bf16x2(v[2], v[3]);
      o.z = pack_bf16x2(v[4], v[5]);
      o.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<u32x4*>(gO + o_off) = o;
    } else {
      *reinterpret_cast<float4*>(gO + o_off) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(gO + o_off + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
  } else {
#pragma unroll
    for (int e = 0; e < OC; ++e)
      if (e < valid) gO[o_off + e] = from_f32<T>(v[e]);
  }
}
