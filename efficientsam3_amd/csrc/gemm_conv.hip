// Implicit-GEMM convolution / linear kernel for gfx950 (CDNA4), bf16 and exact-f32.
//
// Replaces what the reference gets from MIOpen/ATen for every dense contraction on the
// hot path: 1x1 convs (efficientvit/nn/ops.py:39-80), 3x3 convs and 2x2/s2 transposed
// convs of the ViTDet neck (model/necks.py:42-92), the student head
// (model_builder.py:770-775) and the decoder's nn.Linear layers (sam/transformer.py:226-231).
//
// Shape of the computation: out[m][n] = sum_k A[m][k] * Wt[n][k], m = output pixel,
// k = (tap, cin).  The MFMA is issued with the *weights* as the A operand and the
// activations as the B operand (D[i=n][j=m]) so that every lane ends up holding four
// consecutive output channels of one pixel -- a 16-byte LDS write per 4 accumulators --
// and the epilogue re-reads the tile row-wise to do bias / activation / residual and
// 16-byte coalesced NHWC stores.
//
// Tile: BM x BN outputs per 256-thread workgroup (4 wavefronts of 64), K step = 128
// bytes per row (64 bf16 / 32 f32).  Operand tiles are staged global -> VGPR -> LDS with
// the next K tile's global loads in flight during the current tile's MFMAs
// (one barrier per K tile, two LDS buffers).  LDS rows are 128 B; the 16-byte slot index
// is XOR-swizzled with (row>>1)&7 so that both the 8-lane ds_write_b128 groups and the
// 16-lane ds_read_b128 groups hit distinct banks (MI355X_MICROARCH.md, LDS table).
#include "esam3_common.h"

namespace {

// 16-byte register value as a first-class vector (a struct uint4 array is not reliably
// promoted out of scratch by the compiler).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct MmaOps;

template <> struct MmaOps<bf16_t> {
  // one 16-byte fragment = 8 bf16 along K -> one v_mfma_f32_32x32x16_bf16
  static __device__ inline void mma(const u32x4& w, const u32x4& a, f32x16_v& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_v, w),
                                                  __builtin_bit_cast(bf16x8_v, a), acc, 0, 0, 0);
  }
};
template <> struct MmaOps<float> {
  // one 16-byte fragment = 4 f32 along K -> four v_mfma_f32_32x32x2_f32 (exact f32 FMA chain)
  static __device__ inline void mma(const u32x4& w, const u32x4& a, f32x16_v& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.x), __uint_as_float(a.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.y), __uint_as_float(a.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.z), __uint_as_float(a.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(w.w), __uint_as_float(a.w), acc, 0, 0, 0);
  }
};

__device__ inline int swz(int row, int slot) { return (slot ^ ((row >> 1) & 7)) << 4; }

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_gemm_kernel(GemmParams p) {
  constexpr int NT = WM * WN * 64;
  constexpr int EPC = 16 / (int)sizeof(T);   // elements per 16-byte chunk
  constexpr int BKE = 128 / (int)sizeof(T);  // K elements per tile
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_ITERS = BM * 8 / NT, B_ITERS = BN * 8 / NT;
  constexpr int LDN = BN + 4;                // padded fp32 row of the epilogue tile
  static_assert(A_ITERS >= 1 && B_ITERS >= 1, "tile too small for the block");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* lds_a = smem;                       // [2][BM][128 B]
  char* lds_b = smem + 2 * BM * 128;        // [2][BN][128 B]
  float* lds_c = reinterpret_cast<float*>(smem);  // epilogue: [BM][LDN] fp32 (aliases a/b)

  // ---- XCD-aware tile mapping: consecutive tiles (sharing A rows) stay on one XCD ----
  const int tiles_n = (p.N + BN - 1) / BN;
  const int64_t tiles_m = (p.M + BM - 1) / BM;
  const int64_t nblk = tiles_m * tiles_n;
  int64_t bid = blockIdx.x;
  {
    const int64_t q = nblk / 8, r = nblk % 8;
    const int64_t xcd = bid % 8, idx = bid / 8;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_n = (int)(bid % tiles_n);
  const int64_t tile_m = bid / tiles_n;
  const int64_t m0 = tile_m * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

  // ---- per-thread load descriptors ---------------------------------------------------
  const int slot = tid & 7;       // 16-byte slot inside the 128-byte K row
  const int lrow = tid >> 3;      // first tile row this thread stages (step 32)
  const T* __restrict__ gA = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ gW = reinterpret_cast<const T*>(p.Wt);

  int64_t a_off[A_ITERS];         // element offset of the centre pixel's channel 0, or -1
  int a_oh[A_ITERS], a_ow[A_ITERS];
  const int HW = p.H * p.W;
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int64_t m = m0 + lrow + 32 * i;
    if (m < p.M) {
      if (p.ksize == 3) {
        const int64_t b = m / HW;
        const int rem = (int)(m - b * HW);
        a_oh[i] = rem / p.W;
        a_ow[i] = rem - a_oh[i] * p.W;
      } else {
        a_oh[i] = 0;
        a_ow[i] = 0;
      }
      a_off[i] = m * (int64_t)p.lda;
    } else {
      a_off[i] = -1;
      a_oh[i] = 0;
      a_ow[i] = 0;
    }
  }

  u32x4 ra[A_ITERS], rb[B_ITERS];
  int k_c = slot * EPC;  // channel index within the current tap for this thread's slot
  int k_tap = 0;         // current tap (3x3) -- advanced incrementally
  int k_lin = slot * EPC;  // linear k index
  if (p.ksize == 3) {
    while (k_c >= p.Cin) { k_c -= p.Cin; ++k_tap; }
  }

  f32x16_v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int l31 = lane & 31, g = lane >> 5;
  const int nk = p.Kp / BKE;
  // Software pipeline, written inline (no lambdas: captured register arrays end up in
  // scratch): iteration kt issues the global loads of tile kt+1, runs the MFMAs of tile kt
  // from LDS buffer kt&1, then parks tile kt+1 in the other buffer.  kt = -1 is the prologue.
  for (int kt = -1; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
      // weights: rows n0 + lrow + 32*i of the packed [Np][Kp] matrix (always in range)
#pragma unroll
      for (int i = 0; i < B_ITERS; ++i) {
        const int n = n0 + lrow + 32 * i;
        rb[i] = *reinterpret_cast<const u32x4*>(gW + (int64_t)n * p.Kp + (int64_t)(kt + 1) * BKE + slot * EPC);
      }
      const bool k_ok = k_lin < p.K;
      if (p.ksize == 3) {
        const int dh = k_tap / 3 - 1, dw = k_tap - (k_tap / 3) * 3 - 1;
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
          const int ih = a_oh[i] + dh, iw = a_ow[i] + dw;
          const bool okk = k_ok && a_off[i] >= 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          u32x4 v = {0u, 0u, 0u, 0u};
          if (okk) v = *reinterpret_cast<const u32x4*>(gA + a_off[i] + ((int64_t)dh * p.W + dw) * p.lda + k_c);
          ra[i] = v;
        }
      } else {
#pragma unroll
        for (int i = 0; i < A_ITERS; ++i) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (k_ok && a_off[i] >= 0) v = *reinterpret_cast<const u32x4*>(gA + a_off[i] + k_lin);
          ra[i] = v;
        }
      }
      k_lin += BKE;
      if (p.ksize == 3) {
        k_c += BKE;
        while (k_c >= p.Cin) { k_c -= p.Cin; ++k_tap; }
      }
    }
    if (kt >= 0) {
      const char* la = lds_a + (kt & 1) * BM * 128;
      const char* lb = lds_b + (kt & 1) * BN * 128;
#pragma unroll
      for (int ck = 0; ck < 4; ++ck) {
        u32x4 fa[TM], fw[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm0 + i * 32 + l31;
          fa[i] = *reinterpret_cast<const u32x4*>(la + row * 128 + swz(row, ck * 2 + g));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wn0 + j * 32 + l31;
          fw[j] = *reinterpret_cast<const u32x4*>(lb + row * 128 + swz(row, ck * 2 + g));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) MmaOps<T>::mma(fw[j], fa[i], acc[i][j]);
      }
    }
    if (more) {
      const int buf = (kt + 1) & 1;
#pragma unroll
      for (int i = 0; i < A_ITERS; ++i) {
        const int row = lrow + 32 * i;
        *reinterpret_cast<u32x4*>(lds_a + buf * BM * 128 + row * 128 + swz(row, slot)) = ra[i];
      }
#pragma unroll
      for (int i = 0; i < B_ITERS; ++i) {
        const int row = lrow + 32 * i;
        *reinterpret_cast<u32x4*>(lds_b + buf * BN * 128 + row * 128 + swz(row, slot)) = rb[i];
      }
    }
    __syncthreads();
  }

  // ---- epilogue: accumulators -> LDS (fp32, [m][n]) -> coalesced NHWC stores --------
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ml = wm0 + i * 32 + l31;
        const int nl = wn0 + j * 32 + 8 * q + 4 * g;
        float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                               acc[i][j][4 * q + 3]);
        *reinterpret_cast<float4*>(lds_c + ml * LDN + nl) = v;
      }
  __syncthreads();

  constexpr int OC = 8;                 // output channels per thread-chunk
  constexpr int CPR = BN / OC;          // chunks per tile row
  T* __restrict__ gO = reinterpret_cast<T*>(p.out);
  const T* __restrict__ gR = reinterpret_cast<const T*>(p.res);
  for (int c = tid; c < BM * CPR; c += NT) {
    const int ml = c / CPR, cc = c - ml * CPR;
    const int64_t m = m0 + ml;
    const int n = n0 + cc * OC;
    if (m >= p.M || n >= p.N) continue;
    float v[OC];
    {
      const float4 v0 = *reinterpret_cast<const float4*>(lds_c + ml * LDN + cc * OC);
      const float4 v1 = *reinterpret_cast<const float4*>(lds_c + ml * LDN + cc * OC + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
      v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    }
    int64_t o_off, r_off;
    int bias_n = n;
    if (p.out_mode == OUT_CONVT2X2) {
      const int tap = n / p.convt_cout, co = n - tap * p.convt_cout;
      const int64_t b = m / HW;
      const int rem = (int)(m - b * HW);
      const int h = rem / p.W, w = rem - h * p.W;
      const int64_t pin = (int64_t)(2 * h + (tap >> 1)) * (2 * p.W) + 2 * w + (tap & 1);
      const int64_t opix = b * 4 * HW + pin;
      o_off = opix * p.ldc + co;
      const int64_t rb = p.res_bidx ? (int64_t)p.res_bidx[b] : b;
      r_off = (rb * 4 * HW + pin) * p.ldr + co;
      bias_n = co;
    } else {
      o_off = m * (int64_t)p.ldc + n;
      int64_t rrow = m;
      if (p.res_mod > 0) rrow = m % p.res_mod;
      else if (p.res_bidx) {
        const int64_t b = m / HW;
        rrow = (int64_t)p.res_bidx[b] * HW + (m - b * HW);
      }
      r_off = rrow * (int64_t)p.ldr + n;
    }
    const int valid = (p.N - n) < OC ? (p.N - n) : OC;
#pragma unroll
    for (int e = 0; e < OC; ++e) {
      if (e < valid) {
        float x = v[e];
        if (p.bias) x += p.bias[bias_n + e];
        if (gR && !p.res_after_act) x += to_f32<T>(gR[r_off + e]);
        x = act_apply(x, p.act);
        if (gR && p.res_after_act) x += to_f32<T>(gR[r_off + e]);
        v[e] = x;
      }
    }
    const bool vec = (valid == OC) && (((uintptr_t)(gO + o_off)) & 15) == 0;
    if (vec) {
      if constexpr (sizeof(T) == 2) {
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
        o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
        o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
        *reinterpret_cast<uint4*>(gO + o_off) = o;
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
}

template <typename T, int BM, int BN, int WM, int WN>
int launch_cfg(const GemmParams& p, hipStream_t stream) {
  constexpr int LDN = BN + 4;
  constexpr size_t lds_ab = 2 * (size_t)(BM + BN) * 128;
  constexpr size_t lds_c = (size_t)BM * LDN * 4;
  constexpr size_t lds = lds_ab > lds_c ? lds_ab : lds_c;
  static bool attr_set = false;
  auto kern = conv_gemm_kernel<T, BM, BN, WM, WN>;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  if (tiles <= 0) return 0;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(WM * WN * 64), lds, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <typename T>
int launch_gemm_t(const GemmParams& p, hipStream_t stream) {
  if (p.N > 64) return launch_cfg<T, 128, 128, 2, 2>(p, stream);
  if (p.N > 32) return launch_cfg<T, 128, 64, 2, 2>(p, stream);
  return launch_cfg<T, 128, 32, 4, 1>(p, stream);
}

}  // namespace

// Tile height used to pad N in packed weights (every config reads whole BN rows of Wt).
int esam3_gemm_pad_n(int N) {
  const int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
  return (N + bn - 1) / bn * bn;
}
int esam3_gemm_pad_k(int K, int elem_size) {
  const int bke = 128 / elem_size;
  return (K + bke - 1) / bke * bke;
}

int esam3_launch_gemm(int dtype /*0 f32, 1 bf16*/, const GemmParams& p, hipStream_t stream) {
  return dtype == 0 ? launch_gemm_t<float>(p, stream) : launch_gemm_t<bf16_t>(p, stream);
}
