// Implicit-GEMM convolution / linear kernels for gfx950 (CDNA4), bf16 and exact-f32.
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
// 16-byte coalesced NHWC stores (gemm_common.h).
//
// Two kernels:
//  * gemm256_kernel  -- 256x256 outputs per 512-thread workgroup (8 wavefronts, each
//    128 pixels x 64 channels = 4x2 MFMA 32x32 tiles, 128 accumulator registers).  Operand
//    tiles go HBM/L2 -> LDS directly with `global_load_lds` (16 B per lane, no VGPR staging, no
//    ds_write), double buffered, the next K tile's DMA stays in flight across the barrier
//    behind a counted `s_waitcnt vmcnt(8)`.  The XOR bank swizzle is applied on the *source*
//    address (the DMA destination is lane-linear).  A 256^2 tile needs 31 B/clk/CU from L2 at
//    full MFMA rate (the 128^2 tile needs 63 B/clk, more than an XCD's L2 can deliver per CU).
//    Used when N >= 128, K is a multiple of the 128-byte K tile and M >= 256.
//  * conv_gemm_kernel -- 128 x {128,64,32} tiles, 4 wavefronts, global -> VGPR -> LDS staging
//    with bounds checks: thin / ragged layers (backbone 1x1s, decoder token GEMMs).
//
// A 3x3 conv can read its input from a buffer with a 1-pixel zero border (in_pad): every
// tap is then in range and the A loads need no predicates -- which is what lets the DMA path
// serve the 3x3 convs.  Producers write such buffers through out_pad.
#include <cstdlib>

#include "gemm_common.h"

namespace {

// ======================================================================================
// 128 x BN kernel (register staged)
// ======================================================================================
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

  int64_t a_off[A_ITERS];         // element offset of the window origin, or -1
  int a_oh[A_ITERS], a_ow[A_ITERS];
  const int HW = p.H * p.W;
  const int Wp = p.W + 2 * p.in_pad;  // row pitch (pixels) of the input
#pragma unroll
  for (int i = 0; i < A_ITERS; ++i) {
    const int64_t m = m0 + lrow + 32 * i;
    a_oh[i] = 0;
    a_ow[i] = 0;
    if (m < p.M) {
      if (p.ksize == 3) {
        const int64_t b = m / HW;
        const int rem = (int)(m - b * HW);
        a_oh[i] = rem / p.W;
        a_ow[i] = rem - a_oh[i] * p.W;
        if (p.in_pad)  // origin = top-left pixel of the 3x3 window in the padded buffer
          a_off[i] = ((b * (p.H + 2) + a_oh[i]) * (int64_t)Wp + a_ow[i]) * p.lda;
        else           // origin = centre pixel
          a_off[i] = m * (int64_t)p.lda;
      } else {
        a_off[i] = m * (int64_t)p.lda;
      }
    } else {
      a_off[i] = -1;
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
  // Software pipeline, written inline: iteration kt issues the global loads of tile kt+1, runs
  // the MFMAs of tile kt from LDS buffer kt&1, then parks tile kt+1 in the other buffer.
  for (int kt = -1; kt < nk; ++kt) {
    const bool more = kt + 1 < nk;
    if (more) {
#pragma unroll
      for (int i = 0; i < B_ITERS; ++i) {
        const int n = n0 + lrow + 32 * i;
        rb[i] = *reinterpret_cast<const u32x4*>(gW + (int64_t)n * p.Kp + (int64_t)(kt + 1) * BKE + slot * EPC);
      }
      const bool k_ok = k_lin < p.K;
      if (p.ksize == 3) {
        const int kh = k_tap / 3, kw = k_tap - kh * 3;
        if (p.in_pad) {
          const int64_t toff = ((int64_t)kh * Wp + kw) * p.lda + k_c;
#pragma unroll
          for (int i = 0; i < A_ITERS; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (k_ok && a_off[i] >= 0) v = *reinterpret_cast<const u32x4*>(gA + a_off[i] + toff);
            ra[i] = v;
          }
        } else {
          const int dh = kh - 1, dw = kw - 1;
#pragma unroll
          for (int i = 0; i < A_ITERS; ++i) {
            const int ih = a_oh[i] + dh, iw = a_ow[i] + dw;
            const bool okk = k_ok && a_off[i] >= 0 && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (okk) v = *reinterpret_cast<const u32x4*>(gA + a_off[i] + ((int64_t)dh * p.W + dw) * p.lda + k_c);
            ra[i] = v;
          }
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

  constexpr int CPR = BN / 8;  // 8-channel chunks per tile row
  for (int c = tid; c < BM * CPR; c += NT) {
    const int ml = c / CPR, cc = c - ml * CPR;
    const int64_t m = m0 + ml;
    const int n = n0 + cc * 8;
    if (m >= p.M || n >= p.N) continue;
    float v[8];
    const float4 v0 = *reinterpret_cast<const float4*>(lds_c + ml * LDN + cc * 8);
    const float4 v1 = *reinterpret_cast<const float4*>(lds_c + ml * LDN + cc * 8 + 4);
    v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
    v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
    epilogue_chunk<T>(p, m, n, v);
  }
}

// ======================================================================================
// 256 x 256 kernel (LDS-DMA staged)
// ======================================================================================
__device__ __forceinline__ void glds16(const void* gsrc, char* lds_wave_base) {
  // 16 bytes per lane, destination = wave-uniform base + lane*16
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 256, NT = 512;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BKE = 128 / (int)sizeof(T);
  constexpr int STAGE = (BM + BN) * 128;  // 64 KB
  constexpr int LDN = BN + 4;
  constexpr int EROWS = 64;                // epilogue pass height

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds_c = reinterpret_cast<float*>(smem);

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
  const int64_t m0 = (bid / tiles_n) * BM;
  const int n0 = tile_n * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wave tile: pixels [wm*128,+128) x channels [wn*64,+64)
  const int l31 = lane & 31, g = lane >> 5;

  const T* __restrict__ gA = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ gW = reinterpret_cast<const T*>(p.Wt);
  const int HW = p.H * p.W;
  const int Wp = p.W + 2 * p.in_pad;

  // ---- DMA descriptors: wave w fills tile rows [32w, 32w+32) of A and of B, 8 rows (1 KB)
  //      per instruction; lane -> (row = 8j + lane/8, physical slot = lane%8); the logical slot
  //      it fetches is physical ^ ((row>>1)&7)  (source-side swizzle).
  int64_t a_src[4], b_src[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = wave * 32 + j * 8 + (lane >> 3);
    const int lslot = (lane & 7) ^ ((row >> 1) & 7);
    int64_t m = m0 + row;
    if (m >= p.M) m = p.M - 1;  // clamp: rows past M are computed but never stored
    int64_t off;
    if (p.ksize == 3) {
      const int64_t b = m / HW;
      const int rem = (int)(m - b * HW);
      const int oh = rem / p.W, ow = rem - oh * p.W;
      off = ((b * (p.H + 2) + oh) * (int64_t)Wp + ow) * p.lda;  // in_pad is required for ksize 3
    } else {
      off = m * (int64_t)p.lda;
    }
    a_src[j] = off + lslot * EPC;
    b_src[j] = (int64_t)(n0 + row) * p.Kp + lslot * EPC;
  }

  f32x16_v acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BKE;
#define ESAM3_ISSUE_TILE(KT)                                                          \
  do {                                                                                \
    char* sa_ = smem + ((KT) & 1) * STAGE + wave * 32 * 128;                          \
    char* sb_ = sa_ + BM * 128;                                                       \
    int64_t koff_ = (int64_t)(KT) * BKE;                                              \
    if (p.ksize == 3) {                                                               \
      const int k0_ = (KT) * BKE;                                                     \
      const int tap_ = k0_ / p.Cin, c0_ = k0_ - tap_ * p.Cin;                         \
      const int kh_ = tap_ / 3, kw_ = tap_ - kh_ * 3;                                 \
      koff_ = ((int64_t)kh_ * Wp + kw_) * p.lda + c0_;                                \
    }                                                                                 \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                  \
        glds16(gA + a_src[j_] + koff_, sa_ + j_ * 1024);                              \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                  \
        glds16(gW + b_src[j_] + (int64_t)(KT) * BKE, sb_ + j_ * 1024);                \
  } while (0)

  ESAM3_ISSUE_TILE(0);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      ESAM3_ISSUE_TILE(kt + 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile kt landed; tile kt+1 stays in flight
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    const char* la = smem + (kt & 1) * STAGE;
    const char* lb = la + BM * 128;
#pragma unroll
    for (int ck = 0; ck < 4; ++ck) {
      u32x4 fa[4], fw[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 128 + i * 32 + l31;
        fa[i] = *reinterpret_cast<const u32x4*>(la + row * 128 + swz(row, ck * 2 + g));
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int row = wn * 64 + j * 32 + l31;
        fw[j] = *reinterpret_cast<const u32x4*>(lb + row * 128 + swz(row, ck * 2 + g));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) MmaOps<T>::mma(fw[j], fa[i], acc[i][j]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all ds_reads of this buffer retired
    __builtin_amdgcn_s_barrier();                      // before anybody's DMA overwrites it
  }
#undef ESAM3_ISSUE_TILE

  // ---- epilogue in 4 passes of 64 pixels: accumulators -> LDS fp32 -> coalesced stores ----
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    if (wm == (pass >> 1)) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int i = 2 * (pass & 1) + ii;
            const int ml = ii * 32 + l31;
            const int nl = wn * 64 + j * 32 + 8 * q + 4 * g;
            float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                                   acc[i][j][4 * q + 3]);
            *reinterpret_cast<float4*>(lds_c + ml * LDN + nl) = v;
          }
    }
    __syncthreads();
    constexpr int CPR = BN / 8;
#pragma unroll
    for (int it = 0; it < EROWS * CPR / NT; ++it) {
      const int c = tid + it * NT;
      const int ml = c / CPR, cc = c - ml * CPR;
      const int64_t m = m0 + pass * EROWS + ml;
      const int n = n0 + cc * 8;
      if (m < p.M && n < p.N) {
        float v[8];
        const float4 v0 = *reinterpret_cast<const float4*>(lds_c + ml * LDN + cc * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(lds_c + ml * LDN + cc * 8 + 4);
        v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w;
        v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        epilogue_chunk<T>(p, m, n, v);
      }
    }
    __syncthreads();
  }
}

// ======================================================================================
// launchers
// ======================================================================================
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
int launch_256(const GemmParams& p, hipStream_t stream) {
  constexpr size_t lds = 2 * (size_t)(256 + 256) * 128;  // 128 KB (epilogue needs 66.5 KB)
  static bool attr_set = false;
  auto kern = gemm256_kernel<T>;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int64_t tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(512), lds, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <typename T>
bool use_256(const GemmParams& p) {
  constexpr int BKE = 128 / (int)sizeof(T);
  if (p.N < 128 || p.M < 256 || p.K % BKE != 0 || p.K != p.Kp) return false;
  if (p.ksize == 3 && (!p.in_pad || p.Cin % BKE != 0)) return false;
  if (p.lda % (16 / (int)sizeof(T)) != 0) return false;
  return true;
}

template <typename T>
int launch_gemm_t(const GemmParams& p, hipStream_t stream, int force_small) {
  if (!force_small && use_256<T>(p)) return launch_256<T>(p, stream);
  if (p.N > 64) return launch_cfg<T, 128, 128, 2, 2>(p, stream);
  if (p.N > 32) return launch_cfg<T, 128, 64, 2, 2>(p, stream);
  return launch_cfg<T, 128, 32, 4, 1>(p, stream);
}

}  // namespace

// Rows of the packed weight matrix are padded so that every tile configuration that may be
// chosen for this N reads whole tiles: 256 for N >= 128, else the 128-kernel's BN.
int esam3_gemm_pad_n(int N) {
  const int bn = N >= 128 ? 256 : (N > 64 ? 128 : (N > 32 ? 64 : 32));
  return (N + bn - 1) / bn * bn;
}
int esam3_gemm_pad_k(int K, int elem_size) {
  const int bke = 128 / elem_size;
  return (K + bke - 1) / bke * bke;
}

int esam3_launch_gemm(int dtype /*0 f32, 1 bf16*/, const GemmParams& p, hipStream_t stream) {
  static const int force_small = getenv("ESAM3_GEMM_SMALL") ? atoi(getenv("ESAM3_GEMM_SMALL")) : 0;
  return dtype == 0 ? launch_gemm_t<float>(p, stream, force_small)
                    : launch_gemm_t<bf16_t>(p, stream, force_small);
}
