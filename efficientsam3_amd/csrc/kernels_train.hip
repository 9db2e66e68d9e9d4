// Building blocks of the stage-1 student-trunk BACKWARD (stage1/train_image_encoder_stage1.py:196-217: autocast
// forward -> loss -> backward through the student): the gradient kernels of the layers an EfficientViT MBConv / DSConv block is made of
// (backbones/efficientvit/nn/ops.py:39-81 ConvLayer = Conv2d (no bias) -> BatchNorm2d -> activation; :264-360 DSConv / MBConv), on NHWC
// rows like the rest of the engine.  Together with esam3_bn_train_forward / _backward (kernels_stage1.hip) and the forward operators
// (esam3_op_linear, esam3_op_dwconv) they are enough to run one such block forwards and backwards; tests/test_train_blocks.py composes
// them and checks every gradient against torch.autograd; efficientsam3_amd/stage1_train.py sequences them into the training step of every
// stage-1 student (round 4: EfficientViT; round 5: RepViT, TinyViT).  Nothing in the inference engine calls them.
//   esam3_act_forward / _backward      Hardswish | ReLU | GELU (erf) | identity and dx = dy * act'(x)
//   esam3_linear_wgrad                 dW[N][K] = dy[M][N]^T x[M][K]   (1x1 conv / Linear weight gradient; M = pixels is the
//                                      reduction dimension: a "TN" GEMM), optionally dbias[N] = sum_rows dy
//   esam3_colsum                       dbias[N] = sum_rows dy (layers with a conv bias instead of a BatchNorm)
//   esam3_dwconv_wgrad                 depthwise 3x3 | 5x5 weight gradient, stride 1 | 2, padding k / 2
//   esam3_dwconv_dgrad                 depthwise 3x3 | 5x5 data gradient, stride 1 | 2 (for stride 2 a transposed convolution)
//   esam3_lite_mla_backward            backward of LiteMLA's ReLU linear attention core (and its forward output for free)
//   esam3_channel_scale / esam3_batched_coldot   the per-channel scale / shift and the per-image channel reductions RepViT's RepVGGDW and
//                                      SqueezeExcite need forwards and backwards (sam3/backbones/repvit.py:84-161; timm SqueezeExcite)
// The data gradient of a 1x1 conv needs no new kernel: it is esam3_op_linear with the transposed weight.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "../../include/esam3.h"
#include "esam3_common.h"
#include "gemm_common.h"
#include "kernels.h"
#include "train_act.h"

namespace {

template <int DT> struct TElem;  // 0 f32, 1 bf16
template <> struct TElem<0> {
  using type = float;
  struct vec8 { float4 a, b; };   // 8 elements as loaded (unpacked later: lets a kernel issue many loads before the first use)
  static __device__ inline vec8 loadraw(const float* p) { return vec8{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)}; }
  static __device__ inline void unpack(const vec8& q, float* v) {
    v[0] = q.a.x; v[1] = q.a.y; v[2] = q.a.z; v[3] = q.a.w; v[4] = q.b.x; v[5] = q.b.y; v[6] = q.b.z; v[7] = q.b.w;
  }
  static __device__ inline void load8(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  static __device__ inline void store8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
template <> struct TElem<1> {
  using type = uint16_t;
  typedef uint4 vec8;
  static __device__ inline vec8 loadraw(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ inline void unpack(const vec8& q, float* v) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
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

// ---- activations and their derivatives: train_act.h ----
template <int DT, bool BWD>
__global__ __launch_bounds__(256) void act_kernel(const typename TElem<DT>::type* __restrict__ x,
                                                  const typename TElem<DT>::type* __restrict__ dy,
                                                  typename TElem<DT>::type* __restrict__ out, int64_t n8, int act) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
    float v[8], g[8], o[8];
    TElem<DT>::load8(x + i * 8, v);
    if constexpr (BWD) TElem<DT>::load8(dy + i * 8, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = BWD ? g[e] * act_grad(v[e], act) : act_fwd(v[e], act);
    TElem<DT>::store8(out + i * 8, o);
  }
}

// ---- dW[N][K] = dy^T x over the M rows ----------------------------------------------------------------------------------------
// Workgroup = a 64 x 64 tile of dW for one slice of the rows (split-K over the pixels; fp32 partial tiles are summed in a fixed
// order by wgrad_reduce_kernel: deterministic).  The four waves own the four 32 x 32 quadrants.  Rows stream through LDS in tiles
// of 64, both operands row-major [row][64 channels] as they sit in memory (coalesced 16-byte loads).
//   bf16: both MFMA operands want 8 consecutive REDUCTION indices (rows) of one channel per lane, i.e. the transpose of what is in
//         LDS -- the attention kernel's V^T situation on both sides.  Tiles are stored as [16-channel block][64 rows][16 channels]
//         sub-tiles and read with ds_read_b64_tr_b16; dy and x use the same row permutation inside a 16-row step
//         ({4g..4g+3, 8+4g..8+4g+3}), which a dot product does not see.  v_mfma_f32_32x32x16_bf16.
//   fp32 (validation mode): v_mfma_f32_32x32x2_f32 takes ONE reduction index per lane, so a lane reads its scalar straight from the
//         row-major tile; exact fp32 products, fp32 accumulation.
constexpr int WG_TILE = 64, WG_ROWS = 64;
constexpr int WG_VS = WG_ROWS * 32 + 128;  // bf16 sub-tile [64 rows][16 ch] + skew (see attn_mfma_kernel)
typedef short ts16x4_v __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

// GATHER (round 5): the weight gradient of a dense 3x3 conv (padding 1, stride 1 | 2) in ONE launch -- grid.z = 9 taps x splits; for tap
// (ky, kx) the x operand of output pixel (b, oy, ox) is the input pixel (oy s + ky - 1, ox s + kx - 1), zeros outside the image (the nine
// shifted, zero-padded copies the host made before: stage1_train.conv3x3_wgrad of round 4).  partial [tap][split][N][K].
struct WgradGather {
  int on, zs, OH, OW, IH, IW, stride;
  // round 6, XCD-aware tile order: gx > 0 = a 1-D launch of 8 ceil(gx gy gz / 8) workgroups; hardware deals consecutive workgroup ids to the 8
  // XCDs in turn, so workgroup L takes logical tile (L % 8) ceil(T / 8) + L / 8 -- every XCD walks a contiguous range of (row split, tap)
  // groups and the gx gy tiles of a group share ONE L2.  With the plain 3-D grid the 64 tiles of the head's 3x3 conv (one tap, one split)
  // were spread over all eight L2s: each fetched the whole dy slab, 5 GB over the fabric per call (1.5 ms at 400 TFLOP/s).
  int gx, gy, gz;
};
// TW (round 6): the tile of dW a workgroup owns, 64 x 64 or (bf16) 128 x 128 -- a wave then owns a 64 x 64 quadrant as 2 x 2 MFMA blocks.  With
// 64 x 64 tiles every 64 channels of dy and x loaded from L2 feed 32 multiply-adds per byte: the wide weights of the last stages and the head's
// 3x3 conv (N, K >= 128, 32 k - 127 k rows) ran at 280 - 300 TFLOP/s on L2 bandwidth (profiles/r06/roofline_stage1_step_b1_b32_end.md: the 3x3
// alone 2.2 ms of a 42 ms step); the larger tile halves the bytes per product and the LDS reads per MFMA.
template <int DT, int TW = 64>
__global__ __launch_bounds__(256) void wgrad_kernel(const typename TElem<DT>::type* __restrict__ dy, int ldy,
                                                    const typename TElem<DT>::type* __restrict__ x, int ldx, int64_t M, int N, int K,
                                                    int64_t rows_per_split, float* __restrict__ partial /* [splits][N][K] */,
                                                    WgradGather gt = WgradGather{0, 0, 0, 0, 0, 0, 0, 0, 0, 0}) {
  // rows per step: 64 in fp32 (two 16.6 KB tiles), 128 in bf16 (round 6: half the barriers per row; a split is any multiple of 64 rows, the
  // tail of a step past the split's end is staged as zeros)
  static_assert(TW == 64 || (TW == 128 && DT == 1), "128 x 128 tiles in bf16 only");
  constexpr int ROWS = DT == 0 ? 64 : 128, VS = ROWS * 32 + 128;
  constexpr int QW = TW / 2, NBLK = QW / 32;   // a wave's quadrant and its 32 x 32 MFMA blocks per side
  // fp32: plain [row][64 ch] (+1 float pad); bf16: TW / 16 sub-tiles per operand
  __shared__ __attribute__((aligned(16))) char sA[DT == 0 ? ROWS * 65 * 4 : (TW / 16) * VS];
  __shared__ __attribute__((aligned(16))) char sB[DT == 0 ? ROWS * 65 * 4 : (TW / 16) * VS];
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (gt.gx > 0) {
    const int T = gt.gx * gt.gy * gt.gz, T8 = (T + 7) >> 3;
    const int M = ((int)blockIdx.x & 7) * T8 + ((int)blockIdx.x >> 3);
    if (M >= T) return;
    bx = M % gt.gx;
    by = (M / gt.gx) % gt.gy;
    bz = M / (gt.gx * gt.gy);
  }
  const int n0 = by * TW, k0 = bx * TW;
  const int zsplit = gt.on ? bz % gt.zs : bz, tap = gt.on ? bz / gt.zs : 0;
  const int64_t r_begin = (int64_t)zsplit * rows_per_split;
  const int64_t r_end = r_begin + rows_per_split < M ? r_begin + rows_per_split : M;
  const int gky = tap / 3 - 1, gkx = tap % 3 - 1;
  // row of the x operand for output row r: the row itself, or (GATHER) the tap's input pixel (-1: outside the image)
  auto xrow = [&](int64_t r) -> int64_t {
    if (!gt.on) return r;
    // 32-bit arithmetic (the host checks M < 2^31): the 64-bit division this replaced is a few hundred instructions, eight times per thread
    // and 128-row step
    const unsigned hw = (unsigned)(gt.OH * gt.OW), r32 = (unsigned)r;
    const unsigned b = r32 / hw;
    const int rem = (int)(r32 - b * hw);
    const int oy = rem / gt.OW, ox = rem - oy * gt.OW;
    const int iy = oy * gt.stride + gky, ix = ox * gt.stride + gkx;
    if (iy < 0 || iy >= gt.IH || ix < 0 || ix >= gt.IW) return -1;
    return ((int64_t)b * gt.IH + iy) * (int64_t)gt.IW + ix;
  };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, g = lane >> 5;
  const int wn = wave >> 1, wk = wave & 1;  // this wave's quadrant: dW rows n0 + QW wn .., columns k0 + QW wk ..
  f32x16_t accs[NBLK][NBLK];
#pragma unroll
  for (int i = 0; i < NBLK; ++i)
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accs[i][j][r] = 0.f;
  f32x16_t& acc = accs[0][0];

  // Round 6: the global loads of row step s + 1 are issued BEFORE the matrix products of step s (register double buffer).  The loop used to
  // be load -> LDS -> barrier -> 4 MFMAs per wave -> barrier: one exposed trip to memory per 64 rows (0.7 ms for the 8 M-row layers of the
  // stem, 6.5 ms of a 55 ms step over 52 calls, profiles/r06/roofline_stage1_step_b1_b32.md); now the trip hides behind the products.
  constexpr int CPRW = TW / 8;                                          // bf16: 16-byte chunks per tile row
  constexpr int NCH = DT == 0 ? (ROWS * 16) / 256 : (ROWS * CPRW) / 256;   // chunks per thread and operand: 4 (fp32) / 4 | 8 (bf16)
  typedef typename std::conditional<DT == 0, float4, uint4>::type chunk_t;
  chunk_t ra[NCH], rb[NCH];
  // GATHER: the x rows of a step's ROWS output rows, computed once per row by the first ROWS threads (round 6: every thread recomputed the
  // row of each of its 4 - 8 chunks -- two integer divisions each, ~5 000 VALU cycles per 128-row step against 1 024 of MFMA: the 3x3 weight
  // gradient of the head ran at 400 TFLOP/s); double-buffered: step s + 1's table is written between the two barriers of step s
  __shared__ int xr_tab[2][ROWS];
  auto fill = [&](int buf, int64_t r0) {
    if (gt.on && tid < ROWS) xr_tab[buf][tid] = r0 + tid < r_end ? (int)xrow(r0 + tid) : -1;
  };
  auto fetch = [&](int64_t r0, int buf) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i;
      if constexpr (DT == 0) {
        const int row = c >> 4, ch = (c & 15) * 4;
        float4 av = make_float4(0.f, 0.f, 0.f, 0.f), bv = av;
        if (r0 + row < r_end) {
          if (n0 + ch < N) av = *reinterpret_cast<const float4*>(dy + (r0 + row) * ldy + n0 + ch);   // N, K multiples of 8
          const int64_t xr = gt.on ? (int64_t)xr_tab[buf][row] : r0 + row;
          if (k0 + ch < K && xr >= 0) bv = *reinterpret_cast<const float4*>(x + xr * ldx + k0 + ch);
        }
        ra[i] = av; rb[i] = bv;
      } else {
        const int row = c / CPRW, ch8 = c % CPRW;
        uint4 av = make_uint4(0u, 0u, 0u, 0u), bv = av;
        if (r0 + row < r_end) {
          if (n0 + ch8 * 8 < N) av = *reinterpret_cast<const uint4*>(dy + (r0 + row) * ldy + n0 + ch8 * 8);
          const int64_t xr = gt.on ? (int64_t)xr_tab[buf][row] : r0 + row;
          if (k0 + ch8 * 8 < K && xr >= 0) bv = *reinterpret_cast<const uint4*>(x + xr * ldx + k0 + ch8 * 8);
        }
        ra[i] = av; rb[i] = bv;
      }
    }
  };
  auto stash = [&]() {   // ---- 64 rows x 64 channels of dy and of x into LDS (zeros past M / N / K) ----
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int c = tid + 256 * i;
      if constexpr (DT == 0) {
        const int row = c >> 4, ch = (c & 15) * 4;
        float* pa = reinterpret_cast<float*>(sA) + row * 65 + ch;
        float* pb = reinterpret_cast<float*>(sB) + row * 65 + ch;
        pa[0] = ra[i].x; pa[1] = ra[i].y; pa[2] = ra[i].z; pa[3] = ra[i].w;
        pb[0] = rb[i].x; pb[1] = rb[i].y; pb[2] = rb[i].z; pb[3] = rb[i].w;
      } else {
        const int row = c / CPRW, ch8 = c % CPRW;
        const int off = (ch8 >> 1) * VS + row * 32 + (ch8 & 1) * 16;
        *reinterpret_cast<uint4*>(sA + off) = ra[i];
        *reinterpret_cast<uint4*>(sB + off) = rb[i];
      }
    }
  };
  if (gt.on) {
    fill(0, r_begin);
    __syncthreads();
  }
  if (r_begin < r_end) fetch(r_begin, 0);
  int step = 0;
  for (int64_t r0 = r_begin; r0 < r_end; r0 += ROWS, ++step) {
    __syncthreads();
    stash();
    if (r0 + ROWS < r_end) fill((step + 1) & 1, r0 + ROWS);
    __syncthreads();
    if (r0 + ROWS < r_end) fetch(r0 + ROWS, (step + 1) & 1);   // in flight under the products below
    if constexpr (DT == 0) {
      const float* fa = reinterpret_cast<const float*>(sA) + wn * 32 + l31;
      const float* fb = reinterpret_cast<const float*>(sB) + wk * 32 + l31;
#pragma unroll 8
      for (int p = 0; p < ROWS; p += 2)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[(p + g) * 65], fb[(p + g) * 65], acc, 0, 0, 0);
    } else {
      typedef __attribute__((address_space(3))) ts16x4_v* lds_v4;
      const unsigned frag = (unsigned)((l31 >> 4) * VS + (4 * g + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8);
      const auto pa = (__attribute__((address_space(3))) char*)sA + wn * (QW / 16) * VS + frag;
      const auto pb = (__attribute__((address_space(3))) char*)sB + wk * (QW / 16) * VS + frag;
#pragma unroll
      for (int s = 0; s < ROWS / 16; ++s) {
        const int off = s * 16 * 32;
        bf16x8_v af[NBLK], bf[NBLK];
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {   // block i = 32 channels = two 16-channel sub-tiles further on
          const uint2 alo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pa + i * 2 * VS + off)));
          const uint2 ahi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pa + i * 2 * VS + off + 8 * 32)));
          const uint2 blo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pb + i * 2 * VS + off)));
          const uint2 bhi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(pb + i * 2 * VS + off + 8 * 32)));
          const u32x4 a4 = {alo.x, alo.y, ahi.x, ahi.y}, b4 = {blo.x, blo.y, bhi.x, bhi.y};
          af[i] = __builtin_bit_cast(bf16x8_v, a4);
          bf[i] = __builtin_bit_cast(bf16x8_v, b4);
        }
#pragma unroll
        for (int i = 0; i < NBLK; ++i)
#pragma unroll
          for (int j = 0; j < NBLK; ++j) accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], accs[i][j], 0, 0, 0);
      }
    }
  }
  // accumulator layout: column = l31 (the K index), rows (r & 3) + 8 (r >> 2) + 4 g (the N index)
  float* out = partial + (int64_t)bz * N * K;
#pragma unroll
  for (int i = 0; i < NBLK; ++i)
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
      const int kc = k0 + wk * QW + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int nr = n0 + wn * QW + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
        if (nr < N && kc < K) out[(int64_t)nr * K + kc] = accs[i][j][r];
      }
    }
}

// Sum of partial[z][i] over the splits in a FIXED order that is not one dependent chain: the splits are dealt to the four waves of the
// workgroup in contiguous quarters, every wave keeps four interleaved partial sums, and the sixteen sums are added in one written-down
// order.  (The one-thread-per-element loop over up to 512 splits ran at one memory latency per few adds: 65 us per call, 5 ms of a
// stage-1 step over its 80 calls -- profiles/r04/stage1_step_kernel_stats.csv.)  Deterministic: the order depends on (splits) only.
__device__ __forceinline__ float fixed_order_sum(const float* __restrict__ p, int z0, int z1, int64_t stride) {
  float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
  int z = z0;
  for (; z + 3 < z1; z += 4) {
    a += p[(int64_t)z * stride];
    b += p[(int64_t)(z + 1) * stride];
    c += p[(int64_t)(z + 2) * stride];
    d += p[(int64_t)(z + 3) * stride];
  }
  for (; z < z1; ++z) a += p[(int64_t)z * stride];
  return (a + b) + (c + d);
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int64_t n, float* __restrict__ out) {
  __shared__ float sq[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + e;
  const int per = (splits + 3) / 4, z0 = q * per, z1 = min(splits, z0 + per);
  sq[q][e] = i < n ? fixed_order_sum(partial + i, z0, z1, n) : 0.f;
  __syncthreads();
  if (q == 0 && i < n) out[i] = (sq[0][e] + sq[1][e]) + (sq[2][e] + sq[3][e]);
}

// the 3x3 conv's nine taps: out[(n K + k) 9 + tap] = sum over the splits of partial[tap][split][n K + k] (fixed order): dw [N][K][3][3]
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const float* __restrict__ partial, int splits, int64_t nk, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int tap = blockIdx.y;
  if (i >= nk) return;
  out[i * 9 + tap] = fixed_order_sum(partial + (int64_t)tap * splits * nk + i, 0, splits, nk);
}

// dbias[N] = sum over the rows of dy: [splits][N] partials then the same reduce kernel
template <int DT>
__global__ __launch_bounds__(256) void colsum_kernel(const typename TElem<DT>::type* __restrict__ dy, int ldy, int64_t M, int N,
                                                     int64_t rows_per_split, float* __restrict__ partial) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split, r1 = r0 + rows_per_split < M ? r0 + rows_per_split : M;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) {
    if constexpr (DT == 0) s += dy[r * ldy + c];
    else s += __uint_as_float((uint32_t)dy[r * ldy + c] << 16);
  }
  partial[(int64_t)blockIdx.y * N + c] = s;
}

// Round 6: the same column sums with 16-byte loads.  The kernel above gives a thread ONE column and walks its rows with 2-byte (bf16) loads:
// 0.3 TB/s on [508 032][256] (profiles/r06/roofline_stage1_step_b1_b32.md: 3.1 ms of a 65 ms step in 19 launches).  Here a workgroup is
// (N / 8 channel groups) x (256 / (N / 8) row lanes); a thread sums 8 channels of every RL-th row of its split, four rows in flight; the row
// lanes meet in LDS in lane order; partial [split][N] as before (same reduce kernel, fixed order: deterministic).  N % 8 == 0, N <= 2048.
template <int DT>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const typename TElem<DT>::type* __restrict__ dy, int64_t M, int N, int64_t rows_per_split,
                                                         float* __restrict__ partial) {
  __shared__ float red[256 * 8];
  const int C8 = N / 8, RL = 256 / C8;
  const int cg = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_split, r1 = r0 + rows_per_split < M ? r0 + rows_per_split : M;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    int64_t r = r0 + rl;
    for (; r + 3 * (int64_t)RL < r1; r += 4 * (int64_t)RL) {
      float u0[8], u1[8], u2[8], u3[8];
      TElem<DT>::load8(dy + (r * C8 + cg) * 8, u0);
      TElem<DT>::load8(dy + ((r + RL) * C8 + cg) * 8, u1);
      TElem<DT>::load8(dy + ((r + 2 * (int64_t)RL) * C8 + cg) * 8, u2);
      TElem<DT>::load8(dy + ((r + 3 * (int64_t)RL) * C8 + cg) * 8, u3);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += (u0[e] + u1[e]) + (u2[e] + u3[e]);
    }
    for (; r < r1; r += RL) {
      float u[8];
      TElem<DT>::load8(dy + (r * C8 + cg) * 8, u);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += u[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(rl * C8 + cg) * 8 + e] = s[e];
  }
  __syncthreads();
  if (rl == 0) {
    float* dst = partial + (int64_t)blockIdx.x * N + cg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int q = 0; q < RL; ++q) t += red[(q * C8 + cg) * 8 + e];
      dst[e] = t;
    }
  }
}

// ---- per-channel scale / shift of an NHWC tensor, the elementwise half of RepViT's RepVGGDW and SqueezeExcite (round 5) ------------------
//   out[b][p][c] = add[b][p][c] + x[b][p][c] * (mul[b * mul_bs + c] + plus_one) + bias[b * bias_bs + c] * bias_scale
// mul / bias fp32, per channel (batch stride 0) or per (image, channel) (batch stride C); add and bias optional.  What it stands for:
//   RepVGGDW forward   (repvit.py:92-93)  s = conv_bn(x) + conv1(x) + x          add = conv_bn(x), mul = conv1.weight, +1, bias = conv1.bias
//   RepVGGDW backward                     dx = dx_conv + ds * (conv1.weight + 1)
//   SqueezeExcite forward (timm)          y = x * gate[b][c]
//   SqueezeExcite backward                dx = dy * gate[b][c] + d_mean[b][c] / HW  (the mean's share of every pixel)
template <int DT>
__global__ __launch_bounds__(256) void channel_scale_kernel(const typename TElem<DT>::type* __restrict__ x, const float* __restrict__ mul, int mul_bs,
                                                            float plus_one, const float* __restrict__ bias, int bias_bs, float bias_scale,
                                                            const typename TElem<DT>::type* __restrict__ add,
                                                            typename TElem<DT>::type* __restrict__ out, int64_t per_image8, int C8, int64_t total8) {
  // round 6: 32-bit index arithmetic whenever the tensor allows it (the two 64-bit divisions per 8 elements were most of the loop), the
  // multipliers / biases as 16-byte loads (element by element they compile to single-dword loads)
  const bool small = total8 < ((int64_t)1 << 31);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (int64_t)gridDim.x * 256) {
    int64_t b;
    int c;
    if (small) {
      const unsigned iu = (unsigned)i;
      b = (mul_bs | bias_bs) ? (int64_t)(iu / (unsigned)per_image8) : 0;
      c = (int)(iu % (unsigned)C8) * 8;
    } else {
      b = i / per_image8;
      c = (int)(i % C8) * 8;
    }
    float v[8], a[8], o[8], m[8], bi[8];
    TElem<DT>::load8(x + i * 8, v);
    if (add) TElem<DT>::load8(add + i * 8, a);
    TElem<0>::load8(mul + b * mul_bs + c, m);
    if (bias) TElem<0>::load8(bias + b * bias_bs + c, bi);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float r = v[e] * (m[e] + plus_one);
      if (add) r += a[e];
      if (bias) r += bi[e] * bias_scale;
      o[e] = r;
    }
    TElem<DT>::store8(out + i * 8, o);
  }
}

// Round 6: the per-channel (not per-image) case -- every residual addition and BatchNorm-less scale of a training step -- without index
// arithmetic in the loop: the grid stride is a multiple of the row length in 8-element groups (256 % C8 == 0), so a thread keeps ONE channel
// group: its multipliers / biases are loaded once, the loop is two 16-byte loads, eight fused operations and a store.  Same expression per
// element as channel_scale_kernel (bit-identical); that kernel spent its time on two 64-bit divisions and 16 scalar loads per 8 elements
// (2.85 TB/s on the 43 residual additions of a B1 step, 2.3 ms).
template <int DT>
__global__ __launch_bounds__(256) void channel_scale_fast_kernel(const typename TElem<DT>::type* __restrict__ x, const float* __restrict__ mul,
                                                                 float plus_one, const float* __restrict__ bias, float bias_scale,
                                                                 const typename TElem<DT>::type* __restrict__ add,
                                                                 typename TElem<DT>::type* __restrict__ out, int C8, unsigned total8) {
  const unsigned i0 = blockIdx.x * 256u + threadIdx.x;
  const int c = (int)(i0 % (unsigned)C8) * 8;
  float m[8], bi[8];
  {  // 16-byte loads: written element by element these compile to single-dword loads, 16 cache lines of the wave each (see bn_map_kernel)
    const float4 m0 = *reinterpret_cast<const float4*>(mul + c), m1 = *reinterpret_cast<const float4*>(mul + c + 4);
    m[0] = m0.x; m[1] = m0.y; m[2] = m0.z; m[3] = m0.w; m[4] = m1.x; m[5] = m1.y; m[6] = m1.z; m[7] = m1.w;
#pragma unroll
    for (int e = 0; e < 8; ++e) bi[e] = 0.f;
    if (bias) {
      const float4 b0 = *reinterpret_cast<const float4*>(bias + c), b1 = *reinterpret_cast<const float4*>(bias + c + 4);
      bi[0] = b0.x; bi[1] = b0.y; bi[2] = b0.z; bi[3] = b0.w; bi[4] = b1.x; bi[5] = b1.y; bi[6] = b1.z; bi[7] = b1.w;
    }
  }
  const unsigned stride = gridDim.x * 256u;   // a multiple of C8: the channel group of a thread never changes
  for (unsigned i = i0; i < total8; i += stride) {
    float v[8], a[8], o[8];
    TElem<DT>::load8(x + (size_t)i * 8, v);
    if (add) TElem<DT>::load8(add + (size_t)i * 8, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float r = v[e] * (m[e] + plus_one);
      if (add) r += a[e];
      if (bias) r += bi[e] * bias_scale;
      o[e] = r;
    }
    TElem<DT>::store8(out + (size_t)i * 8, o);
  }
}

// ---- out[b][c] = scale * sum over the HW pixels of image b of a[b][p][c] * (b2 ? b2[b][p][c] : 1) -----------------------------------------
// SqueezeExcite's global mean (b2 = NULL, scale = 1 / HW), the gradient of its gate (sum of dy * x), and with B = 1 the gradient of
// RepVGGDW's depthwise 1x1 weight (sum over all rows of ds * x).  grid = (splits, B); a workgroup = (256 / C8) row lanes x C8 groups of 8
// channels; row lanes are summed through LDS and the splits by coldot_reduce_kernel, both in a fixed order (deterministic).
template <int DT>
__global__ __launch_bounds__(256) void batched_coldot_kernel(const typename TElem<DT>::type* __restrict__ a,
                                                             const typename TElem<DT>::type* __restrict__ b2, int64_t HW, int C,
                                                             int64_t rows_per_split, float* __restrict__ partial /* [split][B][C] */) {
  __shared__ float red[256 * 8];
  const int C8 = C / 8, RL = 256 / C8;
  const int cg = threadIdx.x % C8, rl = threadIdx.x / C8;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_split, r1 = r0 + rows_per_split < HW ? r0 + rows_per_split : HW;
  const int64_t base = (int64_t)blockIdx.y * HW;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < RL) {
    // round 6: four rows' loads in flight (the walk was one dependent load per step: 7.6 ms of a 64 ms RepViT-M1.1 step in 39 calls,
    // profiles/r06/r06_kernel_stats_stage1_step_repvit_m1_1_b32.csv); rows still accumulate in ascending order
    int64_t r = r0 + rl;
    for (; r + 3 * (int64_t)RL < r1; r += 4 * (int64_t)RL) {
      typename TElem<DT>::vec8 qa[4], qb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        qa[k] = TElem<DT>::loadraw(a + ((base + r + k * (int64_t)RL) * C8 + cg) * 8);
        if (b2) qb[k] = TElem<DT>::loadraw(b2 + ((base + r + k * (int64_t)RL) * C8 + cg) * 8);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float u[8], w[8];
        TElem<DT>::unpack(qa[k], u);
        if (b2) {
          TElem<DT>::unpack(qb[k], w);
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += u[e] * w[e];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += u[e];
        }
      }
    }
    for (; r < r1; r += RL) {
      float u[8], w[8];
      TElem<DT>::load8(a + ((base + r) * C8 + cg) * 8, u);
      if (b2) {
        TElem<DT>::load8(b2 + ((base + r) * C8 + cg) * 8, w);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += u[e] * w[e];
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += u[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[(rl * C8 + cg) * 8 + e] = s[e];
  }
  __syncthreads();
  if (rl == 0) {
    float* dst = partial + ((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * C + cg * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
      for (int q = 0; q < RL; ++q) t += red[(q * C8 + cg) * 8 + e];
      dst[e] = t;
    }
  }
}
__global__ __launch_bounds__(256) void coldot_reduce_kernel(const float* __restrict__ partial, int splits, int64_t n, float scale, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = fixed_order_sum(partial + i, 0, splits, n) * scale;
}
constexpr int COLDOT_SPLITS_MAX = 64;
int coldot_splits(int B, int64_t HW, int C) {
  const int RL = 256 / (C / 8);
  const int64_t want = 1024 / B > 1 ? 1024 / B : 1;                   // >= ~1024 workgroups over the chip when the image allows it
  const int64_t most = (HW + (int64_t)RL * 8 - 1) / ((int64_t)RL * 8);  // a split has at least 8 rows per row lane
  int64_t sp = want < most ? want : most;
  // round 6: the cap is on splits x B (the workspace holds COLDOT_SPLITS_MAX x max(B, 16) partial rows): with B = 1 -- RepVGGDW's depthwise 1x1
  // weight gradient over ALL rows of the batch -- 64 splits were 64 workgroups on 256 CUs
  const int64_t cap = COLDOT_SPLITS_MAX * (int64_t)(B < 16 ? 16 : B) / B;
  if (sp > cap) sp = cap;
  return (int)(sp < 1 ? 1 : sp);
}

// ---- depthwise k x k weight gradient (k = 3 | 5, padding k / 2):
//      dwd[c][kh][kw] = sum_{b, oy, ox} dy[b][oy][ox][c] x[b][oy s + kh - k/2][ox s + kw - k/2][c] ----------------------------------
// grid = (pixel splits, kernel rows): a workgroup owns ONE kernel row kh, a thread = (8-channel group, row lane) accumulates k x 8 sums
// over the output pixels of its split; row lanes and splits are summed in a fixed order.  partial [split][k k][C].
template <int DT, int KS>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const typename TElem<DT>::type* __restrict__ x,
                                                       const typename TElem<DT>::type* __restrict__ dy, int B, int H, int W, int C, int stride,
                                                       float* __restrict__ partial) {
  extern __shared__ float red[];  // [RL][KS][C]
  constexpr int PAD = KS / 2;
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
  const int CG = C >> 3, RL = 256 / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int kh = blockIdx.y;
  // Round 6: a split is a range of output ROWS (b, oy); the input row iy of this workgroup's kernel row is uniform per output row, and a thread
  // walks the row's pixels with its lane stride -- no per-pixel index arithmetic (the first version recovered (b, oy, ox) from a flat 64-bit
  // pixel index with three divisions per pixel and kernel row: 1.0 TB/s on tensors that stream at 3.4, profiles/r06/roofline_stage1_step_*.md).
  // The summation order inside a split changed with it (rows, then lanes): still fixed, repeats stay bit-identical.
  const int64_t nrows = (int64_t)B * OH;
  const int64_t per = (nrows + gridDim.x - 1) / gridDim.x;
  const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < nrows ? q0 + per : nrows;
  float acc[KS][8];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  if (rl < RL)
    for (int64_t q = q0; q < q1; ++q) {
      const int64_t b = q / OH;
      const int oy = (int)(q - b * OH);
      const int iy = oy * stride + kh - PAD;
      if (iy < 0 || iy >= H) continue;
      const typename TElem<DT>::type* dyr = dy + q * OW * (int64_t)C + cg * 8;
      const typename TElem<DT>::type* xr = x + (b * H + iy) * (int64_t)W * C + cg * 8;
      for (int ox = rl; ox < OW; ox += RL) {   // branch-free taps, all loads of a pixel issued before the first use (see dw_wgrad3_kernel)
        typename TElem<DT>::vec8 raw[KS];
        float cv[KS];
        const typename TElem<DT>::vec8 graw = TElem<DT>::loadraw(dyr + (int64_t)ox * C);
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          const int ix = ox * stride + kw - PAD;
          cv[kw] = (ix >= 0 && ix < W) ? 1.f : 0.f;
          raw[kw] = TElem<DT>::loadraw(xr + (int64_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C);
        }
        float g[8];
        TElem<DT>::unpack(graw, g);
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
          float v[8];
          TElem<DT>::unpack(raw[kw], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[kw][e] = fmaf(g[e] * cv[kw], v[e], acc[kw][e]);
        }
      }
    }
  if (rl < RL)
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(rl * KS + t) * C + cg * 8 + e] = acc[t][e];
  __syncthreads();
  for (int i = threadIdx.x; i < KS * C; i += 256) {
    float s = 0.f;
    for (int l = 0; l < RL; ++l) s += red[l * KS * C + i];
    partial[((int64_t)blockIdx.x * KS + kh) * KS * C + i] = s;
  }
}

// Round 6, stride 1 (the 5x5 aggregation convs of LiteMLA): the kernel above with a SLIDING window along the row.  A thread owns a contiguous
// run of its row's pixels instead of every RL-th one, keeps the last KS - 1 input columns in registers and loads U new columns + U dy values
// per step: 2 U loads in flight carrying U pixels, where the strided walk had KS + 1 loads in flight carrying one (0.6 - 0.7 TB/s on
// [32][63][63][384] and [32][32][32][768], latency-bound at a handful of waves per SIMD: profiles/r06/roofline_stage1_step_b1_b32_end2.md).
// Same grid (row splits x kernel rows), same partial layout and lane reduction; the order inside a split is rows, then the lane's pixels.
template <int DT, int KS>
__global__ __launch_bounds__(256) void dw_wgrad_slide_kernel(const typename TElem<DT>::type* __restrict__ x,
                                                             const typename TElem<DT>::type* __restrict__ dy, int B, int H, int W, int C,
                                                             float* __restrict__ partial) {
  extern __shared__ float red[];  // [RL][KS][C]
  constexpr int PAD = KS / 2, U = 4, NW = KS - 1 + U;
  typedef typename TElem<DT>::vec8 raw_t;
  const int CG = C >> 3, RL = 256 / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int kh = blockIdx.y;
  const int seg = (W + RL - 1) / RL, s0 = rl * seg, s1 = s0 + seg < W ? s0 + seg : W;   // this lane's pixels of every row
  const int64_t nrows = (int64_t)B * H;
  const int64_t per = (nrows + gridDim.x - 1) / gridDim.x;
  const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < nrows ? q0 + per : nrows;
  float acc[KS][8];
#pragma unroll
  for (int t = 0; t < KS; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  if (rl < RL && s0 < s1) {
    int b = (int)(q0 / H), oy = (int)(q0 - (int64_t)b * H);
    for (int64_t q = q0; q < q1; ++q, oy = oy + 1 == H ? 0 : oy + 1, b += oy == 0) {
      const int iy = oy + kh - PAD;
      if (iy < 0 || iy >= H) continue;
      const typename TElem<DT>::type* dyr = dy + q * W * (int64_t)C + cg * 8;
      const typename TElem<DT>::type* xr = x + ((int64_t)b * H + iy) * (int64_t)W * C + cg * 8;
      raw_t win[NW];   // columns s - PAD .. of the current step; 0 .. KS - 2 carried from the step before
      float cvw[NW];   // 1 inside the image, 0 for a clamped column
#pragma unroll
      for (int j = 0; j < KS - 1; ++j) {
        const int ix = s0 - PAD + j;
        cvw[j] = (ix >= 0 && ix < W) ? 1.f : 0.f;
        win[j] = TElem<DT>::loadraw(xr + (int64_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C);
      }
      for (int ox = s0; ox < s1; ox += U) {
        raw_t graw[U];
        float gm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ix = ox + u + PAD;
          cvw[KS - 1 + u] = (ix >= 0 && ix < W) ? 1.f : 0.f;
          win[KS - 1 + u] = TElem<DT>::loadraw(xr + (int64_t)(ix >= W ? W - 1 : ix) * C);
          gm[u] = ox + u < s1 ? 1.f : 0.f;
          graw[u] = TElem<DT>::loadraw(dyr + (int64_t)(ox + u < s1 ? ox + u : s1 - 1) * C);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          __builtin_amdgcn_sched_barrier(0);   // one pixel's unpacking at a time: hoisted, all U x KS unpacked columns cost 240 VGPRs
          float g[8];
          TElem<DT>::unpack(graw[u], g);
#pragma unroll
          for (int kw = 0; kw < KS; ++kw) {
            float v[8];
            TElem<DT>::unpack(win[u + kw], v);
            const float m = gm[u] * cvw[u + kw];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[kw][e] = fmaf(g[e] * m, v[e], acc[kw][e]);
          }
        }
#pragma unroll
        for (int j = 0; j < KS - 1; ++j) {
          win[j] = win[U + j];
          cvw[j] = cvw[U + j];
        }
      }
    }
  }
  if (rl < RL)
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[(rl * KS + t) * C + cg * 8 + e] = acc[t][e];
  __syncthreads();
  for (int i = threadIdx.x; i < KS * C; i += 256) {
    float s = 0.f;
    for (int l = 0; l < RL; ++l) s += red[l * KS * C + i];
    partial[((int64_t)blockIdx.x * KS + kh) * KS * C + i] = s;
  }
}

// Round 6, 3x3 only: ALL nine taps in one workgroup.  The kernel above runs one workgroup per kernel row, so dy and x cross the memory system
// three times each (1.0 - 1.1 TB/s of algorithmic bytes = 3.2 TB/s of traffic: it was at the streaming rate of what it asked for); here a
// thread keeps the 9 x 8 sums of its (8-channel group, pixel lane), reads dy once per pixel and the three input rows out of L1 / L2 (the
// neighbouring output row needs two of them again).  partial [split][9][C], the same layout: the finalize kernel is shared.
template <int DT>
__global__ __launch_bounds__(256) void dw_wgrad3_kernel(const typename TElem<DT>::type* __restrict__ x,
                                                        const typename TElem<DT>::type* __restrict__ dy, int B, int H, int W, int C, int stride,
                                                        float* __restrict__ partial) {
  extern __shared__ float red[];  // [RL][C]: the lanes meet tap by tap (8 KB instead of 72: four workgroups per CU -- the walk is latency-bound)
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
  const int CG = C >> 3, RL = 256 / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int64_t nrows = (int64_t)B * OH;
  const int64_t per = (nrows + gridDim.x - 1) / gridDim.x;
  const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < nrows ? q0 + per : nrows;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  if (rl < RL)
    for (int64_t q = q0; q < q1; ++q) {
      const int64_t b = q / OH;
      const int oy = (int)(q - b * OH);
      const typename TElem<DT>::type* dyr = dy + q * OW * (int64_t)C + cg * 8;
      const typename TElem<DT>::type* xb = x + b * H * (int64_t)W * C + cg * 8;
      const int iy0 = oy * stride - 1;
      // Branch-free taps: every tap's address is clamped into the image and its contribution multiplied by 0 / 1.  With `continue`s on the
      // image border every load sat in its own basic block and the nine loads of a pixel were nine serial trips to L2 -- that, not traffic or
      // occupancy, was what held the first versions of this kernel at 1 TB/s.
      float rv[3];
      int iyc[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int iy = iy0 + kh;
        rv[kh] = (iy >= 0 && iy < H) ? 1.f : 0.f;
        iyc[kh] = iy < 0 ? 0 : (iy >= H ? H - 1 : iy);
      }
      for (int ox = rl; ox < OW; ox += RL) {
        typename TElem<DT>::vec8 raw[9];
        float cv[3];
        int ixc[3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ix = ox * stride + kw - 1;
          cv[kw] = (ix >= 0 && ix < W) ? 1.f : 0.f;
          ixc[kw] = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        }
        const typename TElem<DT>::vec8 graw = TElem<DT>::loadraw(dyr + (int64_t)ox * C);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) raw[kh * 3 + kw] = TElem<DT>::loadraw(xb + ((int64_t)iyc[kh] * W + ixc[kw]) * C);
        float g[8];
        TElem<DT>::unpack(graw, g);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            __builtin_amdgcn_sched_barrier(0);   // one tap's unpacking at a time (hoisted, the nine unpacked taps cost 234 VGPRs in bf16: two waves per SIMD)
            float v[8];
            TElem<DT>::unpack(raw[kh * 3 + kw], v);
            const float m = rv[kh] * cv[kw];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[kh * 3 + kw][e] = fmaf(g[e] * m, v[e], acc[kh * 3 + kw][e]);
          }
      }
    }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (rl < RL)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[rl * C + cg * 8 + e] = acc[t][e];
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
      float s = 0.f;
      for (int l = 0; l < RL; ++l) s += red[l * C + i];
      partial[((int64_t)blockIdx.x * 9 + t) * C + i] = s;
    }
    __syncthreads();
  }
}

// Round 6, 3x3 stride 1: dw_wgrad3_kernel with the sliding window of dw_wgrad_slide_kernel -- a thread owns a contiguous run of the row's pixels,
// keeps the last two input columns (three rows each) in registers and loads U new columns + U dy values per step: 4 U loads in flight carrying
// U pixels where the strided walk had 10 carrying one.  Same partial layout [split][9][C], same tap-by-tap lane reduction.
template <int DT>
__global__ __launch_bounds__(256) void dw_wgrad3_slide_kernel(const typename TElem<DT>::type* __restrict__ x,
                                                              const typename TElem<DT>::type* __restrict__ dy, int B, int H, int W, int C,
                                                              float* __restrict__ partial) {
  extern __shared__ float red[];  // [RL][C]
  constexpr int U = 2, NW = 2 + U;
  typedef typename TElem<DT>::vec8 raw_t;
  const int CG = C >> 3, RL = 256 / CG;
  const int cg = threadIdx.x % CG, rl = threadIdx.x / CG;
  const int seg = (W + RL - 1) / RL, s0 = rl * seg, s1 = s0 + seg < W ? s0 + seg : W;
  const int64_t nrows = (int64_t)B * H;
  const int64_t per = (nrows + gridDim.x - 1) / gridDim.x;
  const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < nrows ? q0 + per : nrows;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
  if (rl < RL && s0 < s1) {
    int b = (int)(q0 / H), oy = (int)(q0 - (int64_t)b * H);
    for (int64_t q = q0; q < q1; ++q, oy = oy + 1 == H ? 0 : oy + 1, b += oy == 0) {
      const typename TElem<DT>::type* dyr = dy + q * W * (int64_t)C + cg * 8;
      const typename TElem<DT>::type* xb = x + (int64_t)b * H * (int64_t)W * C + cg * 8;
      float rv[3];
      const typename TElem<DT>::type* xrow[3];
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int iy = oy - 1 + kh;
        rv[kh] = (iy >= 0 && iy < H) ? 1.f : 0.f;
        xrow[kh] = xb + (int64_t)(iy < 0 ? 0 : (iy >= H ? H - 1 : iy)) * W * C;
      }
      raw_t win[3][NW];
      float cvw[NW];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ix = s0 - 1 + j;
        cvw[j] = (ix >= 0 && ix < W) ? 1.f : 0.f;
        const int64_t off = (int64_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * C;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) win[kh][j] = TElem<DT>::loadraw(xrow[kh] + off);
      }
      for (int ox = s0; ox < s1; ox += U) {
        raw_t graw[U];
        float gm[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ix = ox + u + 1;
          cvw[2 + u] = ix < W ? 1.f : 0.f;
          const int64_t off = (int64_t)(ix >= W ? W - 1 : ix) * C;
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) win[kh][2 + u] = TElem<DT>::loadraw(xrow[kh] + off);
          gm[u] = ox + u < s1 ? 1.f : 0.f;
          graw[u] = TElem<DT>::loadraw(dyr + (int64_t)(ox + u < s1 ? ox + u : s1 - 1) * C);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          float g[8];
          TElem<DT>::unpack(graw[u], g);
#pragma unroll
          for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
              __builtin_amdgcn_sched_barrier(0);   // one tap's unpacking at a time
              float v[8];
              TElem<DT>::unpack(win[kh][u + kw], v);
              const float m = gm[u] * rv[kh] * cvw[u + kw];
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[kh * 3 + kw][e] = fmaf(g[e] * m, v[e], acc[kh * 3 + kw][e]);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          cvw[j] = cvw[U + j];
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) win[kh][j] = win[kh][U + j];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if (rl < RL)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[rl * C + cg * 8 + e] = acc[t][e];
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
      float s = 0.f;
      for (int l = 0; l < RL; ++l) s += red[l * C + i];
      partial[((int64_t)blockIdx.x * 9 + t) * C + i] = s;
    }
    __syncthreads();
  }
}

// depthwise k x k DATA gradient (k = 3 | 5), stride 1 | 2, padding k / 2: dx[b][iy][ix][c] = sum over the taps (kh, kw) whose output
// pixel oy = (iy + k/2 - kh) / s, ox = (ix + k/2 - kw) / s exists (divisible, in range) of dy[b][oy][ox][c] w[c][kh][kw].  One thread per
// input pixel and 8-channel group; w fp32 on the device in PyTorch's [C][1][k][k] layout (where a training engine keeps its weights).
template <int DT, int KS>
__global__ __launch_bounds__(256) void dw_dgrad_kernel(const typename TElem<DT>::type* __restrict__ dy, const float* __restrict__ w,
                                                       typename TElem<DT>::type* __restrict__ dx, int B, int H, int W, int C, int stride) {
  constexpr int PAD = KS / 2;
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
  const int CG = C >> 3;
  const int64_t total = (int64_t)B * H * W * CG;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int cg = (int)(i % CG);
    const int64_t px = i / CG;
    const int ix = (int)(px % W), iy = (int)((px / W) % H);
    const int64_t b = px / ((int64_t)W * H);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
      const int ty = iy + PAD - kh;
      if (ty < 0 || ty % stride) continue;
      const int oy = ty / stride;
      if (oy >= OH) continue;
#pragma unroll
      for (int kw = 0; kw < KS; ++kw) {
        const int tx = ix + PAD - kw;
        if (tx < 0 || tx % stride) continue;
        const int ox = tx / stride;
        if (ox >= OW) continue;
        float g[8];
        TElem<DT>::load8(dy + ((b * OH + oy) * (int64_t)OW + ox) * C + cg * 8, g);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = fmaf(g[e], w[(cg * 8 + e) * KS * KS + kh * KS + kw], acc[e]);
      }
    }
    TElem<DT>::store8(dx + px * C + cg * 8, acc);
  }
}

// Round 6: the stride-2 3x3 data gradient (the transposed convolution behind the four stride-2 MBConvs) with the weights in LDS.  The generic
// kernel above fetched its 8 channels' taps as 72 scalar global loads per thread -- 0.6 TB/s of algorithmic bytes on tensors that stream at
// 3.4 TB/s through the stride-1 path (profiles/r06/roofline_stage1_step_b1_b32.md: 4.1 ms of a 65 ms step in four launches).  Here a
// workgroup stages w as [tap][C] once (two 16-byte LDS reads per tap and thread, conflict-free: consecutive threads = consecutive channel
// groups) and every thread walks input pixels; an input pixel (iy, ix) receives the taps whose parity matches: 1, 2 or 4 of the nine.
template <int DT>
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const typename TElem<DT>::type* __restrict__ dy, const float* __restrict__ w,
                                                          typename TElem<DT>::type* __restrict__ dx, int B, int H, int W, int C) {
  extern __shared__ float sw[];   // [9][C]
  for (int i = threadIdx.x; i < 9 * C; i += 256) sw[(i % 9) * C + i / 9] = w[i];
  __syncthreads();
  const int OH = (H + 1) / 2, OW = (W + 1) / 2;
  const unsigned CG = (unsigned)C >> 3;
  // one workgroup = one input row (b, iy): the row's taps (kh, oy) are workgroup-uniform, the per-item index needs ONE 32-bit division
  // (the first version of this kernel, like the generic one, spent four 64-bit divisions per item on (b, iy, ix, cg))
  for (unsigned row = blockIdx.x; row < (unsigned)(B * H); row += gridDim.x) {
    const unsigned b = row / (unsigned)H, iy = row - b * (unsigned)H;
    const int nkh = (iy & 1) ? 2 : 1;
    int khs[2], oys[2];
    for (int a = 0; a < 2; ++a) {
      khs[a] = (iy & 1) ? 2 * a : 1;
      oys[a] = ((int)iy + 1 - khs[a]) >> 1;
    }
    const typename TElem<DT>::type* dyb = dy + (int64_t)b * OH * OW * C;
    typename TElem<DT>::type* dxr = dx + (int64_t)row * W * C;
    // the row's (up to) two kernel rows, clamped and masked: uniform over the workgroup
    float rm[2];
    int oyc[2];
    for (int a = 0; a < 2; ++a) {
      rm[a] = (a < nkh && oys[a] < OH) ? 1.f : 0.f;
      oyc[a] = oys[a] < OH ? (oys[a] < 0 ? 0 : oys[a]) : OH - 1;
    }
    for (unsigned j = threadIdx.x; j < (unsigned)W * CG; j += 256) {
      const unsigned ix = j / CG, cg = j - ix * CG;
      // (up to) two kernel columns of this pixel, clamped and masked: four loads issued together, a parity that has one tap per axis re-reads
      // a cached line with weight 0 (with `continue`s every load was its own trip to L2)
      const int odd = ix & 1;
      int kws[2], oxc[2];
      float cm[2];
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        kws[c2] = odd ? 2 * c2 : 1;
        const int ox = ((int)ix + 1 - kws[c2]) >> 1;
        cm[c2] = ((odd || c2 == 0) && ox < OW) ? 1.f : 0.f;
        oxc[c2] = ox < OW ? ox : OW - 1;
      }
      typename TElem<DT>::vec8 raw[4];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) raw[a * 2 + c2] = TElem<DT>::loadraw(dyb + ((int64_t)oyc[a] * OW + oxc[c2]) * C + cg * 8);
      float acc[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
          float g[8];
          TElem<DT>::unpack(raw[a * 2 + c2], g);
          const float m = rm[a] * cm[c2];
          const float* wp = sw + (khs[a] * 3 + kws[c2]) * C + cg * 8;
          const float4 w0 = *reinterpret_cast<const float4*>(wp), w1 = *reinterpret_cast<const float4*>(wp + 4);
          acc[0] = fmaf(g[0] * m, w0.x, acc[0]); acc[1] = fmaf(g[1] * m, w0.y, acc[1]); acc[2] = fmaf(g[2] * m, w0.z, acc[2]); acc[3] = fmaf(g[3] * m, w0.w, acc[3]);
          acc[4] = fmaf(g[4] * m, w1.x, acc[4]); acc[5] = fmaf(g[5] * m, w1.y, acc[5]); acc[6] = fmaf(g[6] * m, w1.z, acc[6]); acc[7] = fmaf(g[7] * m, w1.w, acc[7]);
        }
      TElem<DT>::store8(dxr + (int64_t)j * 8, acc);
    }
  }
}

// [k k][C] (tap-major) -> PyTorch's [C][1][k][k]
__global__ __launch_bounds__(256) void dw_wgrad_finalize_kernel(const float* __restrict__ partial, int splits, int C, int taps, float* __restrict__ out) {
  __shared__ float sq[4][64];
  const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e, n = taps * C;
  const int per = (splits + 3) / 4, z0 = q * per, z1 = min(splits, z0 + per);
  sq[q][e] = i < n ? fixed_order_sum(partial + i, z0, z1, n) : 0.f;
  __syncthreads();
  if (q == 0 && i < n) {
    const int t = i / C, c = i - t * C;
    out[c * taps + t] = (sq[0][e] + sq[1][e]) + (sq[2][e] + sq[3][e]);
  }
}

// ---- LiteMLA's ReLU linear attention, backward (backbones/efficientvit/nn/ops.py:584-621 relu_linear_att) ---------------------------
// Per image b and head group g (the 3 DIM channels [q | k | v] of the multi-scale qkv tensor ms [B][N][G 3 DIM]), with Qr = relu(Q),
// Kr = relu(K), Vp = [V; 1]:   S = Vp Kr^T  ((DIM+1) x DIM),  O = S Qr  ((DIM+1) x N),  Y = O[:DIM] / (O[DIM] + eps).
// Given dY:   dO = [dY / D ;  -sum_c dY_c Y_c / D],  D = O[DIM] + eps
//             dS = dO Qr^T,   dQr = S^T dO,   dVp = dS Kr  (dV = its first DIM rows),   dKr = dS^T Vp,   dQ = dQr [Q > 0],  dK = dKr [K > 0].
// One workgroup per (b, g) walks the N tokens three times (S, then dS, then the per-token gradients): the two small matrices live in
// LDS, nothing is reduced across workgroups, so the result is deterministic.  fp32 arithmetic on fp32 / bf16 tensors (the reference
// leaves autocast for this function).  Also writes Y when `y` is given (the forward for free).
template <int DT, int DIM>
__global__ __launch_bounds__(256) void mla_backward_kernel(const typename TElem<DT>::type* __restrict__ ms,
                                                           const typename TElem<DT>::type* __restrict__ dout,
                                                           typename TElem<DT>::type* __restrict__ dms,
                                                           typename TElem<DT>::type* __restrict__ y, int N, int G, float eps) {
  constexpr int TT = 64, D1 = DIM + 1, SE = D1 * DIM;  // tokens per tile, rows of S, elements of S
  __shared__ float tq[TT][DIM + 1], tk[TT][DIM + 1], tv[TT][D1 + 1];  // relu(q), relu(k), [v; 1] of the tile (+1: bank skew)
  __shared__ float tdo[TT][D1 + 1];                                   // dO of the tile
  __shared__ float S[SE], dS[SE];
  typedef typename TElem<DT>::type T;
  const int g = blockIdx.x % G;
  const int64_t b = blockIdx.x / G;
  const int tid = threadIdx.x;
  const int C3 = G * 3 * DIM, CO = G * DIM;
  const T* base = ms + b * N * (int64_t)C3 + g * 3 * DIM;
  auto ldf = [](const T* p) -> float {
    if constexpr (DT == 0) return *p; else return __uint_as_float((uint32_t)*p << 16);
  };
  auto stage = [&](int n0) {  // tile of tokens n0 .. n0 + TT - 1 (zeros past N: they add nothing to S / dS)
    for (int i = tid; i < TT * 3 * DIM; i += 256) {
      const int n = i / (3 * DIM), c = i - n * 3 * DIM;
      float v = 0.f;
      if (n0 + n < N) v = ldf(base + (int64_t)(n0 + n) * C3 + c);
      if (c < DIM) tq[n][c] = v > 0.f ? v : 0.f;
      else if (c < 2 * DIM) tk[n][c - DIM] = v > 0.f ? v : 0.f;
      else tv[n][c - 2 * DIM] = v;
    }
    for (int n = tid; n < TT; n += 256) tv[n][DIM] = n0 + n < N ? 1.f : 0.f;
  };
  // acc[j] += sum over the tile of X[n][a] Z[n][c]  for this thread's elements e = tid + 256 j = a DIM + c
  constexpr int NJ = (SE + 255) / 256;
  auto outer = [&](const float (*X)[D1 + 1], const float (*Z)[DIM + 1], float* acc) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int e = tid + 256 * j;
      if (e < SE) {
        const int a = e / DIM, c = e - a * DIM;
        float s_ = acc[j];
        for (int n = 0; n < TT; ++n) s_ = fmaf(X[n][a], Z[n][c], s_);
        acc[j] = s_;
      }
    }
  };
  // per-token dO from S (in LDS) and dY; also returns D and writes Y
  auto token_dO = [&](int n0, int n, float* dO) {
    float O[D1];
#pragma unroll
    for (int a = 0; a < D1; ++a) {
      float s_ = 0.f;
#pragma unroll
      for (int c = 0; c < DIM; ++c) s_ = fmaf(S[a * DIM + c], tq[n][c], s_);
      O[a] = s_;
    }
    const float D = O[DIM] + eps, inv = 1.f / D;
    float dD = 0.f;
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const float yv = O[c] * inv;
      const float dyv = ldf(dout + ((b * N + n0 + n) * (int64_t)CO) + g * DIM + c);
      dO[c] = dyv * inv;
      dD = fmaf(-dyv, yv, dD);
      if (y) {
        if constexpr (DT == 0) y[(b * N + n0 + n) * (int64_t)CO + g * DIM + c] = yv;
        else y[(b * N + n0 + n) * (int64_t)CO + g * DIM + c] = f32_to_bf16(yv);
      }
    }
    dO[DIM] = dD * inv;
  };

  // ---- pass 1: S = Vp Kr^T ----
  float acc[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
  for (int n0 = 0; n0 < N; n0 += TT) {
    __syncthreads();
    stage(n0);
    __syncthreads();
    outer(tv, tk, acc);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (tid + 256 * j < SE) S[tid + 256 * j] = acc[j];
  // ---- pass 2: dS = dO Qr^T ----
#pragma unroll
  for (int j = 0; j < NJ; ++j) acc[j] = 0.f;
  for (int n0 = 0; n0 < N; n0 += TT) {
    __syncthreads();
    stage(n0);
    __syncthreads();
    if (tid < TT) {
      float dO[D1];
#pragma unroll
      for (int a = 0; a < D1; ++a) dO[a] = 0.f;
      if (n0 + tid < N) token_dO(n0, tid, dO);
#pragma unroll
      for (int a = 0; a < D1; ++a) tdo[tid][a] = dO[a];
    }
    __syncthreads();
    outer(tdo, tq, acc);
  }
#pragma unroll
  for (int j = 0; j < NJ; ++j)
    if (tid + 256 * j < SE) dS[tid + 256 * j] = acc[j];
  // ---- pass 3: per-token gradients ----
  for (int n0 = 0; n0 < N; n0 += TT) {
    __syncthreads();
    stage(n0);
    __syncthreads();
    if (tid < TT && n0 + tid < N) {
      const int n = tid;
      float dO[D1];
      token_dO(n0, n, dO);
      T* o = dms + (b * N + n0 + n) * (int64_t)C3 + g * 3 * DIM;
      auto stf = [&](int c, float v) {
        if constexpr (DT == 0) o[c] = v; else o[c] = f32_to_bf16(v);
      };
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        float dq = 0.f, dk = 0.f, dv = 0.f;
#pragma unroll
        for (int a = 0; a < D1; ++a) {
          dq = fmaf(S[a * DIM + c], dO[a], dq);        // dQr = S^T dO
          dk = fmaf(dS[a * DIM + c], tv[n][a], dk);    // dKr = dS^T Vp
        }
#pragma unroll
        for (int c2 = 0; c2 < DIM; ++c2) dv = fmaf(dS[c * DIM + c2], tk[n][c2], dv);  // dV = (dS Kr)[:DIM], row c
        stf(c, tq[n][c] > 0.f ? dq : 0.f);
        stf(DIM + c, tk[n][c] > 0.f ? dk : 0.f);
        stf(2 * DIM + c, dv);
      }
    }
  }
}

// Round 5: the same three passes with the TOKENS split over P workgroups per (image, head group) -- B x G = 128 - 256 workgroups walking
// up to 3969 tokens three times were 24 % of a training step (profiles/r04/stage1_step_kernel_stats.csv).  MODE 0 writes the partial S of
// its token range, MODE 1 (with the reduced S) the partial dS and Y, MODE 2 (with S and dS) the per-token gradients; the partial
// matrices are summed in range order by mla_backward_reduce_kernel: deterministic, same arithmetic per token as the one-workgroup form.
//
// Round 6 (second form).  The four modes were 6.7 ms of a 50.7 ms B1 batch-32 step against 0.33 ms of HBM time
// (profiles/r06/roofline_stage1_step_b1_b32_end.md): every multiply-add fetched both operands from LDS (S / dS element + token element: the
// LDS pipe, not the VALU, set the time) and every global access was a 2-byte element with the token as the lane (64 cache lines per
// instruction).  Now
//   - S and dS come from the workspace through SCALAR loads: the thread layout is (token = lane, channel quarter = wave), so every S / dS
//     index is wave-uniform -- an SGPR operand of the multiply-add, no LDS read;
//   - a token's vectors (relu(q), dO, [v; 1], relu(k)) are read from LDS once into registers, not once per output channel;
//   - the S / dS partial sums (outer products over the tile's tokens) run on the fp32 matrix unit: v_mfma_f32_16x16x4f32, each wave a
//     quarter of the tile's tokens, the extra row (sum of relu(k) / of dD relu(q)) beside it on the VALU; the four
//     waves' accumulators are added in wave order once at the end (deterministic; fp32 MFMA multiplies and adds in full precision);
//   - the tile is staged with 16-byte loads, dY / Y / the gradients move as 8- or 16-byte pieces (a thread's own channels are contiguous).
// Per token the operations and their order are the ones of the first form; only the S / dS sums associate differently (four token
// quarters per tile instead of one chain).
template <int DT, int NE> struct MlaVec;   // NE consecutive elements of a row, 8 / 16 / 32 bytes
template <> struct MlaVec<1, 4> {
  static __device__ inline void ld(const uint16_t* p, float* v) {
    const uint2 q = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u); v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
  }
  static __device__ inline void st(uint16_t* p, const float* v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
};
template <> struct MlaVec<1, 8> {
  static __device__ inline void ld(const uint16_t* p, float* v) { TElem<1>::load8(p, v); }
  static __device__ inline void st(uint16_t* p, const float* v) { TElem<1>::store8(p, v); }
};
template <> struct MlaVec<0, 4> {
  static __device__ inline void ld(const float* p, float* v) { const float4 q = *reinterpret_cast<const float4*>(p); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
  static __device__ inline void st(float* p, const float* v) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct MlaVec<0, 8> {
  static __device__ inline void ld(const float* p, float* v) { TElem<0>::load8(p, v); }
  static __device__ inline void st(float* p, const float* v) { TElem<0>::store8(p, v); }
};

template <int DT, int DIM, int MODE>
__global__ __launch_bounds__(256) void mla_backward_part_kernel(const typename TElem<DT>::type* __restrict__ ms,
                                                           const typename TElem<DT>::type* __restrict__ dout,
                                                           typename TElem<DT>::type* __restrict__ dms,
                                                           typename TElem<DT>::type* __restrict__ y, int N, int G, float eps, int P, int span,
                                                                const float* __restrict__ Sg, const float* __restrict__ dSg, float* __restrict__ part,
                                                                const typename TElem<DT>::type* __restrict__ ms1 = nullptr,
                                                                typename TElem<DT>::type* __restrict__ dms1 = nullptr) {
  // with `ms1` the multi-scale tensor arrives as its two halves -- head groups 0 .. G/2 - 1 in `ms` (the qkv conv's output), G/2 .. G - 1
  // in `ms1` (the aggregated scale), each [B][N][G/2 * 3 * DIM] -- and the gradient leaves the same way (`dms`, `dms1`): the torch.cat in front of
  // the forward and the two slice copies behind the backward (1.1 ms of a B1 batch-32 step, the ATen kernels of its trace) are gone
  constexpr int TT = 64, D1 = DIM + 1, SE = D1 * DIM;  // tokens per tile, rows of S, elements of S
  constexpr int NB = DIM / 16, CPT = DIM / 4;          // 16 x 16 blocks of S per side; output channels per thread
  __shared__ float tq[TT][DIM + 1], tk[TT][DIM + 1], tv[TT][D1 + 1];  // relu(q), relu(k), [v; 1] of the tile (+1: bank skew)
  __shared__ float tdo[TT][D1 + 1];                                   // dO of the tile
  __shared__ float pdd[4][TT];                                         // the four shares of a token's dD (token_dO)
  __shared__ float wred[(MODE == 0 || MODE == 1) ? 4 : 1][(DIM + 4) * DIM];   // the waves' partial S / dS: DIM rows + 4 shares of the extra row
  typedef typename TElem<DT>::type T;
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const int g = blockIdx.x % G;
  const int64_t b = blockIdx.x / G;
  const int nb = blockIdx.y * span, ne = min(N, nb + span);   // this workgroup's token range
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // the wave = the channel quarter: uniform, so S / dS indices built on it are scalar
  const int C3 = G * 3 * DIM, CO = G * DIM;
  const bool hi = ms1 != nullptr && g >= G / 2;          // this group lives in the second tensor
  const int RS = ms1 ? C3 / 2 : C3;                      // row stride of the tensor(s)
  const int gq = hi ? g - G / 2 : g;                     // group index inside its tensor
  const T* base = (hi ? ms1 : ms) + b * N * (int64_t)RS + gq * 3 * DIM;
  const float* Sw = Sg + (int64_t)blockIdx.x * SE;       // read with wave-uniform indices only
  const float* dSw = dSg + (int64_t)blockIdx.x * SE;
  constexpr int EPC = DT == 0 ? 4 : 8, CPR = 3 * DIM / EPC;   // elements per 16-byte chunk, chunks per token row
  auto stage = [&](int n0) {  // tile of tokens n0 .. n0 + TT - 1 (zeros past the range: they add nothing to S / dS)
    for (int i = tid; i < TT * CPR; i += 256) {
      const int n = i / CPR, c0 = (i - n * CPR) * EPC;
      float v[EPC];
#pragma unroll
      for (int e = 0; e < EPC; ++e) v[e] = 0.f;
      if (n0 + n < ne) MlaVec<DT, EPC>::ld(base + (int64_t)(n0 + n) * RS + c0, v);
      if (c0 < DIM) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) tq[n][c0 + e] = v[e] > 0.f ? v[e] : 0.f;
      } else if (c0 < 2 * DIM) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) tk[n][c0 - DIM + e] = v[e] > 0.f ? v[e] : 0.f;
      } else {
#pragma unroll
        for (int e = 0; e < EPC; ++e) tv[n][c0 - 2 * DIM + e] = v[e];
      }
    }
    if (tid < TT) tv[tid][DIM] = n0 + tid < ne ? 1.f : 0.f;
  };
  // acc[ab][cb] (rows 16 ab .., columns 16 cb ..) += sum over this wave's 16 tokens of X[n][a] Z[n][c];  ext[cb] likewise for the extra row
  // a = DIM, this lane's share (tokens 4 j + lane / 16 of the wave's sixteen)
  f32x4_t acc[NB][NB];
  float ext[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    ext[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  auto outer = [&](const float (*X)[D1 + 1], const float (*Z)[DIM + 1]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = wv * 16 + 4 * j + (lane >> 4);
      float av[NB], bv[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        av[i] = X[n][16 * i + (lane & 15)];
        bv[i] = Z[n][16 * i + (lane & 15)];
      }
      const float xl = X[n][DIM];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        ext[i] = fmaf(xl, bv[i], ext[i]);
#pragma unroll
        for (int k = 0; k < NB; ++k) acc[i][k] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[k], acc[i][k], 0, 0, 0);
      }
    }
  };
  auto flush = [&](float* mine) {   // the four waves' sums, added in wave order
    if constexpr (MODE > 1) return;
    else {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      wred[wv][(DIM + (lane >> 4)) * DIM + 16 * i + (lane & 15)] = ext[i];
#pragma unroll
      for (int k = 0; k < NB; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) wred[wv][(16 * i + 4 * (lane >> 4) + r) * DIM + 16 * k + (lane & 15)] = acc[i][k][r];
    }
    __syncthreads();
    for (int e = tid; e < SE; e += 256) {
      float s_;
      if (e < DIM * DIM) s_ = (wred[0][e] + wred[1][e]) + (wred[2][e] + wred[3][e]);
      else {
        s_ = 0.f;
        for (int w = 0; w < 4; ++w)
          for (int t = 0; t < 4; ++t) s_ += wred[w][(DIM + t) * DIM + (e - DIM * DIM)];
      }
      mine[e] = s_;
    }
    }
  };
  // dO of token n (from S and dY) -> tdo[n][0 .. DIM); writes Y in MODE 1.  Four threads per token (thread = (n = lane, quarter = wave)):
  // every one recomputes D (DIM multiply-adds), handles the DIM / 4 channels a of its quarter and leaves its share of dD in pdd[wave][n];
  // token_dD adds the four shares in quarter order (deterministic).
  auto token_dO = [&](int n0, int n) {
    float qv[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) qv[c] = tq[n][c];
    float od = 0.f;
#pragma unroll
    for (int c = 0; c < DIM; ++c) od = fmaf(Sw[DIM * DIM + c], qv[c], od);
    const float D = od + eps, inv = 1.f / D;
    const int64_t at = (b * N + n0 + n) * (int64_t)CO + g * DIM + wv * CPT;
    float dyv[CPT], yv[CPT];
    MlaVec<DT, CPT>::ld(dout + at, dyv);
    float dD = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int a = wv * CPT + k;
      float s_ = 0.f;
#pragma unroll
      for (int c = 0; c < DIM; ++c) s_ = fmaf(Sw[a * DIM + c], qv[c], s_);
      yv[k] = s_ * inv;
      tdo[n][a] = dyv[k] * inv;
      dD = fmaf(-dyv[k], yv[k], dD);
    }
    if (MODE == 1 && y) MlaVec<DT, CPT>::st(y + at, yv);
    pdd[wv][n] = dD * inv;
  };
  auto token_dD = [&](int n) { tdo[n][DIM] = (pdd[0][n] + pdd[1][n]) + (pdd[2][n] + pdd[3][n]); };

  float* mine = part + ((int64_t)blockIdx.x * P + blockIdx.y) * SE;
  if constexpr (MODE == 0) {   // ---- S partial = Vp Kr^T over this range ----
    for (int n0 = nb; n0 < ne; n0 += TT) {
      __syncthreads();
      stage(n0);
      __syncthreads();
      outer(tv, tk);
    }
    flush(mine);
    return;
  }
  if constexpr (MODE == 3) {   // ---- forward only: Y = O[:DIM] / (O[DIM] + eps) of this range (dout is not read) ----
    for (int n0 = nb; n0 < ne; n0 += TT) {
      __syncthreads();
      stage(n0);
      __syncthreads();
      const int n = lane;
      if (n0 + n < ne) {
        float qv[DIM], yv[CPT];
#pragma unroll
        for (int c = 0; c < DIM; ++c) qv[c] = tq[n][c];
        float od = 0.f;
#pragma unroll
        for (int c = 0; c < DIM; ++c) od = fmaf(Sw[DIM * DIM + c], qv[c], od);
        const float inv = 1.f / (od + eps);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
          const int a = wv * CPT + k;
          float s_ = 0.f;
#pragma unroll
          for (int c = 0; c < DIM; ++c) s_ = fmaf(Sw[a * DIM + c], qv[c], s_);
          yv[k] = s_ * inv;
        }
        MlaVec<DT, CPT>::st(y + (b * N + n0 + n) * (int64_t)CO + g * DIM + wv * CPT, yv);
      }
    }
    return;
  }
  if constexpr (MODE == 1) {   // ---- dS partial = dO Qr^T over this range (also writes Y) ----
    for (int n0 = nb; n0 < ne; n0 += TT) {
      __syncthreads();
      stage(n0);
      __syncthreads();
      {
        const int n = lane;
        if (n0 + n < ne) token_dO(n0, n);
        else {
          for (int k = 0; k < CPT; ++k) tdo[n][wv * CPT + k] = 0.f;
          pdd[wv][n] = 0.f;
        }
      }
      __syncthreads();
      if (tid < TT) token_dD(tid);
      __syncthreads();
      outer(tdo, tq);
    }
    flush(mine);
    return;
  }
  if constexpr (MODE == 2) {
    // ---- per-token gradients of this range: the token's dO first (into LDS, four threads per token), then the gradients with the same
    // thread layout; a thread's DIM / 4 channels of dq, dk, dv are contiguous in the row: three vector stores ----
    for (int n0 = nb; n0 < ne; n0 += TT) {
      __syncthreads();
      stage(n0);
      __syncthreads();
      const int n = lane, c0 = wv * CPT;
      if (n0 + n < ne) token_dO(n0, n);
      __syncthreads();
      if (tid < TT && n0 + tid < ne) token_dD(tid);
      __syncthreads();
      if (n0 + n < ne) {
        float dov[D1], vv[D1], kv[DIM], oq[CPT], ok[CPT], ov[CPT];
#pragma unroll
        for (int a = 0; a < D1; ++a) {
          dov[a] = tdo[n][a];
          vv[a] = tv[n][a];
        }
#pragma unroll
        for (int c = 0; c < DIM; ++c) kv[c] = tk[n][c];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
          const int c = c0 + k;
          float dq = 0.f, dk = 0.f, dv = 0.f;
#pragma unroll
          for (int a = 0; a < D1; ++a) {
            dq = fmaf(Sw[a * DIM + c], dov[a], dq);    // dQr = S^T dO
            dk = fmaf(dSw[a * DIM + c], vv[a], dk);    // dKr = dS^T Vp
          }
#pragma unroll
          for (int c2 = 0; c2 < DIM; ++c2) dv = fmaf(dSw[c * DIM + c2], kv[c2], dv);  // dV = (dS Kr)[:DIM], row c
          oq[k] = tq[n][c] > 0.f ? dq : 0.f;
          ok[k] = tk[n][c] > 0.f ? dk : 0.f;
          ov[k] = dv;
        }
        T* o = (hi ? dms1 : dms) + (b * N + n0 + n) * (int64_t)RS + gq * 3 * DIM + c0;
        MlaVec<DT, CPT>::st(o, oq);
        MlaVec<DT, CPT>::st(o + DIM, ok);
        MlaVec<DT, CPT>::st(o + 2 * DIM, ov);
      }
    }
  }
}


// sum of P partial (DIM + 1) x DIM matrices per (image, head group) in a fixed order (deterministic)
__global__ void mla_backward_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int P, int SE, int64_t total) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (bg, e)
  if (i >= total) return;
  const int64_t bg = i / SE;
  const int e = (int)(i - bg * SE);
  float s_ = 0.f;
  for (int p_ = 0; p_ < P; ++p_) s_ += part[(bg * P + p_) * SE + e];
  out[i] = s_;
}

constexpr int TRAIN_SPLITS_MAX = 256;
int wgrad_splits(int64_t M) {  // every split a whole number of 64-row tiles
  int64_t tiles = (M + WG_ROWS - 1) / WG_ROWS;
  int64_t s = tiles < TRAIN_SPLITS_MAX ? tiles : TRAIN_SPLITS_MAX;
  return (int)(s < 1 ? 1 : s);
}

// Splits of the weight gradient's row reduction: enough workgroups to fill the chip (about 2048 with the N x K tiles), not one per 64-row
// tile -- with 128 - 256 splits of a 1024 x 256 weight the fp32 partial tiles were 128 x the size of dW, and wgrad_reduce_kernel alone 8 %
// of a training step (profiles/r05/stage1_step_b2_kernel_stats_mla4.csv).  A function of (M, N, K) only: the summation order stays fixed.
int wgrad_tile(int dtype, int N, int K) { return dtype == 1 && N >= 128 && K >= 128 ? 128 : WG_TILE; }   // see wgrad_kernel: TW
int wgrad_splits_nk(int64_t M, int N, int K, int tile = WG_TILE) {
  const int64_t tiles_nk = (int64_t)((N + tile - 1) / tile) * ((K + tile - 1) / tile);
  int64_t want = 2048 / tiles_nk;
  if (want < 1) want = 1;
  // round 6: up to 1024 splits (was 256, one workgroup per CU) for a weight of ONE tile (the 8 M-row layers of the first stages): 256
  // workgroups with two 16-byte loads in flight per thread ran at 1.5 - 2 TB/s, bound by latency; four workgroups per CU now (0.36 -> 0.21 ms)
  const int64_t tiles = (M + WG_ROWS - 1) / WG_ROWS;
  const int64_t cap = tiles_nk == 1 ? 1024 : TRAIN_SPLITS_MAX;   // with two or more tiles the extra partial tiles cost more than they hide
  int64_t most = tiles < cap ? (tiles < 1 ? 1 : tiles) : cap;
  if (tile == 128 && most > (M + 1023) / 1024) most = (M + 1023) / 1024;   // at least 8 row steps per workgroup: its partial tile is 64 KB
  return (int)(want < most ? want : most);
}

}  // namespace

extern "C" {

int esam3_act_forward(int dtype, const void* x, void* y, int64_t n, int act, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !y || n <= 0 || n % 8 || act < 0 || act > ACT_SIGMOID) {
    esam3_set_error("esam3_act_forward: bad argument (n a multiple of 8; act none | relu | gelu | hswish | sigmoid)");
    return -1;
  }
  const int64_t n8 = n / 8;
  const unsigned grid = (unsigned)(n8 / 256 + 1 < 16384 ? n8 / 256 + 1 : 16384);
  if (dtype == 0) hipLaunchKernelGGL((act_kernel<0, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)nullptr, (float*)y, n8, act);
  else hipLaunchKernelGGL((act_kernel<1, false>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)nullptr, (uint16_t*)y, n8, act);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_act_backward(int dtype, const void* x, const void* dy, void* dx, int64_t n, int act, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !dy || !dx || n <= 0 || n % 8 || act < 0 || act > ACT_SIGMOID) {
    esam3_set_error("esam3_act_backward: bad argument (n a multiple of 8; act none | relu | gelu | hswish | sigmoid)");
    return -1;
  }
  const int64_t n8 = n / 8;
  const unsigned grid = (unsigned)(n8 / 256 + 1 < 16384 ? n8 / 256 + 1 : 16384);
  if (dtype == 0) hipLaunchKernelGGL((act_kernel<0, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, (float*)dx, n8, act);
  else hipLaunchKernelGGL((act_kernel<1, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)dy, (uint16_t*)dx, n8, act);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int64_t esam3_linear_wgrad_workspace(int64_t M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int s64 = wgrad_splits_nk(M, N, K), s128 = wgrad_splits_nk(M, N, K, 128);   // either tile size (the dtype is not known here)
  return (int64_t)sizeof(float) * (s64 > s128 ? s64 : s128) * ((int64_t)N * K + N);
}

int esam3_linear_wgrad(int dtype, const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* dbias, void* workspace,
                       void* stream) {
  if ((dtype != 0 && dtype != 1) || !dy || !x || !dw || !workspace || M <= 0 || N <= 0 || K <= 0 || N % 8 || K % 8) {
    esam3_set_error("esam3_linear_wgrad: bad argument (fp32 / bf16; N and K multiples of 8)");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int tile = wgrad_tile(dtype, N, K);
  const int splits = wgrad_splits_nk(M, N, K, tile);
  const int64_t tiles = (M + WG_ROWS - 1) / WG_ROWS;
  const int64_t rps = (tiles + splits - 1) / splits * WG_ROWS;
  const int zs = (int)((M + rps - 1) / rps);  // splits actually used
  float* partial = (float*)workspace;
  dim3 grid((unsigned)((K + tile - 1) / tile), (unsigned)((N + tile - 1) / tile), (unsigned)zs);
  WgradGather gt{0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (grid.x * grid.y >= 4 && !esam3_dev_flag("ESAM3_WGRAD_NO_XCD")) {   // several tiles share the rows of a split: keep them on one XCD
    gt.gx = (int)grid.x; gt.gy = (int)grid.y; gt.gz = zs;
    grid = dim3((unsigned)(8 * ((gt.gx * gt.gy * gt.gz + 7) / 8)));
  }
  if (dtype == 0) hipLaunchKernelGGL((wgrad_kernel<0, 64>), grid, dim3(256), 0, s, (const float*)dy, N, (const float*)x, K, M, N, K, rps, partial, gt);
  else if (tile == 128) hipLaunchKernelGGL((wgrad_kernel<1, 128>), grid, dim3(256), 0, s, (const uint16_t*)dy, N, (const uint16_t*)x, K, M, N, K, rps, partial, gt);
  else hipLaunchKernelGGL((wgrad_kernel<1, 64>), grid, dim3(256), 0, s, (const uint16_t*)dy, N, (const uint16_t*)x, K, M, N, K, rps, partial, gt);
  const int64_t nk = (int64_t)N * K;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nk + 63) / 64)), dim3(256), 0, s, partial, zs, nk, dw);
  if (dbias) {
    float* pb = partial + (int64_t)splits * nk;
    const dim3 g2((unsigned)((N + 255) / 256), (unsigned)zs);
    if (dtype == 0) hipLaunchKernelGGL(colsum_kernel<0>, g2, dim3(256), 0, s, (const float*)dy, N, M, N, rps, pb);
    else hipLaunchKernelGGL(colsum_kernel<1>, g2, dim3(256), 0, s, (const uint16_t*)dy, N, M, N, rps, pb);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, pb, zs, (int64_t)N, dbias);
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

static int conv3x3_wgrad_splits(int64_t M, int N, int K, int tile = WG_TILE) {
  const int64_t tiles_nk = 9 * (int64_t)((N + tile - 1) / tile) * ((K + tile - 1) / tile);
  int64_t want = 2048 / tiles_nk;
  if (want < 1) want = 1;
  int64_t most = wgrad_splits(M);
  if (tile == 128 && most > (M + 1023) / 1024) most = (M + 1023) / 1024;
  return (int)(want < most ? want : most);
}

int64_t esam3_conv3x3_wgrad_workspace(int B, int IH, int IW, int Cin, int Cout, int stride) {
  if (B <= 0 || IH <= 0 || IW <= 0 || Cin <= 0 || Cout <= 0 || (stride != 1 && stride != 2)) return 0;
  const int64_t M = (int64_t)B * ((IH + stride - 1) / stride) * ((IW + stride - 1) / stride);
  const int s64 = conv3x3_wgrad_splits(M, Cout, Cin), s128 = conv3x3_wgrad_splits(M, Cout, Cin, 128);
  return (int64_t)sizeof(float) * 9 * (s64 > s128 ? s64 : s128) * (int64_t)Cout * Cin;
}

// dw [Cout][Cin][3][3] fp32 of a dense 3x3 conv (padding 1, stride 1 | 2): x [B][IH][IW][Cin], dy [B][ceil(IH/s)][ceil(IW/s)][Cout]
int esam3_conv3x3_wgrad(int dtype, const void* dy, const void* x, int B, int IH, int IW, int Cin, int Cout, int stride, float* dw, void* workspace,
                        void* stream) {
  if ((dtype != 0 && dtype != 1) || !dy || !x || !dw || !workspace || B <= 0 || IH <= 0 || IW <= 0 || Cin <= 0 || Cout <= 0 || Cin % 8 || Cout % 8 ||
      (stride != 1 && stride != 2)) {
    esam3_set_error("esam3_conv3x3_wgrad: bad argument (fp32 / bf16; Cin and Cout multiples of 8; stride 1 | 2)");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int OH = (IH + stride - 1) / stride, OW = (IW + stride - 1) / stride;
  const int64_t M = (int64_t)B * OH * OW;
  const int N = Cout, K = Cin;
  const int tile = wgrad_tile(dtype, N, K);
  const int splits = conv3x3_wgrad_splits(M, N, K, tile);
  const int64_t tiles = (M + WG_ROWS - 1) / WG_ROWS;
  const int64_t rps = (tiles + splits - 1) / splits * WG_ROWS;
  const int zs = (int)((M + rps - 1) / rps);
  if (9 * zs > 65535 || M >= ((int64_t)1 << 31)) {
    esam3_set_error("esam3_conv3x3_wgrad: %d splits, %lld output pixels", zs, (long long)M);
    return -1;
  }
  float* partial = (float*)workspace;
  WgradGather gt{1, zs, OH, OW, IH, IW, stride, 0, 0, 0};
  dim3 grid((unsigned)((K + tile - 1) / tile), (unsigned)((N + tile - 1) / tile), (unsigned)(9 * zs));
  if (grid.x * grid.y >= 4 && !esam3_dev_flag("ESAM3_WGRAD_NO_XCD")) {
    gt.gx = (int)grid.x; gt.gy = (int)grid.y; gt.gz = 9 * zs;
    grid = dim3((unsigned)(8 * ((gt.gx * gt.gy * gt.gz + 7) / 8)));
  }
  if (dtype == 0) hipLaunchKernelGGL((wgrad_kernel<0, 64>), grid, dim3(256), 0, s, (const float*)dy, N, (const float*)x, K, M, N, K, rps, partial, gt);
  else if (tile == 128) hipLaunchKernelGGL((wgrad_kernel<1, 128>), grid, dim3(256), 0, s, (const uint16_t*)dy, N, (const uint16_t*)x, K, M, N, K, rps, partial, gt);
  else hipLaunchKernelGGL((wgrad_kernel<1, 64>), grid, dim3(256), 0, s, (const uint16_t*)dy, N, (const uint16_t*)x, K, M, N, K, rps, partial, gt);
  const int64_t nk = (int64_t)N * K;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((nk + 255) / 256), 9), dim3(256), 0, s, partial, zs, nk, dw);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int64_t esam3_colsum_workspace(int64_t M, int N) { return M > 0 && N > 0 ? (int64_t)sizeof(float) * wgrad_splits(M) * N : 0; }

// out[N] = sum over the M rows of dy[M][N]: the bias gradient of a conv / Linear (the same two kernels esam3_linear_wgrad runs for dbias)
int esam3_colsum(int dtype, const void* dy, int64_t M, int N, float* out, void* workspace, void* stream) {
  if ((dtype != 0 && dtype != 1) || !dy || !out || !workspace || M <= 0 || N <= 0) {
    esam3_set_error("esam3_colsum: bad argument");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int splits = wgrad_splits(M);
  const int64_t tiles = (M + WG_ROWS - 1) / WG_ROWS;
  const int64_t rps = (tiles + splits - 1) / splits * WG_ROWS;
  const int zs = (int)((M + rps - 1) / rps);
  float* pb = (float*)workspace;
  const dim3 g2((unsigned)((N + 255) / 256), (unsigned)zs);
  if (N % 8 == 0 && N <= 2048 && (((uintptr_t)dy) & 15) == 0) {   // round 6: 16-byte loads, row lanes (colsum_vec_kernel)
    if (dtype == 0) hipLaunchKernelGGL(colsum_vec_kernel<0>, dim3((unsigned)zs), dim3(256), 0, s, (const float*)dy, M, N, rps, pb);
    else hipLaunchKernelGGL(colsum_vec_kernel<1>, dim3((unsigned)zs), dim3(256), 0, s, (const uint16_t*)dy, M, N, rps, pb);
  } else if (dtype == 0) hipLaunchKernelGGL(colsum_kernel<0>, g2, dim3(256), 0, s, (const float*)dy, N, M, N, rps, pb);
  else hipLaunchKernelGGL(colsum_kernel<1>, g2, dim3(256), 0, s, (const uint16_t*)dy, N, M, N, rps, pb);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((N + 63) / 64)), dim3(256), 0, s, pb, zs, (int64_t)N, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_channel_scale(int dtype, const void* x, const float* mul, int mul_per_image, float plus_one, const float* bias, int bias_per_image,
                        float bias_scale, const void* add, void* out, int B, int64_t HW, int C, void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !mul || !out || B <= 0 || HW <= 0 || C <= 0 || C % 8) {
    esam3_set_error("esam3_channel_scale: bad argument (fp32 / bf16; C a multiple of 8)");
    return -1;
  }
  const int C8 = C / 8;
  const int64_t per_image8 = HW * C8, total8 = per_image8 * B;
  const unsigned grid = (unsigned)(total8 / 256 + 1 < 16384 ? total8 / 256 + 1 : 16384);
  hipStream_t s = (hipStream_t)stream;
  if (!mul_per_image && !bias_per_image && 256 % C8 == 0 && total8 < ((int64_t)1 << 31)) {   // one channel group per thread (round 6)
    if (dtype == 0)
      hipLaunchKernelGGL(channel_scale_fast_kernel<0>, dim3(grid < 4096 ? grid : 4096), dim3(256), 0, s, (const float*)x, mul, plus_one, bias, bias_scale, (const float*)add,
                         (float*)out, C8, (unsigned)total8);
    else
      hipLaunchKernelGGL(channel_scale_fast_kernel<1>, dim3(grid < 4096 ? grid : 4096), dim3(256), 0, s, (const uint16_t*)x, mul, plus_one, bias, bias_scale,
                         (const uint16_t*)add, (uint16_t*)out, C8, (unsigned)total8);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  if (dtype == 0)
    hipLaunchKernelGGL(channel_scale_kernel<0>, dim3(grid), dim3(256), 0, s, (const float*)x, mul, mul_per_image ? C : 0, plus_one, bias,
                       bias_per_image ? C : 0, bias_scale, (const float*)add, (float*)out, per_image8, C8, total8);
  else
    hipLaunchKernelGGL(channel_scale_kernel<1>, dim3(grid), dim3(256), 0, s, (const uint16_t*)x, mul, mul_per_image ? C : 0, plus_one, bias,
                       bias_per_image ? C : 0, bias_scale, (const uint16_t*)add, (uint16_t*)out, per_image8, C8, total8);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int64_t esam3_batched_coldot_workspace(int B, int C) { return B > 0 && C > 0 ? (int64_t)sizeof(float) * COLDOT_SPLITS_MAX * (B < 16 ? 16 : B) * C : 0; }

int esam3_batched_coldot(int dtype, const void* a, const void* b2, int B, int64_t HW, int C, float scale, float* out, void* workspace,
                         void* stream) {
  if ((dtype != 0 && dtype != 1) || !a || !out || !workspace || B <= 0 || B > 65535 || HW <= 0 || C <= 0 || C % 8 || C > 2048) {
    esam3_set_error("esam3_batched_coldot: bad argument (fp32 / bf16; C a multiple of 8, at most 2048; B at most 65535)");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int splits = coldot_splits(B, HW, C);
  const int64_t rps = (HW + splits - 1) / splits;
  const int used = (int)((HW + rps - 1) / rps);
  float* partial = (float*)workspace;
  const dim3 grid((unsigned)used, (unsigned)B);
  if (dtype == 0) hipLaunchKernelGGL(batched_coldot_kernel<0>, grid, dim3(256), 0, s, (const float*)a, (const float*)b2, HW, C, rps, partial);
  else hipLaunchKernelGGL(batched_coldot_kernel<1>, grid, dim3(256), 0, s, (const uint16_t*)a, (const uint16_t*)b2, HW, C, rps, partial);
  const int64_t n = (int64_t)B * C;
  hipLaunchKernelGGL(coldot_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, partial, used, n, scale, out);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int64_t esam3_dwconv_wgrad_workspace(int C) { return C > 0 ? (int64_t)sizeof(float) * TRAIN_SPLITS_MAX * 25 * C : 0; }

int esam3_dwconv_wgrad(int dtype, const void* x, const void* dy, int B, int H, int W, int C, int ksize, int stride, float* dw, void* workspace,
                       void* stream) {
  if ((dtype != 0 && dtype != 1) || !x || !dy || !dw || !workspace || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || C > 2048 ||
      (stride != 1 && stride != 2) || (ksize != 3 && ksize != 5)) {
    esam3_set_error("esam3_dwconv_wgrad: bad argument (3x3 | 5x5, padding k / 2, stride 1 | 2, C a multiple of 8 up to 2048)");
    return -1;
  }
  hipStream_t s = (hipStream_t)stream;
  const int OH = (H + stride - 1) / stride, OW = (W + stride - 1) / stride;
  const int64_t npx = (int64_t)B * OH * OW;
  const int splits = (int)(npx < TRAIN_SPLITS_MAX ? npx : TRAIN_SPLITS_MAX);
  const int RL = 256 / (C / 8);
  const size_t lds = sizeof(float) * (size_t)RL * ksize * C;
  float* partial = (float*)workspace;
  if (ksize == 3) {   // round 6: all nine taps in one workgroup (dw_wgrad3_kernel)
    const int64_t nrows = (int64_t)B * OH;
    const int sp3 = (int)(nrows < 704 ? nrows : 704);   // 704 x 9 x C floats fit the workspace sized for 256 x 25 x C
    const size_t lds3 = sizeof(float) * (size_t)RL * C;
    const bool slide = stride == 1 && C >= 64 && !esam3_dev_flag("ESAM3_DWW_OLD");   // round 6: sliding window along the row (narrow C: 32-byte runs per lane, slower)
    if (slide && dtype == 0) hipLaunchKernelGGL(dw_wgrad3_slide_kernel<0>, dim3((unsigned)sp3), dim3(256), lds3, s, (const float*)x, (const float*)dy, B, H, W, C, partial);
    else if (slide) hipLaunchKernelGGL(dw_wgrad3_slide_kernel<1>, dim3((unsigned)sp3), dim3(256), lds3, s, (const uint16_t*)x, (const uint16_t*)dy, B, H, W, C, partial);
    else if (dtype == 0) hipLaunchKernelGGL(dw_wgrad3_kernel<0>, dim3((unsigned)sp3), dim3(256), lds3, s, (const float*)x, (const float*)dy, B, H, W, C, stride, partial);
    else hipLaunchKernelGGL(dw_wgrad3_kernel<1>, dim3((unsigned)sp3), dim3(256), lds3, s, (const uint16_t*)x, (const uint16_t*)dy, B, H, W, C, stride, partial);
    hipLaunchKernelGGL(dw_wgrad_finalize_kernel, dim3((unsigned)((9 * C + 63) / 64)), dim3(256), 0, s, partial, sp3, C, 9, dw);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  const dim3 grid((unsigned)splits, (unsigned)ksize);
#define ESAM3_DWW(DT_, KS_, T_)                                                                                                   \
  do {                                                                                                                            \
    if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(dw_wgrad_kernel<DT_, KS_>), 160 * 1024)) return -1;                     \
    hipLaunchKernelGGL((dw_wgrad_kernel<DT_, KS_>), grid, dim3(256), lds, s, (const T_*)x, (const T_*)dy, B, H, W, C, stride, partial); \
  } while (0)
  if (ksize == 5 && stride == 1 && !esam3_dev_flag("ESAM3_DWW_OLD")) {   // round 6: sliding window along the row
    if (dtype == 0) {
      if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(dw_wgrad_slide_kernel<0, 5>), 160 * 1024)) return -1;
      hipLaunchKernelGGL((dw_wgrad_slide_kernel<0, 5>), grid, dim3(256), lds, s, (const float*)x, (const float*)dy, B, H, W, C, partial);
    } else {
      if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(dw_wgrad_slide_kernel<1, 5>), 160 * 1024)) return -1;
      hipLaunchKernelGGL((dw_wgrad_slide_kernel<1, 5>), grid, dim3(256), lds, s, (const uint16_t*)x, (const uint16_t*)dy, B, H, W, C, partial);
    }
  } else if (dtype == 0 && ksize == 3) ESAM3_DWW(0, 3, float);
  else if (dtype == 0) ESAM3_DWW(0, 5, float);
  else if (ksize == 3) ESAM3_DWW(1, 3, uint16_t);
  else ESAM3_DWW(1, 5, uint16_t);
#undef ESAM3_DWW
  hipLaunchKernelGGL(dw_wgrad_finalize_kernel, dim3((unsigned)((ksize * ksize * C + 63) / 64)), dim3(256), 0, s, partial, splits, C,
                     ksize * ksize, dw);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_lite_mla_backward(int dtype, const void* ms, const void* dout, void* dms, void* y, int B, int N, int groups, int dim, float eps,
                            void* stream) {
  if ((dtype != 0 && dtype != 1) || !ms || !dout || !dms || B <= 0 || N <= 0 || groups <= 0 || (dim != 16 && dim != 32)) {
    esam3_set_error("esam3_lite_mla_backward: bad argument (head dim 16 or 32)");
    return -1;
  }
  const dim3 grid((unsigned)(B * groups));
  hipStream_t s = (hipStream_t)stream;
#define ESAM3_MLA_BWD(DT_, DIM_, T_) \
  hipLaunchKernelGGL((mla_backward_kernel<DT_, DIM_>), grid, dim3(256), 0, s, (const T_*)ms, (const T_*)dout, (T_*)dms, (T_*)y, N, groups, eps)
  if (dtype == 0 && dim == 16) ESAM3_MLA_BWD(0, 16, float);
  else if (dtype == 0) ESAM3_MLA_BWD(0, 32, float);
  else if (dim == 16) ESAM3_MLA_BWD(1, 16, uint16_t);
  else ESAM3_MLA_BWD(1, 32, uint16_t);
#undef ESAM3_MLA_BWD
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

static int mla_bw_parts(int N) { int p_ = (N + 255) / 256; return p_ < 1 ? 1 : (p_ > 64 ? 64 : p_); }

int64_t esam3_lite_mla_backward_workspace(int B, int N, int groups, int dim) {
  if (B <= 0 || N <= 0 || groups <= 0 || (dim != 16 && dim != 32)) return 0;
  const int64_t SE = (int64_t)(dim + 1) * dim;
  return (int64_t)sizeof(float) * B * groups * SE * ((mla_bw_parts(N) > 2 ? mla_bw_parts(N) : 2) + 2);
}

static int mla_backward_ws_impl(int dtype, const void* ms, const void* ms1, const void* dout, void* dms, void* dms1, void* y, int B, int N, int groups,
                               int dim, float eps, void* workspace, void* stream);

int esam3_lite_mla_backward_ws(int dtype, const void* ms, const void* dout, void* dms, void* y, int B, int N, int groups, int dim, float eps,
                               void* workspace, void* stream) {
  return mla_backward_ws_impl(dtype, ms, nullptr, dout, dms, nullptr, y, B, N, groups, dim, eps, workspace, stream);
}

// the multi-scale tensor as its two halves (see mla_backward_part_kernel): ms0 / ms1 [B][N][groups / 2 * 3 * dim], gradients dms0 / dms1 likewise
int esam3_lite_mla_backward_ws2(int dtype, const void* ms0, const void* ms1, const void* dout, void* dms0, void* dms1, void* y, int B, int N,
                                int groups, int dim, float eps, void* workspace, void* stream) {
  if (!ms1 || groups % 2 || (dout && !dms1) || N <= 256) {
    esam3_set_error("esam3_lite_mla_backward_ws2: two halves need an even group count, both gradient tensors and N > 256 tokens");
    return -1;
  }
  return mla_backward_ws_impl(dtype, ms0, ms1, dout, dms0, dms1, y, B, N, groups, dim, eps, workspace, stream);
}

static int mla_backward_ws_impl(int dtype, const void* ms, const void* ms1, const void* dout, void* dms, void* dms1, void* y, int B, int N, int groups,
                               int dim, float eps, void* workspace, void* stream) {
  const bool fwd_only = !dout;   // dout_dev NULL: the forward output y only (dms_dev is not written)
  if ((dtype != 0 && dtype != 1) || !ms || (!fwd_only && !dms) || (fwd_only && !y) || !workspace || B <= 0 || N <= 0 || groups <= 0 ||
      (dim != 16 && dim != 32)) {
    esam3_set_error("esam3_lite_mla_backward_ws: bad argument (head dim 16 or 32, workspace of esam3_lite_mla_backward_workspace bytes)");
    return -1;
  }
  const int P = fwd_only ? (mla_bw_parts(N) > 1 ? mla_bw_parts(N) : 2) : mla_bw_parts(N);
  if (P == 1) return esam3_lite_mla_backward(dtype, ms, dout, dms, y, B, N, groups, dim, eps, stream);
  const int TT = 64;
  const int span = ((N + P - 1) / P + TT - 1) / TT * TT;   // whole 64-token tiles per workgroup
  const int64_t BG = (int64_t)B * groups, SE = (int64_t)(dim + 1) * dim;
  float* part = (float*)workspace;
  float* Sg = part + BG * P * SE;
  float* dSg = Sg + BG * SE;
  const dim3 grid((unsigned)BG, (unsigned)P);
  const unsigned rgrid = (unsigned)((BG * SE + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
#define ESAM3_MLA_PART(DT_, DIM_, T_, MODE_) \
  hipLaunchKernelGGL((mla_backward_part_kernel<DT_, DIM_, MODE_>), grid, dim3(256), 0, s, (const T_*)ms, (const T_*)dout, (T_*)dms, (T_*)y, N, groups, eps, \
                     P, span, Sg, dSg, part, (const T_*)ms1, (T_*)dms1)
#define ESAM3_MLA_ALL(DT_, DIM_, T_)                                                                                  \
  do {                                                                                                               \
    ESAM3_MLA_PART(DT_, DIM_, T_, 0);                                                                                \
    hipLaunchKernelGGL(mla_backward_reduce_kernel, dim3(rgrid), dim3(256), 0, s, part, Sg, P, (int)SE, BG * SE);     \
    if (fwd_only) { ESAM3_MLA_PART(DT_, DIM_, T_, 3); break; }                                                      \
    ESAM3_MLA_PART(DT_, DIM_, T_, 1);                                                                                \
    hipLaunchKernelGGL(mla_backward_reduce_kernel, dim3(rgrid), dim3(256), 0, s, part, dSg, P, (int)SE, BG * SE);    \
    ESAM3_MLA_PART(DT_, DIM_, T_, 2);                                                                                \
  } while (0)
  if (dtype == 0 && dim == 16) ESAM3_MLA_ALL(0, 16, float);
  else if (dtype == 0) ESAM3_MLA_ALL(0, 32, float);
  else if (dim == 16) ESAM3_MLA_ALL(1, 16, uint16_t);
  else ESAM3_MLA_ALL(1, 32, uint16_t);
#undef ESAM3_MLA_ALL
#undef ESAM3_MLA_PART
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

int esam3_dwconv_dgrad(int dtype, const void* dy, const float* w, void* dx, int B, int H, int W, int C, int ksize, int stride, void* stream) {
  if ((dtype != 0 && dtype != 1) || !dy || !w || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (stride != 1 && stride != 2) ||
      (ksize != 3 && ksize != 5)) {
    esam3_set_error("esam3_dwconv_dgrad: bad argument (3x3 | 5x5, padding k / 2, stride 1 | 2, C a multiple of 8)");
    return -1;
  }
  const int64_t total = (int64_t)B * H * W * (C / 8);
  const unsigned grid = (unsigned)(total / 256 + 1 < 32768 ? total / 256 + 1 : 32768);
  hipStream_t s = (hipStream_t)stream;
  if (stride == 2 && ksize == 3 && (size_t)9 * C * 4 <= 64 * 1024) {   // round 6: weights in LDS
    const size_t lds = (size_t)9 * C * 4;
    const unsigned g2 = (unsigned)((int64_t)B * H < 16384 ? (int64_t)B * H : 16384);   // one workgroup per input row (b, iy), strided above that
    if (dtype == 0) hipLaunchKernelGGL(dw_dgrad_s2_kernel<0>, dim3(g2), dim3(256), lds, s, (const float*)dy, w, (float*)dx, B, H, W, C);
    else hipLaunchKernelGGL(dw_dgrad_s2_kernel<1>, dim3(g2), dim3(256), lds, s, (const uint16_t*)dy, w, (uint16_t*)dx, B, H, W, C);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  if (dtype == 0 && ksize == 3) hipLaunchKernelGGL((dw_dgrad_kernel<0, 3>), dim3(grid), dim3(256), 0, s, (const float*)dy, w, (float*)dx, B, H, W, C, stride);
  else if (dtype == 0) hipLaunchKernelGGL((dw_dgrad_kernel<0, 5>), dim3(grid), dim3(256), 0, s, (const float*)dy, w, (float*)dx, B, H, W, C, stride);
  else if (ksize == 3) hipLaunchKernelGGL((dw_dgrad_kernel<1, 3>), dim3(grid), dim3(256), 0, s, (const uint16_t*)dy, w, (uint16_t*)dx, B, H, W, C, stride);
  else hipLaunchKernelGGL((dw_dgrad_kernel<1, 5>), dim3(grid), dim3(256), 0, s, (const uint16_t*)dy, w, (uint16_t*)dx, B, H, W, C, stride);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // extern "C"
