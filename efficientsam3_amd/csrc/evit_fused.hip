// Fused EfficientViT kernels for gfx950 (bf16 engine), round 4.
//
//   mbconv3_kernel   MBConv (efficientvit/nn/ops.py:315-367) = 1x1 expand (+bias, Hardswish) -> depthwise 3x3 (stride 1|2,
//                    +bias, Hardswish) -> 1x1 project (+BN) (+identity shortcut, ops.py:740-770) in one kernel, every
//                    EfficientViT-B0/B1 shape up to 256 channels: the six high-resolution MBConvs of stages 1-3 AND the
//                    local modules of the EfficientViTBlocks of stages 3-4 (ops.py:701-721), which round 3 ran layer by layer.
//                    Same tiling as mbconv_fused2_kernel (mbconv_fused.hip: 8 x 16 output pixels, 64 expanded channels per
//                    chunk, E -> D -> P phases), but
//                      * the depthwise phase runs on the matrix cores (v_mfma_f32_4x4x4_16b_bf16 with diag(w) blocks, the
//                        dwconv_mfma_kernel idiom of kernels_backbone.hip): one LDS read feeds one instruction = one tap of
//                        4 pixels x 64 channels, no bf16 -> f32 unpacking, and the VALU only sees Hardswish + packing;
//                      * the biases are the accumulators' initial values, Hardswish is x * clamp(x / 6 + 0.5, 0, 1)
//                        (fma + med3 + mul), and the "outside the image" zeroing of the expand output is a wave-uniform
//                        branch that interior tiles skip;
//                      * channels are generic: K loops over Cin in 16-channel MFMA steps with bounded register groups,
//                        8-wave workgroups split the project GEMM's output channels for Cout = 256.
//   mla1_kernel      LiteMLA (ops.py:521-671), first pass over a tile: qkv 1x1 conv on the tile + 2-pixel halo -> LDS,
//                    depthwise 5x5 (matrix cores) -> grouped 1x1 (v_mfma_f32_16x16x16_bf16 per 16-channel group) -> LDS;
//                    ReLU(q) of both scales to HBM, and the tile's share of kv = sum_px [v; 1] (x) relu(k) per head as
//                    fp32 partials (MFMA over the pixels, operands by ds_read_b64_tr_b16).  The 3C- and 6C-channel tensors
//                    of the layer-by-layer path (qkv, aggreg output, multi-scale concat) never exist in HBM.
//   mla_kvprep       sums the tile partials in a fixed order (deterministic) and writes the per-image kv matrices as
//                    bf16 hi + lo MFMA operands.
//   mla2_kernel      second pass: att = (kv . relu(q)) / (ksum . relu(q) + eps) per head on the matrix cores (fp32 divide),
//                    proj 1x1 + BN + identity shortcut (ops.py:663-671,740-770) accumulated over 64-channel chunks of att.
#include <type_traits>

#include "gemm_common.h"
#include "kernels.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Hardswish x * relu6(x + 3) / 6 as x * clamp(x / 6 + 0.5, 0, 1): fma + med3 (or a clamp modifier) + mul
__device__ __forceinline__ float hsw(float x) {
  const float t = __builtin_amdgcn_fmed3f(fmaf(x, 1.f / 6.f, 0.5f), 0.f, 1.f);
  return x * t;
}

// the same on pairs: v_pk_fma_f32 + two clamps + v_pk_mul_f32 = two VALU instructions per element instead of three (the fused
// kernels are instruction-issue bound: profiles/r04/pmc_c_summary.txt).  Plain C: an inline-assembly v_pk_fma_f32 ... clamp
// would save one more, but hipcc pads no MFMA-result -> VALU-read wait states for an asm reader (cdna_hip_programming.md 5.7).
__device__ __forceinline__ f32x2_v hsw2(f32x2_v x) {
  f32x2_v t = __builtin_elementwise_fma(x, (f32x2_v)(1.f / 6.f), (f32x2_v)(0.5f));
  t = __builtin_elementwise_min(__builtin_elementwise_max(t, (f32x2_v)(0.f)), (f32x2_v)(1.f));
  return x * t;
}
// Round 5: the clamp as the VOP3P clamp modifier of the packed fma (result clamped to [0, 1]): v_pk_fma_f32 ... clamp + v_pk_mul_f32 =
// ONE VALU instruction per element instead of two (hipcc has no pattern that folds a clamp into a packed f32 op: it emits a
// v_max_f32 ... clamp per element after the v_pk_fma_f32).  The fused MBConv kernels are VALU-issue bound (457 VALU instructions
// per tile and wave in mbconv3s<2,16,32>, 96 of them these clamps: profiles/r05/isa_mbconv3s.txt).  An inline-assembly reader of an
// MFMA result gets no hazard padding from the compiler (cdna_hip_programming.md 5.7), so the asm form is only used BEHIND a
// compiler-generated VALU reader of the same accumulator: `guard` is a value that reader produced, named as an (unused) input, so
// the statement cannot be scheduled above it; once that reader has issued (with the compiler's wait states), the accumulator
// is readable by everybody.
__device__ __forceinline__ f32x2_v hsw2_after(f32x2_v x, float guard) {
#ifdef ESAM3_HSW_C   // A/B builds (tools/dev_variants.sh): the round-4 C form
  return hsw2(x);
#endif
  f32x2_v t;
  const f32x2_v k6 = {1.f / 6.f, 1.f / 6.f}, k05 = {0.5f, 0.5f};
  asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(t) : "v"(x), "v"(k6), "v"(k05), "v"(guard));
  return x * t;
}
template <int N> __device__ __forceinline__ void hsw_n(float (&v)[N]) {   // v: the registers of ONE accumulator
  static_assert(N % 2 == 0, "pairs");
  const f32x2_v r0 = hsw2(f32x2_v{v[0], v[1]});   // the compiler's reader of the accumulator (hazard padding)
#pragma unroll
  for (int i = 2; i < N; i += 2) {
    const f32x2_v r = hsw2_after(f32x2_v{v[i], v[i + 1]}, r0.x);
    v[i] = r.x;
    v[i + 1] = r.y;
  }
  v[0] = r0.x;
  v[1] = r0.y;
}
__device__ __forceinline__ uint2 hsw_pack4(const f32x4& a) {   // 4 accumulators -> Hardswish -> 4 bf16
  const f32x2_v lo = hsw2(f32x2_v{a[0], a[1]}), hi = hsw2_after(f32x2_v{a[2], a[3]}, lo.x);
  return make_uint2(pack_bf16x2(lo.x, lo.y), pack_bf16x2(hi.x, hi.y));
}

// workgroup L of a 1-D grid of nb -> an index such that every XCD (L % 8) owns one contiguous range (bijective)
__device__ __forceinline__ unsigned xcd_contig(unsigned L, unsigned nb) {
  const unsigned q = nb / 8, r = nb % 8, xcd = L % 8, idx = L / 8;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ---- depthwise conv on the matrix cores, software-pipelined by hand -------------------------------------------------------
// One v_mfma_f32_4x4x4_16b_bf16 = one tap of 4 pixels x 64 channels and needs one 8-byte LDS read per lane, so the phase is
// LDS-bandwidth bound (512 bytes per 8-cycle MFMA and SIMD = the LDS's 256 bytes / clock).  hipcc pairs neighbouring 8-byte reads
// into ds_read2_b64, which the LDS serves in 16-lane groups on 32 banks at HALF the rate of ds_read_b64 (MI355X_MICROARCH.md,
// LDS table) and with 2-way conflicts for this tile pitch; there is no switch for that pass.  The reads are therefore inline
// assembly: ds_read_b64 with immediate offsets, the reads of tap t+1 issued before the MFMAs of tap t, one counted
// s_waitcnt lgkmcnt per tap.  The wait statement names the landed registers "+v", so no consumer is scheduled above it
// (cdna_hip_programming.md 5.7, form ii); LDS operations of one wave return in order, a scalar load still in flight only makes
// the counted wait conservative.
template <int OFF>
__device__ __forceinline__ void dsr64(s16x4& d, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
}
template <int N> __device__ __forceinline__ void ds_wait(s16x4 (&c)[3]) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]) : "i"(N));
}
template <int N> __device__ __forceinline__ void ds_wait(s16x4 (&c)[5]) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]) : "i"(N));
}
// A lane block (4 channels x 4 horizontally adjacent pixels) owns ROWS consecutive output rows of ONE pixel run: walking the
// IR = (ROWS - 1) S + KSZ input rows, every loaded operand (input row ir, column tap kx) feeds the up-to-KSZ output rows whose
// window contains that row -- KSZ x IR reads for KSZ^2 x ROWS MFMAs (3 x 10 for 72 at ROWS = 8, stride 1) instead of one read per
// MFMA.  `addr` = LDS byte address of (first input row, pixel pi of the run, channel block); batch = one input row.
template <int KSZ, int ROWS, int S, int HW, int PITCH>
struct DwRows {
  static constexpr int IR = (ROWS - 1) * S + KSZ, NTAP = KSZ * KSZ;
  template <int R, int I = 0> static __device__ __forceinline__ void issue(s16x4 (&buf)[KSZ], unsigned addr) {
    if constexpr (I < KSZ) {
      dsr64<(R * HW + I) * PITCH>(buf[I], addr);
      issue<R, I + 1>(buf, addr);
    }
  }
  // column tap outermost: consecutive MFMAs go to different accumulators (the rows the input row R belongs to)
  template <int R, int KX, int KY> static __device__ __forceinline__ void consume_ky(const s16x4& x, const s16x4 (&wdg)[NTAP],
                                                                                    f32x4 (&acc)[ROWS]) {
    if constexpr (KY < KSZ) {
      if constexpr (R - KY >= 0 && (R - KY) % S == 0 && (R - KY) / S < ROWS)
        acc[(R - KY) / S] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wdg[KY * KSZ + KX], x, acc[(R - KY) / S], 0, 0, 0);
      consume_ky<R, KX, KY + 1>(x, wdg, acc);
    }
  }
  template <int R, int KX = 0> static __device__ __forceinline__ void consume(const s16x4 (&cur)[KSZ], const s16x4 (&wdg)[NTAP],
                                                                              f32x4 (&acc)[ROWS]) {
    if constexpr (KX < KSZ) {
      consume_ky<R, KX, 0>(cur[KX], wdg, acc);
      consume<R, KX + 1>(cur, wdg, acc);
    }
  }
  template <int R = 0> static __device__ __forceinline__ void step(s16x4 (&cur)[KSZ], s16x4 (&nxt)[KSZ], unsigned addr,
                                                                   const s16x4 (&wdg)[NTAP], f32x4 (&acc)[ROWS]) {
    if constexpr (R + 1 < IR) {
      issue<R + 1>(nxt, addr);
      ds_wait<KSZ>(cur);
    } else {
      ds_wait<0>(cur);
    }
    consume<R>(cur, wdg, acc);
    if constexpr (R + 1 < IR) step<R + 1>(nxt, cur, addr, wdg, acc);
  }
  static __device__ __forceinline__ void run(unsigned addr, const s16x4 (&wdg)[NTAP], f32x4 (&acc)[ROWS]) {
    s16x4 a[KSZ], b[KSZ];
    issue<0>(a, addr);
    step<0>(a, b, addr, wdg, acc);
  }
};
// One LDS-DMA piece: 64 lanes x 16 B from (scalar base + per-lane 32-bit byte offset) to LDS [lds .. lds + 1024) (gemm256p.hip's
// idiom).  Inline assembly: M0 is written in the statement that uses it; s_nop 3 covers a freshly written SGPR base.  The
// compiler does not count these loads: every consumer sits behind an explicit s_waitcnt vmcnt(0) + barrier.
__device__ __forceinline__ void dma_piece(const void* base, uint32_t voff, uint32_t lds) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base),
               "s"(__builtin_amdgcn_readfirstlane(lds))
               : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

template <int KSZ, int ROWS, int S, int HW, int PITCH, int AHEAD = (KSZ == 3 ? 3 : 2)>
struct DwRows3 {   // as DwRows, AHEAD input rows of reads in flight (lgkmcnt is a 4-bit counter: AHEAD * KSZ <= 15)
  static constexpr int IR = (ROWS - 1) * S + KSZ, NTAP = KSZ * KSZ, NB = AHEAD + 1;
  typedef DwRows<KSZ, ROWS, S, HW, PITCH> Base;
  template <int R> static __device__ __forceinline__ void step(s16x4 (&buf)[NB][KSZ], unsigned addr, const s16x4 (&wdg)[NTAP],
                                                               f32x4 (&acc)[ROWS]) {
    if constexpr (R + AHEAD < IR) Base::template issue<R + AHEAD>(buf[(R + AHEAD) % NB], addr);
    constexpr int inflight = (IR - 1 - R) < AHEAD ? (IR - 1 - R) : AHEAD;   // rows issued after row R
    ds_wait<inflight * KSZ>(buf[R % NB]);
    Base::template consume<R>(buf[R % NB], wdg, acc);
    if constexpr (R + 1 < IR) step<R + 1>(buf, addr, wdg, acc);
  }
  static __device__ __forceinline__ void run(unsigned addr, const s16x4 (&wdg)[NTAP], f32x4 (&acc)[ROWS]) {
    s16x4 buf[NB][KSZ];
    Base::template issue<0>(buf[0], addr);
    if constexpr (AHEAD >= 2 && IR > 1) Base::template issue<1>(buf[1], addr);
    if constexpr (AHEAD >= 3 && IR > 2) Base::template issue<2>(buf[2], addr);
    step<0>(buf, addr, wdg, acc);
  }
};

__device__ __forceinline__ s16x4 diag_bf16(float w, int pi) {
  const short wb = (short)f32_to_bf16(w);
  return s16x4{(short)(pi == 0 ? wb : 0), (short)(pi == 1 ? wb : 0), (short)(pi == 2 ? wb : 0), (short)(pi == 3 ? wb : 0)};
}

struct Mb3Params {
  const void* x;     // [B][H][W][CIN]
  void* out;         // [B][OH][OW][COUT]
  const void* w1;    // packed [>=Cmid][Kp1] bf16 (expand)
  const float* b1;   // [Cmid]
  const float* wd;   // [9][Cmid] fp32
  const float* bd;   // [Cmid] or null
  const void* w2;    // packed [>=COUT][Kp2] bf16 (project)
  const float* b2;   // [COUT]
  int B, H, W, OH, OW, Cmid, Kp1, Kp2;
  int residual;
  int tiles_x, tiles_y;
  int abl;           // -DESAM3_DEV builds only: phase ablation mask (ESAM3_MB3_ABL), see tools/evit_fused_bench.py
};

#ifdef ESAM3_DEV
#define MB3_ABL(bit) (p.abl & (bit))
#else
#define MB3_ABL(bit) 0
#endif

template <int S, int CIN, int COUT, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 && CIN <= 32 && S == 1) ? 3 : 2) void mbconv3_kernel(Mb3Params p) {
  typedef bf16_t T;
  constexpr int TH = 8, TW = S == 1 ? 16 : 8;          // output tile
  constexpr int OP = TH * TW;                          // 128 / 64 output pixels
  constexpr int HH = TH * S + (S == 1 ? 2 : 1), HW = TW * S + (S == 1 ? 2 : 1);  // halo 10 x 18 / 17 x 17
  constexpr int HP = HH * HW;
  constexpr int NPT = (HP + 31) / 32;                  // expand pixel tiles: 6 / 10
  constexpr int MP = NPT * 32;
  constexpr int KS = CIN / 16;                         // MFMA K steps of the expand GEMM
  constexpr int NT = COUT / 32;                        // project channel tiles
  constexpr int PT = OP / 32;                          // project pixel tiles: 4 / 2
  // bytes per halo pixel of `mid`: 64 channels + padding such that the 4 pixels a depthwise MFMA reads (S apart) start 64 bytes
  // apart modulo the 256-byte bank row (S = 1: 192, 2 x 160 = 320)
  constexpr int PITCH = S == 1 ? 192 : 160;
  static_assert(CIN % 16 == 0 && COUT % 32 == 0 && (NW == 4 || NW == 8), "shape");

  __shared__ __attribute__((aligned(16))) char mid[MP * PITCH];   // [halo pixel][64 ch] bf16, linear
  __shared__ __attribute__((aligned(16))) char dwo[OP * 128];     // [output pixel][64 ch] bf16, GEMM swizzle

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;

  const unsigned tpi = (unsigned)(p.tiles_x * p.tiles_y);
  const unsigned bid = xcd_contig(blockIdx.x, gridDim.x);
  const unsigned b = bid / tpi;
  const unsigned ti = bid - b * tpi;
  const int ty = (int)(ti / (unsigned)p.tiles_x), tx = (int)(ti - ty * p.tiles_x);
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;

  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ gw1 = reinterpret_cast<const T*>(p.w1);
  const T* __restrict__ gw2 = reinterpret_cast<const T*>(p.w2);
  T* __restrict__ go = reinterpret_cast<T*>(p.out);

  // ---- expand GEMM work units: (pixel tile, 32-channel tile) pairs, unit = wave + NW u: a wave always has the same
  //      channel tile (unit & 1 == wave & 1), so its W1 fragments are loaded once per chunk
  constexpr int UPW = (2 * NPT + NW - 1) / NW;
  const int ejt = wave & 1;
  int xoff[UPW];       // element offset of the lane's halo pixel (< 2^31, checked by the launcher), or -1 (outside the image / padding row)
  bool border = false;
#pragma unroll
  for (int u = 0; u < UPW; ++u) {
    const int pt = (wave + NW * u) >> 1;
    const int hp = pt * 32 + l31;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool in = pt < NPT && hp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    xoff[u] = in ? (int)(((b * (unsigned)p.H + iy) * (unsigned)p.W + ix) * (unsigned)CIN) : -1;
    if (pt < NPT) border |= !in;
  }
  // wave-uniform: does any expand output of this wave have to be zeroed (image border / padding rows)?
  const bool any_out = __builtin_amdgcn_ballot_w64(border) != 0ull;

  // project accumulators: PW waves along the pixel tiles, CW = NW / PW groups of channel tiles
  constexpr int PW = PT < NW ? PT : NW, CW = NW / PW;
  constexpr int NTW = (NT + CW - 1) / CW;
  f32x16_v accp[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[t][r] = 0.f;
  const int ppt = wave % PW;
  const int pnt0 = wave / PW;

  // depthwise phase: lane = (block of 4 channels, pixel within a run of 4); a wave owns ONE run of 4 output columns (QN runs per
  // row) and DROWS consecutive output rows of it
  constexpr int QN = TW / 4, WPQ = NW / QN, DROWS = TH / WPQ;
  const int blk = lane >> 2, pi = lane & 3;
  const int dq = wave % QN, drow0 = (wave / QN) * DROWS;
  const char* dbase = mid + ((drow0 * S) * HW + (4 * dq + pi) * S) * PITCH + blk * 8;

  const int nchunks = p.Cmid / 64;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * 64;
    // the depthwise weights of this chunk are requested first: their latency overlaps the expand phase
    float wdf[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wdf[t] = p.wd[t * p.Cmid + c0 + 4 * blk + pi];
    f32x4 bsv = {0.f, 0.f, 0.f, 0.f};
    if (p.bd) {
      const float4 bb = *reinterpret_cast<const float4*>(p.bd + c0 + 4 * blk);
      bsv = f32x4{bb.x, bb.y, bb.z, bb.w};
    }
    // ================= E: mid[halo px][64] = hswish(W1[c0..c0+64) . x + b1), 0 outside the image =================
    if (!MB3_ABL(2)) {
      u32x4 fw[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        fw[ks] = *reinterpret_cast<const u32x4*>(gw1 + (int64_t)(c0 + ejt * 32 + l31) * p.Kp1 + (2 * ks + g) * 8);
      f32x16_v binit;   // bias in the accumulator layout: register 4q + e = channel 8q + 4g + e of the 32-channel tile
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(p.b1 + c0 + ejt * 32 + 8 * q + 4 * g);
        binit[4 * q + 0] = bb.x; binit[4 * q + 1] = bb.y; binit[4 * q + 2] = bb.z; binit[4 * q + 3] = bb.w;
      }
      // units in groups of UG, K in blocks of KB steps: the pixel fragments of a group are requested together
      constexpr int UG = (UPW * KS <= 12) ? UPW : (KS <= 4 ? 2 : 1);
      constexpr int KB = UG == 1 && KS > 8 ? 8 : KS;
#pragma unroll
      for (int u0 = 0; u0 < UPW; u0 += UG) {
        f32x16_v acc[UG];
#pragma unroll
        for (int uu = 0; uu < UG; ++uu) acc[uu] = binit;
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += KB) {
          u32x4 fa[UG][KB];
#pragma unroll
          for (int uu = 0; uu < UG; ++uu)
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) {
              fa[uu][kk] = u32x4{0u, 0u, 0u, 0u};
              const int u = u0 + uu < UPW ? u0 + uu : 0;
              if (u0 + uu < UPW && xoff[u] >= 0 && !MB3_ABL(1))
                fa[uu][kk] = *reinterpret_cast<const u32x4*>(gx + xoff[u] + (2 * (k0 + kk) + g) * 8);
            }
#pragma unroll
          for (int uu = 0; uu < UG; ++uu)
#pragma unroll
            for (int kk = 0; kk < KB; ++kk) MmaOps<T>::mma(fw[k0 + kk], fa[uu][kk], acc[uu]);
        }
#pragma unroll
        for (int uu = 0; uu < UG; ++uu) {
          const int u = u0 + uu;
          if (u >= UPW) continue;
          const int pt = (wave + NW * u) >> 1;
          if (pt >= NPT) continue;  // wave-uniform
          const int hp = pt * 32 + l31;
          float v[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = hsw(acc[uu][e]);
          if (any_out) {
            const bool in = xoff[u] >= 0;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = in ? v[e] : 0.f;
          }
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            const uint32_t a0 = pack_bf16x2(v[8 * qp + 0], v[8 * qp + 1]), a1 = pack_bf16x2(v[8 * qp + 2], v[8 * qp + 3]);
            const uint32_t c0_ = pack_bf16x2(v[8 * qp + 4], v[8 * qp + 5]), c1_ = pack_bf16x2(v[8 * qp + 6], v[8 * qp + 7]);
            auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};  // channels ejt*32 + 16qp + 8g .. +8 of halo pixel hp
            const int c = ejt * 4 + qp * 2 + g;
            *reinterpret_cast<u32x4*>(mid + hp * PITCH + (c << 4)) = o;
          }
        }
      }
    }
    __syncthreads();

    // ================= D: dwo[out px][64] = hswish(dw3x3(mid) + bd) on v_mfma_f32_4x4x4_16b_bf16 =================
    if (!MB3_ABL(4)) {
      s16x4 wdg[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wdg[t] = diag_bf16(wdf[t], pi);
      f32x4 acc[DROWS];
#pragma unroll
      for (int i = 0; i < DROWS; ++i) acc[i] = bsv;
      DwRows<3, DROWS, S, HW, PITCH>::run(lds_addr(dbase), wdg, acc);
#pragma unroll
      for (int r = 0; r < DROWS; ++r) {
        const int op = (drow0 + r) * TW + 4 * dq + pi;
        uint2 o;
        o.x = pack_bf16x2(hsw(acc[r][0]), hsw(acc[r][1]));
        o.y = pack_bf16x2(hsw(acc[r][2]), hsw(acc[r][3]));
        *reinterpret_cast<uint2*>(dwo + op * 128 + swz(op, blk >> 1) + (blk & 1) * 8) = o;
      }
    }
    __syncthreads();

    // ================= P: acc[out px][Cout] += dwo . W2[:, c0..c0+64)^T =================
    if (!MB3_ABL(8)) {
      const int prow = ppt * 32 + l31;
      u32x4 fd[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) fd[kc] = *reinterpret_cast<const u32x4*>(dwo + prow * 128 + swz(prow, kc * 2 + g));
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int nt = pnt0 + t * CW;
        if (nt >= NT) continue;  // wave-uniform
        u32x4 fw2[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
          fw2[kc] = *reinterpret_cast<const u32x4*>(gw2 + (int64_t)(nt * 32 + l31) * p.Kp2 + c0 + (kc * 2 + g) * 8);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) MmaOps<T>::mma(fw2[kc], fd[kc], accp[t]);
      }
    }
    // (the barrier after the next chunk's expand phase orders these dwo reads before the next depthwise phase)
  }

  // ================= out = acc + b2 (+ x): 16-byte NHWC stores =================
  if (!MB3_ABL(16) || accp[0][0] == 12345.f) {
    const int op = ppt * 32 + l31;
    const int oy = oy0 + op / TW, ox = ox0 + op % TW;
    const bool ok = oy < p.OH && ox < p.OW;
    const int64_t opix = ((int64_t)b * p.OH + oy) * p.OW + ox;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int nt = pnt0 + t * CW;
      if (nt >= NT) continue;
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(p.b2 + nt * 32 + 8 * q + 4 * g);
        v[4 * q + 0] = accp[t][4 * q + 0] + bb.x; v[4 * q + 1] = accp[t][4 * q + 1] + bb.y;
        v[4 * q + 2] = accp[t][4 * q + 2] + bb.z; v[4 * q + 3] = accp[t][4 * q + 3] + bb.w;
        if (p.residual && ok) {  // identity shortcut: the same pixel of the input (stride 1, Cin == Cout)
          const uint2 u = *reinterpret_cast<const uint2*>(gx + opix * CIN + nt * 32 + 8 * q + 4 * g);
          v[4 * q + 0] += __uint_as_float(u.x << 16); v[4 * q + 1] += __uint_as_float(u.x & 0xffff0000u);
          v[4 * q + 2] += __uint_as_float(u.y << 16); v[4 * q + 3] += __uint_as_float(u.y & 0xffff0000u);
        }
      }
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const uint32_t a0 = pack_bf16x2(v[8 * qp + 0], v[8 * qp + 1]), a1 = pack_bf16x2(v[8 * qp + 2], v[8 * qp + 3]);
        const uint32_t c0_ = pack_bf16x2(v[8 * qp + 4], v[8 * qp + 5]), c1_ = pack_bf16x2(v[8 * qp + 6], v[8 * qp + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
        if (ok) {
          const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(go + opix * COUT + nt * 32 + 16 * qp + 8 * g) = o;
        }
      }
    }
  }
}

// =====================================================================================================================
// LiteMLA, pass 1.  Workgroup = 8 waves, one 8 x 16 tile of output pixels of one image; dim = 16, heads = C / 16.
// Loop over chunks of TWO heads = 96 qkv channels ([q16 k16 v16] per head, the reference's channel order, ops.py:596-612):
//   E   qkv chunk = Wqkv[96 rows] . x on the tile + 2-pixel halo (12 x 20 = 240 -> 8 pixel tiles of 32, one per wave, the
//       x fragments stay in registers for all chunks) -> mid[halo px][96] bf16 (no bias: pixels outside the image give 0 =
//       the depthwise conv's zero padding)
//   D   aggreg.0.0: depthwise 5x5 of mid -> dwo[px][96] on v_mfma_f32_4x4x4_16b_bf16; block slot = (4-channel group of 24,
//       run of 4 columns, all 8 rows of it): 96 slots = 6 waves; every operand read feeds up to 5 output rows
//   P   aggreg.0.1: grouped 1x1 (16 -> 16 per group) dwo -> ago[px][96]: one v_mfma_f32_16x16x16_bf16 per (group, tile row)
//   KVQ relu(q) of the two scales (mid centre, ago) -> qms[b][token][chunk*64 + scale*32 + head*16 + d];
//       kv partial of each of the 4 (scale, head) groups over this tile's pixels: D[dv][dk] += V^T . relu(K) with the pixels as
//       the MFMA's K dimension (operands by ds_read_b64_tr_b16 from the [px][ch] LDS rows), ksum with an all-ones operand
// LDS pitch of the three tiles = 192 bytes = the 96 channels: 4 horizontally adjacent pixels start 64 bytes apart modulo the
// 256-byte bank row, so the depthwise reads are conflict-free without padding.
// =====================================================================================================================
struct Mla1Params {
  const void* x;        // [B][H][W][C] bf16
  const void* wqkv;     // packed [>=3C][Kpq] bf16
  const float* wdw;     // [25][3C] fp32 (aggreg.0.0)
  const void* wgrp;     // [3C][Kpg] bf16: row = output channel, 16 inputs of its group (aggreg.0.1)
  void* qms;            // bf16 relu(q): [B][tiles][heads / 2 chunks][4 groups = scale*2 + head&1][8 tile rows][16 px][16 ch]
  float* kvp;           // [B][tiles*2][2*heads][272] fp32 partials, [dv][dk] with row 16 = ksum
  int B, H, W, Kpq, Kpg;
  int tiles_x, tiles_y;
  int abl;              // -DESAM3_DEV builds only: phase ablation mask (ESAM3_MLA1_ABL)
};
#ifdef ESAM3_DEV
#define MLA1_ABL(bit) (p.abl & (bit))
#else
#define MLA1_ABL(bit) 0
#endif

typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
__device__ __forceinline__ s16x4 lds_tr16(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p));
}
__device__ __forceinline__ s16x4 relu_bf16x4(s16x4 v) {   // sign bit set -> 0 (v_pk_max_i16 with 0)
  return __builtin_elementwise_max(v, s16x4{0, 0, 0, 0});
}

template <int C>
__global__ __launch_bounds__(512, 2) void mla1_kernel(Mla1Params p) {
  typedef bf16_t T;
  constexpr int HEADS = C / 16, C3 = 3 * C, KS = C / 16, NCH = HEADS / 2;
  constexpr int TH = 8, TW = 16, HH = TH + 4, HW = TW + 4, HP = HH * HW;   // 12 x 20 halo
  constexpr int PITCH = 192;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* mid = smem;                    // [256 halo px][192 B]
  char* dwo = mid + 256 * PITCH;       // [128 px][192 B]
  char* ago = dwo + 128 * PITCH;       // [128 px][192 B]
  char* wqs = ago + 128 * PITCH;       // this chunk's 96 rows of Wqkv: [96][C] bf16, 16-byte slot ^ (row & 15), filled by LDS-DMA

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5, l15 = lane & 15, kg = lane >> 4;

  const unsigned tpi = (unsigned)(p.tiles_x * p.tiles_y);
  const unsigned bid = xcd_contig(blockIdx.x, gridDim.x);
  const unsigned b = bid / tpi;
  const unsigned ti = bid - b * tpi;
  const int ty = (int)(ti / (unsigned)p.tiles_x), tx = (int)(ti - ty * p.tiles_x);
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int N = p.H * p.W;

  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ gwq = reinterpret_cast<const T*>(p.wqkv);
  const T* __restrict__ gwg = reinterpret_cast<const T*>(p.wgrp);
  T* __restrict__ gq = reinterpret_cast<T*>(p.qms);

  // weights of a chunk: 96 rows x C columns = 12 / 24 KB = 12 / 24 DMA pieces (wave w issues pieces w, w + 8, ...); the swizzle
  // is applied to the source address
  const uint32_t wq_lds = lds_addr(wqs);
  auto dma_wq = [&](int ch) {
#pragma unroll
    for (int j = 0; j < (96 * C * 2 / 1024 + 7) / 8; ++j) {
      const int piece = wave + 8 * j;
      if (piece < 96 * C * 2 / 1024) {   // wave-uniform
        const int byte = piece * 1024 + lane * 16;
        const int row = byte / (C * 2), pslot = (byte - row * (C * 2)) >> 4;
        const uint32_t voff = (uint32_t)(((ch * 96 + row) * p.Kpq + ((pslot ^ (row & 15)) << 3)) * 2);
        dma_piece(gwq, voff, wq_lds + (uint32_t)piece * 1024u);
      }
    }
  };
  dma_wq(0);
  // ---- x fragments of this wave's halo pixel tile (pixels 32 wave .. + 31), kept for all chunks ----
  u32x4 fa[KS];
  {
    const int hp = wave * 32 + l31;
    const int hy = hp / HW, hx = hp - hy * HW;
    const int iy = oy0 - 2 + hy, ix = ox0 - 2 + hx;
    const bool in = hp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const T* px = gx + ((int64_t)(b * (unsigned)p.H + (in ? iy : 0)) * p.W + (in ? ix : 0)) * C;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fa[ks] = u32x4{0u, 0u, 0u, 0u};
      if (in) fa[ks] = *reinterpret_cast<const u32x4*>(px + (2 * ks + g) * 8);
    }
  }
  // depthwise phase: waves 0..5, block slot = wave*16 + blk -> (channel group cg of 24, column run dq of 4)
  const int blk = lane >> 2, pi = lane & 3;
  const int dslot = wave * 16 + blk;
  const int cg = dslot % 24, dq = dslot / 24;   // (4-channel group, run of 4 columns): all 8 output rows of that run
  const char* dbase = mid + (4 * dq + pi) * PITCH + cg * 8;
  // this lane's output pixel in the P phase (tile row = wave) inside the image?
  const bool p_in = (oy0 + wave) < p.H && (ox0 + l15) < p.W;

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // Wqkv chunk 0 is in LDS
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 96;   // first qkv channel of the chunk (heads 2ch, 2ch + 1)
    // the grouped conv's six 16 x 16 weight blocks of this chunk (A operands of the P phase): requested now, used three
    // phases later (the P phase cost 35 of 127 us waiting for them, profiles/r04/mla_ablation_e.txt)
    s16x4 wga[6];
#pragma unroll
    for (int gi = 0; gi < 6; ++gi) wga[gi] = *reinterpret_cast<const s16x4*>(gwg + (int64_t)(c0 + gi * 16 + l15) * p.Kpg + 4 * kg);
    // ================= E: mid[halo px][96] = Wqkv[c0 .. c0+96) . x =================
    if (!MLA1_ABL(1))
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      f32x16_v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int wr = j * 32 + l31;
      const char* wrow = wqs + wr * (C * 2);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 fw = *reinterpret_cast<const u32x4*>(wrow + (((2 * ks + g) ^ (wr & 15)) << 4));
        MmaOps<T>::mma(fw, fa[ks], acc);
      }
      const int hp = wave * 32 + l31;
      u32x4 o[2];
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const uint32_t a0 = pack_bf16x2(acc[8 * qp + 0], acc[8 * qp + 1]), a1 = pack_bf16x2(acc[8 * qp + 2], acc[8 * qp + 3]);
        const uint32_t c0_ = pack_bf16x2(acc[8 * qp + 4], acc[8 * qp + 5]), c1_ = pack_bf16x2(acc[8 * qp + 6], acc[8 * qp + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
        o[qp] = u32x4{s0[0], s1[0], s0[1], s1[1]};  // channels j*32 + 16qp + 8g .. +8 of halo pixel hp
      }
      // lane-dependent order of the two pieces: 2-way instead of 4-way bank conflicts (pixel pitch 192 B = 64 mod 128)
      const bool flip = (l31 >> 1) & 1;
      const u32x4 w0 = flip ? o[1] : o[0], w1v = flip ? o[0] : o[1];
      char* rowp = mid + hp * PITCH + ((j * 4 + g) << 4);
      *reinterpret_cast<u32x4*>(rowp + (flip ? 32 : 0)) = w0;
      *reinterpret_cast<u32x4*>(rowp + (flip ? 0 : 32)) = w1v;
    }
    __syncthreads();
    if (ch + 1 < NCH) dma_wq(ch + 1);   // the expand phase is done with the buffer; lands under the other three phases

    // ================= D: dwo[px][96] = dw5x5(mid) =================
    if (wave < 6 && !MLA1_ABL(2)) {
      s16x4 wdg[25];
#pragma unroll
      for (int t = 0; t < 25; ++t) wdg[t] = diag_bf16(p.wdw[t * C3 + c0 + 4 * cg + pi], pi);
      f32x4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      DwRows3<5, 8, 1, HW, PITCH, 2>::run(lds_addr(dbase), wdg, acc);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int op = r * TW + 4 * dq + pi;
        uint2 o;
        o.x = pack_bf16x2(acc[r][0], acc[r][1]);
        o.y = pack_bf16x2(acc[r][2], acc[r][3]);
        *reinterpret_cast<uint2*>(dwo + op * PITCH + cg * 8) = o;
      }
    }
    __syncthreads();

    // ================= P: ago[px][96] = grouped 1x1 of dwo; tile row = wave, 6 groups of 16 channels =================
    if (!MLA1_ABL(4)) {
      const int op = wave * 16 + l15;
#pragma unroll
      for (int gi = 0; gi < 6; ++gi) {
        const s16x4 xb = *reinterpret_cast<const s16x4*>(dwo + op * PITCH + gi * 32 + kg * 8);
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wga[gi], xb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        uint2 o;   // channels gi*16 + 4kg .. +4 of pixel op; pixels of the tile that lie outside the image must not reach kv
        o.x = p_in ? pack_bf16x2(d[0], d[1]) : 0u;
        o.y = p_in ? pack_bf16x2(d[2], d[3]) : 0u;
        *reinterpret_cast<uint2*>(ago + op * PITCH + gi * 32 + kg * 8) = o;
      }
    }
    __syncthreads();

    // ================= KVQ =================
    // relu(q) -> qms[b][tile][chunk][gl = scale*2 + head][tile row][px][16 ch]: the 16 pixels x 32 bytes of a (group, tile row) are
    // contiguous = exactly what one MFMA B-operand load of pass 2 reads (the token-major layout gave it 32-byte pieces 512
    // bytes apart).  item = (gl, row, px, half): 32 consecutive lanes write one 512-byte run
    if (!MLA1_ABL(8))
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + 512 * it;
      const int half = item & 1, pxx = (item >> 1) & 15, py = (item >> 5) & 7, gl = item >> 8;
      const int scale = gl >> 1, hh = gl & 1;
      const int px = py * 16 + pxx;
      const char* src = (scale ? ago + px * PITCH : mid + ((py + 2) * HW + pxx + 2) * PITCH) + hh * 96 + half * 16;
      const uint4 v = *reinterpret_cast<const uint4*>(src);
      const s16x4 lo = relu_bf16x4(__builtin_bit_cast(s16x4, make_uint2(v.x, v.y)));
      const s16x4 hi = relu_bf16x4(__builtin_bit_cast(s16x4, make_uint2(v.z, v.w)));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      *reinterpret_cast<uint4*>(gq + (((int64_t)b * tpi + ti) * NCH + ch) * 8192 + item * 8) = make_uint4(l2.x, l2.y, h2.x, h2.y);
    }
    // kv partials: wave = (group gsel of 4, pixel half); 4 tile rows of 16 pixels each = 4 MFMA steps
    if (!MLA1_ABL(16)) {
      const int gsel = wave & 3, half = wave >> 2;
      const int scale = gsel >> 1, hh = gsel & 1;
      f32x4 akv = {0.f, 0.f, 0.f, 0.f}, aks = {0.f, 0.f, 0.f, 0.f};
      const s16x4 ones = {(short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80};
      const int jpx = 4 * kg + (l15 >> 2);   // the pixel this lane addresses for the transposing read
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int ry = half * 4 + st;
        const char* rb = (scale ? ago + (ry * 16 + jpx) * PITCH : mid + ((ry + 2) * HW + 2 + jpx) * PITCH) + hh * 96 + (lane & 3) * 8;
        const s16x4 kf = relu_bf16x4(lds_tr16(rb + 32));
        const s16x4 vf = lds_tr16(rb + 64);
        akv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, kf, akv, 0, 0, 0);
        aks = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ones, kf, aks, 0, 0, 0);
      }
      const int gnat = scale * HEADS + 2 * ch + hh;
      float* o = p.kvp + (((int64_t)b * tpi + ti) * 2 + half) * (int64_t)(2 * HEADS * 272) + gnat * 272;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[(4 * kg + i) * 16 + l15] = akv[i];
      if (kg == 0) o[256 + l15] = aks[0];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next chunk's weights
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// mla1v (round 6): pass 1 rebuilt around its phase ablation (profiles/r06/mla1_ablation_before.txt: at 63^2 x 128 the depthwise phase
// was 44 of the 84 us that the five phases account for and the grouped 1x1 22, the qkv GEMM 10; nothing saturated, every phase a chain of
// latencies at two waves per SIMD).  Same arithmetic in the same order as mla1_kernel (bit-identical qms / kvp); what changed:
//   * depthwise 5x5 and grouped 1x1 are ONE phase: lane = (4-channel block c4 of the 16-channel group, pixel column n of 16), so the
//     depthwise accumulator of a lane (4 channels of one pixel) is, packed to bf16, exactly the B operand of the grouped conv's
//     v_mfma_f32_16x16x16_bf16 (k = 4 c4 .. + 3 of column n): the dwo tile, its barrier, its 4-way conflicted reads and the two idle waves
//     are gone.  Work units = (group, output row); the 6 x 8 = 48 of a chunk go to the 8 waves as runs of 6 consecutive rows (a run
//     crosses at most one group boundary: segments of 6, 2 + 4, 4 + 2, 6 rows -- every loaded operand still feeds up to 5 rows);
//   * pixel pitch of the LDS tiles 208 B (= 16 mod 256): 16 consecutive pixels x 16 bytes tile the 64 banks exactly, so the depthwise
//     ds_read_b64, the expand phase's ds_write_b128 and the grouped conv's writes are conflict-free (192 B: 4 distinct residues for 16
//     pixels; SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.50 - 0.57);
//   * the 25 depthwise taps of a chunk are staged cooperatively (5 loads per thread, one chunk ahead) as bf16 into a double-buffered LDS
//     table: the depthwise phase no longer starts with 25 global loads + 300 VALU instructions of operand building per lane;
//   * relu(q) and the kv partials of scale 0 (which only need the qkv tile) are issued at the start of the depthwise phase; the wait for
//     the next chunk's weight DMA sits BEFORE the late stores, so no wave ends a chunk waiting for its own stores to be acknowledged.
// LDS: mid 240 x 208 + ago 128 x 208 + Wqkv chunk 96 x C x 2 + taps 2 x 25 x 96 x 2 = 108 / 132 KB.
// ---------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512, 2) void mla1v_kernel(Mla1Params p) {
  typedef bf16_t T;
  constexpr int HEADS = C / 16, C3 = 3 * C, KS = C / 16, NCH = HEADS / 2;
  constexpr int TH = 8, TW = 16, HH = TH + 4, HW = TW + 4, HP = HH * HW;   // 12 x 20 halo
  constexpr int PITCH = 208;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* mid = smem;                    // [240 halo px][208 B]: 96 qkv channels of the chunk
  char* ago = mid + HP * PITCH;        // [128 px][208 B]: the aggregated (second scale) 96 channels
  char* wqs = ago + 128 * PITCH;       // this chunk's 96 rows of Wqkv: [96][C] bf16, 16-byte slot ^ (row & 15), filled by LDS-DMA
  uint16_t* swd = reinterpret_cast<uint16_t*>(wqs + 96 * C * 2);   // [2][25 taps][96 ch] bf16

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5, l15 = lane & 15, kg = lane >> 4;

  // persistent: XCD xcd owns the contiguous tile range [first, first + cnt), its workgroups stride through it (63 x 63 x 128: four tiles per
  // workgroup -- the next tile's pixel fragments are requested under the last chunk's depthwise phase, the weight / tap streams run across
  // the tile boundary; 32 x 32 x 256: one tile each)
  const unsigned tpi = (unsigned)(p.tiles_x * p.tiles_y), ntiles = tpi * (unsigned)p.B;
  const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const unsigned nx = nwg / 8 + (xcd < nwg % 8 ? 1u : 0u);
  const unsigned tq_ = ntiles / 8, tr_ = ntiles % 8;
  const unsigned first = xcd < tr_ ? xcd * (tq_ + 1) : tr_ * (tq_ + 1) + (xcd - tr_) * tq_, cnt = tq_ + (xcd < tr_ ? 1u : 0u);
  if (wi >= cnt) return;
  unsigned b = 0, ti = 0;
  int oy0 = 0, ox0 = 0;
  bool col_in = false;

  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ gwq = reinterpret_cast<const T*>(p.wqkv);
  const T* __restrict__ gwg = reinterpret_cast<const T*>(p.wgrp);
  T* __restrict__ gq = reinterpret_cast<T*>(p.qms);

  const uint32_t wq_lds = lds_addr(wqs);
  auto dma_wq = [&](int ch) {
#pragma unroll
    for (int j = 0; j < (96 * C * 2 / 1024 + 7) / 8; ++j) {
      const int piece = wave + 8 * j;
      if (piece < 96 * C * 2 / 1024) {   // wave-uniform
        const int byte = piece * 1024 + lane * 16;
        const int row = byte / (C * 2), pslot = (byte - row * (C * 2)) >> 4;
        const uint32_t voff = (uint32_t)(((ch * 96 + row) * p.Kpq + ((pslot ^ (row & 15)) << 3)) * 2);
        dma_piece(gwq, voff, wq_lds + (uint32_t)piece * 1024u);
      }
    }
  };
  // depthwise taps of chunk ch: 25 x 96 values, element e = tap * 96 + channel, thread t owns e = t + 512 k
  float wst[5];
  auto load_taps = [&](int ch) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int e = tid + 512 * k;
      const int tap = e / 96, cc = e - tap * 96;
      wst[k] = e < 2400 ? p.wdw[tap * C3 + ch * 96 + cc] : 0.f;
    }
  };
  auto store_taps = [&](int buf) {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int e = tid + 512 * k;
      if (e < 2400) swd[buf * 2400 + e] = f32_to_bf16(wst[k]);
    }
  };
  dma_wq(0);
  load_taps(0);
  // ---- x fragments of this wave's halo pixel tile (pixels 32 wave .. + 31) of tile `tile`, kept for all its chunks ----
  u32x4 fa[KS];
  const int hp_e = wave * 32 + l31;
  auto load_fa = [&](unsigned tile) {
    const unsigned b_ = tile / tpi, ti_ = tile - b_ * tpi;
    const int ty_ = (int)(ti_ / (unsigned)p.tiles_x), tx_ = (int)(ti_ - ty_ * p.tiles_x);
    const int hy = hp_e / HW, hx = hp_e - hy * HW;
    const int iy = ty_ * TH - 2 + hy, ix = tx_ * TW - 2 + hx;
    const bool in = hp_e < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
    const T* px = gx + ((int64_t)(b_ * (unsigned)p.H + (in ? iy : 0)) * p.W + (in ? ix : 0)) * C;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      fa[ks] = u32x4{0u, 0u, 0u, 0u};
      if (in) fa[ks] = *reinterpret_cast<const u32x4*>(px + (2 * ks + g) * 8);
    }
  };
  auto set_tile = [&](unsigned tile) {
    b = tile / tpi;
    ti = tile - b * tpi;
    const int ty_ = (int)(ti / (unsigned)p.tiles_x), tx_ = (int)(ti - ty_ * p.tiles_x);
    oy0 = ty_ * TH; ox0 = tx_ * TW;
    col_in = (ox0 + l15) < p.W;
  };
  set_tile(first + wi);
  load_fa(first + wi);
  // depthwise + grouped phase: lane = (c4 = 4-channel block of a 16-channel group, column n); the wave's run of 6 (group, row) units
  const int pi = lane & 3;
  const int wq4 = wave & 3, gi0 = (wave >> 2) * 3;
  const int giA = gi0 + (wq4 >= 2 ? (wq4 == 2 ? 1 : 2) : 0);   // group of the run's first segment
  const int giB = gi0 + (wq4 == 1 ? 1 : 2);                    // group of the second segment (waves 1, 2 of a quad)

  store_taps(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();   // Wqkv chunk 0 and the taps of chunk 0 are in LDS
  static_assert(NCH % 2 == 0, "the tap table's buffer parity runs across tile boundaries");
  constexpr bool PERSIST = C == 128;   // C = 256 is at the register limit: launched with one workgroup per tile (25 spilled registers otherwise)
  for (unsigned t = wi; t < cnt; t += nx) {
  const bool has_next = PERSIST && t + nx < cnt;
  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 96;   // first qkv channel of the chunk (heads 2ch, 2ch + 1)
    const bool more = ch + 1 < NCH || has_next;   // the flat (tile, chunk) stream has a next chunk
    const int nxt = ch + 1 < NCH ? ch + 1 : 0;
    // the grouped conv's 16 x 16 weight blocks of this wave's (at most two) groups: requested now, used after the depthwise walk
    const s16x4 wgaA = *reinterpret_cast<const s16x4*>(gwg + (int64_t)(c0 + giA * 16 + l15) * p.Kpg + 4 * kg);
    const s16x4 wgaB = *reinterpret_cast<const s16x4*>(gwg + (int64_t)(c0 + giB * 16 + l15) * p.Kpg + 4 * kg);
    if (more) load_taps(nxt);
    // ================= E: mid[halo px][96] = Wqkv[c0 .. c0+96) . x =================
    if (!MLA1_ABL(1))
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      f32x16_v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int wr = j * 32 + l31;
      const char* wrow = wqs + wr * (C * 2);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 fw = *reinterpret_cast<const u32x4*>(wrow + (((2 * ks + g) ^ (wr & 15)) << 4));
        MmaOps<T>::mma(fw, fa[ks], acc);
      }
      u32x4 o[2];
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const uint32_t a0 = pack_bf16x2(acc[8 * qp + 0], acc[8 * qp + 1]), a1 = pack_bf16x2(acc[8 * qp + 2], acc[8 * qp + 3]);
        const uint32_t c0_ = pack_bf16x2(acc[8 * qp + 4], acc[8 * qp + 5]), c1_ = pack_bf16x2(acc[8 * qp + 6], acc[8 * qp + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
        o[qp] = u32x4{s0[0], s1[0], s0[1], s1[1]};  // channels j*32 + 16qp + 8g .. +8 of halo pixel hp
      }
      if (hp_e < HP) {
        char* rowp = mid + hp_e * PITCH + ((j * 4 + g) << 4);
        *reinterpret_cast<u32x4*>(rowp) = o[0];
        *reinterpret_cast<u32x4*>(rowp + 32) = o[1];
      }
    }
    if (more) store_taps((ch + 1) & 1);   // read by the NEXT chunk's depthwise phase (behind two barriers)
    __syncthreads();
    if (more) dma_wq(nxt);   // the expand phase is done with the buffer; lands under the other phases
    if (ch + 1 == NCH && has_next) load_fa(first + t + nx);   // this tile's fragments are dead: the next tile's arrive under the phases below

    // ================= early KVQ: scale 0 (the qkv tile itself) =================
    auto q_store = [&](int item) {   // relu(q) -> qms[b][tile][chunk][gl = scale*2 + head][tile row][px][16 ch], item = (gl, row, px, half)
      const int half = item & 1, pxx = (item >> 1) & 15, py = (item >> 5) & 7, gl = item >> 8;
      const int scale = gl >> 1, hh = gl & 1;
      const int px = py * 16 + pxx;
      const char* src = (scale ? ago + px * PITCH : mid + ((py + 2) * HW + pxx + 2) * PITCH) + hh * 96 + half * 16;
      const uint4 v = *reinterpret_cast<const uint4*>(src);
      const s16x4 lo = relu_bf16x4(__builtin_bit_cast(s16x4, make_uint2(v.x, v.y)));
      const s16x4 hi = relu_bf16x4(__builtin_bit_cast(s16x4, make_uint2(v.z, v.w)));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      *reinterpret_cast<uint4*>(gq + (((int64_t)b * tpi + ti) * NCH + ch) * 8192 + item * 8) = make_uint4(l2.x, l2.y, h2.x, h2.y);
    };
    auto kv_partial = [&](int scale, int hh, int half) {   // 4 tile rows of 16 pixels each = 4 MFMA steps
      f32x4 akv = {0.f, 0.f, 0.f, 0.f}, aks = {0.f, 0.f, 0.f, 0.f};
      const s16x4 ones = {(short)0x3F80, (short)0x3F80, (short)0x3F80, (short)0x3F80};
      const int jpx = 4 * kg + (l15 >> 2);   // the pixel this lane addresses for the transposing read
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int ry = half * 4 + st;
        const char* rb = (scale ? ago + (ry * 16 + jpx) * PITCH : mid + ((ry + 2) * HW + 2 + jpx) * PITCH) + hh * 96 + (lane & 3) * 8;
        const s16x4 kf = relu_bf16x4(lds_tr16(rb + 32));
        const s16x4 vf = lds_tr16(rb + 64);
        akv = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf, kf, akv, 0, 0, 0);
        aks = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ones, kf, aks, 0, 0, 0);
      }
      const int gnat = scale * HEADS + 2 * ch + hh;
      float* o = p.kvp + (((int64_t)b * tpi + ti) * 2 + half) * (int64_t)(2 * HEADS * 272) + gnat * 272;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[(4 * kg + i) * 16 + l15] = akv[i];
      if (kg == 0) o[256 + l15] = aks[0];
    };
    if (!MLA1_ABL(8)) q_store(tid);
    if (!MLA1_ABL(16) && wave < 4) kv_partial(0, wave & 1, wave >> 1);

    // ================= DP: ago[px][96] = grouped 1x1 of dw5x5(mid), in registers =================
    if (!MLA1_ABL(2)) {
      const uint16_t* tb = swd + (ch & 1) * 2400;
      auto segment = [&](auto rows_c, int gi, int r0, const s16x4& wga) {
        constexpr int R = decltype(rows_c)::value;
        s16x4 wdg[25];
#pragma unroll
        for (int t = 0; t < 25; ++t) {
          const uint64_t wv = (uint64_t)tb[t * 96 + gi * 16 + 4 * kg + pi] << (16 * pi);
          wdg[t] = __builtin_bit_cast(s16x4, wv);
        }
        f32x4 acc[R];
#pragma unroll
        for (int i = 0; i < R; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        DwRows3<5, R, 1, HW, PITCH, 2>::run(lds_addr(mid + (r0 * HW + l15) * PITCH + gi * 32 + kg * 8), wdg, acc);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          s16x4 xb;
          if (MLA1_ABL(4)) {
            xb = s16x4{0, 0, 0, 0};
          } else {
            const uint2 pk = make_uint2(pack_bf16x2(acc[r][0], acc[r][1]), pack_bf16x2(acc[r][2], acc[r][3]));
            xb = __builtin_bit_cast(s16x4, pk);
          }
          const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wga, xb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          // channels gi*16 + 4kg .. +4 of pixel (r0 + r, l15); pixels of the tile that lie outside the image must not reach kv
          const bool p_in = col_in && (oy0 + r0 + r) < p.H;
          uint2 o;
          o.x = p_in ? pack_bf16x2(d[0], d[1]) : 0u;
          o.y = p_in ? pack_bf16x2(d[2], d[3]) : 0u;
          *reinterpret_cast<uint2*>(ago + ((r0 + r) * 16 + l15) * PITCH + gi * 32 + kg * 8) = o;
        }
      };
      if (wq4 == 0) {
        segment(std::integral_constant<int, 6>{}, giA, 0, wgaA);
      } else if (wq4 == 1) {
        segment(std::integral_constant<int, 2>{}, giA, 6, wgaA);
        segment(std::integral_constant<int, 4>{}, giB, 0, wgaB);
      } else if (wq4 == 2) {
        segment(std::integral_constant<int, 4>{}, giA, 4, wgaA);
        segment(std::integral_constant<int, 2>{}, giB, 0, wgaB);
      } else {
        segment(std::integral_constant<int, 6>{}, giA, 2, wgaA);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next chunk's weights (issued a phase ago) have landed
    __syncthreads();

    // ================= late KVQ: scale 1 =================
    if (!MLA1_ABL(8)) q_store(512 + tid);
    if (!MLA1_ABL(16) && wave >= 4) kv_partial(1, wave & 1, (wave >> 1) & 1);
    // no barrier here: these reads touch `ago` only, which the next chunk writes behind ITS first barrier; `mid`, the weight buffer and
    // the tap table the next expand phase writes were last read in front of the barrier above
  }
  if (!PERSIST) break;
  if (has_next) set_tile(first + t + nx);
  }
}

// kv[b][g] = sum of the tile partials in a fixed order; written as the bf16 hi / lo MFMA operands of pass 2:
// tab[b][g][op][lane][4], op 0/1 = kv hi / lo (lane (m = dv, kq): kv[m][4kq .. 4kq+3]), op 2/3 = ksum hi / lo (every row m)
// Round 6: 1024 threads per (image, group), SAME summation order as the 256-thread form of round 4 (kept below for the dev-build A/B): that
// form keeps eight interleaved partial sums s_j = p[j] + p[j + 8] + ... and combines them as ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
// here thread quarter q owns s_2q and s_2q+1 (two rows of loads in flight per step, the steps unrolled by four so that sixteen loads are
// out before the first add), forms t_q = s_2q + s_2q+1 and the quarters meet in LDS as (t0 + t1) + (t2 + t3) -- bit-identical tables.  The
// 256-thread form walked the 64 partials of a 63 x 63 image in 8 dependent rounds of L2 latency (13.7 us per launch, 90 % of the wave time
// parked: profiles/r06/pmc_backbone_before.txt); 9.4 us now.
__global__ __launch_bounds__(1024) void mla_kvprep_kernel(const float* __restrict__ kvp, bf16_t* __restrict__ tab, int P, int G) {
  __shared__ float red[2][3][256];
  const int bg = blockIdx.x;
  const int b = bg / G, gi = bg - b * G;
  const int t = threadIdx.x & 255, part = threadIdx.x >> 8, m = t >> 4, k = t & 15;
  const float* src = kvp + ((int64_t)b * P * G + gi) * 272;
  const int64_t st = (int64_t)G * 272;
  float sa = 0.f, sb = 0.f, ka = 0.f, kb = 0.f;   // s_2q, s_2q+1 and the ksum rows of the same partials
  const int j0 = 2 * part, j1 = 2 * part + 1;
  int q0 = 0;
  for (; q0 + 32 <= P; q0 += 32) {   // four steps of eight partials: all sixteen loads first, the adds in the original order
    float va[4], vb[4], wa[4], wb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      va[u] = src[(q0 + 8 * u + j0) * st + m * 16 + k];
      vb[u] = src[(q0 + 8 * u + j1) * st + m * 16 + k];
      wa[u] = src[(q0 + 8 * u + j0) * st + 256 + k];
      wb[u] = src[(q0 + 8 * u + j1) * st + 256 + k];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { sa += va[u]; sb += vb[u]; ka += wa[u]; kb += wb[u]; }
  }
  for (; q0 < P; q0 += 8) {
    if (q0 + j0 < P) { sa += src[(q0 + j0) * st + m * 16 + k]; ka += src[(q0 + j0) * st + 256 + k]; }
    if (q0 + j1 < P) { sb += src[(q0 + j1) * st + m * 16 + k]; kb += src[(q0 + j1) * st + 256 + k]; }
  }
  float s = sa + sb, sk = ka + kb;
  if (part) {
    red[0][part - 1][t] = s;
    red[1][part - 1][t] = sk;
  }
  __syncthreads();
  if (part) return;
  s = (s + red[0][0][t]) + (red[0][1][t] + red[0][2][t]);
  sk = (sk + red[1][0][t]) + (red[1][1][t] + red[1][2][t]);
  bf16_t* o = tab + (int64_t)bg * 4 * 256;
  const int li = (m + 16 * (k >> 2)) * 4 + (k & 3);
  const bf16_t h = f32_to_bf16(s), hk = f32_to_bf16(sk);
  o[li] = h;
  o[256 + li] = f32_to_bf16(s - bf16_to_f32(h));
  o[512 + li] = hk;
  o[768 + li] = f32_to_bf16(sk - bf16_to_f32(hk));
}

// the round-4 form (dev builds: ESAM3_KVPREP_OLD=1)
__global__ __launch_bounds__(256) void mla_kvprep256_kernel(const float* __restrict__ kvp, bf16_t* __restrict__ tab, int P, int G) {
  const int bg = blockIdx.x;
  const int b = bg / G, gi = bg - b * G;
  const int t = threadIdx.x, m = t >> 4, k = t & 15;
  const float* src = kvp + ((int64_t)b * P * G + gi) * 272;
  float s8[8], k8[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s8[j] = k8[j] = 0.f;
  for (int q0 = 0; q0 < P; q0 += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (q0 + j < P) {
        s8[j] += src[(int64_t)(q0 + j) * G * 272 + m * 16 + k];
        k8[j] += src[(int64_t)(q0 + j) * G * 272 + 256 + k];
      }
    }
  }
  const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  const float sk = ((k8[0] + k8[1]) + (k8[2] + k8[3])) + ((k8[4] + k8[5]) + (k8[6] + k8[7]));
  bf16_t* o = tab + (int64_t)bg * 4 * 256;
  const int li = (m + 16 * (k >> 2)) * 4 + (k & 3);
  const bf16_t h = f32_to_bf16(s), hk = f32_to_bf16(sk);
  o[li] = h;
  o[256 + li] = f32_to_bf16(s - bf16_to_f32(h));
  o[512 + li] = hk;
  o[768 + li] = f32_to_bf16(sk - bf16_to_f32(hk));
}

// =====================================================================================================================
// LiteMLA, pass 2 (+ proj + BN + shortcut).  Workgroup = 128 consecutive tokens of one image.  Per chunk of 4 groups (64 att
// channels, the order pass 1 wrote q in):
//   A  att[d][token] = (kv . relu(q)) / (ksum . relu(q) + eps): v_mfma_f32_16x16x16_bf16 with kv as bf16 hi + lo operands
//      (fp32 accumulation, ~2^-17 relative operand error), q straight from HBM as the B operand, fp32 divide -> att tile in LDS
//   P  acc[token][co] += Wproj[co][those 64 channels] . att      (columns of Wproj taken in the reference's channel order)
// two att buffers -> one barrier per chunk.  out = acc + b (BN folded) + x.
// =====================================================================================================================
struct Mla2Params {
  const void* qms;      // pass 1's layout: [B][tiles][chunks][4][8][16][16]
  const void* tab;      // [B][2*heads][4][256] bf16
  const void* wp;       // packed [>=C][Kpp] bf16 (proj, BN folded)
  const float* bp;      // [C]
  const void* x;        // [B][H][W][C] shortcut
  void* out;            // [B][H][W][C]
  int B, H, W, Kpp, tiles_x, tiles_y;
};

template <int C, int NW>
__global__ __launch_bounds__(NW * 64, 2) void mla2_kernel(Mla2Params p) {
  typedef bf16_t T;
  constexpr int HEADS = C / 16, G = 2 * HEADS, NCH = (2 * C) / 64, NT = C / 32;
  constexpr int CW = NW / 4, NTW = NT / CW;
  __shared__ __attribute__((aligned(16))) char atl[2][128 * 128];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5, l15 = lane & 15, kg = lane >> 4;
  const unsigned bid = xcd_contig(blockIdx.x, gridDim.x);
  const unsigned tpi = (unsigned)(p.tiles_x * p.tiles_y);
  const unsigned b = bid / tpi;
  const unsigned ti = bid - b * tpi;
  const int oy0 = (int)(ti / (unsigned)p.tiles_x) * 8, ox0 = (int)(ti % (unsigned)p.tiles_x) * 16;
  const T* __restrict__ gq = reinterpret_cast<const T*>(p.qms);
  const T* __restrict__ gt = reinterpret_cast<const T*>(p.tab);
  const T* __restrict__ gw = reinterpret_cast<const T*>(p.wp);
  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x);
  T* __restrict__ go = reinterpret_cast<T*>(p.out);

  f32x16_v accp[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[t][r] = 0.f;
  const int ppt = wave & 3, pnt0 = wave >> 2;
  constexpr int IPW = 32 / NW;   // (token block, group) items per wave and chunk

  // every q fragment of the tile is requested before the first chunk (one HBM round trip per workgroup instead of one per
  // chunk: the kernel was 84 % parked at waitcnt, profiles/r04/pmc_c_summary.txt)
  s16x4 qall[NCH][IPW];
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
    for (int it = 0; it < IPW; ++it) {
      const int item = wave * IPW + it;
      const int pb = item >> 2, gl = item & 3;   // tile row, group within the chunk
      qall[ch][it] = *reinterpret_cast<const s16x4*>(gq + (((int64_t)b * tpi + ti) * NCH + ch) * 8192 + ((gl * 8 + pb) * 16 + l15) * 16 + 4 * kg);
    }
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    char* at = atl[ch & 1];
    // ---- A: att chunk -> LDS ----
#pragma unroll
    for (int it = 0; it < IPW; ++it) {
      const int item = wave * IPW + it;
      const int pb = item >> 2, gl = item & 3;           // token block of 16, group within the chunk (scale*2 + head&1)
      const int gnat = (gl >> 1) * HEADS + 2 * ch + (gl & 1);
      const s16x4 qf = qall[ch][it];
      const T* tb = gt + ((int64_t)b * G + gnat) * 1024 + lane * 4;
      const s16x4 ah = *reinterpret_cast<const s16x4*>(tb), al = *reinterpret_cast<const s16x4*>(tb + 256);
      const s16x4 kh = *reinterpret_cast<const s16x4*>(tb + 512), kl = *reinterpret_cast<const s16x4*>(tb + 768);
      f32x4 num = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      num = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, qf, num, 0, 0, 0);
      f32x4 den = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kh, qf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      den = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(kl, qf, den, 0, 0, 0);
      float a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = num[i] / (den[i] + 1e-15f);
      const int row = pb * 16 + l15;
      *reinterpret_cast<uint2*>(at + row * 128 + swz(row, gl * 2 + (kg >> 1)) + (kg & 1) * 8) =
          make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
    }
    __syncthreads();
    // ---- P: acc += Wproj[:, chunk] . att ----
    {
      const int prow = ppt * 32 + l31;
      u32x4 fd[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) fd[kc] = *reinterpret_cast<const u32x4*>(at + prow * 128 + swz(prow, kc * 2 + g));
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int nt = pnt0 + t * CW;
        u32x4 fw2[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const int gnat = (kc >> 1) * HEADS + 2 * ch + (kc & 1);
          fw2[kc] = *reinterpret_cast<const u32x4*>(gw + (int64_t)(nt * 32 + l31) * p.Kpp + gnat * 16 + g * 8);
        }
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) MmaOps<T>::mma(fw2[kc], fd[kc], accp[t]);
      }
    }
  }
  // ---- out = acc + b + x ----
  {
    const int op = ppt * 32 + l31;
    const int oy = oy0 + (op >> 4), ox = ox0 + (op & 15);
    const bool ok = oy < p.H && ox < p.W;
    const int64_t row = ((int64_t)b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int nt = pnt0 + t * CW;
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bp + nt * 32 + 8 * q + 4 * g);
        v[4 * q + 0] = accp[t][4 * q + 0] + bb.x; v[4 * q + 1] = accp[t][4 * q + 1] + bb.y;
        v[4 * q + 2] = accp[t][4 * q + 2] + bb.z; v[4 * q + 3] = accp[t][4 * q + 3] + bb.w;
        if (ok) {
          const uint2 u = *reinterpret_cast<const uint2*>(gx + row * C + nt * 32 + 8 * q + 4 * g);
          v[4 * q + 0] += __uint_as_float(u.x << 16); v[4 * q + 1] += __uint_as_float(u.x & 0xffff0000u);
          v[4 * q + 2] += __uint_as_float(u.y << 16); v[4 * q + 3] += __uint_as_float(u.y & 0xffff0000u);
        }
      }
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const uint32_t a0 = pack_bf16x2(v[8 * qp + 0], v[8 * qp + 1]), a1 = pack_bf16x2(v[8 * qp + 2], v[8 * qp + 3]);
        const uint32_t c0_ = pack_bf16x2(v[8 * qp + 4], v[8 * qp + 5]), c1_ = pack_bf16x2(v[8 * qp + 6], v[8 * qp + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
        if (ok) {
          const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(go + row * C + nt * 32 + 16 * qp + 8 * g) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// mla2d (round 5): pass 2 rebuilt around what its counters showed (profiles/r04/pmc_c_summary.txt: 84 % of the wave cycles parked at
// s_waitcnt, 0.16 TB/s -- a chain of L2 round trips, two per 64-channel chunk: the kv operands in front of the attention MFMAs and
// the projection's weight fragments in front of the projection MFMAs; the weight fragments were 16-byte pieces of 32 different
// rows per instruction, i.e. a quarter of every cache line fetched, through a 32 KB L1 that a chunk's 64 - 128 KB of lines
// thrashes).  Same arithmetic, same token / chunk order, same rounding points as mla2_kernel (bit-identical results); what changed:
//   * the chunk's projection weights [C rows][64 att channels] go global -> LDS by LDS-DMA (whole 128-byte rows, the GEMM swizzle
//     applied to the source address), double-buffered: chunk c+1's are requested right after chunk c's barrier and land under its
//     projection phase and the next attention phase -- no phase starts by waiting for a load it has just issued;
//   * the kv / ksum operands of chunk c+1 and the q fragments of chunk c+2 are requested into registers at the top of chunk c;
//   * 8 waves for both widths (a wave = one 16-token block x the chunk's 4 groups in the attention phase, one 32-token tile x half of
//     the output channel tiles in the projection), ONE barrier per chunk.
// LDS: att 2 x 16 KB + weights 2 x 16 / 32 KB = 64 / 96 KB.
// ---------------------------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(512, C == 128 ? 2 : 1) void mla2d_kernel(Mla2Params p) {
  typedef bf16_t T;
  constexpr int HEADS = C / 16, G = 2 * HEADS, NCH = (2 * C) / 64, NT = C / 32;
  constexpr int NW = 8, CW = NW / 4, NTW = NT / CW;
  constexpr int WB = C * 128;   // bytes of one chunk's weights in LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* atl = smem;               // [2][128 tokens][128 B] GEMM swizzle
  char* wpl = smem + 2 * 16384;   // [2][C rows][128 B] GEMM swizzle
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5, l15 = lane & 15, kg = lane >> 4;
  const unsigned bid = xcd_contig(blockIdx.x, gridDim.x);
  const unsigned tpi = (unsigned)(p.tiles_x * p.tiles_y);
  const unsigned b = bid / tpi;
  const unsigned ti = bid - b * tpi;
  const int oy0 = (int)(ti / (unsigned)p.tiles_x) * 8, ox0 = (int)(ti % (unsigned)p.tiles_x) * 16;
  const T* __restrict__ gq = reinterpret_cast<const T*>(p.qms);
  const T* __restrict__ gt = reinterpret_cast<const T*>(p.tab);
  const T* __restrict__ gw = reinterpret_cast<const T*>(p.wp);
  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x);
  T* __restrict__ go = reinterpret_cast<T*>(p.out);

  // ---- weights of chunk ch -> wpl[buf]: C x 128 B = C / 8 DMA pieces, wave w issues pieces w, w + 8, ...; LDS slot s of row r holds
  //      the row's logical slot s ^ ((r >> 1) & 7) = (group kc of the chunk, half) = 8 columns of Wproj at gnat(kc) * 16 + half * 8
  const uint32_t wp_lds = lds_addr(wpl);
  auto dma_w = [&](int ch, int buf) {
#pragma unroll
    for (int j = 0; j < WB / 8192; ++j) {
      const int piece = wave + 8 * j;
      const int byte = piece * 1024 + lane * 16;
      const int row = byte >> 7, ls = ((byte & 127) >> 4) ^ ((row >> 1) & 7);
      const int kc = ls >> 1;
      const int gnat = (kc >> 1) * HEADS + 2 * ch + (kc & 1);
      const uint32_t voff = (uint32_t)((row * p.Kpp + gnat * 16 + (ls & 1) * 8) * 2);
      dma_piece(gw, voff, wp_lds + (uint32_t)(buf * WB + piece * 1024));
    }
  };
  // kv hi / lo, ksum hi / lo operands of the chunk's 4 groups (this lane's 8 bytes of each 512-byte operand block)
  auto load_tab = [&](int ch, s16x4 (&t)[4][4]) {
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) {
      const int gnat = (gl >> 1) * HEADS + 2 * ch + (gl & 1);
      const T* tb = gt + ((int64_t)b * G + gnat) * 1024 + lane * 4;
#pragma unroll
      for (int o = 0; o < 4; ++o) t[gl][o] = *reinterpret_cast<const s16x4*>(tb + 256 * o);
    }
  };
  // relu(q) of token block `wave`, the chunk's 4 groups (pass 1's layout: [chunk][group][tile row][16 px][16 ch])
  auto load_q = [&](int ch, s16x4 (&q)[4]) {
#pragma unroll
    for (int gl = 0; gl < 4; ++gl)
      q[gl] = *reinterpret_cast<const s16x4*>(gq + (((int64_t)b * tpi + ti) * NCH + ch) * 8192 + ((gl * 8 + wave) * 16 + l15) * 16 + 4 * kg);
  };

  f32x16_v accp[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) accp[t][r] = 0.f;
  const int ppt = wave & 3, pnt0 = wave >> 2;

  dma_w(0, 0);
  s16x4 qf[3][4], tabf[2][4][4];
  load_q(0, qf[0]);
  if (NCH > 1) load_q(1, qf[1]);
  load_tab(0, tabf[0]);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    char* at = atl + (ch & 1) * 16384;
    // this wave's DMA pieces of W(ch) (issued one chunk ago) have landed; the barrier below publishes everybody's
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ch + 1 < NCH) load_tab(ch + 1, tabf[(ch + 1) & 1]);
    if (ch + 2 < NCH) load_q(ch + 2, qf[(ch + 2) % 3]);
    // ---- A: att[token block = wave][4 groups x 16 channels] -> LDS ----
#pragma unroll
    for (int gl = 0; gl < 4; ++gl) {
      const s16x4 q = qf[ch % 3][gl];
      const s16x4(&t)[4] = tabf[ch & 1][gl];
      f32x4 num = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(t[0], q, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      num = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(t[1], q, num, 0, 0, 0);
      f32x4 den = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(t[2], q, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      den = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(t[3], q, den, 0, 0, 0);
      float a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = num[i] / (den[i] + 1e-15f);
      const int row = wave * 16 + l15;
      *reinterpret_cast<uint2*>(at + row * 128 + swz(row, gl * 2 + (kg >> 1)) + (kg & 1) * 8) =
          make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
    }
    __syncthreads();   // att(ch) and W(ch) complete; everybody is past P(ch - 1): the other weight / att buffers are free
    if (ch + 1 < NCH) dma_w(ch + 1, (ch + 1) & 1);
    // ---- P: acc += Wproj[:, chunk] . att ----
    {
      const char* wb = wpl + (ch & 1) * WB;
      const int prow = ppt * 32 + l31;
      u32x4 fd[4];
#pragma unroll
      for (int kc = 0; kc < 4; ++kc) fd[kc] = *reinterpret_cast<const u32x4*>(at + prow * 128 + swz(prow, kc * 2 + g));
#pragma unroll
      for (int t = 0; t < NTW; ++t) {
        const int wr = (pnt0 + t * CW) * 32 + l31;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          const u32x4 fw2 = *reinterpret_cast<const u32x4*>(wb + wr * 128 + swz(wr, kc * 2 + g));
          MmaOps<T>::mma(fw2, fd[kc], accp[t]);
        }
      }
    }
  }
  // ---- out = acc + b + x ----
  {
    const int op = ppt * 32 + l31;
    const int oy = oy0 + (op >> 4), ox = ox0 + (op & 15);
    const bool ok = oy < p.H && ox < p.W;
    const int64_t row = ((int64_t)b * p.H + oy) * p.W + ox;
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int nt = pnt0 + t * CW;
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(p.bp + nt * 32 + 8 * q + 4 * g);
        v[4 * q + 0] = accp[t][4 * q + 0] + bb.x; v[4 * q + 1] = accp[t][4 * q + 1] + bb.y;
        v[4 * q + 2] = accp[t][4 * q + 2] + bb.z; v[4 * q + 3] = accp[t][4 * q + 3] + bb.w;
        if (ok) {
          const uint2 u = *reinterpret_cast<const uint2*>(gx + row * C + nt * 32 + 8 * q + 4 * g);
          v[4 * q + 0] += __uint_as_float(u.x << 16); v[4 * q + 1] += __uint_as_float(u.x & 0xffff0000u);
          v[4 * q + 2] += __uint_as_float(u.y << 16); v[4 * q + 3] += __uint_as_float(u.y & 0xffff0000u);
        }
      }
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const uint32_t a0 = pack_bf16x2(v[8 * qp + 0], v[8 * qp + 1]), a1 = pack_bf16x2(v[8 * qp + 2], v[8 * qp + 3]);
        const uint32_t c0_ = pack_bf16x2(v[8 * qp + 4], v[8 * qp + 5]), c1_ = pack_bf16x2(v[8 * qp + 6], v[8 * qp + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
        if (ok) {
          const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(go + row * C + nt * 32 + 16 * qp + 8 * g) = o;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// mbconv3s: the same three phases for Cin <= 64 (the six high-resolution MBConvs), rebuilt around what the phase ablation of
// mbconv3 showed (profiles/r04/evit_fused_bench_b.txt: every phase's cost ADDS UP -- the workgroup is a chain of memory / LDS
// latencies, not a throughput problem):
//   * PERSISTENT workgroups walk the tiles of their XCD's contiguous range; the pixel fragments of a tile (UPW x KS 16-byte
//     registers) are loaded ONCE, kept for all chunks, and the NEXT tile's are requested as soon as the last expand phase of
//     the current tile has consumed them -- the HBM latency hides behind the depthwise + project phases and the epilogue;
//   * W1 fragments + expand bias of chunk c+1 are requested at the start of chunk c's depthwise phase, W2 fragments of chunk
//     c before its expand phase, the shortcut pixels at the start of the last depthwise phase: no phase starts by waiting
//     for a global load it has just issued;
//   * the expand epilogue writes its two 16-byte pieces in a lane-dependent order (2-way instead of 4-way bank conflicts);
//   * the depthwise LDS reads run three input rows ahead of the MFMAs (DwRows3).
// ---------------------------------------------------------------------------------------------------------------------
template <int S, int CIN, int COUT, bool GELU = false>
__global__ __launch_bounds__(256, 2) void mbconv3s_kernel(Mb3Params p) {
  typedef bf16_t T;
  constexpr int NW = 4;
  constexpr int TH = 8, TW = S == 1 ? 16 : 8, OP = TH * TW;
  constexpr int HH = TH * S + (S == 1 ? 2 : 1), HW = TW * S + (S == 1 ? 2 : 1), HP = HH * HW;
  constexpr int NPT = (HP + 31) / 32, MP = NPT * 32;
  constexpr int KS = CIN / 16, NT = COUT / 32, PT = OP / 32;
  constexpr int PITCH = S == 1 ? 192 : 160;
  static_assert(CIN <= 64 && CIN % 16 == 0 && COUT % 32 == 0, "shape");
  constexpr bool PF_RES = CIN <= 32;   // shortcut pixels prefetched during the last depthwise phase (registers permitting)
  __shared__ __attribute__((aligned(16))) char mid[MP * PITCH];
  __shared__ __attribute__((aligned(16))) char dwo[OP * 128];
  // per-channel vectors of the whole layer, staged once per (persistent) workgroup: expand bias, depthwise bias (fp32) and the
  // nine depthwise taps as ready-made bf16 MFMA operands; Cmid <= 256
  __shared__ __attribute__((aligned(16))) float sb1[256];
  __shared__ __attribute__((aligned(16))) float sbd[256];
  __shared__ __attribute__((aligned(16))) uint64_t swd[9 * 256];   // diag operand of channel c: bf16(w) in element c % 4

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ gw1 = reinterpret_cast<const T*>(p.w1);
  const T* __restrict__ gw2 = reinterpret_cast<const T*>(p.w2);
  T* __restrict__ go = reinterpret_cast<T*>(p.out);

  // ---- this workgroup's tiles: XCD xcd owns the contiguous range [first, first + cnt), its workgroups stride through it ----
  const unsigned tpi = (unsigned)(p.tiles_x * p.tiles_y), ntiles = tpi * (unsigned)p.B;
  const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const unsigned nx = nwg / 8 + (xcd < nwg % 8 ? 1u : 0u);
  const unsigned tq = ntiles / 8, tr = ntiles % 8;
  const unsigned first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1u : 0u);
  if (wi >= cnt) return;
  for (int c = tid; c < p.Cmid; c += 256) {
    sb1[c] = p.b1[c];
    sbd[c] = p.bd ? p.bd[c] : 0.f;
#pragma unroll
    for (int k9 = 0; k9 < 9; ++k9) swd[k9 * 256 + c] = (uint64_t)f32_to_bf16(p.wd[k9 * p.Cmid + c]) << (16 * (c & 3));
  }

  constexpr int UPW = (2 * NPT + NW - 1) / NW;
  const int ejt = wave & 1;
  constexpr int PW = PT < NW ? PT : NW, CW = NW / PW, NTW = (NT + CW - 1) / CW;
  const int ppt = wave % PW, pnt0 = wave / PW;
  constexpr int QN = TW / 4, WPQ = NW / QN, DROWS = TH / WPQ;
  const int blk = lane >> 2, pi = lane & 3;
  const int dq = wave % QN, drow0 = (wave / QN) * DROWS;
  const unsigned daddr = lds_addr(mid + ((drow0 * S) * HW + (4 * dq + pi) * S) * PITCH + blk * 8);
  const int nchunks = p.Cmid / 64;

  // pixel fragments of a tile -> xf; xin bit u = unit u's halo pixel is inside the image
  u32x4 xf[UPW][KS];
  unsigned xin = 0;
  auto load_x = [&](unsigned tile, unsigned& b_, int& oy0_, int& ox0_) {
    b_ = tile / tpi;
    const unsigned ti = tile - b_ * tpi;
    const int ty = (int)(ti / (unsigned)p.tiles_x), tx = (int)(ti - ty * p.tiles_x);
    oy0_ = ty * TH; ox0_ = tx * TW;
    const int iy0 = oy0_ * S - 1, ix0 = ox0_ * S - 1;
    xin = 0;
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
      const int pt = (wave + NW * u) >> 1;
      const int hp = pt * 32 + l31;
      const int hy = hp / HW, hx = hp - hy * HW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool in = pt < NPT && hp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const T* px = gx + (int64_t)(((b_ * (unsigned)p.H + (in ? iy : 0)) * (unsigned)p.W + (in ? ix : 0)) * (unsigned)CIN);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        xf[u][ks] = u32x4{0u, 0u, 0u, 0u};
        if (in) xf[u][ks] = *reinterpret_cast<const u32x4*>(px + (2 * ks + g) * 8);
      }
      if (in) xin |= 1u << u;
    }
  };
  u32x4 fw[KS];
  auto load_w1 = [&](int c0) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      fw[ks] = *reinterpret_cast<const u32x4*>(gw1 + (int64_t)(c0 + ejt * 32 + l31) * p.Kp1 + (2 * ks + g) * 8);
  };

  unsigned b = 0;
  int oy0 = 0, ox0 = 0;
  load_x(first + wi, b, oy0, ox0);
  load_w1(0);
  __syncthreads();   // the staged vectors

  for (unsigned t = wi; t < cnt; t += nx) {
    const bool has_next = t + nx < cnt;
    unsigned nb = b;
    int noy0 = oy0, nox0 = ox0;
    // units with a pixel outside the image / padding rows exist in this tile for this wave?
    unsigned all_units = 0;
#pragma unroll
    for (int u = 0; u < UPW; ++u) all_units |= ((wave + NW * u) >> 1) < NPT ? 1u << u : 0u;
    const bool any_out = __builtin_amdgcn_ballot_w64((xin & all_units) != all_units) != 0ull;
    const unsigned xin_t = xin;   // this tile's flags (xin is rewritten by the next tile's load_x)

    f32x16_v accp[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accp[tt][r] = 0.f;
    uint2 resv[NTW][4];   // shortcut pixels (requested during the last depthwise phase)
    const int opp = ppt * 32 + l31;
    const int ooy = oy0 + opp / TW, oox = ox0 + opp % TW;
    const bool ok = ooy < p.OH && oox < p.OW;
    const int64_t opix = ((int64_t)b * p.OH + ooy) * p.OW + oox;

    for (int ch = 0; ch < nchunks; ++ch) {
      const int c0 = ch * 64;
      // project weights of this chunk and the depthwise weights: requested before the expand phase
      u32x4 fw2[NTW][4];
#pragma unroll
      for (int tt = 0; tt < NTW; ++tt) {
        const int nt = pnt0 + tt * CW;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
          fw2[tt][kc] = u32x4{0u, 0u, 0u, 0u};
          if (nt < NT) fw2[tt][kc] = *reinterpret_cast<const u32x4*>(gw2 + (int64_t)(nt * 32 + l31) * p.Kp2 + c0 + (kc * 2 + g) * 8);
        }
      }
      // ================= E =================
      f32x16_v binit;   // bias in the accumulator layout: register 4q + e = channel 8q + 4g + e of the 32-channel tile
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(sb1 + c0 + ejt * 32 + 8 * q + 4 * g);
        binit[4 * q + 0] = bb.x; binit[4 * q + 1] = bb.y; binit[4 * q + 2] = bb.z; binit[4 * q + 3] = bb.w;
      }
      // Round 5: ONE basic block per expand phase.  Every wave owns exactly UPW units here (2 NPT is a multiple of the four waves for both
      // tile shapes), so the per-unit "pt >= NPT" exit is gone, and the border zeroing picks one of two straight-line versions of the loop
      // instead of branching inside every unit: the scheduler can put unit u + 1's MFMA under unit u's Hardswish / pack / store chain
      // (each unit used to be MFMA -> wait -> 30 dependent VALU instructions -> two LDS stores, back to back).
      static_assert((2 * NPT) % NW == 0, "every wave owns the same number of expand units");
      auto expand_units = [&](auto zero_border) {
        // software pipeline by hand: unit u + 1's MFMAs are ISSUED before unit u's VALU chain (left to itself the register allocator
        // reuses one accumulator and serialises MFMA -> wait -> VALU per unit); the scheduling barrier keeps the order
        f32x16_v accs[2];
        accs[0] = binit;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) MmaOps<T>::mma(fw[ks], xf[0][ks], accs[0]);
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
          const int pt = (wave + NW * u) >> 1;
          const int hp = pt * 32 + l31;
          if (u + 1 < UPW) {
            accs[(u + 1) & 1] = binit;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) MmaOps<T>::mma(fw[ks], xf[u + 1][ks], accs[(u + 1) & 1]);
          }
          __builtin_amdgcn_sched_barrier(0);
          const f32x16_v acc = accs[u & 1];
          float v[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = acc[e];
          if constexpr (GELU) gelu_fast_n<16>(v);   // TinyViT MBConv (tiny_vit.py:73-108): GELU after conv1 / conv2 and after the shortcut add
          else hsw_n<16>(v);
          if constexpr (decltype(zero_border)::value) {
            const bool in = (xin_t >> u) & 1u;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = in ? v[e] : 0.f;
          }
          u32x4 o[2];
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            const uint32_t a0 = pack_bf16x2(v[8 * qp + 0], v[8 * qp + 1]), a1 = pack_bf16x2(v[8 * qp + 2], v[8 * qp + 3]);
            const uint32_t c0_ = pack_bf16x2(v[8 * qp + 4], v[8 * qp + 5]), c1_ = pack_bf16x2(v[8 * qp + 6], v[8 * qp + 7]);
            auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
            o[qp] = u32x4{s0[0], s1[0], s0[1], s1[1]};  // channels ejt*32 + 16qp + 8g .. +8 of halo pixel hp
          }
          // S = 1: lanes whose pixel index has bit 1 set write the qp = 1 piece first: the 8 lanes of a ds_write_b128 group then hit
          // 4 distinct 16-byte bank slots instead of 2 (pixel pitch 192 B = 64 mod 128, pieces 32 B apart).  S = 2 (pitch 160 B =
          // 32 mod 128): consecutive pixels already land on 4 distinct slots, and the swap made it WORSE (lanes 0, 3, 4, 7 on one slot:
          // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.60, profiles/r04/pmc_c_summary.txt) -- no swap, and 8 selects fewer per unit
#ifdef ESAM3_HSW_C
          const bool flip = (l31 >> 1) & 1;
#else
          const bool flip = S == 1 && ((l31 >> 1) & 1);
#endif
          const u32x4 w0 = flip ? o[1] : o[0], w1v = flip ? o[0] : o[1];
          char* rowp = mid + hp * PITCH + ((ejt * 4 + g) << 4);
          *reinterpret_cast<u32x4*>(rowp + (flip ? 32 : 0)) = w0;
          *reinterpret_cast<u32x4*>(rowp + (flip ? 0 : 32)) = w1v;
        }
      };
      if (any_out) expand_units(std::true_type{});
      else expand_units(std::false_type{});
      __syncthreads();
      // next chunk's W1 fragments + bias (chunk 0 again for the next tile); after the last expand phase of this tile the next
      // tile's pixel fragments
      const bool last = ch + 1 == nchunks;
      if (!last) load_w1(c0 + 64);
      else if (has_next) {
        if (nchunks > 1) load_w1(0);
        load_x(first + t + nx, nb, noy0, nox0);
      }
      if (PF_RES && last && p.residual) {
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const int nt = pnt0 + tt * CW;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            resv[tt][q] = make_uint2(0u, 0u);
            if (ok && nt < NT) resv[tt][q] = *reinterpret_cast<const uint2*>(gx + opix * CIN + nt * 32 + 8 * q + 4 * g);
          }
        }
      }
      // ================= D =================
      {
        s16x4 wdg[9];   // diag(w[tap][4 blk ..]): this lane is row pi of its block, element pi = its channel's tap
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) wdg[k9] = __builtin_bit_cast(s16x4, swd[k9 * 256 + c0 + 4 * blk + pi]);
        const float4 bb = *reinterpret_cast<const float4*>(sbd + c0 + 4 * blk);
        f32x4 acc[DROWS];
#pragma unroll
        for (int i = 0; i < DROWS; ++i) acc[i] = f32x4{bb.x, bb.y, bb.z, bb.w};
        DwRows3<3, DROWS, S, HW, PITCH, (CIN >= 64 ? 2 : 3)>::run(daddr, wdg, acc);
#pragma unroll
        for (int r = 0; r < DROWS; ++r) {
          const int op = (drow0 + r) * TW + 4 * dq + pi;
          uint2 packed;
          if constexpr (GELU) {
            float a4[4] = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
            gelu_fast_n<4>(a4);
            packed = make_uint2(pack_bf16x2(a4[0], a4[1]), pack_bf16x2(a4[2], a4[3]));
          } else {
            packed = hsw_pack4(acc[r]);
          }
          *reinterpret_cast<uint2*>(dwo + op * 128 + swz(op, blk >> 1) + (blk & 1) * 8) = packed;
        }
      }
      __syncthreads();
      // ================= P =================
      {
        const int prow = ppt * 32 + l31;
        u32x4 fd[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) fd[kc] = *reinterpret_cast<const u32x4*>(dwo + prow * 128 + swz(prow, kc * 2 + g));
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          const int nt = pnt0 + tt * CW;
          if (nt >= NT) continue;  // wave-uniform
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) MmaOps<T>::mma(fw2[tt][kc], fd[kc], accp[tt]);
        }
      }
    }
    // ================= out = acc + b2 (+ x) =================
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      const int nt = pnt0 + tt * CW;
      if (nt >= NT) continue;
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(p.b2 + nt * 32 + 8 * q + 4 * g);
        v[4 * q + 0] = accp[tt][4 * q + 0] + bb.x; v[4 * q + 1] = accp[tt][4 * q + 1] + bb.y;
        v[4 * q + 2] = accp[tt][4 * q + 2] + bb.z; v[4 * q + 3] = accp[tt][4 * q + 3] + bb.w;
        if (p.residual) {
          uint2 u2 = resv[tt][q];
          if constexpr (!PF_RES) {
            u2 = make_uint2(0u, 0u);
            if (ok) u2 = *reinterpret_cast<const uint2*>(gx + opix * CIN + nt * 32 + 8 * q + 4 * g);
          }
          v[4 * q + 0] += __uint_as_float(u2.x << 16); v[4 * q + 1] += __uint_as_float(u2.x & 0xffff0000u);
          v[4 * q + 2] += __uint_as_float(u2.y << 16); v[4 * q + 3] += __uint_as_float(u2.y & 0xffff0000u);
        }
      }
      if constexpr (GELU) gelu_fast_n<16>(v);   // the block's closing activation follows the shortcut add
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const uint32_t a0 = pack_bf16x2(v[8 * qp + 0], v[8 * qp + 1]), a1 = pack_bf16x2(v[8 * qp + 2], v[8 * qp + 3]);
        const uint32_t c0_ = pack_bf16x2(v[8 * qp + 4], v[8 * qp + 5]), c1_ = pack_bf16x2(v[8 * qp + 6], v[8 * qp + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
        if (ok) {
          const u32x4 o4 = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(go + opix * COUT + nt * 32 + 16 * qp + 8 * g) = o4;
        }
      }
    }
    b = nb; oy0 = noy0; ox0 = nox0;
  }
}

template <int S, int CIN, int COUT, bool GELU = false>
int launch_mb3s(Mb3Params p, hipStream_t stream) {
  constexpr int TH = 8, TW = S == 1 ? 16 : 8;
  p.tiles_x = (p.OW + TW - 1) / TW;
  p.tiles_y = (p.OH + TH - 1) / TH;
  const unsigned ntiles = (unsigned)p.B * p.tiles_x * p.tiles_y;
  // persistent: two workgroups per CU (LDS 52 - 58 KB each), a multiple of the 8 XCDs
  unsigned grid = 256 * 2;
  const int gd = esam3_dev_flag("ESAM3_MB3S_GRID");
  if (gd > 0) grid = (unsigned)gd;
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL((mbconv3s_kernel<S, CIN, COUT, GELU>), dim3(grid), dim3(256), 0, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// mbconv3b: stride-1 MBConv with Cin = Cout = 128 / 256 (the local modules of the EfficientViTBlocks of stages 3 - 4; the
// expanded tensor has 512 / 1024 channels).  The generic kernel re-reads the tile's pixel fragments and every weight fragment
// from L2 for every 64-channel chunk and every wave that needs them (0.45 - 0.5 MB per chunk through one CU's L1): the x loads
// and the expand phase were 0.15 of 0.24 ms in its ablation.  Here, per persistent 8-wave workgroup:
//   * waves 0..5 own one 32-pixel tile of the 10 x 18 halo each and keep its pixel fragments in registers for all chunks
//     (the next tile's are requested after the last expand phase);
//   * the chunk's weights go global -> registers -> LDS ONCE per workgroup: W1[64][Cin] (16-byte slot ^ (row & 15): conflict-
//     free ds_read_b128 A fragments) and W2[Cout][64] (GEMM swizzle); the registers are loaded one chunk ahead (W2 of chunk
//     c+1 and W1 of chunk c+2 are requested after chunk c's first barrier and written to LDS after chunk c+1's), so no phase
//     waits for a load it has just issued, and the (tile, chunk) sequence is one flat stream across tile boundaries;
//   * per-channel vectors (b1, bd, the nine depthwise taps as bf16) are staged in LDS once per workgroup.
// LDS: mid 36 KB + dwo 16 KB + W1 16 / 32 KB + W2 16 / 32 KB + vectors 6.5 / 13 KB (Cmid 512 / 1024).
// ---------------------------------------------------------------------------------------------------------------------
#ifndef ESAM3_MB3B_UNITS_CIN
#define ESAM3_MB3B_UNITS_CIN 0   // A/B builds (tools/dev_variants.sh -DESAM3_MB3B_UNITS_CIN=16|32|64): measured no gain (profiles/r06/mb3b_units_ab.txt), whole tiles stay
#endif
// ACTM: 0 = Hardswish after the expand and the depthwise conv (EfficientViT), 1 = GELU there AND after the shortcut add (TinyViT MBConv),
// 2 = GELU there, none at the end (TinyViT PatchMerging, tiny_vit.py:128-154: conv1 + BN -> GELU -> depthwise 3x3 stride 2 + BN -> GELU ->
// conv3 + BN; its middle width is the OUTPUT width, CMID_ = Cout, not 4 Cin).
template <int CIN, int S = 1, int COUT = CIN, int ACTM = 0, int CMID_ = 0>
__global__ __launch_bounds__(512, (CIN < 64 || (CIN == 64 && S == 1)) ? 4 : 2) void mbconv3b_kernel(Mb3Params p) {
  constexpr bool GELU = ACTM != 0;
  typedef bf16_t T;
  // Round 6: also the two stride-2 blocks that open stages 3 and 4 (64 -> 256 -> 128 at 126^2 and 128 -> 512 -> 256 at 63^2), until now on the
  // generic one-tile-per-workgroup kernel (0.147 + 0.161 ms, 0.06 - 0.08 of their floors: pixel and weight fragments re-read from L2 for
  // every chunk).  S = 2: 8 x 8 output tile, 17 x 17 halo = 10 pixel tiles of 32 -- waves 0, 1 own two of them --, pitch 160 B.
  constexpr int CMID = CMID_ ? CMID_ : 4 * CIN, NW = 8;
  constexpr int TH = 8, TW = S == 1 ? 16 : 8, OP = TH * TW;
  constexpr int HH = TH * S + (S == 1 ? 2 : 1), HW = TW * S + (S == 1 ? 2 : 1), HP = HH * HW;
  // expand work: UNITS (narrow blocks): unit u = (pixel tile u >> 1, channel half u & 1), wave w owns units w, w + 8, ... -- its channel half is
  // w & 1 throughout and every SIMD (waves s, s + 4) carries the same number of units (3 of 12 at stride 1, 5 of 20 at stride 2; whole-tile
  // ownership gave SIMDs 0 / 1 four resp. six); wide blocks keep whole pixel tiles per wave (waves 0 .. NPT - 1): their fragments fill the registers
  constexpr bool UNITS = ESAM3_MB3B_UNITS_CIN >= CIN && !(CIN == 64 && S == 1);   // (64 -> 256 -> 64 at 128 VGPRs: 39 - 55 spilled registers)
  constexpr int NPT = (HP + 31) / 32, MP = NPT * 32;
  constexpr int TPW = UNITS ? (2 * NPT + NW - 1) / NW : (NPT + NW - 1) / NW;   // pixel tiles a wave holds fragments of
  constexpr int KS = CIN / 16, NT = COUT / 32, NCH = CMID / 64;
  constexpr int PITCH = S == 1 ? 192 : 160;
  constexpr int W1B = 64 * CIN * 2, W2B = COUT * 128;              // bytes of one chunk's weights
  static_assert(W1B % 1024 == 0 && W2B % 1024 == 0 && (S == 1 || S == 2) && (S == 2 || COUT == CIN), "shape");
  constexpr int W1P = W1B / 1024, W2P = W2B / 1024;   // 1 KB LDS-DMA pieces per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* mid = smem;                                  // [halo px][PITCH]
  char* dwo = mid + MP * PITCH;                      // [OP px][128 B] GEMM swizzle
  char* w1s = dwo + OP * 128;                        // [64 rows][CIN] bf16, slot ^ (row & 15)
  char* w2s = w1s + W1B;                             // [COUT rows][64] bf16, swz(row, slot)
  float* sb1 = reinterpret_cast<float*>(w2s + W2B);  // [CMID]
  float* sbd = sb1 + CMID;                           // [CMID]
  uint16_t* swd = reinterpret_cast<uint16_t*>(sbd + CMID);  // [9][CMID] bf16

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, g = lane >> 5;
  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ gw1 = reinterpret_cast<const T*>(p.w1);
  const T* __restrict__ gw2 = reinterpret_cast<const T*>(p.w2);
  T* __restrict__ go = reinterpret_cast<T*>(p.out);

  const unsigned tpi = (unsigned)(p.tiles_x * p.tiles_y), ntiles = tpi * (unsigned)p.B;
  const unsigned nwg = gridDim.x, xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  const unsigned nx = nwg / 8 + (xcd < nwg % 8 ? 1u : 0u);
  const unsigned tq = ntiles / 8, tr = ntiles % 8;
  const unsigned first = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1u : 0u);
  if (wi >= cnt) return;
  for (int c = tid; c < CMID; c += 512) {
    sb1[c] = p.b1[c];
    sbd[c] = p.bd ? p.bd[c] : 0.f;
#pragma unroll
    for (int k9 = 0; k9 < 9; ++k9) swd[k9 * CMID + c] = f32_to_bf16(p.wd[k9 * CMID + c]);
  }

  // ---- weight staging by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B -> 1 KB of LDS, lane-linear): the swizzles are
  //      applied to the SOURCE address (cdna_hip_programming.md 5.4 rule 21); wave w issues pieces w, w + 8, ...
  const uint32_t w1_lds = lds_addr(w1s), w2_lds = lds_addr(w2s);
  // 16-byte slot of a W1 row in LDS: rows of >= 256 B: slot ^ (row & 15); 128-byte rows (Cin = 64) hold 8 slots and two rows share a bank
  // row: the GEMM swizzle slot ^ ((row >> 1) & 7) -- both conflict-free for the 32-row ds_read_b128 A fragments
  // (64-byte rows, Cin = 32: four rows per bank row, slot ^ ((row >> 2) & 3); 32-byte rows, Cin = 16: eight consecutive rows already tile it)
  auto w1slot = [](int slot, int row) {
    return CIN >= 128 ? slot ^ (row & 15) : CIN == 64 ? slot ^ ((row >> 1) & 7) : CIN == 32 ? slot ^ ((row >> 2) & 3) : slot;
  };
  auto dma_w1 = [&](int ch) {   // W1 rows ch*64 .. +64, all CIN columns -> [row][CIN], swizzled slots
#pragma unroll
    for (int j = 0; j < (W1P + 7) / 8; ++j) {
      const int piece = wave + 8 * j;
      if constexpr (W1P % 8 != 0) { if (piece >= W1P) continue; }   // wave-uniform
      const int byte = piece * 1024 + lane * 16;
      const int row = byte / (CIN * 2), pslot = (byte - row * (CIN * 2)) >> 4;
      const uint32_t voff = (uint32_t)(((ch * 64 + row) * p.Kp1 + (w1slot(pslot, row) << 3)) * 2);
      dma_piece(gw1, voff, w1_lds + (uint32_t)piece * 1024u);
    }
  };
  auto dma_w2 = [&](int ch) {   // W2 rows 0 .. COUT, columns ch*64 .. +64 -> [row][128 B] with the GEMM swizzle
#pragma unroll
    for (int j = 0; j < (W2P + 7) / 8; ++j) {
      const int piece = wave + 8 * j;
      if constexpr (W2P % 8 != 0) { if (piece >= W2P) continue; }   // wave-uniform
      const int byte = piece * 1024 + lane * 16;
      const int row = byte >> 7, pslot = (byte & 127) >> 4;
      const uint32_t voff = (uint32_t)((row * p.Kp2 + ch * 64 + ((pslot ^ ((row >> 1) & 7)) << 3)) * 2);
      dma_piece(gw2, voff, w2_lds + (uint32_t)piece * 1024u);
    }
  };

  // ---- pixel fragments: wave w owns the halo pixel tiles w, w + 8, ... < NPT ----
  u32x4 fa[TPW][KS];
  unsigned xin = 0;   // bit k: the lane's pixel of its k-th tile is inside the image
  auto load_x = [&](unsigned tile, unsigned& b_, int& oy0_, int& ox0_) {
    b_ = tile / tpi;
    const unsigned ti = tile - b_ * tpi;
    const int ty = (int)(ti / (unsigned)p.tiles_x), tx = (int)(ti - ty * p.tiles_x);
    oy0_ = ty * TH; ox0_ = tx * TW;
    xin = 0;
#pragma unroll
    for (int k = 0; k < TPW; ++k) {
      const int pt = UNITS ? (wave >> 1) + (NW / 2) * k : wave + NW * k;
      const int hp = pt * 32 + l31;
      const int hy = hp / HW, hx = hp - hy * HW;
      const int iy = oy0_ * S - 1 + hy, ix = ox0_ * S - 1 + hx;
      const bool in = pt < NPT && hp < HP && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
      const T* px = gx + (int64_t)(((b_ * (unsigned)p.H + (in ? iy : 0)) * (unsigned)p.W + (in ? ix : 0)) * (unsigned)CIN);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        fa[k][ks] = u32x4{0u, 0u, 0u, 0u};
        if (in) fa[k][ks] = *reinterpret_cast<const u32x4*>(px + (2 * ks + g) * 8);
      }
      if (in) xin |= 1u << k;
    }
  };

  // project: pixel tile = wave % PW, channel tiles (wave / PW) + CW t
  constexpr int PT = OP / 32, PW = PT, CW = NW / PW, NTW = (NT + CW - 1) / CW;   // (narrow outputs: waves with pnt0 >= NT sit the phase out)
  const int ppt = wave % PW, pnt0 = wave / PW;
  // depthwise: run of 4 columns = wave % QN, rows DROWS (wave / QN) .. + DROWS
  constexpr int QN = TW / 4, DROWS = TH / (NW / QN);
  const int blk = lane >> 2, pi = lane & 3;
  const int dq = wave % QN, drow0 = (wave / QN) * DROWS;
  const unsigned daddr = lds_addr(mid + ((drow0 * S) * HW + (4 * dq + pi) * S) * PITCH + blk * 8);

  unsigned b = 0;
  int oy0 = 0, ox0 = 0;
  load_x(first + wi, b, oy0, ox0);
  // prologue of the weight stream: W1(0) -> LDS
  dma_w1(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (unsigned t = wi; t < cnt; t += nx) {
    const bool has_next = t + nx < cnt;
    unsigned nb = b;
    int noy0 = oy0, nox0 = ox0;
    unsigned own = 0;
#pragma unroll
    for (int k = 0; k < TPW; ++k) own |= (UNITS ? (wave >> 1) + (NW / 2) * k : wave + NW * k) < NPT ? 1u << k : 0u;
    const bool any_out = __builtin_amdgcn_ballot_w64((xin & own) != own) != 0ull;
    const unsigned xin_t = xin;
    f32x16_v accp[NTW];
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) accp[tt][r] = 0.f;
    const int opp = ppt * 32 + l31;
    const int ooy = oy0 + opp / TW, oox = ox0 + opp % TW;
    const bool ok = ooy < p.OH && oox < p.OW;
    const int64_t opix = ((int64_t)b * p.OH + ooy) * p.OW + oox;

    for (int ch = 0; ch < NCH; ++ch) {
      const int c0 = ch * 64;
      if (!MB3_ABL(32) || (t == wi && ch == 0)) dma_w2(ch);   // W2 buffer is free (barrier C of the previous chunk); lands under the expand phase
      // ================= E: mid[px tiles of this wave][64] =================
#pragma unroll
      for (int k = 0; k < TPW; ++k) {
        const int pt = UNITS ? (wave >> 1) + (NW / 2) * k : wave + NW * k;
        if (pt >= NPT) continue;   // wave-uniform
        const int hp = pt * 32 + l31;
#pragma unroll
        for (int ej = 0; ej < (UNITS ? 1 : 2); ++ej) {
          const int ejt = UNITS ? (wave & 1) : ej;
          f32x16_v acc;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 bb = *reinterpret_cast<const float4*>(sb1 + c0 + ejt * 32 + 8 * q + 4 * g);
            acc[4 * q + 0] = bb.x; acc[4 * q + 1] = bb.y; acc[4 * q + 2] = bb.z; acc[4 * q + 3] = bb.w;
          }
          const int row = ejt * 32 + l31;
          const char* wrow = w1s + row * (CIN * 2);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const u32x4 fwv = *reinterpret_cast<const u32x4*>(wrow + (w1slot(2 * ks + g, row) << 4));
            MmaOps<T>::mma(fwv, fa[k][ks], acc);
          }
          float v[16];
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] = acc[e];
          if constexpr (GELU) gelu_fast_n<16>(v);   // TinyViT MBConv (tiny_vit.py:73-108): GELU after conv1 / conv2 and after the shortcut add
          else hsw_n<16>(v);
          if (any_out) {
            const bool in = (xin_t >> k) & 1u;
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = in ? v[e] : 0.f;
          }
          u32x4 o[2];
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {
            const uint32_t a0 = pack_bf16x2(v[8 * qp + 0], v[8 * qp + 1]), a1 = pack_bf16x2(v[8 * qp + 2], v[8 * qp + 3]);
            const uint32_t c0_ = pack_bf16x2(v[8 * qp + 4], v[8 * qp + 5]), c1_ = pack_bf16x2(v[8 * qp + 6], v[8 * qp + 7]);
            auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
            o[qp] = u32x4{s0[0], s1[0], s0[1], s1[1]};
          }
          const bool flip = S == 1 && ((l31 >> 1) & 1);   // (flipping by pixel bit 2 instead measured the same: profiles/r06/flip_ab.txt)
          const u32x4 w0 = flip ? o[1] : o[0], w1v = flip ? o[0] : o[1];
          char* rowp = mid + hp * PITCH + ((ejt * 4 + g) << 4);
          *reinterpret_cast<u32x4*>(rowp + (flip ? 32 : 0)) = w0;
          *reinterpret_cast<u32x4*>(rowp + (flip ? 0 : 32)) = w1v;
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of W2(ch) have landed
      __syncthreads();   // A: mid and W2(ch) complete; W1 buffer free (expand done)
      {
        const bool last = ch + 1 == NCH;
        if ((!last || has_next) && !MB3_ABL(32)) dma_w1((ch + 1) % NCH);   // flat (tile, chunk) stream; lands under depthwise + project
        if (last && has_next) load_x(first + t + nx, nb, noy0, nox0);
      }
      // ================= D =================
      {
        s16x4 wdg[9];
#pragma unroll
        for (int k9 = 0; k9 < 9; ++k9) {
          const uint64_t wv = (uint64_t)swd[k9 * CMID + c0 + 4 * blk + pi] << (16 * pi);
          wdg[k9] = __builtin_bit_cast(s16x4, wv);
        }
        const float4 bb = *reinterpret_cast<const float4*>(sbd + c0 + 4 * blk);
        f32x4 acc[DROWS];
#pragma unroll
        for (int i = 0; i < DROWS; ++i) acc[i] = f32x4{bb.x, bb.y, bb.z, bb.w};
        DwRows3<3, DROWS, S, HW, PITCH, 2>::run(daddr, wdg, acc);
#pragma unroll
        for (int r = 0; r < DROWS; ++r) {
          const int op = (drow0 + r) * TW + 4 * dq + pi;
          uint2 packed;
          if constexpr (GELU) {
            float a4[4] = {acc[r][0], acc[r][1], acc[r][2], acc[r][3]};
            gelu_fast_n<4>(a4);
            packed = make_uint2(pack_bf16x2(a4[0], a4[1]), pack_bf16x2(a4[2], a4[3]));
          } else {
            packed = hsw_pack4(acc[r]);
          }
          *reinterpret_cast<uint2*>(dwo + op * 128 + swz(op, blk >> 1) + (blk & 1) * 8) = packed;
        }
      }
      __syncthreads();   // B: dwo complete
      // ================= P =================
      {
        const int prow = ppt * 32 + l31;
        u32x4 fd[4];
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) fd[kc] = *reinterpret_cast<const u32x4*>(dwo + prow * 128 + swz(prow, kc * 2 + g));
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
          if constexpr (NT % CW != 0) { if (pnt0 + tt * CW >= NT) continue; }   // wave-uniform
          const int wr = (pnt0 + tt * CW) * 32 + l31;
#pragma unroll
          for (int kc = 0; kc < 4; ++kc) {
            const u32x4 fw2 = *reinterpret_cast<const u32x4*>(w2s + wr * 128 + swz(wr, kc * 2 + g));
            MmaOps<T>::mma(fw2, fd[kc], accp[tt]);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of W1(ch + 1) (and the next tile's pixels) have landed
      __syncthreads();   // C: W1(ch + 1) complete, W2 buffer free
    }
    // ================= out = acc + b2 (+ x) =================
#pragma unroll
    for (int tt = 0; tt < NTW; ++tt) {
      const int nt = pnt0 + tt * CW;
      if constexpr (NT % CW != 0) { if (nt >= NT) continue; }   // wave-uniform
      float v[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(p.b2 + nt * 32 + 8 * q + 4 * g);
        v[4 * q + 0] = accp[tt][4 * q + 0] + bb.x; v[4 * q + 1] = accp[tt][4 * q + 1] + bb.y;
        v[4 * q + 2] = accp[tt][4 * q + 2] + bb.z; v[4 * q + 3] = accp[tt][4 * q + 3] + bb.w;
        if (S == 1 && p.residual && ok) {
          const uint2 u2 = *reinterpret_cast<const uint2*>(gx + opix * CIN + nt * 32 + 8 * q + 4 * g);
          v[4 * q + 0] += __uint_as_float(u2.x << 16); v[4 * q + 1] += __uint_as_float(u2.x & 0xffff0000u);
          v[4 * q + 2] += __uint_as_float(u2.y << 16); v[4 * q + 3] += __uint_as_float(u2.y & 0xffff0000u);
        }
      }
      if constexpr (ACTM == 1) gelu_fast_n<16>(v);   // the block's closing activation follows the shortcut add
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const uint32_t a0 = pack_bf16x2(v[8 * qp + 0], v[8 * qp + 1]), a1 = pack_bf16x2(v[8 * qp + 2], v[8 * qp + 3]);
        const uint32_t c0_ = pack_bf16x2(v[8 * qp + 4], v[8 * qp + 5]), c1_ = pack_bf16x2(v[8 * qp + 6], v[8 * qp + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
        if (ok) {
          const u32x4 o4 = {s0[0], s1[0], s0[1], s1[1]};
          *reinterpret_cast<u32x4*>(go + opix * COUT + nt * 32 + 16 * qp + 8 * g) = o4;
        }
      }
    }
    b = nb; oy0 = noy0; ox0 = nox0;
  }
}

template <int CIN, int S = 1, int COUT = CIN, int ACTM = 0, int CMID_ = 0>
int launch_mb3b(Mb3Params p, hipStream_t stream) {
  constexpr int TW = S == 1 ? 16 : 8, HP = S == 1 ? 180 : 289, MP = (HP + 31) / 32 * 32, PITCH = S == 1 ? 192 : 160;
  p.tiles_x = (p.OW + TW - 1) / TW;
  p.tiles_y = (p.OH + 7) / 8;
  const unsigned ntiles = (unsigned)p.B * p.tiles_x * p.tiles_y;
  constexpr size_t lds = (size_t)MP * PITCH + 8 * TW * 128 + 64 * CIN * 2 + COUT * 128 + (size_t)(CMID_ ? CMID_ : 4 * CIN) * (4 + 4 + 18);
  auto kern = mbconv3b_kernel<CIN, S, COUT, ACTM, CMID_>;
  if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds)) return -1;
  // persistent: one workgroup per CU, a multiple of the 8 XCDs; the narrow blocks (<= 128 VGPRs, <= 80 KB of LDS) run TWO per CU = four
  // waves per SIMD (profiles/r06/mb3b_small_ab.txt: 256 / 512 / 768 workgroups)
  unsigned grid = (lds <= 80 * 1024 && (CIN < 64 || (CIN == 64 && S == 1))) ? 512 : 256;
  if (esam3_dev_flag("ESAM3_MB3B_GRID") > 0) grid = (unsigned)esam3_dev_flag("ESAM3_MB3B_GRID");
  if (grid > ntiles) grid = ntiles;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

template <int S, int CIN, int COUT, int NW>
int launch_mb3(Mb3Params p, hipStream_t stream) {
  constexpr int TH = 8, TW = S == 1 ? 16 : 8;
  p.tiles_x = (p.OW + TW - 1) / TW;
  p.tiles_y = (p.OH + TH - 1) / TH;
  const unsigned grid = (unsigned)p.B * p.tiles_x * p.tiles_y;
  hipLaunchKernelGGL((mbconv3_kernel<S, CIN, COUT, NW>), dim3(grid), dim3(NW * 64), 0, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

}  // namespace

// shapes the round-4 fused MBConv is instantiated for (bf16): EfficientViT-B1 / B0 widths
// TinyViT PatchMerging as one launch (esam3_launch_mbconv3 with residual = 4): the two shapes of TinyViT-5M / -11M whose tiles fit the LDS
bool esam3_patch_merging_fused_ok(int dtype, int Cin, int Cout) {
  return dtype == 1 && ((Cin == 64 && Cout == 128) || (Cin == 128 && Cout == 256));
}

bool esam3_mbconv3_ok(int dtype, int Cin, int Cmid, int Cout, int stride) {
  if (dtype != 1 || Cmid % 64) return false;
  // the launcher's own limits (esam3_launch_mbconv3): the persistent Cin <= 64 kernels stage per-channel vectors for Cmid <= 256;
  // the engine falls back to the layer-by-layer path for anything else (e.g. a checkpoint with a larger expand ratio)
  if (Cin <= 64 && Cmid > 256 && !(stride == 2 && Cin == 64)) return false;
  if (stride == 2)
    return (Cin == 16 && Cout == 32) || (Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 128) || (Cin == 128 && Cout == 256);
  if (stride == 1)
    return (Cin == 32 && Cout == 32) || (Cin == 64 && Cout == 64) || (Cin == 128 && Cout == 128) || (Cin == 256 && Cout == 256);
  return false;
}

int esam3_launch_mbconv3(const void* x, void* out, const void* w1, int Kp1, const float* b1, const float* wd, const float* bd,
                         const void* w2, int Kp2, const float* b2, int B, int H, int W, int Cin, int Cmid, int Cout, int stride,
                         int residual, hipStream_t stream) {
  if (residual == 4) {   // TinyViT PatchMerging (round 6): Cin -> Cout (GELU) -> depthwise 3x3 stride 2 (GELU) -> Cout, no shortcut, no closing activation
    if (!esam3_patch_merging_fused_ok(1, Cin, Cout) || Cmid != Cout || stride != 2 || Kp1 < Cin || Kp2 < Cmid ||
        (int64_t)B * H * W * Cin >= ((int64_t)1 << 31)) {
      esam3_set_error("mbconv3: PatchMerging variant unsupported for %d -> %d -> %d stride %d", Cin, Cmid, Cout, stride);
      return -1;
    }
    Mb3Params q{};
    q.x = x; q.out = out; q.w1 = w1; q.b1 = b1; q.wd = wd; q.bd = bd; q.w2 = w2; q.b2 = b2;
    q.B = B; q.H = H; q.W = W; q.OH = (H + 1) / 2; q.OW = (W + 1) / 2;
    q.Cmid = Cmid; q.Kp1 = Kp1; q.Kp2 = Kp2; q.residual = 0;
    q.abl = esam3_dev_flag("ESAM3_MB3_ABL");
    if (Cin == 64) return launch_mb3b<64, 2, 128, 2, 128>(q, stream);
    return launch_mb3b<128, 2, 256, 2, 256>(q, stream);
  }
  if (!esam3_mbconv3_ok(1, Cin, Cmid, Cout, stride) || Kp1 < Cin || Kp2 < Cmid || (int64_t)B * H * W * Cin >= ((int64_t)1 << 31)) {
    esam3_set_error("mbconv3: unsupported configuration %d -> %d -> %d stride %d", Cin, Cmid, Cout, stride);
    return -1;
  }
  Mb3Params q{};
  q.x = x; q.out = out; q.w1 = w1; q.b1 = b1; q.wd = wd; q.bd = bd; q.w2 = w2; q.b2 = b2;
  q.B = B; q.H = H; q.W = W; q.OH = (H + stride - 1) / stride; q.OW = (W + stride - 1) / stride;
  q.Cmid = Cmid; q.Kp1 = Kp1; q.Kp2 = Kp2; q.residual = residual & 1;
  q.abl = esam3_dev_flag("ESAM3_MB3_ABL");
  if (residual & 2) {   // TinyViT MBConv (tiny_vit.py:73-108), round 6: GELU after conv1, conv2 and the shortcut add; one shape, 64 -> 256 -> 64
    if (Cin != 64 || Cout != 64 || Cmid != 256 || stride != 1 || !(residual & 1)) {
      esam3_set_error("mbconv3: the GELU variant is built for 64 -> 256 -> 64 channels, stride 1, with the shortcut");
      return -1;
    }
    if (esam3_dev_flag("ESAM3_MB3S")) return launch_mb3s<1, 64, 64, true>(q, stream);
    return launch_mb3b<64, 1, 64, 1>(q, stream);
  }
  if (Cin <= 64 && Cmid == 4 * Cin && !esam3_dev_flag("ESAM3_MB3_GENERIC") && !esam3_dev_flag("ESAM3_MB3S")) {
    // round 6: the narrow blocks too run the 8-wave LDS-weight kernel, two workgroups per CU (0.264 -> 0.236, 0.267 -> 0.232, 0.137 ->
    // 0.122, 0.184 -> 0.142 ms against the 4-wave mbconv3s; dev builds: ESAM3_MB3S=1 selects the latter)
    if (stride == 2 && Cin == 16) return launch_mb3b<16, 2, 32>(q, stream);
    if (stride == 2 && Cin == 32) return launch_mb3b<32, 2, 64>(q, stream);
    if (stride == 1 && Cin == 32) return launch_mb3b<32, 1, 32>(q, stream);
    if (stride == 1 && Cin == 64) return launch_mb3b<64, 1, 64>(q, stream);
  }
  if (Cin <= 64 && !esam3_dev_flag("ESAM3_MB3_GENERIC")) {   // persistent + prefetching 4-wave variant (any expand ratio up to Cmid 256)
    if (Cmid > 256) { esam3_set_error("mbconv3s: Cmid %d > 256", Cmid); return -1; }
    if (stride == 2 && Cin == 16) return launch_mb3s<2, 16, 32>(q, stream);
    if (stride == 2 && Cin == 32) return launch_mb3s<2, 32, 64>(q, stream);
    if (stride == 1 && Cin == 32) return launch_mb3s<1, 32, 32>(q, stream);
    if (stride == 1) return launch_mb3s<1, 64, 64>(q, stream);
    // 64 -> 128 stride 2: 5 x 4 pixel fragments per lane do not fit next to the prefetches: the 8-wave kernel below
  }
  if (stride == 2 && Cin >= 64 && Cmid == 4 * Cin && Cout == 2 * Cin && !esam3_dev_flag("ESAM3_MB3_GENERIC") &&
      !esam3_dev_flag("ESAM3_MB3S")) {   // round 6 (dev builds: ESAM3_MB3S=1 restores the whole round-5 dispatch)
    if (Cin == 64) return launch_mb3b<64, 2, 128>(q, stream);
    return launch_mb3b<128, 2, 256>(q, stream);
  }
  if (stride == 1 && Cin >= 128 && Cmid == 4 * Cin && !esam3_dev_flag("ESAM3_MB3_GENERIC")) {   // weights through LDS, resident pixels
    if (Cin == 128) return launch_mb3b<128>(q, stream);
    return launch_mb3b<256>(q, stream);
  }
  if (stride == 2) {
    if (Cin == 16) return launch_mb3<2, 16, 32, 4>(q, stream);
    if (Cin == 32) return launch_mb3<2, 32, 64, 4>(q, stream);
    if (Cin == 64) return launch_mb3<2, 64, 128, 4>(q, stream);
    return launch_mb3<2, 128, 256, 8>(q, stream);
  }
  if (Cin == 32) return launch_mb3<1, 32, 32, 4>(q, stream);
  if (Cin == 64) return launch_mb3<1, 64, 64, 4>(q, stream);
  if (Cin == 128) return launch_mb3<1, 128, 128, 4>(q, stream);
  return launch_mb3<1, 256, 256, 8>(q, stream);
}

// ---- LiteMLA context module (ops.py:521-671 inside ResidualBlock, :740-770), fused: C = 128 / 256 channels, dim = 16 ----
bool esam3_mla_fused_ok(int dtype, int C, int dim) { return dtype == 1 && dim == 16 && (C == 128 || C == 256); }

// scratch: qms bf16 [B][tiles][chunks][4][128][16], kvp fp32 [B][tiles*2][2*heads][272], tab bf16 [B][2*heads][1024]
void esam3_mla_fused_scratch(int B, int H, int W, int C, size_t* qms_bytes, size_t* kvp_bytes, size_t* tab_bytes) {
  const int tiles = ((W + 15) / 16) * ((H + 7) / 8), G = 2 * (C / 16);
  *qms_bytes = (size_t)B * tiles * (C / 32) * 8192 * 2;   // [tiles][heads / 2 chunks][4 groups][128 px][16 ch] bf16
  *kvp_bytes = (size_t)B * tiles * 2 * G * 272 * 4;
  *tab_bytes = (size_t)B * G * 1024 * 2;
}

// x [B][H][W][C] -> out = x + proj(att(qkv(x))) (BN folded into wproj / bproj).  wqkv [>=3C][Kpq], wgrp [3C][Kpg] (16 used),
// wproj [>=C][Kpp] packed bf16; wdw fp32 [25][3C]
int esam3_launch_mla_fused(const void* x, void* out, const void* wqkv, int Kpq, const float* wdw, const void* wgrp, int Kpg,
                           const void* wproj, int Kpp, const float* bproj, void* qms, float* kvp, void* tab, int B, int H, int W,
                           int C, hipStream_t stream) {
  if (!esam3_mla_fused_ok(1, C, 16) || Kpq < C || Kpg < 16 || Kpp < 2 * C || (int64_t)B * (H + 8) * (W + 16) * 2 * C >= ((int64_t)1 << 31)) {
    esam3_set_error("mla_fused: unsupported configuration C = %d", C);
    return -1;
  }
  Mla1Params a{};
  a.x = x; a.wqkv = wqkv; a.wdw = wdw; a.wgrp = wgrp; a.qms = qms; a.kvp = kvp;
  a.B = B; a.H = H; a.W = W; a.Kpq = Kpq; a.Kpg = Kpg;
  a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 7) / 8;
  a.abl = esam3_dev_flag("ESAM3_MLA1_ABL");
  const int tiles = a.tiles_x * a.tiles_y, G = 2 * (C / 16);
  const size_t lds1 = (size_t)(256 + 128 + 128) * 192 + (size_t)96 * C * 2;
  const size_t lds1v = (size_t)(240 + 128) * 208 + (size_t)96 * C * 2 + 2 * 2400 * 2;
  if (!esam3_dev_flag("ESAM3_MLA1_OLD")) {   // round 6: depthwise + grouped conv in registers (A/B in dev builds: the round-4 kernel)
    if (C == 128) {
      if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(mla1v_kernel<128>), (int)lds1v)) return -1;
      hipLaunchKernelGGL((mla1v_kernel<128>), dim3((unsigned)(B * tiles < 256 ? B * tiles : 256)), dim3(512), lds1v, stream, a);
    } else {
      if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(mla1v_kernel<256>), (int)lds1v)) return -1;
      hipLaunchKernelGGL((mla1v_kernel<256>), dim3((unsigned)(B * tiles)), dim3(512), lds1v, stream, a);   // one tile per workgroup (see PERSIST)
    }
  } else if (C == 128) {
    if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(mla1_kernel<128>), (int)lds1)) return -1;
    hipLaunchKernelGGL((mla1_kernel<128>), dim3((unsigned)(B * tiles)), dim3(512), lds1, stream, a);
  } else {
    if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(mla1_kernel<256>), (int)lds1)) return -1;
    hipLaunchKernelGGL((mla1_kernel<256>), dim3((unsigned)(B * tiles)), dim3(512), lds1, stream, a);
  }
  HIP_CHECK_RET(hipGetLastError());
  if (esam3_dev_flag("ESAM3_KVPREP_OLD"))
    hipLaunchKernelGGL(mla_kvprep256_kernel, dim3((unsigned)(B * G)), dim3(256), 0, stream, kvp, reinterpret_cast<bf16_t*>(tab), tiles * 2, G);
  else
    hipLaunchKernelGGL(mla_kvprep_kernel, dim3((unsigned)(B * G)), dim3(1024), 0, stream, kvp, reinterpret_cast<bf16_t*>(tab), tiles * 2, G);
  HIP_CHECK_RET(hipGetLastError());
  Mla2Params q{};
  q.qms = qms; q.tab = tab; q.wp = wproj; q.bp = bproj; q.x = x; q.out = out;
  q.B = B; q.H = H; q.W = W; q.Kpp = Kpp; q.tiles_x = a.tiles_x; q.tiles_y = a.tiles_y;
  if (esam3_dev_flag("ESAM3_MLA2_OLD")) {   // A/B (dev builds): the round-4 kernel, bit-identical output
    if (C == 128) hipLaunchKernelGGL((mla2_kernel<128, 4>), dim3((unsigned)(B * tiles)), dim3(256), 0, stream, q);
    else hipLaunchKernelGGL((mla2_kernel<256, 8>), dim3((unsigned)(B * tiles)), dim3(512), 0, stream, q);
  } else {
    const size_t lds2 = (size_t)2 * 16384 + (size_t)2 * C * 128;
    if (C == 128) {
      if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(mla2d_kernel<128>), (int)lds2)) return -1;
      hipLaunchKernelGGL((mla2d_kernel<128>), dim3((unsigned)(B * tiles)), dim3(512), lds2, stream, q);
    } else {
      if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(mla2d_kernel<256>), (int)lds2)) return -1;
      hipLaunchKernelGGL((mla2d_kernel<256>), dim3((unsigned)(B * tiles)), dim3(512), lds2, stream, q);
    }
  }
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
