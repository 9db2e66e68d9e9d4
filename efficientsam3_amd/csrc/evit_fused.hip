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
template <int N> __device__ __forceinline__ void ds_wait(s16x4 (&c)[2]) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(c[0]), "+v"(c[1]) : "i"(N));
}
template <int N> __device__ __forceinline__ void ds_wait(s16x4 (&c)[4]) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]) : "i"(N));
}
template <int N> __device__ __forceinline__ void ds_wait(s16x4 (&c)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]) : "i"(N));
}
// KSZ x KSZ taps, accumulator i = (row r = i / QN of the wave's RPW rows, run q = i % QN of 4 pixels); `addr` = LDS byte address
// of (first row, pixel pi, channel block) ; halo row pitch HW pixels, pixel pitch PITCH bytes, stride S
template <int KSZ, int RPW, int QN, int S, int HW, int PITCH>
struct DwMfma {
  static constexpr int NACC = RPW * QN, NTAP = KSZ * KSZ;
  template <int TAP, int I> static constexpr int off() {
    return (((I / QN) * S + TAP / KSZ) * HW + (I % QN) * 4 * S + TAP % KSZ) * PITCH;
  }
  template <int TAP, int I = 0> static __device__ __forceinline__ void issue(s16x4 (&buf)[NACC], unsigned addr) {
    if constexpr (I < NACC) {
      dsr64<off<TAP, I>()>(buf[I], addr);
      issue<TAP, I + 1>(buf, addr);
    }
  }
  template <int TAP = 0> static __device__ __forceinline__ void step(s16x4 (&cur)[NACC], s16x4 (&nxt)[NACC], unsigned addr,
                                                                     const s16x4 (&wdg)[NTAP], f32x4 (&acc)[NACC]) {
    if constexpr (TAP + 1 < NTAP) {
      issue<TAP + 1>(nxt, addr);
      ds_wait<NACC>(cur);
    } else {
      ds_wait<0>(cur);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(wdg[TAP], cur[i], acc[i], 0, 0, 0);
    if constexpr (TAP + 1 < NTAP) step<TAP + 1>(nxt, cur, addr, wdg, acc);
  }
  static __device__ __forceinline__ void run(unsigned addr, const s16x4 (&wdg)[NTAP], f32x4 (&acc)[NACC]) {
    s16x4 a[NACC], b[NACC];
    issue<0>(a, addr);
    step<0>(a, b, addr, wdg, acc);
  }
};
__device__ __forceinline__ unsigned lds_addr(const char* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}

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
};

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

  // depthwise phase: lane = (block of 4 channels, pixel within a run of 4); a wave owns RPW output rows
  constexpr int RPW = TH / NW, QN = TW / 4;
  const int blk = lane >> 2, pi = lane & 3;
  const char* dbase = mid + ((wave * RPW * S) * HW + pi * S) * PITCH + blk * 8;

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
    {
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
              if (u0 + uu < UPW && xoff[u] >= 0)
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
    {
      s16x4 wdg[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wdg[t] = diag_bf16(wdf[t], pi);
      f32x4 acc[RPW * QN];
#pragma unroll
      for (int i = 0; i < RPW * QN; ++i) acc[i] = bsv;
      DwMfma<3, RPW, QN, S, HW, PITCH>::run(lds_addr(dbase), wdg, acc);
#pragma unroll
      for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int op = (wave * RPW + r) * TW + 4 * q + pi;
          uint2 o;
          o.x = pack_bf16x2(hsw(acc[r * QN + q][0]), hsw(acc[r * QN + q][1]));
          o.y = pack_bf16x2(hsw(acc[r * QN + q][2]), hsw(acc[r * QN + q][3]));
          *reinterpret_cast<uint2*>(dwo + op * 128 + swz(op, blk >> 1) + (blk & 1) * 8) = o;
        }
    }
    __syncthreads();

    // ================= P: acc[out px][Cout] += dwo . W2[:, c0..c0+64)^T =================
    {
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
  {
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
//       row pair of 4): 96 slots = 6 waves
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
  void* qms;            // [B][H*W][2C] bf16: relu(q), channel = chunk*64 + scale*32 + (head & 1)*16 + d
  float* kvp;           // [B][tiles*2][2*heads][272] fp32 partials, [dv][dk] with row 16 = ksum
  int B, H, W, Kpq, Kpg;
  int tiles_x, tiles_y;
};

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
  __shared__ __attribute__((aligned(16))) char mid[256 * PITCH];
  __shared__ __attribute__((aligned(16))) char dwo[128 * PITCH];
  __shared__ __attribute__((aligned(16))) char ago[128 * PITCH];

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
  // depthwise phase: waves 0..5, block slot = wave*16 + blk -> (channel group cg of 24, row pair sub of 4)
  const int blk = lane >> 2, pi = lane & 3;
  const int dslot = wave * 16 + blk;
  const int cg = dslot % 24, sub = dslot / 24;
  const char* dbase = mid + ((2 * sub) * HW + pi) * PITCH + cg * 8;
  // this lane's output pixel in the P phase (tile row = wave) inside the image?
  const bool p_in = (oy0 + wave) < p.H && (ox0 + l15) < p.W;

  for (int ch = 0; ch < NCH; ++ch) {
    const int c0 = ch * 96;   // first qkv channel of the chunk (heads 2ch, 2ch + 1)
    // ================= E: mid[halo px][96] = Wqkv[c0 .. c0+96) . x =================
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      f32x16_v acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const T* wrow = gwq + (int64_t)(c0 + j * 32 + l31) * p.Kpq + g * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 fw = *reinterpret_cast<const u32x4*>(wrow + ks * 16);
        MmaOps<T>::mma(fw, fa[ks], acc);
      }
      const int hp = wave * 32 + l31;
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const uint32_t a0 = pack_bf16x2(acc[8 * qp + 0], acc[8 * qp + 1]), a1 = pack_bf16x2(acc[8 * qp + 2], acc[8 * qp + 3]);
        const uint32_t c0_ = pack_bf16x2(acc[8 * qp + 4], acc[8 * qp + 5]), c1_ = pack_bf16x2(acc[8 * qp + 6], acc[8 * qp + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0_, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1_, false, false);
        const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};  // channels j*32 + 16qp + 8g .. +8 of halo pixel hp
        *reinterpret_cast<u32x4*>(mid + hp * PITCH + ((j * 4 + qp * 2 + g) << 4)) = o;
      }
    }
    __syncthreads();

    // ================= D: dwo[px][96] = dw5x5(mid) =================
    if (wave < 6) {
      s16x4 wdg[25];
#pragma unroll
      for (int t = 0; t < 25; ++t) wdg[t] = diag_bf16(p.wdw[t * C3 + c0 + 4 * cg + pi], pi);
      f32x4 acc[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      DwMfma<5, 2, 4, 1, HW, PITCH>::run(lds_addr(dbase), wdg, acc);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int op = (2 * sub + r) * TW + 4 * q + pi;
          uint2 o;
          o.x = pack_bf16x2(acc[r * 4 + q][0], acc[r * 4 + q][1]);
          o.y = pack_bf16x2(acc[r * 4 + q][2], acc[r * 4 + q][3]);
          *reinterpret_cast<uint2*>(dwo + op * PITCH + cg * 8) = o;
        }
    }
    __syncthreads();

    // ================= P: ago[px][96] = grouped 1x1 of dwo; tile row = wave, 6 groups of 16 channels =================
    {
      const int op = wave * 16 + l15;
#pragma unroll
      for (int gi = 0; gi < 6; ++gi) {
        const s16x4 wa = *reinterpret_cast<const s16x4*>(gwg + (int64_t)(c0 + gi * 16 + l15) * p.Kpg + 4 * kg);
        const s16x4 xb = *reinterpret_cast<const s16x4*>(dwo + op * PITCH + gi * 32 + kg * 8);
        const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa, xb, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        uint2 o;   // channels gi*16 + 4kg .. +4 of pixel op; pixels of the tile that lie outside the image must not reach kv
        o.x = p_in ? pack_bf16x2(d[0], d[1]) : 0u;
        o.y = p_in ? pack_bf16x2(d[2], d[3]) : 0u;
        *reinterpret_cast<uint2*>(ago + op * PITCH + gi * 32 + kg * 8) = o;
      }
    }
    __syncthreads();

    // ================= KVQ =================
    // relu(q): 128 pixels x 8 16-byte pieces [scale][head][half]
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = tid + 512 * it;
      const int px = item >> 3, sl = item & 7;
      const int scale = sl >> 2, hh = (sl >> 1) & 1, half = sl & 1;
      const int py = px >> 4, pxx = px & 15;
      const char* src = (scale ? ago + px * PITCH : mid + ((py + 2) * HW + pxx + 2) * PITCH) + hh * 96 + half * 16;
      const uint4 v = *reinterpret_cast<const uint4*>(src);
      const s16x4 lo = relu_bf16x4(__builtin_bit_cast(s16x4, make_uint2(v.x, v.y)));
      const s16x4 hi = relu_bf16x4(__builtin_bit_cast(s16x4, make_uint2(v.z, v.w)));
      const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
      const int oy = oy0 + py, ox = ox0 + pxx;
      if (oy < p.H && ox < p.W)
        *reinterpret_cast<uint4*>(gq + ((int64_t)b * N + (int64_t)oy * p.W + ox) * (2 * C) + ch * 64 + sl * 8) =
            make_uint4(l2.x, l2.y, h2.x, h2.y);
    }
    // kv partials: wave = (group gsel of 4, pixel half); 4 tile rows of 16 pixels each = 4 MFMA steps
    {
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
    __syncthreads();
  }
}

// kv[b][g] = sum of the tile partials in a fixed order; written as the bf16 hi / lo MFMA operands of pass 2:
// tab[b][g][op][lane][4], op 0/1 = kv hi / lo (lane (m = dv, kq): kv[m][4kq .. 4kq+3]), op 2/3 = ksum hi / lo (every row m)
__global__ __launch_bounds__(256) void mla_kvprep_kernel(const float* __restrict__ kvp, bf16_t* __restrict__ tab, int P, int G) {
  const int bg = blockIdx.x;
  const int b = bg / G, gi = bg - b * G;
  const int t = threadIdx.x, m = t >> 4, k = t & 15;
  const float* src = kvp + ((int64_t)b * P * G + gi) * 272;
  float s = 0.f, sk = 0.f;
  for (int q = 0; q < P; ++q) {
    s += src[(int64_t)q * G * 272 + m * 16 + k];
    sk += src[(int64_t)q * G * 272 + 256 + k];
  }
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
  const void* qms;      // [B][N][2C]
  const void* tab;      // [B][2*heads][4][256] bf16
  const void* wp;       // packed [>=C][Kpp] bf16 (proj, BN folded)
  const float* bp;      // [C]
  const void* x;        // [B][N][C] shortcut
  void* out;            // [B][N][C]
  int B, N, Kpp, tiles;
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
  const unsigned b = bid / (unsigned)p.tiles;
  const int n0 = (int)(bid - b * (unsigned)p.tiles) * 128;
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

  for (int ch = 0; ch < NCH; ++ch) {
    char* at = atl[ch & 1];
    // ---- A: att chunk -> LDS ----
#pragma unroll
    for (int it = 0; it < IPW; ++it) {
      const int item = wave * IPW + it;
      const int pb = item >> 2, gl = item & 3;           // token block of 16, group within the chunk (scale*2 + head&1)
      const int gnat = (gl >> 1) * HEADS + 2 * ch + (gl & 1);
      const int n = n0 + pb * 16 + l15;
      s16x4 qf = {0, 0, 0, 0};
      if (n < p.N) qf = *reinterpret_cast<const s16x4*>(gq + ((int64_t)b * p.N + n) * (2 * C) + ch * 64 + gl * 16 + 4 * kg);
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
    const int n = n0 + ppt * 32 + l31;
    const bool ok = n < p.N;
    const int64_t row = (int64_t)b * p.N + n;
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
bool esam3_mbconv3_ok(int dtype, int Cin, int Cmid, int Cout, int stride) {
  if (dtype != 1 || Cmid % 64) return false;
  if (stride == 2)
    return (Cin == 16 && Cout == 32) || (Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 128) || (Cin == 128 && Cout == 256);
  if (stride == 1)
    return (Cin == 32 && Cout == 32) || (Cin == 64 && Cout == 64) || (Cin == 128 && Cout == 128) || (Cin == 256 && Cout == 256);
  return false;
}

int esam3_launch_mbconv3(const void* x, void* out, const void* w1, int Kp1, const float* b1, const float* wd, const float* bd,
                         const void* w2, int Kp2, const float* b2, int B, int H, int W, int Cin, int Cmid, int Cout, int stride,
                         int residual, hipStream_t stream) {
  if (!esam3_mbconv3_ok(1, Cin, Cmid, Cout, stride) || Kp1 < Cin || Kp2 < Cmid || (int64_t)B * H * W * Cin >= ((int64_t)1 << 31)) {
    esam3_set_error("mbconv3: unsupported configuration %d -> %d -> %d stride %d", Cin, Cmid, Cout, stride);
    return -1;
  }
  Mb3Params q{};
  q.x = x; q.out = out; q.w1 = w1; q.b1 = b1; q.wd = wd; q.bd = bd; q.w2 = w2; q.b2 = b2;
  q.B = B; q.H = H; q.W = W; q.OH = (H + stride - 1) / stride; q.OW = (W + stride - 1) / stride;
  q.Cmid = Cmid; q.Kp1 = Kp1; q.Kp2 = Kp2; q.residual = residual & 1;
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

// scratch: qms bf16 [B][N][2C], kvp fp32 [B][tiles*2][2*heads][272], tab bf16 [B][2*heads][1024]
void esam3_mla_fused_scratch(int B, int H, int W, int C, size_t* qms_bytes, size_t* kvp_bytes, size_t* tab_bytes) {
  const int tiles = ((W + 15) / 16) * ((H + 7) / 8), G = 2 * (C / 16);
  *qms_bytes = (size_t)B * H * W * 2 * C * 2;
  *kvp_bytes = (size_t)B * tiles * 2 * G * 272 * 4;
  *tab_bytes = (size_t)B * G * 1024 * 2;
}

// x [B][H][W][C] -> out = x + proj(att(qkv(x))) (BN folded into wproj / bproj).  wqkv [>=3C][Kpq], wgrp [3C][Kpg] (16 used),
// wproj [>=C][Kpp] packed bf16; wdw fp32 [25][3C]
int esam3_launch_mla_fused(const void* x, void* out, const void* wqkv, int Kpq, const float* wdw, const void* wgrp, int Kpg,
                           const void* wproj, int Kpp, const float* bproj, void* qms, float* kvp, void* tab, int B, int H, int W,
                           int C, hipStream_t stream) {
  if (!esam3_mla_fused_ok(1, C, 16) || Kpq < C || Kpg < 16 || Kpp < 2 * C || (int64_t)B * H * W * 2 * C >= ((int64_t)1 << 31)) {
    esam3_set_error("mla_fused: unsupported configuration C = %d", C);
    return -1;
  }
  Mla1Params a{};
  a.x = x; a.wqkv = wqkv; a.wdw = wdw; a.wgrp = wgrp; a.qms = qms; a.kvp = kvp;
  a.B = B; a.H = H; a.W = W; a.Kpq = Kpq; a.Kpg = Kpg;
  a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 7) / 8;
  const int tiles = a.tiles_x * a.tiles_y, G = 2 * (C / 16);
  if (C == 128) hipLaunchKernelGGL((mla1_kernel<128>), dim3((unsigned)(B * tiles)), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((mla1_kernel<256>), dim3((unsigned)(B * tiles)), dim3(512), 0, stream, a);
  HIP_CHECK_RET(hipGetLastError());
  hipLaunchKernelGGL(mla_kvprep_kernel, dim3((unsigned)(B * G)), dim3(256), 0, stream, kvp, reinterpret_cast<bf16_t*>(tab), tiles * 2, G);
  HIP_CHECK_RET(hipGetLastError());
  Mla2Params q{};
  q.qms = qms; q.tab = tab; q.wp = wproj; q.bp = bproj; q.x = x; q.out = out;
  q.B = B; q.N = H * W; q.Kpp = Kpp; q.tiles = (H * W + 127) / 128;
  if (C == 128) hipLaunchKernelGGL((mla2_kernel<128, 4>), dim3((unsigned)(B * q.tiles)), dim3(256), 0, stream, q);
  else hipLaunchKernelGGL((mla2_kernel<256, 8>), dim3((unsigned)(B * q.tiles)), dim3(512), 0, stream, q);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
