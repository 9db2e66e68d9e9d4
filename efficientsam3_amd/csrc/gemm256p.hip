// gemm256p_kernel: bf16 256x256x64 implicit GEMM for gfx950, phase-interleaved schedule.
//
// Serves the same layers as gemm256_kernel<bf16> in gemm_conv.hip (neck 3x3 / 1x1 / ConvT-k2s2 convs of
// model/necks.py:42-92, the head 3x3 of model_builder.py:770-775, the ViT-H Linears of model/vitdet.py:339-515)
// with the same operand convention (weights = MFMA A operand, pixels = B operand; a lane owns 4 consecutive
// channels of one pixel).  What differs is the schedule of the main loop and the epilogue:
//
//  * The eight wavefronts form two groups (wm = 0 / 1, one wave of each group per SIMD) that run half a phase
//    apart: while one group issues its 8 MFMAs of a phase, the other issues its ds_reads and LDS-DMA pieces.
//    A K tile (64 deep) is four phases, one 64-pixel x 32-channel quadrant of the wave's 128 x 64 block each:
//        phase 1: read B0 (4 x ds_read_b128) + A0 (8)   MFMA acc[0..1][0]      stage A1 of K tile t+1
//        phase 2: read B1 (4)                           MFMA acc[0..1][1]      stage B0 of K tile t+2
//        phase 3: read A1 (8, into A0's registers)      MFMA acc[2..3][1]      stage A0 of K tile t+2
//        phase 4: -                                     MFMA acc[2..3][0]      stage B1 of K tile t+2, vmcnt(6)
//    "A0/A1" = the rows every wave reads in phase 1 / 3 (its upper / lower 64 pixels), "B0/B1" = its first /
//    second 32 channels: the LDS image of a K tile is four 16 KB half tiles laid out by *reader*, so that a half
//    tile is dead for the whole workgroup one phase after it was read and can be re-staged while the rest of the
//    buffer is still in use.  One half tile (2 LDS-DMA pieces per wave) is staged per phase; three half tiles
//    stay in flight across the barriers behind the counted vmcnt(6) of phase 4.
//  * The staging stream is decoupled from the output tile: it runs ahead across the end of the K loop into the
//    next output tile of the persistent workgroup, so the DMA of the next tile's first two K tiles is already
//    in flight when the epilogue starts.
//  * Epilogue: bias / activation / residual in the accumulator layout, packed bf16 transposed through a
//    wave-private 4 KB LDS strip so that every global store instruction writes 8 complete 128-byte lines
//    (the accumulator layout itself gives 32 partial lines per instruction).
//
// Hazards (DMA write -> ds_read, ds_read -> DMA overwrite) are ordered by counted waits followed by a barrier
// that every reader / writer passes; the derivation is in DESIGN.md ("gemm256p: phase schedule").
#include "gemm_common.h"
#include "kernels.h"

#ifdef ESAM3_P_TRACE  /* dev build only (tools/dev_variants.sh): cycle stamps of workgroups 0-7, waves 0 and 4 */
__device__ unsigned long long g_trace[8 * 2 * 16 * 16];
#define ESAM3_TRACE(PT)                                                                              \
  do {                                                                                               \
    if (blockIdx.x < 8 && (wave & 3) == 0 && w < 16 && lane == 0)                                    \
      g_trace[((blockIdx.x * 2 + (wave >> 2)) * 16 + w) * 16 + (PT)] = clock64();                    \
  } while (0)
extern "C" int esam3_dev_read_trace(unsigned long long* host, int n) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), sizeof(unsigned long long) * n);
}
#else
#define ESAM3_TRACE(PT)
#endif

namespace {

// One LDS-DMA piece: 64 lanes x 16 B from (scalar base + per-lane 32-bit byte offset) to LDS [m0 .. m0+1024).
// Inline assembly on purpose: the compiler's wait-count pass must not see these (it would fence every following
// ds_read with vmcnt(0)).  s_nop 3: five wait states between a VALU-written SGPR base and the VMEM that reads it.
__device__ __forceinline__ void dma_piece(const void* base, uint32_t voff, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr)
               : "memory");
}

// Division by a launch constant as multiply-high + shift (the kernel divides by H*W, W, tiles per image, tiles per row and
// tiles along N several times per output tile -- in the stream look-ahead, the lane table of the store groups and the
// residual addresses -- and a 32-bit hardware-less division is ~25 VALU / SALU instructions).  Exact for n < 2^31 (the
// launcher guarantees M < 2^31): d a power of two -> shift; else s = ceil(log2 d), mul = ceil(2^(31+s) / d) < 2^32 and
// floor(n / d) = umulhi(n, mul) >> (s - 1)  [error term n * e / (d * 2^(31+s)) < 1 / d for e < d, n < 2^31].
struct FastDiv {
  uint32_t mul, shr;
};
inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f{0u, 0u};
  if (d == 0) d = 1;
  if ((d & (d - 1)) == 0) {
    while ((1u << f.shr) < d) ++f.shr;
    return f;  // mul == 0: shift only
  }
  uint32_t s = 0;
  while ((1ull << s) < d) ++s;
  f.mul = (uint32_t)((((unsigned __int128)1 << (31 + s)) + d - 1) / d);
  f.shr = s - 1;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
  return f.mul ? (__umulhi(n, f.mul) >> f.shr) : (n >> f.shr);
}
struct GemmDivs {
  FastDiv hw, w, tiles_img, tiles_x, tiles_n;
};

template <int ACT, bool RES, bool OUT32 = false>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmParams p, GemmDivs dv) {
  typedef bf16_t T;
  constexpr int BKE = 64;                  // K elements per tile (128 bytes)
  constexpr uint32_t HALF = 16384u;        // one half tile: 128 rows x 128 B
  constexpr uint32_t BUF = 65536u;         // A0 | A1 | B0 | B1
  constexpr uint32_t EPI = 131072u;        // 8 x 4 KB epilogue strips

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tiles_n = (p.N + 255) / 256;
  const unsigned tiles_m = (unsigned)((p.M + 255) / 256);
  const unsigned nblk = tiles_m * (unsigned)tiles_n;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wave block: pixels [wm*128,+128) x channels [wn*64,+64)
  const int l31 = lane & 31, g = lane >> 5;

  const T* __restrict__ gA = reinterpret_cast<const T*>(p.A);
  const T* __restrict__ gW = reinterpret_cast<const T*>(p.Wt);
  T* __restrict__ gO = reinterpret_cast<T*>(p.out);
  const T* __restrict__ gR = reinterpret_cast<const T*>(p.res);
  const int HW = p.H * p.W;
  const int Wp = p.W + 2 * p.in_pad;
  const int P = p.out_pad;
  const bool convt = p.out_mode == OUT_CONVT2X2;
  const int nk = p.K / BKE;
  const unsigned M32 = (unsigned)p.M;

  // 3x3 convs: a 256-row tile is a 16x16 pixel patch when the image tiles evenly, else 256 consecutive pixels
  const bool patch = p.ksize >= 2 && (p.H % 16 == 0) && (p.W % 16 == 0);
  const unsigned tiles_x = patch ? p.W / 16 : 1, tiles_img = patch ? (p.H / 16) * tiles_x : 1;
  auto row_to_m = [&](unsigned m0, int row) -> unsigned {
    if (!patch) return m0 + (unsigned)row;
    const unsigned t = m0 >> 8;
    const unsigned b = fdiv(t, dv.tiles_img);
    const unsigned ti = t - b * tiles_img;
    const unsigned ty = fdiv(ti, dv.tiles_x), tx = ti - ty * tiles_x;
    return b * (unsigned)HW + (ty * 16 + ((unsigned)row >> 4)) * (unsigned)p.W + tx * 16 + ((unsigned)row & 15);
  };
  auto a_row_off = [&](unsigned m) -> int64_t {  // element offset of pixel / token row m in A
    if (p.ksize >= 2) {  // zero-bordered input: padded pixel (oh, ow) = top-left tap of the 3x3 (or of class (0,0)'s 2x2)
      const unsigned b = fdiv(m, dv.hw);
      const unsigned rem = m - b * (unsigned)HW;
      const unsigned oh = fdiv(rem, dv.w), ow = rem - oh * (unsigned)p.W;
      return ((int64_t)(b * (unsigned)(p.H + 2) + oh) * Wp + ow) * p.lda;  // in_pad is required for ksize 2 / 3
    }
    return (int64_t)m * p.lda;
  };

  // ---- persistent workgroups, XCD-contiguous logical tile order (as gemm256_kernel) ----------------------
  const unsigned nwg = gridDim.x;
  const unsigned xcd = blockIdx.x % 8, wg_in_xcd = blockIdx.x / 8;
  const unsigned wgs_this_xcd = nwg / 8 + (xcd < nwg % 8 ? 1 : 0);
  const unsigned q_ = nblk / 8, r_ = nblk % 8;
  const unsigned xcd_first = xcd < r_ ? xcd * (q_ + 1) : r_ * (q_ + 1) + (xcd - r_) * q_;
  const unsigned xcd_count = q_ + (xcd < r_ ? 1 : 0);
  if (wg_in_xcd >= xcd_count) return;
  auto tile_of = [&](unsigned w, unsigned& m0, int& n0) {
    const unsigned lt = xcd_first + wg_in_xcd + w * wgs_this_xcd;
    const unsigned mt = fdiv(lt, dv.tiles_n);
    m0 = mt * 256u;
    n0 = (int)(lt - mt * (unsigned)tiles_n) * 256;
  };
  auto tile_exists = [&](unsigned w) -> bool { return wg_in_xcd + w * wgs_this_xcd < xcd_count; };

  // ---- staging stream -------------------------------------------------------------------------------------
  // Wave w fills physical rows [16w, 16w+16) of every half tile, 8 rows (1 KB) per piece; lane -> (row = 8j +
  // lane/8, physical 16-byte slot = lane%8); the logical slot it fetches is physical ^ ((row>>1)&7).
  // Physical row pr of half h holds:  A: tile row (pr>>6)*128 + h*64 + (pr&63)   B: channel (pr>>5)*64 + h*32 + (pr&31)
  uint32_t a_off[2][2], b_off[2][2];  // byte offsets from the wave-uniform bases below
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pr = wave * 16 + j * 8 + (lane >> 3);
      const int lslot = (lane & 7) ^ ((pr >> 1) & 7);
      const int ch = (pr >> 5) * 64 + h * 32 + (pr & 31);
      b_off[h][j] = (uint32_t)(((int64_t)ch * p.Kp + lslot * 8) * 2);
    }
  auto set_a_off = [&](unsigned m0, int64_t tileA) {
    int ln = lane;  // laundered: the row / slot values below are not to be hoisted out of the tile loop and kept (or spilled)
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pr = wave * 16 + j * 8 + (ln >> 3);
        const int lslot = (ln & 7) ^ ((pr >> 1) & 7);
        const int row = (pr >> 6) * 128 + h * 64 + (pr & 63);
        unsigned m = row_to_m(m0, row);
        if (m >= M32) m = M32 - 1;  // rows past M are computed but never stored
        a_off[h][j] = (uint32_t)((a_row_off(m) - tileA + lslot * 8) * 2);
      }
  };
  // tiles whose per-lane offsets differ from the generic full tile: ragged last M tile, 3x3 without patch tiling
  const bool a_off_varies = (p.ksize >= 2 && !patch) || (M32 % 256u) != 0;

  // ksize 2 = the "up-conv" gather (a ConvTranspose2d k2 s2 composed with the 3x3 conv that follows it, necks.py:42-92):
  // output pixel (2y + dy, 2x + dx) of class (dy, dx) = n0 / convt_cout depends on the 2 x 2 input pixels
  // (y - 1 + dy + kh, x - 1 + dx + kw); the N tile's class shifts the gather by (dy, dx), the K order is channel-chunk
  // major with 4 taps per chunk, the store is the ConvT pixel shuffle with tap = class.
  const int ntap = p.ksize == 2 ? 4 : 9;
  auto class_shift = [&](int n0_) -> int64_t {
    if (p.ksize != 2) return 0;
    const int cls = n0_ / p.convt_cout;
    return ((int64_t)(cls >> 1) * Wp + (cls & 1)) * p.lda;
  };
  int s_kt = 0, s_tap = 0, s_chunk = 0;
  uint32_t s_par = 0;          // stream: LDS buffer of the K tile being staged
  bool s_ok = true;            // stream not exhausted
  int64_t s_tileA = 0, s_tileB = 0, s_cls = 0;
  const T* sA = gA;
  const T* sB = gW;
  auto stream_bases = [&]() {
    int64_t koff = (int64_t)s_kt * BKE;
    if (p.ksize == 3) {  // K order: channel chunk major, 9 taps per chunk (korder 1)
      const int kh = (s_tap * 11) >> 5, kw = s_tap - kh * 3;
      koff = ((int64_t)kh * Wp + kw) * p.lda + s_chunk * BKE;
    } else if (p.ksize == 2) {
      koff = ((int64_t)(s_tap >> 1) * Wp + (s_tap & 1)) * p.lda + s_chunk * BKE + s_cls;
    }
    sA = gA + s_tileA + koff;
    sB = gW + s_tileB + (int64_t)s_kt * BKE;
  };
  // The wave-uniform bases of the NEXT output tile the stream will enter are worked out ahead of time, inside the
  // epilogue (whose waits hide the scalar divisions), not when the stream crosses the tile boundary in the K loop.
  int64_t nx_tileA = 0, nx_tileB = 0, nx_cls = 0;
  unsigned nx_m0 = 0;
  bool nx_ok = false;
  auto stream_look_ahead = [&](unsigned w) {  // tile ordinal w of this workgroup
    nx_ok = tile_exists(w);
    if (nx_ok) {
      int n0;
      tile_of(w, nx_m0, n0);
      nx_tileA = a_row_off(row_to_m(nx_m0, 0));
      nx_tileB = (int64_t)n0 * p.Kp;
      nx_cls = class_shift(n0);
    }
  };
  auto stream_advance = [&]() {
    if (!s_ok) return;
    ++s_kt;
    if (++s_tap == ntap) { s_tap = 0; ++s_chunk; }
    if (s_kt == nk) {
      s_kt = 0; s_tap = 0; s_chunk = 0;
      if (!nx_ok) { s_ok = false; return; }
      s_tileA = nx_tileA;
      s_tileB = nx_tileB;
      s_cls = nx_cls;
      if (a_off_varies) set_a_off(nx_m0, s_tileA);
    }
    s_par ^= 1u;
    stream_bases();
  };
  auto stream_advance_in_tile = [&]() {  // s_ok and s_kt + 1 < nk are known
    ++s_kt;
    if (++s_tap == ntap) { s_tap = 0; ++s_chunk; }
    s_par ^= 1u;
    stream_bases();
  };
#define ESAM3_STAGE(HSEL)                                                              \
  do {                                                                                 \
    if (s_ok) {                                                                        \
      const uint32_t dst_ = lds0 + s_par * BUF + (HSEL) * HALF + (uint32_t)wave * 2048u; \
      if ((HSEL) < 2) {                                                                \
        dma_piece(sA, a_off[(HSEL) & 1][0], dst_);                                     \
        dma_piece(sA, a_off[(HSEL) & 1][1], dst_ + 1024u);                             \
      } else {                                                                         \
        dma_piece(sB, b_off[(HSEL) & 1][0], dst_);                                     \
        dma_piece(sB, b_off[(HSEL) & 1][1], dst_ + 1024u);                             \
      }                                                                                \
    }                                                                                  \
  } while (0)
  // half selectors: 0 = A0, 1 = A1, 2 = B0, 3 = B1 (also the order of the half tiles inside a buffer)

  // ---- fragment read addresses (per lane, without the buffer base) ---------------------------------------
  uint32_t rdA[4], rdB[4];
#pragma unroll
  for (int ck = 0; ck < 4; ++ck) {
    rdA[ck] = (uint32_t)((wm * 64 + l31) * 128 + swz(l31, ck * 2 + g));
    rdB[ck] = (uint32_t)(2 * HALF + (wn * 32 + l31) * 128 + swz(l31, ck * 2 + g));
  }

  // ---- bias: folded into the accumulator initialisation -----------------------------------------------------
  // A lane's 32 bias values (channels 8q + 4g .. + 3 of both 32-channel halves of the wave's block) are the same for its
  // four pixel blocks.  They are fetched one output tile ahead -- here for the first tile, at the top of every epilogue for
  // the next one, where the epilogue's vmcnt(0) covers them -- so the K loop starts from acc = bias with nothing to wait
  // for and the epilogue has no bias arithmetic.
  // Residual variants are at the register limit in their epilogue (the values above would be live next to the residual
  // addresses): they keep acc = 0 and add this tile's bias in the epilogue.
  constexpr bool BIAS_INIT = !RES || OUT32;
  f32x4_v bq[2][4];
  auto load_bias = [&](int n0_) {
    const int nbw_ = n0_ + wn * 64;
    const int bn_ = (convt ? nbw_ % p.convt_cout : nbw_) + 4 * g;
    const bool on = p.bias && nbw_ < p.N;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4_v z = {0.f, 0.f, 0.f, 0.f};
        bq[j][q] = on ? *reinterpret_cast<const f32x4_v*>(p.bias + bn_ + j * 32 + 8 * q) : z;
      }
  };
  if constexpr (BIAS_INIT) {
    unsigned m0_;
    int n0_;
    tile_of(0, m0_, n0_);
    load_bias(n0_);
  }

  // ---- prologue: K tiles 0 and 1 of the first output tile ---------------------------------------------------
  stream_look_ahead(0);
  s_tileA = nx_tileA;
  s_tileB = nx_tileB;
  s_cls = nx_cls;
  set_a_off(nx_m0, s_tileA);
  stream_look_ahead(1);
  stream_bases();
  ESAM3_STAGE(2); ESAM3_STAGE(0); ESAM3_STAGE(3); ESAM3_STAGE(1);
  stream_advance();
  ESAM3_STAGE(2); ESAM3_STAGE(0); ESAM3_STAGE(3); ESAM3_STAGE(1);
  if (s_ok) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  uint32_t c_par = 0;       // compute: LDS buffer of the current K tile
  bool c_landed = false;    // the K tile after the current one is already known to have landed (set by the epilogue)
  char* const epi = smem + EPI + wave * 4096;
  // store side of the epilogue: lane -> (sp = lane >> 3: pixel 8k + sp of a 32-pixel block, sc = lane & 7: 16-byte chunk of its row)

  for (unsigned w = 0;; ++w) {
    unsigned m0;
    int n0;
    tile_of(w, m0, n0);
    const bool has_next = tile_exists(w + 1);

    f32x16_v acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = BIAS_INIT ? bq[j][r >> 2][r & 3] : 0.f;

    if (p.border_corr) {
      // up-conv: the composed bias assumes all nine taps of the 3x3 see the ConvT's output; at the ring of the OUTPUT image
      // the taps outside contribute nothing (the 3x3 pads with zeros, not with the ConvT's bias), so their share
      // sum_m W3[o][m][tap] * b_t[m] is taken back: border_corr[class][0 row edge | 1 column edge | 2 both][convt_cout],
      // added to the accumulators of the ring pixels before the K loop (exact: one fp32 rounding, as for the bias).
      const int cls = n0 / p.convt_cout, dy = cls >> 1, dx = cls & 1;
      const unsigned t = m0 >> 8;
      const unsigned b = fdiv(t, dv.tiles_img);
      const unsigned ti = t - b * tiles_img;
      const unsigned ty = fdiv(ti, dv.tiles_x), tx = ti - ty * tiles_x;
      const bool edge_tile = (dy ? ty == (unsigned)(p.H / 16 - 1) : ty == 0u) || (dx ? tx == tiles_x - 1 : tx == 0u);
      if (edge_tile && n0 + wn * 64 < p.N) {
        const int cbase = (n0 + wn * 64) % p.convt_cout + 4 * g;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wm * 128 + i * 32 + l31;
          const int py = (int)ty * 16 + (row >> 4), px = (int)tx * 16 + (row & 15);
          const bool er = dy ? py == p.H - 1 : py == 0, ec = dx ? px == p.W - 1 : px == 0;
          if (er || ec) {
            const float* cp = p.border_corr + ((int64_t)cls * 3 + (er ? (ec ? 2 : 0) : 1)) * p.convt_cout + cbase;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4_v c4 = *reinterpret_cast<const f32x4_v*>(cp + j * 32 + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += c4[e];
              }
          }
        }
      }
    }
    ESAM3_TRACE(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0

#define ESAM3_LDS16(OFF) (*reinterpret_cast<const u32x4*>(lbuf + (OFF)))
// The MFMA builtins are pure values to the compiler: without the two register pins it sinks them below the
// closing barrier (and below the next phase's ds_reads).  s_setprio sits outside the pins for the same reason.
#define ESAM3_MFMA8(FW, I0, J)                                                       \
  do {                                                                               \
    __builtin_amdgcn_s_barrier();                                                    \
    __builtin_amdgcn_s_setprio(1);                                                   \
    asm volatile("" : "+v"(acc[(I0)][(J)]), "+v"(acc[(I0) + 1][(J)]));               \
    _Pragma("unroll") for (int ck = 0; ck < 4; ++ck) {                               \
      MmaOps<T>::mma(FW[ck], fa[0][ck], acc[(I0)][(J)]);                             \
      MmaOps<T>::mma(FW[ck], fa[1][ck], acc[(I0) + 1][(J)]);                         \
    }                                                                                \
    asm volatile("" : "+v"(acc[(I0)][(J)]), "+v"(acc[(I0) + 1][(J)]));               \
    __builtin_amdgcn_s_setprio(0);                                                   \
    __builtin_amdgcn_s_barrier();                                                    \
  } while (0)
// One K tile = four phases.  ADVANCE moves the staging stream to the K tile two ahead of the one being computed;
// that K tile is staged in phases 2-4 (B0 and A1 one phase after their last read: the reads are retired by a counted
// lgkmcnt before the phase's first barrier, so the other group's DMA cannot overtake them).
#define ESAM3_KTILE(ADVANCE)                                                         \
  do {                                                                               \
    const char* lbuf = smem + c_par * BUF;                                           \
    /* ---- phase 1: B0 + A0 -> acc[0..1][0] ---- */                                 \
    _Pragma("unroll") for (int ck = 0; ck < 4; ++ck) fb0[ck] = ESAM3_LDS16(rdB[ck]); \
    __builtin_amdgcn_sched_barrier(0);                                               \
    _Pragma("unroll") for (int ck = 0; ck < 4; ++ck) {                               \
      fa[0][ck] = ESAM3_LDS16(rdA[ck]);                                              \
      fa[1][ck] = ESAM3_LDS16(rdA[ck] + 4096u);                                      \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                               \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); /* B0 reads (issued first) retired */ \
    ESAM3_MFMA8(fb0, 0, 0);                                                          \
    /* ---- phase 2: B1 -> acc[0..1][1]; stage B0 of K tile t+2 ---- */              \
    _Pragma("unroll") for (int ck = 0; ck < 4; ++ck) fb1[ck] = ESAM3_LDS16(rdB[ck] + HALF); \
    __builtin_amdgcn_sched_barrier(0);                                               \
    ADVANCE;                                                                         \
    ESAM3_STAGE(2);                                                                  \
    ESAM3_MFMA8(fb1, 0, 1);                                                          \
    /* ---- phase 3: A1 (into A0's registers) -> acc[2..3][1]; stage A0 of K tile t+2 ---- */ \
    _Pragma("unroll") for (int ck = 0; ck < 4; ++ck) {                               \
      fa[0][ck] = ESAM3_LDS16(rdA[ck] + HALF);                                       \
      fa[1][ck] = ESAM3_LDS16(rdA[ck] + HALF + 4096u);                               \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                               \
    ESAM3_STAGE(0);                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* A1 reads retired: A1 is re-staged in the next phase */ \
    ESAM3_MFMA8(fb1, 2, 1);                                                          \
    /* ---- phase 4: no reads -> acc[2..3][0]; stage B1 + A1 of K tile t+2; K tile t+1 has landed ---- */ \
    ESAM3_STAGE(3);                                                                  \
    ESAM3_STAGE(1);                                                                  \
    if (c_landed) c_landed = false;                                                  \
    else if (s_ok) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            \
    ESAM3_MFMA8(fb0, 2, 0);                                                          \
    c_par ^= 1u;                                                                     \
  } while (0)

    u32x4 fa[2][4], fb0[4], fb1[4];
    // steady state: the stream stays inside this output tile (K tile kt+2 exists)
    for (int kt = 0; kt < nk - 2; ++kt) {
      ESAM3_KTILE(stream_advance_in_tile());
      if (kt == 0) ESAM3_TRACE(1);
    }
    ESAM3_TRACE(2);
    // last two K tiles: the stream crosses into the next output tile (or ends)
    ESAM3_KTILE(stream_advance());
    ESAM3_KTILE(stream_advance());
#undef ESAM3_KTILE
#undef ESAM3_MFMA8
#undef ESAM3_LDS16
    ESAM3_TRACE(3);
    // Lane-derived epilogue values are recomputed here from a laundered copy of the lane id: hoisted out of the tile loop
    // (they are loop invariants) they would live through the K loop, which has no register to spare.
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int l31 = lane_e & 31, g = lane_e >> 5, sp = lane_e >> 3, sc = lane_e & 7;
    // the NEXT tile's bias values are requested first: they land under the last barrier and the address set-up
    const int nbw = n0 + wn * 64;  // first channel of this wave's 64-channel block (wave-uniform)
    if constexpr (BIAS_INIT) {  // (without a next tile this re-reads the current tile's values: the registers are not loop-carried state then)
      unsigned nm0_;
      int nn0_;
      tile_of(has_next ? w + 1 : w, nm0_, nn0_);
      load_bias(nn0_);
    } else {
      load_bias(n0);
    }
#define ESAM3_PIN_BIAS()                                                                                   \
  if constexpr (BIAS_INIT)                                                                                 \
  asm volatile("" : "+v"(bq[0][0]), "+v"(bq[0][1]), "+v"(bq[0][2]), "+v"(bq[0][3]), "+v"(bq[1][0]), "+v"(bq[1][1]), \
               "+v"(bq[1][2]), "+v"(bq[1][3]))
    if (wm == 0) __builtin_amdgcn_s_barrier();  // both groups leave the K loop behind the same barrier
    ESAM3_TRACE(4);

    // =========================== epilogue ===========================
    bool looked = false;  // stream_look_ahead(w + 2) done (it is placed between the store blocks, under their waits)
#ifdef ESAM3_P_NOSTORE  /* ablation build (tools/dev_variants.sh): keep the accumulators alive, no epilogue (wrong results) */
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) asm volatile("" ::"v"(acc[i][j]));
    if (false) {
#else
    if (nbw < p.N) {
#endif
     if constexpr (OUT32) {
      // fp32 output (+ fp32 residual), plain rows, no activation.  One 32-pixel x 32-channel block of fp32 values is exactly
      // a 4 KB strip: it goes through the wave-private LDS strip like the bf16 blocks do, so that every global access (the
      // residual read and the store) covers complete 128-byte lines (8 lanes per row) instead of 32-byte pieces.
      float* __restrict__ gO32 = reinterpret_cast<float*>(p.out);
      const float* __restrict__ gR32 = reinterpret_cast<const float*>(p.res);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next tile's first K tiles have landed (see the bf16 path)
      ESAM3_PIN_BIAS();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
            const int c = 2 * q + g;  // 16-byte chunk of pixel l31's 128-byte row: channels 8q + 4g .. + 3
            *reinterpret_cast<float4*>(epi + l31 * 128 + ((c ^ (l31 & 7)) << 4)) = v;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            float4 v = *reinterpret_cast<const float4*>(epi + (8 * k + sp) * 128 + ((sc ^ sp) << 4));
            const unsigned m = m0 + (unsigned)(wm * 128 + i * 32 + 8 * k + sp);
            const int n = nbw + j * 32 + 4 * sc;
            if (m < M32) {
              if constexpr (RES) {
                const unsigned rrow = p.res_mod > 0 ? m % (unsigned)p.res_mod : m;
                const float4 r = *reinterpret_cast<const float4*>(gR32 + (int64_t)rrow * p.ldr + n);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
              }
              *reinterpret_cast<float4*>(gO32 + (int64_t)m * p.ldc + n) = v;
            }
          }
        }
        if (i == 1 && has_next) { stream_look_ahead(w + 2); looked = true; }
      }
     } else {
      // channel-direction part of the addresses (a 64-channel block never straddles a ConvT tap: Cout % 64 == 0)
      int64_t ocol = nbw, rcol = nbw;
      if (convt) {
        const int tap = nbw / p.convt_cout, co = nbw - tap * p.convt_cout;
        const int OWp = 2 * p.W + 2 * P;
        ocol = ((int64_t)(tap >> 1) * OWp + (tap & 1)) * p.ldc + co;
        rcol = ((int64_t)(tap >> 1) * (2 * p.W) + (tap & 1)) * p.ldr + co;
      }
      // residual rows in the accumulator layout (lane's own pixel l31 of block i)
      int64_t rbase[4];
      bool rok[4];
      if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned m = row_to_m(m0, wm * 128 + i * 32 + l31);
          rok[i] = m < M32;
          const unsigned mm = rok[i] ? m : 0u;
          const unsigned b = fdiv(mm, dv.hw);
          const unsigned rem = mm - b * (unsigned)HW;
          if (convt) {
            const unsigned h = fdiv(rem, dv.w), ww = rem - h * (unsigned)p.W;
            const unsigned rb = p.res_bidx ? (unsigned)p.res_bidx[b] : b;
            rbase[i] = ((int64_t)(rb * 2u * p.H + 2 * h) * (2 * p.W) + 2 * ww) * p.ldr + rcol;
          } else {
            unsigned rrow = mm;
            if (p.res_mod > 0) rrow = mm % (unsigned)p.res_mod;
            else if (p.res_bidx) rrow = (unsigned)p.res_bidx[b] * (unsigned)HW + rem;
            rbase[i] = (int64_t)rrow * p.ldr + rcol;
          }
        }
      }
      // Store rows R = wm*128 + 8*s + sp, s = 0..15, walked incrementally: the element offset of the row advances by
      // inc[s & 1] per step (patch tiles: 8 pixels to the right, then down one image row and 8 back; linear rows: 8
      // rows).  Outputs with a border or a ConvT pixel shuffle on linear rows also carry an (image row, column)
      // walker: when the column wraps the offset skips the border / the second output row (d1), when the row wraps
      // it skips the border rows between images (d2).  W >= 8, so a step wraps at most once.
      //
      // When every 8-row store group lies inside one image row (patch tiles; plain rows; W % 8 == 0) and the tile is full,
      // the 16 group offsets of the wave are computed ONCE, lane-parallel (lane s & 15 -> group s: one pair of divisions),
      // as 32-bit byte offsets from the wave's first group; a store then costs v_readlane + v_add + `global_store_dwordx4
      // voff, data, s[base]` with the lane's constant offset (row sp of the group, 16-byte chunk sc).  The per-lane walk
      // (64-bit VALU add, compare, select and an exec-mask branch per store) remains for the other shapes.
      const bool walker = !patch && (convt || P);
      bool ufast = m0 + 256u <= M32 && (!walker || (p.W % 8) == 0);
      uint32_t grel = 0;          // lane (s & 15): byte offset of store group s from the wave's group 0
      char* tb = nullptr;         // wave-uniform: address of group 0, channel block included
      const uint32_t lane_off = (uint32_t)(((uint32_t)sp * (uint32_t)(convt ? 2 * p.ldc : p.ldc) + (uint32_t)sc * 8u) * 2u);
      if (ufast) {
        const unsigned mg = row_to_m(m0, wm * 128 + 8 * (lane_e & 15));
        int64_t goff;
        if (!patch && !walker) {
          goff = (int64_t)mg * p.ldc;
        } else {
          const unsigned b = fdiv(mg, dv.hw);
          const unsigned rem = mg - b * (unsigned)HW;
          const unsigned h = fdiv(rem, dv.w), ww = rem - h * (unsigned)p.W;
          if (convt) goff = ((int64_t)(b * (unsigned)(2 * p.H + 2 * P) + 2 * h + P) * (2 * p.W + 2 * P) + 2 * ww + P) * p.ldc;
          else goff = ((int64_t)(b * (unsigned)(p.H + 2 * P) + h + P) * (p.W + 2 * P) + ww + P) * p.ldc;
        }
        const uint32_t b_lo = __builtin_amdgcn_readfirstlane((uint32_t)goff);
        const uint32_t b_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)goff >> 32));
        const int64_t gbase = (int64_t)(((uint64_t)b_hi << 32) | b_lo);  // lane 0 holds group 0
        const uint64_t rel = (uint64_t)(goff - gbase) * 2u;
        ufast = __all(rel < (1ull << 31));
        grel = (uint32_t)rel;
        tb = reinterpret_cast<char*>(gO) + ((gbase + ocol) << 1);
      }
      // ---- per-lane walk ----
      unsigned mrow = m0 + (unsigned)(wm * 128 + sp);
      unsigned ph = 0, pw = 0;
      int64_t cur = 0, inc0 = 0, inc1 = 0, d1 = 0, d2 = 0;
      if (!ufast) {
        if (patch) {
          const unsigned t = m0 >> 8;
          const unsigned b = fdiv(t, dv.tiles_img);
          const unsigned ti = t - b * tiles_img;
          const unsigned ty = fdiv(ti, dv.tiles_x), tx = ti - ty * tiles_x;
          const int64_t pitch = (int64_t)(p.W + 2 * P);
          cur = (((int64_t)b * (p.H + 2 * P) + ty * 16 + P + wm * 8) * pitch + tx * 16 + P + sp) * p.ldc;
          inc0 = 8 * (int64_t)p.ldc;
          inc1 = (pitch - 8) * p.ldc;
        } else if (!walker) {
          cur = (int64_t)mrow * p.ldc;
          inc0 = inc1 = 8 * (int64_t)p.ldc;
        } else {
          const unsigned pb = fdiv(mrow, dv.hw);
          const unsigned rem = mrow - pb * (unsigned)HW;
          ph = fdiv(rem, dv.w);
          pw = rem - ph * (unsigned)p.W;
          if (convt) {
            const int64_t OWp = 2 * p.W + 2 * P;
            cur = ((int64_t)(pb * (unsigned)(2 * p.H + 2 * P) + 2 * ph + P) * OWp + 2 * pw + P) * p.ldc;
            inc0 = inc1 = 16 * (int64_t)p.ldc;
            d1 = (2 * OWp - 2 * p.W) * p.ldc;
            d2 = 2 * P * OWp * p.ldc;
          } else {
            cur = ((int64_t)(pb * (unsigned)(p.H + 2) + ph + 1) * (p.W + 2) + pw + 1) * p.ldc;
            inc0 = inc1 = 8 * (int64_t)p.ldc;
            d1 = 2 * (int64_t)p.ldc;
            d2 = 2 * (int64_t)(p.W + 2) * p.ldc;
          }
        }
        cur += ocol + sc * 8;
      }
      ESAM3_TRACE(5);
      // block i of the wave (32 pixels x 64 channels): residual / activation in the accumulator layout, packed bf16,
      // half-wave exchange, into the wave's LDS strip (pixel rows of 128 B, 16-byte chunks XOR-swizzled by the row)
      auto pack_block = [&](int i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int qp = 0; qp < 2; ++qp) {  // 8 values at a time (channels 16qp + 8q' + 4g + e): bounded live ranges under GELU
            float v[8], r8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
              v[e] = BIAS_INIT ? acc[i][j][8 * qp + e] : acc[i][j][8 * qp + e] + bq[j][2 * qp + (e >> 2)][e & 3];
            if constexpr (RES) {
#pragma unroll
              for (int q = 0; q < 2; ++q) {
                uint2 u = make_uint2(0u, 0u);
                if (rok[i]) u = *reinterpret_cast<const uint2*>(gR + rbase[i] + j * 32 + 8 * (2 * qp + q) + 4 * g);
                r8[4 * q + 0] = __uint_as_float(u.x << 16); r8[4 * q + 1] = __uint_as_float(u.x & 0xffff0000u);
                r8[4 * q + 2] = __uint_as_float(u.y << 16); r8[4 * q + 3] = __uint_as_float(u.y & 0xffff0000u);
              }
              if (!p.res_after_act) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r8[e];
              }
            }
            act_apply_n<8>(v, ACT);
            if constexpr (RES) {
              if (p.res_after_act) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += r8[e];
              }
            }
            const uint32_t a0 = pack_bf16x2(v[0], v[1]);
            const uint32_t a1 = pack_bf16x2(v[2], v[3]);
            const uint32_t b0 = pack_bf16x2(v[4], v[5]);
            const uint32_t b1 = pack_bf16x2(v[6], v[7]);
            // half-wave exchange: lanes 0-31 end with channels 16qp..16qp+7, lanes 32-63 with +8..+15
            auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
            const u32x4 o = {s0[0], s1[0], s0[1], s1[1]};
            const int c = j * 4 + qp * 2 + g;  // 16-byte chunk of pixel l31's 128-byte row
            *reinterpret_cast<u32x4*>(epi + l31 * 128 + ((c ^ (l31 & 7)) << 4)) = o;
          }
        }
      };
      // what sits between a block's packing and its stores
      auto between = [&](int i) {
        if (i == 0) { asm volatile("" ::"v"(epi)); ESAM3_TRACE(6); }
        if (i == 1 && has_next) { stream_look_ahead(w + 2); looked = true; }
        if (i == 0) {
          // Every LDS-DMA piece issued so far (the next output tile's K tiles 0 and 1) has had the last phases and
          // this block's arithmetic to land; waiting for them HERE, before the first store, lets the next tile's
          // first K tile skip its vmcnt wait, which would otherwise also wait for all of the stores below
          // (vmcnt counts stores).  The next tile's bias values (load_bias above) are covered by the same wait.
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          ESAM3_PIN_BIAS();
          ESAM3_TRACE(7);
        }
      };
      // rows back out of the strip: 8 complete 128-byte lines per store instruction
      if (ufast) {  // full tile: no bounds checks, group offsets out of the lane table
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pack_block(i);
          between(i);
          u32x4 o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = *reinterpret_cast<const u32x4*>(epi + (8 * k + sp) * 128 + ((sc ^ sp) << 4));
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t voff = (uint32_t)__builtin_amdgcn_readlane((int)grel, 4 * i + k) + lane_off;
            *reinterpret_cast<u32x4*>(tb + voff) = o[k];
          }
          __builtin_amdgcn_sched_barrier(0);  // one block at a time: interleaving the blocks costs more registers than there are
          ESAM3_TRACE(8 + i);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pack_block(i);
          between(i);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const u32x4 o = *reinterpret_cast<const u32x4*>(epi + (8 * k + sp) * 128 + ((sc ^ sp) << 4));
            if (mrow < M32) *reinterpret_cast<u32x4*>(gO + cur) = o;
            // next store row
            mrow += 8;
            cur += (k & 1) ? inc1 : inc0;
            if (walker) {
              pw += 8;
              const bool c1 = pw >= (unsigned)p.W;
              pw -= c1 ? (unsigned)p.W : 0u;
              ph += c1 ? 1u : 0u;
              const bool c2 = ph >= (unsigned)p.H;
              ph -= c2 ? (unsigned)p.H : 0u;
              cur += (c1 ? d1 : 0) + (c2 ? d2 : 0);
            }
          }
          ESAM3_TRACE(8 + i);
        }
      }
     }
      c_landed = has_next;
    }
    if (!has_next) break;
    if (!looked) stream_look_ahead(w + 2);
  }
#undef ESAM3_STAGE
}

}  // namespace

bool esam3_gemm256p_ok(const GemmParams& p) {
  // the caller has already checked the gemm256 conditions (use_256); what this kernel needs on top of them
  if (p.K < 128) return false;
  if (p.N % 64 != 0) return false;
  if (p.out_mode == OUT_CONVT2X2 && p.convt_cout % 64 != 0) return false;
  if (p.ksize == 3 && !p.korder) return false;
  if (p.ksize == 2) {  // up-conv gather: one class per 256-wide N tile, patch tiles, full tiles, no residual / fp32 output
    if (!p.in_pad || p.out_mode != OUT_CONVT2X2 || p.convt_cout % 256 != 0 || p.N != 4 * p.convt_cout) return false;
    if (p.H % 16 != 0 || p.W % 16 != 0 || p.M % 256 != 0 || p.Cin % 64 != 0 || p.K != 4 * p.Cin) return false;
    if (p.res || p.out_f32 || p.res_bidx) return false;
    if (p.border_corr && (((uintptr_t)p.border_corr) & 15)) return false;
  } else if (p.border_corr) {
    return false;
  }
  if ((p.out_mode == OUT_CONVT2X2 || p.out_pad) && p.W < 8) return false;  // epilogue row walker: one wrap per 8-pixel step
  if ((((uintptr_t)p.out) & 15) || (((uintptr_t)p.A) & 15) || (((uintptr_t)p.Wt) & 15)) return false;
  if (p.bias && (((uintptr_t)p.bias) & 15)) return false;
  if (p.out_f32 && (p.act != ACT_NONE || p.out_mode != OUT_PLAIN || p.out_pad || p.ksize != 1 || p.res_bidx || p.ldc % 4 || (p.res && p.ldr % 4) ||
                    (p.res && (((uintptr_t)p.res) & 15))))
    return false;
  return true;
}

int esam3_launch_gemm256p(const GemmParams& p, hipStream_t stream) {
  constexpr size_t lds = 163840;  // 2 x 64 KB K-tile buffers + 8 x 4 KB epilogue strips = all of a CU's LDS
  GemmDivs dv;
  {
    const bool patch = p.ksize >= 2 && p.H % 16 == 0 && p.W % 16 == 0;
    const uint32_t tiles_x = patch ? (uint32_t)p.W / 16 : 1, tiles_img = patch ? (uint32_t)(p.H / 16) * tiles_x : 1;
    dv.hw = make_fastdiv((uint32_t)p.H * (uint32_t)p.W);
    dv.w = make_fastdiv((uint32_t)p.W);
    dv.tiles_img = make_fastdiv(tiles_img);
    dv.tiles_x = make_fastdiv(tiles_x);
    dv.tiles_n = make_fastdiv((uint32_t)((p.N + 255) / 256));
  }
  void (*kerns[10])(GemmParams, GemmDivs) = {
      gemm256p_kernel<ACT_NONE, false>, gemm256p_kernel<ACT_RELU, false>, gemm256p_kernel<ACT_GELU, false>,
      gemm256p_kernel<ACT_HSWISH, false>, gemm256p_kernel<ACT_SIGMOID, false>,
      gemm256p_kernel<ACT_NONE, true>, gemm256p_kernel<ACT_RELU, true>, gemm256p_kernel<ACT_GELU, true>,
      gemm256p_kernel<ACT_HSWISH, true>, gemm256p_kernel<ACT_SIGMOID, true>};
  if (p.act < 0 || p.act > 4) { esam3_set_error("gemm: bad activation %d", p.act); return -1; }
  const int64_t tiles = ((p.M + 255) / 256) * ((p.N + 255) / 256);
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    HIP_CHECK_RET(hipGetDevice(&dev));
    HIP_CHECK_RET(hipGetDeviceProperties(&prop, dev));
    n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  int64_t grid = tiles < n_cu ? tiles : n_cu;  // persistent: one workgroup per CU
  // -DESAM3_DEV builds: ESAM3_P_GRID=n caps the persistent grid (how much of a power-bound launch's time depends on the CU count?)
  if (const int cap = esam3_dev_flag("ESAM3_P_GRID")) grid = grid < cap ? grid : cap;
  if (p.out_f32) {  // fp32 output / residual stream (esam3_gemm256p_ok has checked: no activation, plain rows)
    void (*k32)(GemmParams, GemmDivs) = p.res ? gemm256p_kernel<ACT_NONE, true, true> : gemm256p_kernel<ACT_NONE, false, true>;
    if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(k32), (int)lds)) return -1;
    hipLaunchKernelGGL(k32, dim3((unsigned)grid), dim3(512), lds, stream, p, dv);
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  }
  if (esam3_allow_dyn_lds(reinterpret_cast<const void*>(kerns[p.act + (p.res ? 5 : 0)]), (int)lds)) return -1;
  hipLaunchKernelGGL(kerns[p.act + (p.res ? 5 : 0)], dim3((unsigned)grid), dim3(512), lds, stream, p, dv);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
