// COCO run-length encoding of binary masks on the device: the evaluation writers' mask -> RLE step
// (scripts/eval/gold/eval_efficientsam3_all_subsets.py:124-135 via pycocotools; sam3/sam3/train/masks_ops.py:161-230
// rle_encode does the same with torch ops).  Byte / integer work, HBM-bound: every mask byte is read three times
// (transpose, count, emit) and written once.
//   masks u8 [n][H][W] (non-zero = foreground)  ->  runs in column-major (Fortran) order: counts of zeros, ones,
//   zeros, ... (the first count is 0 when the mask starts with a one), all masks back to back + offsets [n+1].
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/esam3.h"
#include "esam3_common.h"
#include "kernels.h"

namespace {

constexpr int RLE_EPT = 16;                 // bytes per thread
constexpr int RLE_CHUNK = 256 * RLE_EPT;    // bytes per workgroup

// [n][H][W] -> [n][Lp] column-major (x major, y minor); Lp = H*W rounded up to 16
__global__ __launch_bounds__(256) void rle_transpose_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int H,
                                                            int W, int64_t Lp) {
  __shared__ uint8_t tile[64][65];
  const int n = blockIdx.z;
  const int x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const uint8_t* src = in + (int64_t)n * H * W;
  for (int r = ty; r < 64; r += 4) {
    const int y = y0 + r, x = x0 + tx;
    tile[r][tx] = (y < H && x < W) ? (src[(int64_t)y * W + x] != 0) : 0;
  }
  __syncthreads();
  uint8_t* dst = out + (int64_t)n * Lp;
  for (int r = ty; r < 64; r += 4) {
    const int x = x0 + r, y = y0 + tx;
    if (x < W && y < H) dst[(int64_t)x * H + y] = tile[tx][r];
  }
}

// transitions inside the 16 bytes that start at f0 (bit e = the value changes at f0 + e; position 0 counts as a
// change when the mask starts with a one)
__device__ inline uint32_t rle_flags16(const uint8_t* __restrict__ m, int64_t f0, int64_t L) {
  if (f0 >= L) return 0u;
  uint8_t v[RLE_EPT];
  if (f0 + RLE_EPT <= L) {
    const uint4 q = *reinterpret_cast<const uint4*>(m + f0);
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < RLE_EPT; ++e) v[e] = (uint8_t)((w[e >> 2] >> (8 * (e & 3))) & 0xffu);
  } else {
#pragma unroll
    for (int e = 0; e < RLE_EPT; ++e) v[e] = (f0 + e < L) ? m[f0 + e] : 0;
  }
  uint8_t prev = f0 > 0 ? m[f0 - 1] : 0;
  uint32_t flags = 0;
#pragma unroll
  for (int e = 0; e < RLE_EPT; ++e) {
    if (f0 + e < L && v[e] != prev) flags |= 1u << e;
    prev = v[e];
  }
  return flags;
}

__global__ __launch_bounds__(256) void rle_count_kernel(const uint8_t* __restrict__ tr, int64_t L, int64_t Lp, int chunks,
                                                        uint32_t* __restrict__ chunk_count) {
  __shared__ uint32_t sred[4];
  const int n = blockIdx.y, c = blockIdx.x;
  const int64_t f0 = ((int64_t)c * 256 + threadIdx.x) * RLE_EPT;
  uint32_t cnt = __popc(rle_flags16(tr + (int64_t)n * Lp, f0, L));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if ((threadIdx.x & 63) == 0) sred[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) chunk_count[(int64_t)n * chunks + c] = sred[0] + sred[1] + sred[2] + sred[3];
}

// per mask: exclusive scan of its chunk counts (in place) and its number of runs = transitions + 1
__global__ __launch_bounds__(256) void rle_scan_mask_kernel(uint32_t* __restrict__ chunk_count, int chunks,
                                                            uint32_t* __restrict__ runs) {
  __shared__ uint32_t swave[4];
  __shared__ uint32_t carry;
  const int n = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t* cc = chunk_count + (int64_t)n * chunks;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < chunks; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < chunks ? cc[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) swave[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += swave[w];
    const uint32_t c0 = carry;
    if (i < chunks) cc[i] = c0 + woff + inc - v;
    __syncthreads();
    if (threadIdx.x == 255) carry = c0 + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) runs[n] = carry + 1u;
}

// offsets[0..n] = exclusive scan of runs (one workgroup; n is small)
__global__ __launch_bounds__(256) void rle_scan_offsets_kernel(const uint32_t* __restrict__ runs, int n,
                                                               int32_t* __restrict__ offsets) {
  __shared__ uint32_t swave[4];
  __shared__ uint32_t carry;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 256) {
    const int i = base + threadIdx.x;
    const uint32_t v = i < n ? runs[i] : 0u;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) swave[wave] = inc;
    __syncthreads();
    uint32_t woff = 0;
    for (int w = 0; w < wave; ++w) woff += swave[w];
    const uint32_t c0 = carry;
    if (i < n) offsets[i] = (int32_t)(c0 + woff + inc - v);
    __syncthreads();
    if (threadIdx.x == 255) carry = c0 + woff + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[n] = (int32_t)carry;
}

// positions of the transitions, compacted: pos[offsets[n] + k] = k-th transition of mask n
__global__ __launch_bounds__(256) void rle_emit_kernel(const uint8_t* __restrict__ tr, int64_t L, int64_t Lp, int chunks,
                                                       const uint32_t* __restrict__ chunk_base,
                                                       const int32_t* __restrict__ offsets, uint32_t* __restrict__ pos,
                                                       int64_t capacity) {
  __shared__ uint32_t swave[4];
  const int n = blockIdx.y, c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t f0 = ((int64_t)c * 256 + threadIdx.x) * RLE_EPT;
  const uint32_t flags = rle_flags16(tr + (int64_t)n * Lp, f0, L);
  const uint32_t v = __popc(flags);
  uint32_t inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) swave[wave] = inc;
  __syncthreads();
  uint32_t woff = 0;
  for (int w = 0; w < wave; ++w) woff += swave[w];
  int64_t idx = (int64_t)offsets[n] + chunk_base[(int64_t)n * chunks + c] + woff + inc - v;
  uint32_t fl = flags;
  while (fl) {
    const int e = __ffs(fl) - 1;
    fl &= fl - 1;
    if (idx < capacity) pos[idx] = (uint32_t)(f0 + e);
    ++idx;
  }
}

// run lengths from the transition positions: counts[j] = pos[j] - pos[j-1], first = pos[0], last = L - pos[last]
__global__ __launch_bounds__(256) void rle_diff_kernel(const uint32_t* __restrict__ pos, const int32_t* __restrict__ offsets,
                                                       uint32_t L, uint32_t* __restrict__ counts, int64_t capacity) {
  const int n = blockIdx.y;
  const int64_t o0 = offsets[n], nr = (int64_t)offsets[n + 1] - o0;  // runs of this mask = transitions + 1
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < nr; j += (int64_t)gridDim.x * 256) {
    if (o0 + j >= capacity) return;
    const uint32_t hi = j < nr - 1 ? pos[o0 + j] : L;
    const uint32_t lo = j > 0 ? pos[o0 + j - 1] : 0u;
    counts[o0 + j] = hi - lo;
  }
}

}  // namespace

static inline int64_t rle_lp(int H, int W) { return (((int64_t)H * W + 15) / 16) * 16; }
static inline int rle_chunks(int H, int W) { return (int)(((int64_t)H * W + RLE_CHUNK - 1) / RLE_CHUNK); }

int64_t esam3_rle_scratch_bytes(int n, int H, int W, int64_t capacity) {
  // transposed masks | chunk counts | runs per mask | transition positions
  return (int64_t)n * rle_lp(H, W) + 16 + ((int64_t)n * rle_chunks(H, W) + n + 4) * 4 + capacity * 4 + 64;
}

int esam3_launch_rle_encode(const uint8_t* masks, int n, int H, int W, uint32_t* counts, int64_t capacity, int32_t* offsets,
                            void* scratch, hipStream_t s) {
  const int64_t L = (int64_t)H * W, Lp = rle_lp(H, W);
  if (L >= (int64_t)1 << 32) { esam3_set_error("rle_encode: %d x %d masks are too large", H, W); return -1; }
  const int chunks = rle_chunks(H, W);
  uint8_t* tr = (uint8_t*)scratch;
  uint32_t* chunk_count = (uint32_t*)(tr + (((int64_t)n * Lp + 15) / 16) * 16);
  uint32_t* runs = chunk_count + (int64_t)n * chunks;
  uint32_t* pos = runs + n + 4 - (n % 4);
  hipLaunchKernelGGL(rle_transpose_kernel, dim3((W + 63) / 64, (H + 63) / 64, n), dim3(256), 0, s, masks, tr, H, W, Lp);
  hipLaunchKernelGGL(rle_count_kernel, dim3(chunks, n), dim3(256), 0, s, tr, L, Lp, chunks, chunk_count);
  hipLaunchKernelGGL(rle_scan_mask_kernel, dim3(n), dim3(256), 0, s, chunk_count, chunks, runs);
  hipLaunchKernelGGL(rle_scan_offsets_kernel, dim3(1), dim3(256), 0, s, runs, n, offsets);
  hipLaunchKernelGGL(rle_emit_kernel, dim3(chunks, n), dim3(256), 0, s, tr, L, Lp, chunks, chunk_count, offsets, pos, capacity);
  hipLaunchKernelGGL(rle_diff_kernel, dim3(64, n), dim3(256), 0, s, pos, offsets, (uint32_t)L, counts, capacity);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
