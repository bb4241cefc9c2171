// Per-row ascending sorts of int32 keys staged row by row (start[r] .. start[r + 1]): rows of at most 64 keys by one wave
// (rank by readlane, ds_permute into place), longer rows by one workgroup drawn from a ticket (LDS bitonic network up to
// CC_LDS keys, in place in memory beyond).  Used by the COO -> CSR build (coo_csr.hip) and by the transposed entry lists
// of the deterministic pooling backward (pool_bwd.hip).  `static`: every translation unit that includes this gets its own
// kernels.
#pragma once
#include "common.h"

#define CC_WAVE_MAX 64
#define CC_LDS 12288            /* keys of a long row sorted in LDS (48 KiB) */

// ascending in-place sort of a[0..n) by one workgroup: the all-ascending bitonic network (partners beyond n act as +inf)
template <typename P>
__device__ __forceinline__ void cc_wg_sort(P a, int n) {
  int pow2 = 1;
  while (pow2 < n) pow2 <<= 1;
  for (int k = 2; k <= pow2; k <<= 1) {
    const int hk = k >> 1;
    for (int i = threadIdx.x; i < (pow2 >> 1); i += blockDim.x) {
      const int blk = i / hk, o = i - blk * hk;
      const int lo = blk * k + o, hi = blk * k + k - 1 - o;
      if (hi < n) { const int32_t x = a[lo], y = a[hi]; if (y < x) { a[lo] = y; a[hi] = x; } }
    }
    __syncthreads();
    for (int j = hk >> 1; j >= 1; j >>= 1) {
      for (int i = threadIdx.x; i < (pow2 >> 1); i += blockDim.x) {
        const int blk = i / j, o = i - blk * j;
        const int lo = blk * 2 * j + o, hi = lo + j;
        if (hi < n) { const int32_t x = a[lo], y = a[hi]; if (y < x) { a[lo] = y; a[hi] = x; } }
      }
      __syncthreads();
    }
  }
}

// Rows of at most 64 entries: one wave each — lane l ranks its column among the row's (ties by lane: duplicates stay
// adjacent), pushes it to lane rank; the sorted row goes back in place, ucount[r] = number of distinct columns.
// Longer rows are appended to long_list.
static __global__ __launch_bounds__(OCN_BLOCK) void cc_sort_short_kernel(const i64* __restrict__ start, i64 n_rows, int32_t* __restrict__ stage,
                                                                  int32_t* __restrict__ ucount, int32_t* __restrict__ long_list,
                                                                  int32_t* __restrict__ n_long) {
  const int lane = threadIdx.x & 63;
  for (i64 r = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6); r < n_rows; r += (i64)gridDim.x * OCN_WPB) {
    const i64 b = start[r];
    const int n = (int)(start[r + 1] - b);
    if (n > CC_WAVE_MAX) {
      if (lane == 0) long_list[atomicAdd(n_long, 1)] = (int32_t)r;
      continue;
    }
    if (n == 0) { if (ucount && lane == 0) ucount[r] = 0; continue; }
    const int32_t c = lane < n ? stage[b + lane] : 0x7fffffff;
    int rank = 0;
    for (int m = 0; m < n; ++m) {
      const int32_t o = __builtin_amdgcn_readlane(c, m);
      rank += (o < c) | ((o == c) & (m < lane));
    }
    const int32_t s = __builtin_amdgcn_ds_permute((lane < n ? rank : lane) << 2, c);      // lane q: q-th smallest
    if (lane < n) stage[b + lane] = s;
    if (ucount) {
      const int32_t prev = __shfl_up(s, 1, OCN_WAVE);
      const bool first = lane < n && (lane == 0 || s != prev);
      const int u = __popcll(__ballot(first));
      if (lane == 0) ucount[r] = u;
    }
  }
}

static __global__ __launch_bounds__(OCN_BLOCK) void cc_sort_long_kernel(const i64* __restrict__ start, int32_t* __restrict__ stage,
                                                                 int32_t* __restrict__ ucount, const int32_t* __restrict__ long_list,
                                                                 const int32_t* __restrict__ n_long, int32_t* __restrict__ ticket) {
  __shared__ int32_t s_key[CC_LDS];
  __shared__ int s_item;
  __shared__ i64 s_scan[2 * OCN_WPB];
  const int total = n_long[0];
  for (;;) {
    if (threadIdx.x == 0) s_item = atomicAdd(ticket, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= total) break;                            // every wave reaches this: the grid drains
    const i64 r = long_list[item];
    const i64 b = start[r];
    const int n = (int)(start[r + 1] - b);
    int32_t* keys = stage + b;
    i64 uniq = 0;
    if (n <= CC_LDS) {
      for (int q = threadIdx.x; q < n; q += OCN_BLOCK) s_key[q] = keys[q];
      __syncthreads();
      cc_wg_sort(&s_key[0], n);
      i64 mine = 0;
      for (int q = threadIdx.x; q < n; q += OCN_BLOCK) {
        const int32_t v = s_key[q];
        keys[q] = v;
        mine += (q == 0 || s_key[q - 1] != v);
      }
      block_excl_scan(mine, s_scan, &uniq);
    } else {
      cc_wg_sort(keys, n);                               // __syncthreads between the steps orders the global accesses of one workgroup
      i64 mine = 0;
      for (int q = threadIdx.x; q < n; q += OCN_BLOCK) mine += (q == 0 || keys[q - 1] != keys[q]);
      block_excl_scan(mine, s_scan, &uniq);
    }
    if (ucount && threadIdx.x == 0) ucount[r] = (int32_t)uniq;
    __syncthreads();
  }
}
