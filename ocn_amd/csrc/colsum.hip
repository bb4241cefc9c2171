// Order-exact column sums for cn5 / cn6 when the predictor's `innerprod` buffer is non-zero (every trained
// checkpoint).  The reference sums the orthogonalised values v = cn2 - nip * ncn1 of a column entry by entry in
// fp32 with index_add_ over the coalesced COO pattern (model.py:2405-2406; cn6 :2912-2913), i.e. in ascending
// batch-row order; where colsum(cn2) ~ nip the result depends on that order.  Here the union pattern is
// transposed into per-column entry lists — count (the histogram's n_union field, or an atomic count when a
// second flag array joins) -> scan -> fill -> sort by flag position (positions ascend with the batch row) — and
// one lane chain adds each column's values in that order, every product / difference rounded separately.
// Only columns whose values are not integers need any of that: a column with fewer than two cn1 entries has t = nip *
// inv1 = 0 (model.py:2263-2266), all its v are the cn2 values themselves (1.0, or the walk count), and their fp32 sum is
// exact in any order below 2^24 — it is written from the histogram's counts, and its entries are neither counted,
// filled, sorted nor chained (cn5; cn6's second stage has non-integer values in every column of the union).
#include "common.h"

#define CS_WAVE_MAX 64          /* columns with at most this many entries: one wave each */
#ifndef OCN_X_CS_LDS
#define OCN_X_CS_LDS 4096       /* longer columns up to this many entries are sorted in LDS, beyond in place in memory */
#endif
#define CS_CHUNK 2048           /* values staged in LDS per accumulation round of a long column */
#define CS_BINS 256             /* long columns: one bucket pass by position range, then every thread sorts one small bucket */
#define CS_BIN_MAX 48           /* ... unless a bucket is longer than this (clustered positions): then the bitonic network */

// ---------------------------------------------------------------------------------------------
// entry lists
// ---------------------------------------------------------------------------------------------
// does the order of a column's entries matter?  (t != 0: non-integer values; or an integer sum that leaves fp32's exact range)
__device__ __forceinline__ bool cs_ordered(u64 pk, u64 walks, float nip, bool valued) {
  const int n1 = hf_n1(pk);
  const float t = __fmul_rn(nip, n1 >= 2 ? 1.0f / (float)n1 : 0.0f);
  return t != 0.0f || (valued ? walks : (u64)hf_n2(pk)) >= (1ull << 24);
}

// One thread per column: which columns need an ordered sum.  FROM_HIST: counts[c] = the histogram's n_union for an ordered
// column, 0 otherwise (else: counts come from the counting pass).  An unordered, touched column gets its exact closed
// form right here; an ordered column without a local entry (an edge shard) passes the earlier shards' partial sum
// through.  ALL: every column of the union is ordered (cn6).  (A compacted list of the ordered columns for the short
// kernel was tried: its appends — one atomic per wave on one counter — cost 40 us and saved nothing, the short kernel's
// time is the ordered columns' own load chains.)
template <bool FROM_HIST, bool ALL>
__global__ __launch_bounds__(OCN_BLOCK) void colsum_classify_kernel(const u64* __restrict__ hist, i64 N, const float* __restrict__ innerprod,
                                                                    const int32_t* __restrict__ scalars, int valued,
                                                                    int32_t* __restrict__ counts, float* __restrict__ s2,
                                                                    const float* __restrict__ s2_init) {
  const float nip = cn5_nip(scalars[0], innerprod[0]);
  for (i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (i64)gridDim.x * blockDim.x) {
    const u64 pk = hist[2 * c], walks = hist[2 * c + 1];
    const bool ordered = ALL || cs_ordered(pk, walks, nip, valued);
    int n;
    if (FROM_HIST) { n = ordered ? hf_nu(pk) : 0; counts[c] = n; }
    else n = counts[c];
    if (n == 0) {
      if (!ALL && pk != 0 && !ordered) s2[c] = valued ? (float)walks : (float)hf_n2(pk);
      else if (s2_init) s2[c] = s2_init[c];
    }
  }
}

__global__ __launch_bounds__(OCN_BLOCK) void colsum_zero_kernel(int32_t* __restrict__ p, i64 n) {
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (i64)gridDim.x * blockDim.x) p[q] = 0;
}

// One wave per batch row.  FILL = false: counts[k] += 1 per union entry; FILL = true: the entry's flag position
// goes to the next free slot of its column (cursor zero on entry).
// (FILL: a column without a list — col_off[k] == col_off[k + 1] — is an unordered one; counting: `hist` != NULL selects
// the ordered columns by the same predicate as colsum_nu_kernel, NULL counts every column)
template <bool FILL>
__global__ __launch_bounds__(OCN_BLOCK) void colsum_entries_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA, const i64* __restrict__ src, i64 B,
    const i64* __restrict__ off, const uint8_t* __restrict__ flagsA, const uint8_t* __restrict__ flagsB, i64 cap,
    const i64* __restrict__ col_off, int32_t* __restrict__ cursor, uint32_t* __restrict__ entries,
    const u64* __restrict__ hist, const float* __restrict__ innerprod, const int32_t* __restrict__ scalars, int valued) {
  const int lane = threadIdx.x & 63;
  const float nip = hist ? cn5_nip(scalars[0], innerprod[0]) : 0.0f;
  for (i64 e = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6); e < B; e += (i64)gridDim.x * OCN_WPB) {
    const i64 i = src[e];
    const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0, base = off[e];
    if (base + da > cap) continue;                       // row beyond the flag capacity: nothing was written for it
    for (i64 p = lane; p < da; p += OCN_WAVE) {
      unsigned f = flagsA[base + p];
      if (flagsB) f |= (unsigned)(flagsB[base + p] & OCN_F_CN1) << 2;
      if (!f) continue;
      const int32_t k = colA[a0 + p];
      if (FILL) {
        const i64 c0 = col_off[k];
        if (col_off[k + 1] == c0) continue;
        entries[c0 + atomicAdd(cursor + k, 1)] = (uint32_t)(base + p);
      } else {
        if (hist && !cs_ordered(hist[2 * (i64)k], hist[2 * (i64)k + 1], nip, valued)) continue;
        atomicAdd(cursor + k, 1);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// values and chains
// ---------------------------------------------------------------------------------------------
struct ColCtx {
  const uint8_t* flagsA;
  const uint8_t* flagsB;
  const int32_t* wc;
  float nip;
};

// v2 of the entry at flag position p in a column with t = nip * inv1 (model.py:2380-2384); in2 = the entry
// belongs to the stage-1 union (cn1 or cn2), a3 = its cn3 value (cn6)
__device__ __forceinline__ float entry_v2(const ColCtx& cx, uint32_t p, float t, bool& in2, float& a3, float& tt) {
  const unsigned fa = cx.flagsA[p];
  const float c = (fa & OCN_F_CN2) ? (cx.wc ? (float)cx.wc[p] : 1.0f) : 0.0f;
  tt = (fa & OCN_F_CN1) ? t : 0.0f;
  in2 = fa != 0;
  a3 = (cx.flagsB && (cx.flagsB[p] & OCN_F_CN1)) ? 1.0f : 0.0f;
  return __fsub_rn(c, tt);
}

// acc + v[0] + v[1] + ... + v[cnt-1] in that order, v[r] held by lane r (wave-uniform result)
__device__ __forceinline__ float chain_add(float acc, float v, int cnt) {
  const int vi = __builtin_bit_cast(int, v);
  for (int r = 0; r < cnt; ++r) acc = __fadd_rn(acc, __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, r)));
  return acc;
}

__device__ __forceinline__ float col_t(const u64* hist, i64 c, float nip) {
  const int n1 = hf_n1(hist[2 * c]);
  const float inv1 = n1 >= 2 ? 1.0f / (float)n1 : 0.0f;               // model.py:2263-2266
  return __fmul_rn(nip, inv1);
}

// Columns with <= 64 entries: one wave each (rank sort in registers, ds_permute into rank order, lane chain);
// longer columns are appended to long_list for colsum_long_kernel.
__global__ __launch_bounds__(OCN_BLOCK) void colsum_short_kernel(
    const u64* __restrict__ hist, i64 N, const i64* __restrict__ col_off, const uint32_t* __restrict__ entries,
    ColCtx cx, const float* __restrict__ innerprod, const int32_t* __restrict__ scalars,
    float* __restrict__ s2, float* __restrict__ s3, int32_t* __restrict__ long_list, int32_t* __restrict__ n_long,
    const float* __restrict__ s2_init) {
  const int lane = threadIdx.x & 63;
  cx.nip = cn5_nip(scalars[0], innerprod[0]);
  for (i64 c = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6); c < N; c += (i64)gridDim.x * OCN_WPB) {
    const i64 b = col_off[c];
    const int n = (int)(col_off[c + 1] - b);
    if (n == 0) continue;                                // an unordered or untouched column: colsum_classify_kernel's
    if (n > CS_WAVE_MAX) {
      if (lane == 0) long_list[atomicAdd(n_long, 1)] = (int32_t)c;
      continue;
    }
    const uint32_t p = lane < n ? entries[b + lane] : 0xffffffffu;
    int rank = 0;
    for (int m = 0; m < n; ++m) rank += (uint32_t)__builtin_amdgcn_readlane((int)p, m) < p;
    const float t = col_t(hist, c, cx.nip);
    bool in2 = false;
    float a3 = 0.f, tt = 0.f, v2 = 0.f;
    if (lane < n) v2 = entry_v2(cx, p, t, in2, a3, tt);
    // lane l pushes its values to lane rank_l: afterwards lane r holds the r-th entry in ascending position order
    const int dst = (lane < n ? rank : lane) << 2;
    const float v2s = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dst, __builtin_bit_cast(int, in2 ? v2 : 0.0f)));
    float S2 = chain_add(s2_init ? s2_init[c] : 0.0f, v2s, n);   // entries outside the stage-1 union add +0.0: no effect
    s2[c] = S2;
    if (s3) {
      if (S2 == 0.0f) S2 = 1.0f;                         // model.py:2409 / :2705
      const float inv2 = 1.0f / S2;
      // v3 = cn3 - nip * ncn1 - nip * ncn2' (:2895-2899), ncn2' = v2 / S2 on the stage-1 union, 0 elsewhere
      const float a2n = in2 ? __fmul_rn(v2, inv2) : 0.0f;
      const float v3 = __fsub_rn(__fsub_rn(a3, tt), __fmul_rn(cx.nip, a2n));
      const float v3s = __builtin_bit_cast(float, __builtin_amdgcn_ds_permute(dst, __builtin_bit_cast(int, lane < n ? v3 : 0.0f)));
      s3[c] = chain_add(0.0f, v3s, n);
    }
  }
}

// Columns with 65 .. CS_MED_MAX entries: ONE WAVE each, four columns per workgroup, no workgroup barrier.  At the collab shape a
// trained model's 65 536-edge batch has ~3 000 such columns and none longer (the longest: ~400 entries); a whole workgroup per column
// (colsum_long_kernel: ticket, bucket sort with five barriers, staged values, wave 0's chain while three waves wait) spent ~20 us of
// latencies on each.  Here the wave ranks its keys against all the column's keys read back from LDS (positions are distinct: rank = number of smaller
// keys), scatters them into rank order, stages the values in that order and runs the same lane chain.
// Longer columns go on to long2_list for colsum_long_kernel.
#define CS_MED_MAX 512
#define CS_MED_KPL (CS_MED_MAX / OCN_WAVE)      /* keys per lane */
template <int NR>
__device__ __forceinline__ void med_rank(int (&rank)[CS_MED_KPL], const uint32_t (&key)[CS_MED_KPL], const uint32_t* keys, int n) {
  for (int m0 = 0; m0 < n; m0 += 8) {
    uint32_t km[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) km[u] = keys[m0 + u];                     // (past n: 0xffffffff, smaller than no key)
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < NR; ++j) rank[j] += km[u] < key[j];
  }
}

__global__ __launch_bounds__(OCN_BLOCK) void colsum_medium_kernel(
    const u64* __restrict__ hist, const i64* __restrict__ col_off, const uint32_t* __restrict__ entries,
    ColCtx cx, const float* __restrict__ innerprod, const int32_t* __restrict__ scalars,
    float* __restrict__ s2, float* __restrict__ s3, const int32_t* __restrict__ long_list, const int32_t* __restrict__ n_long,
    int32_t* __restrict__ long2_list, int32_t* __restrict__ n_long2, const float* __restrict__ s2_init) {
  __shared__ uint32_t s_key[OCN_WPB][CS_MED_MAX];      // the column's entry positions as loaded (padded with 0xffffffff) ...
  __shared__ uint32_t s_srt[OCN_WPB][CS_MED_MAX];      // ... and in rank order
  __shared__ float s_val[OCN_WPB][CS_MED_MAX];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  cx.nip = cn5_nip(scalars[0], innerprod[0]);
  const int total = n_long[0];
  for (int item = blockIdx.x * OCN_WPB + w; item < total; item += gridDim.x * OCN_WPB) {
    const i64 c = long_list[item];
    const i64 b = col_off[c];
    const int n = (int)(col_off[c + 1] - b);
    if (n > CS_MED_MAX) {
      if (lane == 0) long2_list[atomicAdd(n_long2, 1)] = (int32_t)c;
      continue;
    }
    uint32_t key[CS_MED_KPL];
#pragma unroll
    for (int j = 0; j < CS_MED_KPL; ++j) {
      const int q = lane + OCN_WAVE * j;
      key[j] = q < n ? entries[b + q] : 0xffffffffu;
      s_key[w][q] = key[j];                              // (the whole array: the ranking loop below reads it in blocks of eight)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // rank = the number of smaller keys (positions are distinct).  Key m is a broadcast LDS read; eight of them are requested before
    // the first is compared — one read per step put the LDS latency on every step (66 us for the ~3 000 columns of the collab batch;
    // a bitonic network on the wave's array: 84 us, broadcasting out of registers with v_readlane: 96 us).
    int rank[CS_MED_KPL];
#pragma unroll
    for (int j = 0; j < CS_MED_KPL; ++j) rank[j] = 0;
    // (only the registers that hold keys take part: a column of 100 entries fills two of the eight)
    if (n <= 2 * OCN_WAVE) med_rank<2>(rank, key, &s_key[w][0], n);
    else if (n <= 4 * OCN_WAVE) med_rank<4>(rank, key, &s_key[w][0], n);
    else med_rank<CS_MED_KPL>(rank, key, &s_key[w][0], n);
#pragma unroll
    for (int j = 0; j < CS_MED_KPL; ++j)
      if (lane + OCN_WAVE * j < n) s_srt[w][rank[j]] = key[j];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float t = col_t(hist, c, cx.nip);
    // the entries' flag bytes (and walk counts) in rank order: ALL requested at once — one byte per entry from all over the flag
    // buffer, up to eight trips to memory if each waited for the one before
    unsigned fa[CS_MED_KPL], fb[CS_MED_KPL];
    float cw[CS_MED_KPL];
#pragma unroll
    for (int j = 0; j < CS_MED_KPL; ++j) {
      const int q = lane + OCN_WAVE * j;
      const uint32_t p = q < n ? s_srt[w][q] : 0u;
      fa[j] = q < n ? cx.flagsA[p] : 0u;
      fb[j] = (q < n && cx.flagsB) ? cx.flagsB[p] : 0u;
      cw[j] = (q < n && cx.wc) ? (float)cx.wc[p] : 1.0f;
    }
    float inv2 = 0.0f;
    for (int pass = 0; pass < (s3 ? 2 : 1); ++pass) {
#pragma unroll
      for (int j = 0; j < CS_MED_KPL; ++j) {
        const int q = lane + OCN_WAVE * j;
        if (q < n) {                                     // entry_v2's arithmetic on the bytes fetched above
          const float cc = (fa[j] & OCN_F_CN2) ? cw[j] : 0.0f;
          const float tt = (fa[j] & OCN_F_CN1) ? t : 0.0f;
          const bool in2 = fa[j] != 0;
          const float a3 = (fb[j] & OCN_F_CN1) ? 1.0f : 0.0f;
          const float v2 = __fsub_rn(cc, tt);
          float v = in2 ? v2 : 0.0f;
          if (pass) v = __fsub_rn(__fsub_rn(a3, tt), __fmul_rn(cx.nip, in2 ? __fmul_rn(v2, inv2) : 0.0f));
          s_val[w][q] = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      float acc = (pass == 0 && s2_init) ? s2_init[c] : 0.0f;
      for (int r0 = 0; r0 < n; r0 += OCN_WAVE) {
        const float v = r0 + lane < n ? s_val[w][r0 + lane] : 0.0f;
        acc = chain_add(acc, v, n - r0 < OCN_WAVE ? n - r0 : OCN_WAVE);
      }
      if (pass == 0) {
        if (lane == 0) s2[c] = acc;
        inv2 = 1.0f / (acc == 0.0f ? 1.0f : acc);
      } else if (lane == 0) {
        s3[c] = acc;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
}

// In-place sort of a[0..n) (LDS or global), any n: the bitonic network in its all-ascending form (first step of
// a merge mirrors, the rest are strides; every compare-exchange leaves the minimum at the lower index), so
// partners beyond n behave as +inf and are simply skipped.
template <typename P>
__device__ __forceinline__ void wg_sort(P a, int n) {
  int pow2 = 1;
  while (pow2 < n) pow2 <<= 1;
  for (int k = 2; k <= pow2; k <<= 1) {
    const int hk = k >> 1;
    for (int i = threadIdx.x; i < (pow2 >> 1); i += blockDim.x) {
      const int blk = i / hk, o = i - blk * hk;
      const int lo = blk * k + o, hi = blk * k + k - 1 - o;
      if (hi < n) { const uint32_t x = a[lo], y = a[hi]; if (y < x) { a[lo] = y; a[hi] = x; } }
    }
    __syncthreads();
    for (int j = hk >> 1; j >= 1; j >>= 1) {
      for (int i = threadIdx.x; i < (pow2 >> 1); i += blockDim.x) {
        const int blk = i / j, o = i - blk * j;
        const int lo = blk * 2 * j + o, hi = lo + j;
        if (hi < n) { const uint32_t x = a[lo], y = a[hi]; if (y < x) { a[lo] = y; a[hi] = x; } }
      }
      __syncthreads();
    }
  }
}

// One workgroup per long column (ticket-drawn from long_list): sort its entry positions, then wave 0 adds the
// values in that order, CS_CHUNK at a time through LDS.
__global__ __launch_bounds__(OCN_BLOCK) void colsum_long_kernel(
    const u64* __restrict__ hist, const i64* __restrict__ col_off, uint32_t* __restrict__ entries,
    ColCtx cx, const float* __restrict__ innerprod, const int32_t* __restrict__ scalars,
    float* __restrict__ s2, float* __restrict__ s3, const int32_t* __restrict__ long_list,
    const int32_t* __restrict__ n_long, int32_t* __restrict__ ticket, const float* __restrict__ s2_init, i64 cap) {
  __shared__ uint32_t s_key[OCN_X_CS_LDS];
  __shared__ uint32_t s_out[OCN_X_CS_LDS];
  __shared__ int s_cnt[CS_BINS], s_start[CS_BINS];
  __shared__ i64 s_scan[2 * OCN_WPB];
  __shared__ int s_big;
  __shared__ float s_val[CS_CHUNK];
  __shared__ int s_item;
  __shared__ float s_inv2;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  cx.nip = cn5_nip(scalars[0], innerprod[0]);
  const int total = n_long[0];
  for (;;) {                                             // (items dealt round robin instead of ticket-drawn: 231 against 207 µs, round 4)
    if (threadIdx.x == 0) s_item = atomicAdd(ticket, 1);
    __syncthreads();
    const int item = s_item;
    if (item >= total) break;                            // every wave reaches this: the grid drains
    const i64 c = long_list[item];
    const i64 b = col_off[c];
    const int n = (int)(col_off[c + 1] - b);
    uint32_t* keys = entries + b;
    const bool in_lds = n <= OCN_X_CS_LDS;
    const uint32_t* sorted = keys;                         // where the sorted positions end up
    if (in_lds) {
      // A column's entries come from all over the batch, so their flag positions spread over [0, cap): CS_BINS position
      // ranges hold a handful of entries each.  Count, scan, scatter, then thread t insertion-sorts bucket t — four
      // barriers where the bitonic network of this size needs fifty.
      for (int q = threadIdx.x; q < n; q += OCN_BLOCK) s_key[q] = keys[q];
      if (threadIdx.x < CS_BINS) s_cnt[threadIdx.x] = 0;
      if (threadIdx.x == 0) s_big = 0;
      __syncthreads();
      for (int q = threadIdx.x; q < n; q += OCN_BLOCK) atomicAdd(&s_cnt[(int)(((u64)s_key[q] * CS_BINS) / (u64)cap)], 1);
      __syncthreads();
      const int mine = threadIdx.x < CS_BINS ? s_cnt[threadIdx.x] : 0;
      if (mine > CS_BIN_MAX) s_big = 1;
      i64 tot;
      const i64 start = block_excl_scan((i64)mine, s_scan, &tot);
      if (threadIdx.x < CS_BINS) { s_start[threadIdx.x] = (int)start; s_cnt[threadIdx.x] = (int)start; }
      __syncthreads();
      if (!s_big) {
        for (int q = threadIdx.x; q < n; q += OCN_BLOCK) {
          const uint32_t k = s_key[q];
          s_out[atomicAdd(&s_cnt[(int)(((u64)k * CS_BINS) / (u64)cap)], 1)] = k;
        }
        __syncthreads();
        if (threadIdx.x < CS_BINS) {
          const int lo = s_start[threadIdx.x], hi = lo + mine;
          for (int i = lo + 1; i < hi; ++i) {
            const uint32_t k = s_out[i];
            int j = i - 1;
            while (j >= lo && s_out[j] > k) { s_out[j + 1] = s_out[j]; --j; }
            s_out[j + 1] = k;
          }
        }
        __syncthreads();
        sorted = &s_out[0];
      } else {
        wg_sort(&s_key[0], n);
        sorted = &s_key[0];
      }
    } else {
      wg_sort(keys, n);
    }
    const float t = col_t(hist, c, cx.nip);
    for (int pass = 0; pass < (s3 ? 2 : 1); ++pass) {
      float acc = (pass == 0 && s2_init) ? s2_init[c] : 0.0f;
      const float inv2 = pass ? s_inv2 : 0.0f;
      for (int q0 = 0; q0 < n; q0 += CS_CHUNK) {
        const int m = n - q0 < CS_CHUNK ? n - q0 : CS_CHUNK;
        for (int q = threadIdx.x; q < m; q += OCN_BLOCK) {
          const uint32_t p = sorted[q0 + q];
          bool in2;
          float a3, tt;
          const float v2 = entry_v2(cx, p, t, in2, a3, tt);
          float v = in2 ? v2 : 0.0f;
          if (pass) v = __fsub_rn(__fsub_rn(a3, tt), __fmul_rn(cx.nip, in2 ? __fmul_rn(v2, inv2) : 0.0f));
          s_val[q] = v;
        }
        __syncthreads();
        if (w == 0) {
          for (int r0 = 0; r0 < m; r0 += OCN_WAVE) {
            const float v = r0 + lane < m ? s_val[r0 + lane] : 0.0f;
            acc = chain_add(acc, v, m - r0 < OCN_WAVE ? m - r0 : OCN_WAVE);
          }
        }
        __syncthreads();
      }
      if (threadIdx.x == 0) {
        if (pass == 0) {
          s2[c] = acc;
          s_inv2 = 1.0f / (acc == 0.0f ? 1.0f : acc);
        } else {
          s3[c] = acc;
        }
      }
      __syncthreads();
    }
  }
}

extern "C" {

int64_t ocn_cn_colsum_workspace_bytes(int64_t N, int64_t flags_cap) {
  // col_off int64[N+1] | counts/cursor int32[N] | long_list int32[N] | tickets int32[4] | scan state | entries uint32[cap] | long2_list int32[N]
  const int64_t a = ((N + 1) * 8 + 15) / 16 * 16, b = (N * 4 + 15) / 16 * 16;
  return a + 2 * b + 16 + (ocn_scan_workspace_bytes(N) + 15) / 16 * 16 + (flags_cap * 4 + 15) / 16 * 16 + b + 64;
}

int ocn_cn_colsum_exact(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, int64_t B,
                        const int64_t* off, const uint8_t* flagsA, const uint8_t* flagsB, const int32_t* wc,
                        int64_t flags_cap, const uint64_t* hist, int64_t N, const float* innerprod, int32_t* scalars,
                        const float* s2_init, float* s2, float* s3, void* workspace, void* stream) {
  if (B < 0 || N < 0 || flags_cap < 0 || flags_cap > 0xfffffffell) return OCN_EINVAL;
  if (N == 0) return 0;
  if (!hist || !innerprod || !scalars || !s2 || !workspace) return OCN_EINVAL;
  if (B > 0 && (!rowptrA || !src || !off || !flagsA)) return OCN_EINVAL;
  if ((flagsB != nullptr) != (s3 != nullptr) || (flagsB && (wc || s2_init))) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int64_t a = ((N + 1) * 8 + 15) / 16 * 16, b = (N * 4 + 15) / 16 * 16;
  i64* col_off = (i64*)ws;
  int32_t* counts = (int32_t*)(ws + a);
  int32_t* long_list = (int32_t*)(ws + a + b);
  int32_t* tickets = (int32_t*)(ws + a + 2 * b);                    // [0] columns longer than a wave's 64, [1] work ticket of the long kernel, [2] columns longer than CS_MED_MAX
  void* scan_ws = (void*)(ws + a + 2 * b + 16);
  const int64_t sw = (ocn_scan_workspace_bytes(N) + 15) / 16 * 16;
  uint32_t* entries = (uint32_t*)(ws + a + 2 * b + 16 + sw);
  int32_t* long2_list = (int32_t*)(ws + a + 2 * b + 16 + sw + (flags_cap * 4 + 15) / 16 * 16);
  const int gridN = grid_for((N + OCN_BLOCK - 1) / OCN_BLOCK, 2048);
  const int gridB = grid_for((B + OCN_WPB - 1) / OCN_WPB, 1 << 16);
  int rc = ocn_cn5_column_stats(hist, N, scalars, stream);           // nip needs the batch's scale (idempotent)
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_zero_kernel, dim3(1), dim3(OCN_BLOCK), 0, st, tickets, (i64)(4 + sw / 4));   // + the scan state
  const int valued = wc != nullptr;
#define CLASSIFY(FROM_HIST, ALL)                                                                                          \
  hipLaunchKernelGGL((colsum_classify_kernel<FROM_HIST, ALL>), dim3(gridN), dim3(OCN_BLOCK), 0, st, (const u64*)hist, (i64)N, \
                     innerprod, (const int32_t*)scalars, valued, counts, s2, s2_init)
  if (flagsB || s2_init) {        // cn6: the union is wider than histA's n_union; a shard: hist holds the GLOBAL counts
    hipLaunchKernelGGL(colsum_zero_kernel, dim3(gridN), dim3(OCN_BLOCK), 0, st, counts, (i64)N);
    if (B > 0)
      hipLaunchKernelGGL((colsum_entries_kernel<false>), dim3(gridB), dim3(OCN_BLOCK), 0, st, (const i64*)rowptrA, colA,
                         (const i64*)src, (i64)B, (const i64*)off, flagsA, flagsB, (i64)flags_cap,
                         (const i64*)nullptr, counts, (uint32_t*)nullptr, flagsB ? (const u64*)nullptr : (const u64*)hist,
                         innerprod, (const int32_t*)scalars, valued);
    if (flagsB) CLASSIFY(false, true); else CLASSIFY(false, false);
  } else {
    CLASSIFY(true, false);
  }
#undef CLASSIFY
  rc = ocn_scan_i32(counts, N, (int64_t*)col_off, scan_ws, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(colsum_zero_kernel, dim3(gridN), dim3(OCN_BLOCK), 0, st, counts, (i64)N);
  if (B > 0)
    hipLaunchKernelGGL((colsum_entries_kernel<true>), dim3(gridB), dim3(OCN_BLOCK), 0, st, (const i64*)rowptrA, colA,
                       (const i64*)src, (i64)B, (const i64*)off, flagsA, flagsB, (i64)flags_cap, (const i64*)col_off,
                       counts, entries, (const u64*)nullptr, innerprod, (const int32_t*)scalars, valued);
  ColCtx cx{flagsA, flagsB, wc, 0.0f};
  hipLaunchKernelGGL(colsum_short_kernel, dim3(grid_for((N + OCN_WPB - 1) / OCN_WPB, 1 << 15)), dim3(OCN_BLOCK), 0, st,
                     (const u64*)hist, (i64)N, (const i64*)col_off, (const uint32_t*)entries, cx, innerprod,
                     (const int32_t*)scalars, s2, s3, long_list, tickets, s2_init);
  hipLaunchKernelGGL(colsum_medium_kernel, dim3(256 * 4), dim3(OCN_BLOCK), 0, st, (const u64*)hist, (const i64*)col_off,
                     (const uint32_t*)entries, cx, innerprod, (const int32_t*)scalars, s2, s3, (const int32_t*)long_list,
                     (const int32_t*)tickets, long2_list, tickets + 2, s2_init);
  hipLaunchKernelGGL(colsum_long_kernel, dim3(256 * 3), dim3(OCN_BLOCK), 0, st, (const u64*)hist, (const i64*)col_off,
                     entries, cx, innerprod, (const int32_t*)scalars, s2, s3, (const int32_t*)long2_list,
                     (const int32_t*)(tickets + 2), tickets + 1, s2_init, (i64)(flags_cap > 0 ? flags_cap : 1));
  return launch_status();
}

}  // extern "C"
