// The common-neighbour stage: intersection (K1, pattern and walk-count forms), per-column weights
// (K2) and embedding pooling (K3).  See include/ocn_hip.h for the reference call sites.
#include "common.h"


// ---------------------------------------------------------------------------------------------
// sorted-list membership
// ---------------------------------------------------------------------------------------------
// Branch-uniform binary search over a[0..n) (global or LDS): every lane of a wave searches the same
// row, so the trip count is wave-uniform and the top levels of the tree are shared cache lines.
template <typename P>
__device__ __forceinline__ bool sorted_has(P a, i64 n, int32_t key) {
  i64 lo = 0, hi = n;
  bool found = false;
  while (lo < hi) {
    const i64 mid = (lo + hi) >> 1;
    const int32_t v = a[mid];
    found |= (v == key);
    if (v < key) lo = mid + 1; else hi = mid;
  }
  return found;
}

// Two-level search of a long row: `samp` (LDS) holds the row's elements at positions (s*n)>>6,
// s = 0..63; six LDS probes pick the segment, the remaining log2(n/64) probes go to memory.
__device__ __forceinline__ bool sampled_has(const int32_t* samp, const int32_t* __restrict__ row, i64 n,
                                            int32_t key) {
  int lo = 0, hi = OCN_WAVE;                 // upper bound: lo = number of samples <= key
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (samp[mid] <= key) lo = mid + 1; else hi = mid;
  }
  if (lo == 0) return false;
  const int s = lo - 1;
  const i64 p0 = ((i64)s * n) >> 6, p1 = ((i64)(s + 1) * n) >> 6;
  return sorted_has(row + p0, p1 - p0, key);
}

// ---------------------------------------------------------------------------------------------
// K1 (pattern): flags for N(src) against the rows of dst in T1 (and T2)
// ---------------------------------------------------------------------------------------------
#define T1_CAP 1024
#define REC_LEN_SHIFT 40                     /* slot record word 2: row start (nnz < 2^40) | row length << 40 */
#define REC_FULL2_BIT 61     /* record word 3: every position of the source row is a cn2 entry (cnt2 == row length) */
#ifndef OCN_X_G
#define OCN_X_G 64     /* lanes per candidate edge; tools/kbench.py overrides it for timing experiments */
#endif
// a group's cost: the largest entry count among its four candidates (a wave walks one candidate), in buckets of 32 —
// measured at the collab shape (tools/_ab notes in DESIGN.md): sum / max and 4 .. 256-entry buckets all land within
// 0.184 - 0.200 ms against 0.206 unscheduled; coarse buckets keep more of the source order's L2 locality
#define SCHED_COST(t, c) ((t) > (c) ? (t) : (c))
#ifndef OCN_X_SCHED_SHIFT
#define OCN_X_SCHED_SHIFT 5
#endif
#ifndef OCN_X_POOL_LPE
#define OCN_X_POOL_LPE 64    /* lanes per candidate of the H = 256 pooling (64: one candidate per wave) */
#endif
#define POOL_FOLD (OCN_WAVE / OCN_X_POOL_LPE)      /* 4-slot cost groups of the intersection pass per pooling workgroup */
#define SCHED_GROUP (OCN_BLOCK / OCN_X_G)    /* slots per scheduling group: a workgroup of the intersection pass == one of the H = 256 pooling */

// G lanes cooperate on one candidate edge (64/G edges per wave).  Measured on the collab-shaped
// batch (tools/kbench.py): G = 64 / 32 / 16 / 8 -> 208 / 242 / 327 / 494 us.  The kernel is bound by
// the number of distinct cache lines its scattered probes touch per wave instruction, not by the
// latency of the chains: lanes that search the SAME rows share the top-of-tree lines, lanes of
// different edges do not, so one edge per wave wins although most source rows are < 64 long.
// LH: the column histogram of this workgroup is kept in LDS (n_cols <= LH_MAX_COLS, i.e. Cora /
// Citeseer / ddi-sized graphs, where tens of millions of CN entries would otherwise hammer a few
// thousand global addresses) and flushed once at the end of the workgroup's grid-stride loop.
#define LH_MAX_COLS 8192

template <int G, bool HAS_T2, bool LH>
__global__ __launch_bounds__(OCN_BLOCK) void cn_flags_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ rowptrT1, const int32_t* __restrict__ colT1,
    const i64* __restrict__ rowptrT2, const int32_t* __restrict__ colT2,
    const unsigned* __restrict__ bmT1, i64 bm1_stride, const unsigned* __restrict__ bmT2, i64 bm_stride,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    i64 n_cols, const i64* __restrict__ off, uint8_t* __restrict__ flags, i64 cap,
    u64* __restrict__ hist, int32_t* __restrict__ cnt1, int32_t* __restrict__ cnt2,
    int32_t* __restrict__ status, u64* __restrict__ rec, int32_t* __restrict__ gcost) {
  constexpr int GPB = OCN_BLOCK / G;
  __shared__ int s_cost[GPB];
  __shared__ int32_t s_t1[GPB][T1_CAP];
  __shared__ int32_t s_t2[GPB][OCN_WAVE];
  extern __shared__ __attribute__((aligned(16))) u64 s_hist[];      // LH only: n_cols words
  const int gl = threadIdx.x % G, g = threadIdx.x / G;
  if (blockIdx.x == 0 && threadIdx.x == 0) status_raise(status, off[B], cap);
  const bool void_batch = off[B] < 0;        // the offsets come from a scan that gave up (ocn_hip.h: OCN_SCAN_POISON): the batch is treated as
                                             // empty — zero counts, empty slot records — so that nothing behind this pass indexes with them
  if (LH) {
    for (i64 c = threadIdx.x; c < n_cols; c += OCN_BLOCK) s_hist[c] = 0ull;
    __syncthreads();
  }
  for (i64 e0 = (i64)blockIdx.x * GPB; e0 < B; e0 += (i64)gridDim.x * GPB) {
    const i64 slot = e0 + g;                  // processing slot; `order` maps it to a batch row
    const bool act = slot < B;
    const i64 e = act ? (order ? order[slot] : slot) : 0;
    i64 a0 = 0, da = 0, b0 = 0, db = 0, c0 = 0, dc = 0, base = 0;
    const unsigned* bm_row = nullptr;         // bit row of dst in T2, when T2 comes with a dense bitmap
    const unsigned* bm1_row = nullptr;        // ... and in T1 (small dense graphs: a membership test is one probe)
    if (act) {
      const i64 i = src[e], j = dst[e];
      a0 = rowptrA[i]; da = rowptrA[i + 1] - a0;
      if (bmT1) bm1_row = bmT1 + j * bm1_stride;
      else { b0 = rowptrT1[j]; db = rowptrT1[j + 1] - b0; }
      if (HAS_T2) {
        if (bmT2) bm_row = bmT2 + j * bm_stride;
        if (rowptrT2 && (LH || !bmT2)) { c0 = rowptrT2[j]; dc = rowptrT2[j + 1] - c0; }   // (small graphs with bit rows too: a FULL row — a dense A², ogbl-ddi —
                                                                                  // needs no probe at all; on large graphs the two loads would only lengthen the chain)
      }
      base = off[e];
      if (void_batch) { da = 0; db = 0; dc = 0; base = 0; }
    }
    // stage the short target row, and a 64-point sample of the long one, in this group's LDS slice
    const bool t1_lds = db <= T1_CAP;
    if (t1_lds)
      for (i64 q = gl; q < db; q += G) s_t1[g][q] = colT1[b0 + q];
    const bool t2_full = HAS_T2 && rowptrT2 && dc == n_cols;   // a full row (dense A², e.g. ddi) contains every column
    if (HAS_T2 && !t2_full && !bmT2) {
      if (dc > OCN_WAVE) {
#pragma unroll
        for (int q = gl; q < OCN_WAVE; q += G) s_t2[g][q] = colT2[c0 + (((i64)q * dc) >> 6)];
      } else {
        for (int q = gl; q < dc; q += G) s_t2[g][q] = colT2[c0 + q];
      }
    }
    __syncthreads();
    const bool fits = base + da <= cap;
    int c1 = 0, c2 = 0;
    for (i64 p = gl; p < da; p += G) {
      const int32_t k = colA[a0 + p];
#ifdef OCN_X_NOT1   /* OCN_X_*: timing experiments of tools/kbench.py, never defined in the product build */
      const bool f1 = (k & 7) == 0;
#else
      const bool f1 = bmT1 ? (bool)((bm1_row[k >> 5] >> (k & 31)) & 1u)
                           : (t1_lds ? sorted_has(&s_t1[g][0], db, k) : sorted_has(colT1 + b0, db, k));
#endif
      bool f2 = false;
#ifdef OCN_X_NOT2
      f2 = (k & 1) == 0;
#else
      if (HAS_T2) {
        if (t2_full) f2 = true;
        else if (bmT2) f2 = (bm_row[k >> 5] >> (k & 31)) & 1u;    // one probe
        else f2 = dc > OCN_WAVE ? sampled_has(&s_t2[g][0], colT2 + c0, dc, k) : sorted_has(&s_t2[g][0], dc, k);
      }
#endif
#ifndef OCN_X_NOFLAGS
      if (fits) flags[base + p] = (uint8_t)((f1 ? OCN_F_CN1 : 0u) | (f2 ? OCN_F_CN2 : 0u));
#endif
#ifndef OCN_X_NOATOMIC
      if (f1 | f2) {
        const u64 inc = (u64)f1 | ((u64)f2 << HF_BITS) | (1ull << (2 * HF_BITS));
#ifdef OCN_X_ATOMIC_SPREAD   /* timing experiment: the same number of atomics on uniformly spread addresses (no hot column) */
        if (LH) atomicAdd(s_hist + k, inc); else atomicAdd(hist + 2 * (i64)((((u64)k * 2654435761ull) ^ ((u64)(base + p) * 40503ull)) % (u64)n_cols), inc);
#elif defined(OCN_X_ATOMIC_WG)  /* timing experiment: the atomics resolved in the issuing XCD's L2 (results wrong across XCDs) */
        if (LH) atomicAdd(s_hist + k, inc); else __hip_atomic_fetch_add(hist + 2 * (i64)k, inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
        if (LH) atomicAdd(s_hist + k, inc); else atomicAdd(hist + 2 * (i64)k, inc);
#endif
      }
#endif
      c1 += f1;
      c2 += f2;
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) {
      c1 += __shfl_xor(c1, o, OCN_WAVE);
      c2 += __shfl_xor(c2, o, OCN_WAVE);
    }
    if (act && gl == 0) {
      cnt1[e] = c1;
      if (cnt2) cnt2[e] = c2;
      if (rec) {                              // what the pooling needs to know about this slot, in ONE 32-byte record
        const i64 i = src[e], j = dst[e];
        u64* r = rec + 4 * slot;
        r[0] = (u64)e;
        r[1] = (u64)i | ((u64)j << 32);
        r[2] = (u64)a0 | ((u64)da << REC_LEN_SHIFT);
        r[3] = (u64)base | ((u64)(da > 0 && (i64)c2 == da) << REC_FULL2_BIT) | ((u64)(c1 > 0) << 62) | ((u64)(c2 > 0) << 63);
      }
    }
    if (gcost) {
      // what this group of GPB consecutive slots will cost the pooling (entries to gather): the pooling visits its
      // groups longest first (ocn_gather_schedule), so that no straggler ends its kernel
      if (gl == 0) s_cost[g] = act ? c1 + c2 : 0;
      __syncthreads();
      if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int q = 0; q < GPB; ++q) t = SCHED_COST(t, s_cost[q]);
        gcost[e0 / GPB] = t;
      }
    }
    __syncthreads();
  }
  if (LH) {
    for (i64 c = threadIdx.x; c < n_cols; c += OCN_BLOCK) {
      const u64 v = s_hist[c];
      if (v) atomicAdd(hist + 2 * c, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K1 (walk counts): the pygho route of NeighborOverlap_large_ppa.py:147-173 without A².
// cn1 = N(i) ∩ N(j); cn2[e,k] = |N(k) ∩ N(j)| for k in N(i) (number of 2-walks j -> k), kept if > 0.
// ---------------------------------------------------------------------------------------------
// The probed neighbour set is a 256 Kbit Bloom-style bitmap in LDS (one multiplicative hash): building
// it is one atomicOr per member, probing one LDS read, and it does not care how long the row is (an
// 8 300-neighbour hub gives 3 % false positives, a typical row 0.1 %).  Elements that pass it — the
// true hits (~1 % of the swept elements) plus the false positives — are queued in LDS and resolved in
// bulk, one per thread, by binary search in the sorted CSR row; done inline that search would run with
// a handful of live lanes on nearly every wave iteration.
#ifndef OCN_X_WALK_BM_BITS
#define OCN_X_WALK_BM_BITS 17
#endif
#define WALK_BM_BITS OCN_X_WALK_BM_BITS
#define WALK_BM_WORDS (1 << (WALK_BM_BITS - 5))
#ifndef OCN_X_WALK_Q
#define OCN_X_WALK_Q 1024
#endif
#define WALK_Q OCN_X_WALK_Q     /* queue entries; flushed when half full, overflow resolves in place */
// The probed row itself is kept in LDS beside its bitmap (it passes through the workgroup's hands anyway when the bitmap
// is built): resolving a queued element, and the cn1 test of the finalise step, are then binary searches in LDS —
// ~10 dependent LDS reads instead of ~10 dependent trips to L2, which were two thirds of an item's chain of dependent
// loads.  Rows longer than WALK_SET (hubs) keep the search in memory.  The room comes from halving the bitmap
// (128 Kbit: twice the false positives, each now one cheap LDS search).
#ifndef OCN_X_WALK_SET
#define OCN_X_WALK_SET 4096
#endif
#define WALK_SET OCN_X_WALK_SET

__device__ __forceinline__ unsigned walk_bit(int32_t v) { return ((unsigned)v * 2654435761u) >> (32 - WALK_BM_BITS); }
__device__ __forceinline__ void walk_bm_add(unsigned* bm, int32_t v) {
  const unsigned b = walk_bit(v);
  atomicOr(&bm[b >> 5], 1u << (b & 31));
}
__device__ __forceinline__ bool walk_bm_maybe(const unsigned* bm, int32_t v) {
  const unsigned b = walk_bit(v);
  return (bm[b >> 5] >> (b & 31)) & 1u;
}

// position of key in the sorted row a[0..n), or -1
__device__ __forceinline__ i64 sorted_find(const int32_t* __restrict__ a, i64 n, int32_t key) {
  i64 lo = 0, hi = n;
  while (lo < hi) {
    const i64 mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return (lo < n && a[lo] == key) ? lo : -1;
}

// position of key in the probed row (LDS copy when it fits, else the CSR row in memory), or -1
__device__ __forceinline__ i64 walk_set_find(const int32_t* s_set, const int32_t* __restrict__ set_g, i64 ds, int32_t key) {
#if WALK_SET > 0
  if (ds <= WALK_SET) {
    int lo = 0, hi = (int)ds;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (s_set[mid] < key) lo = mid + 1; else hi = mid;
    }
    return (lo < (int)ds && s_set[lo] == key) ? lo : -1;
  }
#endif
  return sorted_find(set_g, ds, key);
}

// batch slot of a work item: last slot with item_off[slot] <= item (item_off[0] = 0 <= item <
// item_off[B]).  Called by one whole wave: 64 probes per round instead of a dependent chain of
// log2(B) single loads (2 rounds for B = 2048, 3 for 65 536).
__device__ __forceinline__ i64 walk_item_slot(const i64* __restrict__ item_off, i64 B, i64 item, int lane) {
  i64 lo = 0, hi = B;                        // item_off[lo] <= item < item_off[hi]
  while (hi - lo > 1) {
    const i64 step = (hi - lo + OCN_WAVE - 1) / OCN_WAVE;
    const i64 idx = lo + (i64)(lane + 1) * step;
    const bool le = idx < hi && item_off[idx] <= item;
    const int c = __popcll(__ballot(le));
    lo += (i64)c * step;
    hi = lo + step < hi ? lo + step : hi;
  }
  return lo;
}

#ifndef OCN_X_WALK_THREADS
#define OCN_X_WALK_THREADS 512
#endif
#define WALK_THREADS OCN_X_WALK_THREADS   /* threads per walk work item */
#define WALK_WAVES (WALK_THREADS / OCN_WAVE)
#define WALK_ROWS (WALK_WAVES * WALK_CHUNK)   /* rows a forward item can take: one 64-row chunk per wave */

// The rows of one item, flattened.  Wave w loads the ids / starts / lengths of rows [64w, 64w+64) of
// the item (`first` = index of the item's first row id in colA) and scans the lengths; after the
// barrier the chunk totals are folded in, so that element x of the concatenation belongs to the last
// row t with s_pre[t] <= x (s_pre[n_rows] = INT_MAX).  Returns the number of elements.
__device__ __forceinline__ int walk_item_rows(const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
                                              i64 first, int n_rows, int* s_pre, i64* s_r0, int32_t* s_r,
                                              int* s_ctot) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int t = threadIdx.x;                 // WALK_THREADS == WALK_ROWS: one row per thread
  int32_t r = 0;
  i64 r0 = 0, dr = 0;
  if (t < n_rows) { r = colA[first + t]; r0 = rowptrA[r]; dr = rowptrA[r + 1] - r0; }
  const i64 incl = wave_incl_scan(dr, lane);                 // an item's elements stay far below 2^31
  if (lane == OCN_WAVE - 1) s_ctot[w] = (int)incl;
  s_r[t] = r; s_r0[t] = r0;
  __syncthreads();
  int before = 0, total = 0;
#pragma unroll
  for (int q = 0; q < WALK_WAVES; ++q) {
    const int c = s_ctot[q];
    if (q < w) before += c;
    total += c;
  }
  s_pre[t] = t < n_rows ? before + (int)(incl - dr) : 0x7fffffff;
  if (t == 0) s_pre[WALK_ROWS] = 0x7fffffff;
  __syncthreads();
  return total;
}

// One work item of either direction: `rows` = the neighbours of the sweeping endpoint prepared by
// walk_item_rows, `set_g[0..ds)` = the sorted neighbour row of the other endpoint whose bitmap is in
// s_bm.  Calls hit(key, row_in_item, position_in_set) for every swept element that is a member of
// the set.
template <typename Hit>
__device__ __forceinline__ void walk_sweep(const int32_t* __restrict__ colA, const unsigned* s_bm, const int* s_pre,
                                           const i64* s_r0, int total, const int32_t* s_set, const int32_t* __restrict__ set_g, i64 ds,
                                           int32_t* s_qk, uint16_t* s_qr, int* s_nq, Hit hit) {
#ifndef OCN_X_WALK_WU
#define OCN_X_WALK_WU 8
#endif
  constexpr int WU = OCN_X_WALK_WU;          // independent element loads in flight per thread
  int lo = 0;                                // a thread's elements come in increasing x: the row pointer only moves forward
  for (int x0 = 0; x0 < total; x0 += WU * WALK_THREADS) {              // workgroup-uniform trip count
    int row[WU];
    int32_t m[WU];
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      const int x = x0 + u * WALK_THREADS + threadIdx.x;
      m[u] = -1;
      if (x < total) {
        while (s_pre[lo + 1] <= x) ++lo;     // <= WALK_ROWS advances over the whole item
        m[u] = colA[s_r0[lo] + (x - s_pre[lo])];
      }
      row[u] = lo;
    }
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      if (m[u] >= 0 && walk_bm_maybe(s_bm, m[u])) {
        const int q = atomicAdd(s_nq, 1);
        if (q < WALK_Q) { s_qk[q] = m[u]; s_qr[q] = (uint16_t)row[u]; }
        else {                                                      // queue full (dense overlap): resolve in place
          const i64 pos = walk_set_find(s_set, set_g, ds, m[u]);
          if (pos >= 0) hit(m[u], row[u], pos);
        }
      }
    }
    // The flush decision must be workgroup-uniform (the branch holds barriers): every thread reads
    // the queue fill inside the barrier itself, before any wave can append for the next round.
    const int flush = __syncthreads_or(*s_nq > WALK_Q / 2 || x0 + WU * WALK_THREADS >= total);
    if (flush) {
      const int nq = *s_nq < WALK_Q ? *s_nq : WALK_Q;
      for (int q = threadIdx.x; q < nq; q += WALK_THREADS) {
        const i64 pos = walk_set_find(s_set, set_g, ds, s_qk[q]);
        if (pos >= 0) hit(s_qk[q], (int)s_qr[q], pos);
      }
      __syncthreads();
      if (threadIdx.x == 0) *s_nq = 0;
      __syncthreads();
    }
  }
}

#define WALK_SHARED                                     \
  __shared__ unsigned s_bm[WALK_BM_WORDS];              \
  __shared__ int32_t s_set[WALK_SET > 0 ? WALK_SET : 1]; \
  __shared__ int s_pre[WALK_ROWS + 1];                  \
  __shared__ i64 s_r0[WALK_ROWS];                       \
  __shared__ int32_t s_r[WALK_ROWS];                    \
  __shared__ int s_ctot[WALK_WAVES];                    \
  __shared__ int32_t s_qk[WALK_Q];                      \
  __shared__ uint16_t s_qr[WALK_Q];                     \
  __shared__ int s_nq;                                  \
  __shared__ i64 s_slot, s_item

// draw the next work item (ticket counter) and find its batch slot, while the other waves clear the bitmap
#define WALK_NEXT_ITEM(TICKET, ITEM_OFF) WALK_NEXT_ITEM_X(TICKET, ITEM_OFF, (void)0)
#define WALK_NEXT_ITEM_X(TICKET, ITEM_OFF, ONEXIT)                                                   \
  if (w == 0) {                                                                                      \
    i64 t = 0;                                                                                       \
    if (lane == 0) t = atomicAdd((TICKET), 1);                                                       \
    t = __shfl(t, 0, OCN_WAVE);                                                                      \
    const i64 sl = t < n_items ? walk_item_slot((ITEM_OFF), B, t, lane) : 0;                         \
    if (lane == 0) { s_item = t; s_slot = sl; s_nq = 0; }                                            \
  } else {                                                                                           \
    for (int q = threadIdx.x - OCN_WAVE; q < WALK_BM_WORDS; q += WALK_THREADS - OCN_WAVE) s_bm[q] = 0u; \
  }                                                                                                  \
  __syncthreads();                                                                                   \
  const i64 item = s_item;                                                                           \
  if (item >= n_items) { ONEXIT; break; }                                                            \
  const i64 slot = s_slot;                                                                           \
  const i64 e = order ? order[slot] : slot;                                                          \
  const i64 i = src[e], j = dst[e];                                                                  \
  const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0;                                               \
  const i64 b0 = rowptrA[j], db = rowptrA[j + 1] - b0;                                               \
  const i64 base = off[e]

// Reverse sweep (runs first, only for the batch rows walk_reverse() selects): work item = (batch row,
// chunk of WALK_REV_CHUNK neighbours m of j).  The members k' of the rows N(m) are probed against
// N(i); a hit adds one walk to wc[off[e] + position of k' in N(i)] (wc is zero on entry).  The forward
// kernel's items of that batch row then only finalise it.  Items differ 100x in cost: workgroups draw
// them from a ticket counter.
__global__ __launch_bounds__(WALK_THREADS) void cn_walk_rev_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    const i64* __restrict__ rev_off, const i64* __restrict__ off, int32_t* __restrict__ wc, i64 cap,
    int32_t* __restrict__ ticket) {
  WALK_SHARED;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const i64 n_items = rev_off[B];
  for (;;) {
    WALK_NEXT_ITEM(ticket, rev_off);
    const i64 p_lo = (item - rev_off[slot]) * WALK_REV_CHUNK;
    const int nm = (int)(((p_lo + WALK_REV_CHUNK) < db ? (p_lo + WALK_REV_CHUNK) : db) - p_lo);
    const int32_t* ni_g = colA + a0;
    for (i64 q = threadIdx.x; q < da; q += WALK_THREADS) {
      const int32_t v = ni_g[q];
      walk_bm_add(s_bm, v);
      if (q < WALK_SET) s_set[q] = v;
    }
    int total = walk_item_rows(rowptrA, colA, b0 + p_lo, nm, s_pre, s_r0, s_r, s_ctot);
#ifdef OCN_X_WALK_NOSWEEP   /* timing experiment: per-item overhead only */
    total = 0;
#endif
    if (base + da > cap) total = 0;
    int32_t* wrow = wc + base;
    walk_sweep(colA, s_bm, s_pre, s_r0, total, s_set, ni_g, da, s_qk, s_qr, &s_nq,
               [wrow](int32_t, int, i64 pos) { atomicAdd(wrow + pos, 1); });
    __syncthreads();
  }
}

#ifdef OCN_X_WALK_STAMPS
__device__ unsigned long long g_walk_stamps[256 * 8];
extern "C" int ocn_debug_walk_stamps(unsigned long long* out, int reset) {
  if (reset) { static unsigned long long z[256 * 8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_walk_stamps), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_walk_stamps), 256 * 8 * sizeof(unsigned long long));
}
#endif
// Forward sweep.  Work item = (batch row, group of <= WALK_WAVES consecutive 64-row chunks of N(i));
// items are enumerated through the exclusive scan chunk_off[] so that a hub source node is spread
// over many workgroups instead of serialising one, while a light row is a single item (one round of
// dependent loads for all its rows).  The members of the item's rows N(k) are FLATTENED: the threads
// sweep the concatenation of the rows, so short rows do not idle lanes, no load waits on a per-row
// pointer chase, and a hub k costs what its length costs; each is probed against N(j), and a hit
// bumps the row's counter with an LDS atomic.  For a batch row the reverse sweep has already counted
// (walk_reverse()), the item only reads its counts back from wc and finalises flags, histogram and
// per-edge counts.
__global__ __launch_bounds__(WALK_THREADS) void cn_walk_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA, const i64* __restrict__ nds,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    const i64* __restrict__ chunk_off, const i64* __restrict__ off, uint8_t* __restrict__ flags,
    int32_t* __restrict__ wc, i64 cap, u64* __restrict__ hist, int32_t* __restrict__ cnt1,
    int32_t* __restrict__ cnt2, int32_t* __restrict__ status) {   // status[0] flags, [1] / [2] item tickets (zero on entry)
  WALK_SHARED;
  __shared__ int s_walks[WALK_ROWS];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    status_raise(status, off[B], cap);
    if (chunk_off[B] < 0) status_raise(status, chunk_off[B], cap);
  }
  const i64 n_items = chunk_off[B];
#ifdef OCN_X_WALK_STAMPS   /* diagnostic build: where an item's time goes (tools/walkstamps.py) */
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define WSTAMP(k) do { __builtin_amdgcn_s_waitcnt(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); tph[k] += tn - tprev; tprev = tn; } while (0)
#define WSTAMP_OUT() do { if (threadIdx.x == 0 && blockIdx.x < 256) for (int q = 0; q < 6; ++q) g_walk_stamps[blockIdx.x * 8 + q] += tph[q]; } while (0)
#else
#define WSTAMP(k) do {} while (0)
#define WSTAMP_OUT() do {} while (0)
#endif
  for (;;) {
    if (w == 0) { WSTAMP(5); }
    WALK_NEXT_ITEM_X(status + 1, chunk_off, WSTAMP_OUT());
    WSTAMP(0);
    const bool rev = walk_reverse(nds, i, j, da, db);      // workgroup-uniform
    const int32_t* nj_g = colA + b0;
#ifndef OCN_X_WALK_NOBM
    for (i64 q = threadIdx.x; q < db; q += WALK_THREADS) {
      const int32_t v = nj_g[q];
      walk_bm_add(s_bm, v);
      if (q < WALK_SET) s_set[q] = v;
    }
#endif
    const i64 n_chunks = (da + WALK_CHUNK - 1) / WALK_CHUNK;
    const i64 cg = walk_group(nds, i, da);
    const i64 p_lo = (item - chunk_off[slot]) * cg * WALK_CHUNK;
    const i64 p_hi = p_lo + cg * WALK_CHUNK < da ? p_lo + cg * WALK_CHUNK : da;
    (void)n_chunks;
    const int nk = (int)(p_hi - p_lo);
    const bool in_cap = base + da <= cap;
    s_walks[threadIdx.x] = 0;
#ifdef OCN_X_WALK_NOROWS
    int total = 0; __syncthreads();
#else
    int total = walk_item_rows(rowptrA, colA, a0 + p_lo, nk, s_pre, s_r0, s_r, s_ctot);
#endif
#ifdef OCN_X_WALK_NOSWEEP
    total = 0;
#endif
    if (rev) total = 0;
    WSTAMP(1);
    int* walks_of = s_walks;
    walk_sweep(colA, s_bm, s_pre, s_r0, total, s_set, nj_g, db, s_qk, s_qr, &s_nq,
               [walks_of](int32_t, int row, i64) { atomicAdd(walks_of + row, 1); });
    __syncthreads();
    WSTAMP(2);
#ifndef OCN_X_WALK_NOFIN
    {                                          // finalise: one row per thread
      const int t = threadIdx.x;
      bool f1 = false, f2 = false;
      if (t < nk) {
        const int32_t k = s_r[t];
        const int walks = rev ? (in_cap ? wc[base + p_lo + t] : 0) : s_walks[t];
        f1 = walk_bm_maybe(s_bm, k) && walk_set_find(s_set, nj_g, db, k) >= 0;
        f2 = walks > 0;
        if (in_cap) {
          flags[base + p_lo + t] = (uint8_t)((f1 ? OCN_F_CN1 : 0u) | (f2 ? OCN_F_CN2 : 0u));
          if (!rev) wc[base + p_lo + t] = walks;
        }
        if (f1 | f2) {
          atomicAdd(hist + 2 * (i64)k, (u64)f1 | ((u64)f2 << HF_BITS) | (1ull << (2 * HF_BITS)));
          if (f2) atomicAdd(hist + 2 * (i64)k + 1, (u64)walks);
        }
      }
      const int c1 = __popcll(__ballot(f1)), c2 = __popcll(__ballot(f2));
      if (lane == 0) {                         // cnt1 / cnt2 are zero on entry; a row spans several waves and items
        if (c1) atomicAdd(cnt1 + e, c1);
        if (c2) atomicAdd(cnt2 + e, c2);
      }
    }
#endif
    WSTAMP(3);
    __syncthreads();
    WSTAMP(4);
    if (threadIdx.x == 0) {
#ifdef OCN_X_WALK_STAMPS
      if (blockIdx.x < 256) g_walk_stamps[blockIdx.x * 8 + 6] += 1;
#endif
    }
  }
}

// wc[0 .. min(off[B], cap)) = 0: the reverse sweep accumulates into it
__global__ __launch_bounds__(OCN_BLOCK) void walk_zero_kernel(const i64* __restrict__ off, i64 B, i64 cap,
                                                              int32_t* __restrict__ wc) {
  i64 n = off[B];
  if (n > cap) n = cap;
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (i64)gridDim.x * blockDim.x) wc[q] = 0;
}

// nds[v] = Σ_{u ∈ N(v)} deg(u): the number of elements a sweep of v's neighbour rows touches
__global__ __launch_bounds__(OCN_BLOCK) void neighbor_degree_sum_kernel(const i64* __restrict__ rowptr,
                                                                        const int32_t* __restrict__ col, i64 n,
                                                                        i64* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  for (i64 v = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; v < n; v += ((i64)gridDim.x * blockDim.x) >> 6) {
    i64 s = 0;
    for (i64 p = rowptr[v] + lane; p < rowptr[v + 1]; p += OCN_WAVE) {
      const int32_t u = col[p];
      s += rowptr[u + 1] - rowptr[u];
    }
    s = wave_sum(s);
    if (lane == 0) out[v] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// K2: per-column weights {w1, t, inv2, 0}, in place over the histogram
//   pooled xcn1 uses w1;  a union entry with cn2 value c (1, or the walk count) contributes
//   (c·[in cn2] − t·[in cn1]) · inv2 to xcn2.
// ---------------------------------------------------------------------------------------------
// scalars[0] (zero on entry) ends as: 0 = no union entry at all; -1 = union entries but no column
// with n1 >= 2; otherwise min{n1 : n1 >= 2} - INT_MAX - 1 (<= -2).  One atomicMin per workgroup,
// skipped when the word already holds something at least as small.
__global__ __launch_bounds__(OCN_BLOCK) void cn5_column_stats(const u64* __restrict__ hist, i64 N,
                                                              int32_t* __restrict__ scalars) {
  __shared__ int sh[OCN_WPB];
  int v = 0;
  for (i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (i64)gridDim.x * blockDim.x) {
    const u64 pk = hist[2 * c];
    int t = pk ? -1 : 0;
    const int n1 = hf_n1(pk);
    if (n1 >= 2) t = n1 - 0x7fffffff - 1;
    v = t < v ? t : v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int t = __shfl_xor(v, o, OCN_WAVE);
    v = t < v ? t : v;
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < OCN_WPB; ++i) v = sh[i] < v ? sh[i] : v;
    if (v < 0 && v < __hip_atomic_load(scalars, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMin(scalars, v);
  }
}

__global__ __launch_bounds__(OCN_BLOCK) void cn5_column_weights(u64* __restrict__ hist, i64 N,
                                                                const float* __restrict__ innerprod,
                                                                const int32_t* __restrict__ scalars,
                                                                int valued, const float* __restrict__ s2_exact) {
  // model.py:2370-2376: scale = max |ncn1| over the union-aligned vector (1.0 if it is empty)
  const float nip = cn5_nip(scalars[0], innerprod[0]);
  float4* wout = reinterpret_cast<float4*>(hist);
  for (i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (i64)gridDim.x * blockDim.x) {
    const u64 pk = hist[2 * c];
    if (pk == 0) continue;                               // untouched column: never read by the gather
    const u64 walks = hist[2 * c + 1];
    const int n1 = hf_n1(pk), n2 = hf_n2(pk), nb = n1 + n2 - hf_nu(pk);
    const float inv1 = n1 >= 2 ? 1.0f / (float)n1 : 0.0f;              // :2263-2266 (Q2)
    const float t = __fmul_rn(nip, inv1);                              // nip * ncn1 value
    // :2405-2406 column sum of v = cn2 − nip·ncn1 over the union pattern.  The reference adds the entries one by
    // one in fp32, in ascending batch-row order (index_add_ over the coalesced COO): s2_exact holds exactly that
    // sum (ocn_cn_colsum_exact) whenever nip != 0.  For nip == 0 every v is an integer (1.0, or the walk count),
    // the sequential fp32 sum is the integer count itself as long as it stays below 2^24, and the closed form
    // below is that same number.
    float S2;
    if (s2_exact) {
      S2 = s2_exact[c];
    } else {
      double s2d;
      if (!valued) {
        const float v_both = __fsub_rn(1.0f, t);                         // :2380-2384
        const float v_only2 = __fsub_rn(1.0f, __fmul_rn(nip, 0.0f));
        const float v_only1 = __fsub_rn(0.0f, t);
        s2d = (double)(n2 - nb) * (double)v_only2 + (double)nb * (double)v_both +
              (double)(n1 - nb) * (double)v_only1;
      } else {
        s2d = (double)walks - (double)n1 * (double)t;
      }
      S2 = (float)s2d;
    }
    if (S2 == 0.0f) S2 = 1.0f;                                         // :2409
    wout[c] = make_float4(inv1, t, 1.0f / S2, 0.0f);                   // :2410-2413
  }
}

// cn6 (model.py:2535-2951), pattern route: stage 1 is cn5's (histA = {n1, n2, n_union} of cn1 / cn2,
// rewritten in place as {inv1, t, inv2, 0}); stage 2 orthogonalises cn3 (histB: its n1 field counts the
// cn3 entries of the column) against both normalised matrices,
//   v3 = [in cn3] - nip*inv1*[in cn1] - nip*ncn2,     S3 = column sum of v3 (0 -> 1),
// and histB is rewritten as {1/S3, 0, 0, 0}.  The column sums are formed from the integer counts in
// fp64 (exact for nip == 0: S2 = n2, S3 = n3), as for cn5.  nip_out[0] receives nip for the gather.
__global__ __launch_bounds__(OCN_BLOCK) void cn6_column_weights(u64* __restrict__ histA, u64* __restrict__ histB,
                                                                i64 N, const float* __restrict__ innerprod,
                                                                const int32_t* __restrict__ scalars,
                                                                float* __restrict__ nip_out,
                                                                const float* __restrict__ s2_exact,
                                                                const float* __restrict__ s3_exact) {
  const float nip = cn5_nip(scalars[0], innerprod[0]);
  if (blockIdx.x == 0 && threadIdx.x == 0) nip_out[0] = nip;
  float4* wa = reinterpret_cast<float4*>(histA);
  float4* wb = reinterpret_cast<float4*>(histB);
  for (i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (i64)gridDim.x * blockDim.x) {
    const u64 pa = histA[2 * c], pb = histB[2 * c];
    if (pa == 0 && pb == 0) continue;
    const int n1 = hf_n1(pa), n2 = hf_n2(pa), nb = n1 + n2 - hf_nu(pa), n3 = hf_n1(pb);
    const float inv1 = n1 >= 2 ? 1.0f / (float)n1 : 0.0f;
    const float t = __fmul_rn(nip, inv1);
    const float v_both = __fsub_rn(1.0f, t), v_only2 = __fsub_rn(1.0f, __fmul_rn(nip, 0.0f)), v_only1 = __fsub_rn(0.0f, t);
    float S2 = s2_exact ? s2_exact[c]
                        : (float)((double)(n2 - nb) * (double)v_only2 + (double)nb * (double)v_both +
                                  (double)(n1 - nb) * (double)v_only1);
    if (S2 == 0.0f) S2 = 1.0f;
    const float inv2 = 1.0f / S2;
    float S3;
    if (s3_exact) {
      S3 = s3_exact[c];
    } else {
      // column sum of the normalised cn2' values (1 up to rounding, or 0)
      const double s2n = (double)(n2 - nb) * (double)__fmul_rn(v_only2, inv2) + (double)nb * (double)__fmul_rn(v_both, inv2) +
                         (double)(n1 - nb) * (double)__fmul_rn(v_only1, inv2);
      S3 = (float)((double)n3 - (double)n1 * (double)t - (double)nip * s2n);
    }
    if (S3 == 0.0f) S3 = 1.0f;
    wa[c] = make_float4(inv1, t, inv2, 0.0f);
    wb[c] = make_float4(1.0f / S3, 0.0f, 0.0f, 0.0f);
  }
}

// d1 / d2 (or NULL = all ones): the diagonals diag(T_k(linspace(-1, 1, N))) the reference multiplies the normalised cn1 and
// the raw cn2 by (evaluate_polynomial, model.py:2995-3019; spspmm with the diagonal at :3141-3165 and :3186-3209 — one fp32
// product per entry).  The drivers hard-wire k = 0 (T0 = 1, the --polyfirst / --polysecond flags are parsed and ignored, Q4).
__global__ __launch_bounds__(OCN_BLOCK) void cn7_column_weights(u64* __restrict__ hist, i64 N,
                                                                float sum_fill, const float* __restrict__ d1,
                                                                const float* __restrict__ d2) {
  float4* wout = reinterpret_cast<float4*>(hist);
  for (i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x; c < N; c += (i64)gridDim.x * blockDim.x) {
    const u64 pk = hist[2 * c];
    if (pk == 0) continue;
    const int n1 = hf_n1(pk);
    const float inv1 = n1 >= 2 ? 1.0f / (float)n1 : sum_fill;          // model.py:3116-3120
    // cn1: ncn1 x T_k1 (model.py:3141-3165); cn2 raw (Q5) x T_k2 (:3186-3209): t = 0, "inv2" = the diagonal's entry, so that
    // an entry's weight (c - 0) * inv2 is the one product c * T_k2 the reference forms
    wout[c] = make_float4(__fmul_rn(inv1, d1 ? d1[c] : 1.0f), 0.0f, d2 ? d2[c] : 1.0f, 0.0f);
  }
}

// ---------------------------------------------------------------------------------------------
// K3: pooling — gather embedding rows over the flagged neighbours
// ---------------------------------------------------------------------------------------------
// (entry_weights: common.h)

// Pool the flagged neighbours at positions [p_begin, p_end) of the source row into acc1 / acc2, in
// ascending position (= column) order.  LPE lanes cooperate; each lane owns NV float4 of the
// H = LPE*NV*4 features; four embedding rows are in flight per group.
#ifndef OCN_X_GATHER_UNR
#define OCN_X_GATHER_UNR 4
#endif
// (eight for the one-wave-per-candidate layout of H >= 256 — one candidate's gathers are all a wave has in flight:
// 0.214 -> 0.206 ms at the collab shape; four where several candidates share a wave)
template <int LPE, int NV, int UNR = (LPE >= 64 ? 2 * OCN_X_GATHER_UNR : OCN_X_GATHER_UNR)>
__device__ __forceinline__ void pool_range(i64 p_begin, i64 p_end, i64 a0, i64 base, int gl, int gbase,
                                           const int32_t* __restrict__ colA, const uint8_t* __restrict__ flags,
                                           const int32_t* __restrict__ wc, const float4* __restrict__ weights,
                                           const float4* __restrict__ h4, i64 rowq, float4 (&acc1)[NV],
                                           float4 (&acc2)[NV], bool full2 = false) {
  // Narrow groups (small H) would otherwise pay one dependent load chain (column id -> column weights)
  // per LPE positions: a lane fetches PT positions per round, so a round always covers 64 of them
  // (hub rows of the ppa shape: 101 -> 40 us).
  constexpr int PT = (OCN_WAVE / LPE) < 8 ? (OCN_WAVE / LPE) : 8;
  f32x2 acc[NV][4];                          // {acc1, acc2} component pairs: one packed multiply + add per pair
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    acc[v][0] = f32x2{acc1[v].x, acc2[v].x}; acc[v][1] = f32x2{acc1[v].y, acc2[v].y};
    acc[v][2] = f32x2{acc1[v].z, acc2[v].z}; acc[v][3] = f32x2{acc1[v].w, acc2[v].w};
  }
  for (i64 p0 = p_begin; p0 < p_end; p0 += LPE * PT) {
    int32_t k[PT];
    unsigned f[PT];
    int32_t cv[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      const i64 p = p0 + t * LPE + gl;
      k[t] = 0; f[t] = 0; cv[t] = 1;
      if (p < p_end) {
        k[t] = colA[a0 + p]; f[t] = flags[base + p];
        if (wc) cv[t] = wc[base + p];
      }
      if (full2) f[t] &= ~OCN_F_CN2;           // the whole row is cn2: its pooled vector is the row sum (rowsum), not summed here
    }
    float wa[PT], wb[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
      wa[t] = wb[t] = 0.f;
      if (f[t]) entry_weights(f[t], weights[k[t]], (float)cv[t], wa[t], wb[t]);
    }
#pragma unroll
    for (int t = 0; t < PT; ++t) {             // ascending position order: tile t, then lane
      const bool need = (wa[t] != 0.f) | (wb[t] != 0.f);
      unsigned long long m = __ballot(need);
      if (LPE < 64) m = (m >> gbase) & ((1ull << (LPE & 63)) - 1ull);
      while (m) {
        int bsel[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          bsel[u] = m ? (__ffsll((long long)m) - 1) : -1;
          m &= m - 1;                          // no-op once m == 0
        }
        int32_t kk[UNR];
        float wwa[UNR], wwb[UNR];
        float4 x[UNR][NV];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          const int sl = gbase + (bsel[u] < 0 ? 0 : bsel[u]);
          kk[u] = __shfl(k[t], sl, OCN_WAVE);
          wwa[u] = __shfl(wa[t], sl, OCN_WAVE);
          wwb[u] = __shfl(wb[t], sl, OCN_WAVE);
#ifdef OCN_X_ROWSKIP   /* timing experiment (results wrong): only every OCN_X_ROWSKIP-th entry fetches its row, the others reuse it — what perfect in-register row sharing would leave of the launch */
          if (bsel[u] >= 0 && (u % OCN_X_ROWSKIP) != 0) {
#pragma unroll
            for (int v = 0; v < NV; ++v) x[u][v] = x[u - (u % OCN_X_ROWSKIP)][v];
          } else
#endif
          if (bsel[u] >= 0) {
#ifdef OCN_X_ROWMASK   /* timing experiment (results wrong): every row fetch folded onto a table of OCN_X_ROWMASK + 1 rows — what the kernel costs when its rows are cache-resident */
            const float4* row = h4 + (i64)(kk[u] & OCN_X_ROWMASK) * rowq + gl;
#else
            const float4* row = h4 + (i64)kk[u] * rowq + gl;
#endif
#pragma unroll
            for (int v = 0; v < NV; ++v) x[u][v] = row[v * LPE];
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (bsel[u] >= 0) {
            const f32x2 w = {wwa[u], wwb[u]};
#pragma unroll
            for (int v = 0; v < NV; ++v) {
              axpy_pair(acc[v][0], w, x[u][v].x);
              axpy_pair(acc[v][1], w, x[u][v].y);
              axpy_pair(acc[v][2], w, x[u][v].z);
              axpy_pair(acc[v][3], w, x[u][v].w);
            }
          }
        }
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    acc1[v] = make_float4(acc[v][0].x, acc[v][1].x, acc[v][2].x, acc[v][3].x);
    acc2[v] = make_float4(acc[v][0].y, acc[v][1].y, acc[v][2].y, acc[v][3].y);
  }
}

template <int LPE, int NV>
__device__ __forceinline__ void pool_store(i64 e, i64 i, i64 j, int gl, const float4* __restrict__ h4, i64 rowq,
                                           const float4 (&acc1)[NV], const float4 (&acc2)[NV],
                                           float* __restrict__ xcn1, float* __restrict__ xcn2,
                                           float* __restrict__ xij, bool st1 = true, bool st2 = true) {
  const float4* hi = h4 + i * rowq + gl;
  const float4* hj = h4 + j * rowq + gl;
  float4* o1 = reinterpret_cast<float4*>(xcn1) + e * rowq + gl;
  float4* o2 = reinterpret_cast<float4*>(xcn2) + e * rowq + gl;
  float4* o3 = reinterpret_cast<float4*>(xij) + e * rowq + gl;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float4 a = hi[v * LPE], b = hj[v * LPE];
    if (st1) o1[v * LPE] = acc1[v];
    if (st2) o2[v * LPE] = acc2[v];
    o3[v * LPE] = make_float4(__fmul_rn(a.x, b.x), __fmul_rn(a.y, b.y), __fmul_rn(a.z, b.z),
                              __fmul_rn(a.w, b.w));
  }
}

// Source rows longer than this are pooled by a whole workgroup (cn_gather_long_kernel): its lane
// groups take contiguous segments of the row and the partial sums are added in segment order.
// Rows up to LONG_ROW keep the strictly sequential ascending-column sum of the reference's spmm.
#define LONG_ROW 1024
#ifndef GATHER_SLICE_MIN_BATCH
#define GATHER_SLICE_MIN_BATCH 16384     /* candidates from which the pooling of H >= 256 runs one feature slice per XCD */
#endif

#ifdef OCN_X_POOL_STAMPS   /* diagnostic build (tools/poolstamps.py): per processing slot {start, end, row length, xcc | hw id} of its pooling wave */
__device__ unsigned long long g_pool_stamps[4 << 17];
extern "C" int ocn_debug_pool_stamps(unsigned long long* out, long long n_slots) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pool_stamps), (size_t)n_slots * 4 * sizeof(unsigned long long));
}
#endif
// LPE lanes cooperate on one edge (64/LPE edges per wave).
// SLICED: the H features are cut into 8 slices of LPE*NV*4 and workgroup b pools slice b % 8 of its candidates.
// Workgroups are dealt round-robin over the 8 XCDs, so XCD x only ever reads feature slice x of the embedding
// table: every row slice has ONE home L2 (a row shared by candidates on different XCDs is no longer fetched up to
// eight times) and that L2 holds 8x as many rows.  Every feature is still summed on its own in ascending column
// order, so the result does not change by a bit.
template <int LPE, int NV, bool SLICED = false>
__global__ __launch_bounds__(OCN_BLOCK) void cn_gather_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    const i64* __restrict__ off, const uint8_t* __restrict__ flags, const int32_t* __restrict__ wc,
    const float4* __restrict__ weights, const float* __restrict__ h, int H,
    float* __restrict__ xcn1, float* __restrict__ xcn2, float* __restrict__ xij,
    const i64* __restrict__ out_row,     // out_row[batch row] = output row (class-major heads), or NULL
    const int32_t* __restrict__ cnt1, const int32_t* __restrict__ cnt2,     // per-row CN counts, or NULL
    const u64* __restrict__ rec,         // slot records of the intersection pass (then order/src/dst/off/cnt are not read), or NULL
    const int32_t* __restrict__ perm,    // ocn_gather_schedule's visiting order of the slot groups (longest first per XCD), or NULL
    const float* __restrict__ rowsum) {  // (A h)[i] rows for candidates whose whole source row is cn2 with weight 1 (ocn_hip.h), or NULL
  constexpr int GPW = OCN_WAVE / LPE;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPE;
  const int gbase = lane - gl;
  // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  The batch rows are
  // visited in source-node order, so give every XCD one contiguous eighth of that order: rows
  // with neighbouring sources then share an L2 instead of being spread over all eight.
  i64 bid = blockIdx.x;
  int slice = 0;
  if (SLICED) {
    slice = (int)(bid & 7);
    bid >>= 3;
  } else {
#ifndef OCN_X_NOXCD
    if ((gridDim.x & 7) == 0) {
      bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
      // ... and inside its eighth an XCD takes the groups in the order of the schedule: the longest jobs first (groups of
      // one source have one cost and stay neighbours: the L2 locality of the source order is kept)
      if (perm) bid = perm[bid];
    }
#endif
  }
  const i64 slot = (bid * OCN_WPB + (threadIdx.x >> 6)) * GPW + lane / LPE;
  if (slot >= B) return;                    // whole group leaves together
#ifdef OCN_X_POOL_STAMPS
  const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
  // One dependent load instead of three (order -> src / dst / off / counts -> rowptr) in front of the first gather:
  // the intersection pass left everything about this slot in a 32-byte record.
  i64 e, i, j, a0, da, base;
  bool has1, has2, full2;
  if (rec) {
    const ulonglong2* rp = reinterpret_cast<const ulonglong2*>(rec + 4 * slot);
    const ulonglong2 ra = rp[0], rb = rp[1];
    e = (i64)ra.x; i = (i64)(ra.y & 0xffffffffull); j = (i64)(ra.y >> 32);
    a0 = (i64)(rb.x & ((1ull << REC_LEN_SHIFT) - 1)); da = (i64)(rb.x >> REC_LEN_SHIFT);
    base = (i64)(rb.y & ((1ull << REC_FULL2_BIT) - 1)); has1 = (rb.y >> 62) & 1ull; has2 = rb.y >> 63;
    full2 = rowsum && ((rb.y >> REC_FULL2_BIT) & 1ull);
  } else {
    e = order ? order[slot] : slot;
    i = src[e]; j = dst[e];
    a0 = rowptrA[i]; da = rowptrA[i + 1] - a0;
    base = off[e];
    // a candidate without any CN entry (half of an evaluation batch) has nothing to pool; with class-major
    // output rows the heads never read its xcn1 / xcn2 rows (nor the xcn1 row of one without cn1 entries)
    has1 = !cnt1 || cnt1[e] > 0; has2 = !cnt2 || cnt2[e] > 0;
    full2 = rowsum && cnt2 && da > 0 && (i64)cnt2[e] == da;
  }
  if (da > LONG_ROW && !full2) return;      // cn_gather_long_kernel's (a hub row whose cn2 is the whole row only has its cn1 entries left)
  const float4* h4 = reinterpret_cast<const float4*>(h) + slice * (LPE * NV);
  const i64 rowq = H >> 2;                  // float4 per row
  if (SLICED) { xcn1 += slice * (LPE * NV * 4); xcn2 += slice * (LPE * NV * 4); xij += slice * (LPE * NV * 4); }
  float4 acc1[NV], acc2[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc1[v] = acc2[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has1 | (has2 & !full2)) pool_range<LPE, NV>(0, da, a0, base, gl, gbase, colA, flags, wc, weights, h4, rowq, acc1, acc2, full2);
  if (full2) {
    const float4* rs = reinterpret_cast<const float4*>(rowsum) + slice * (LPE * NV) + i * rowq + gl;
#pragma unroll
    for (int v = 0; v < NV; ++v) acc2[v] = rs[v * LPE];
  }
  pool_store<LPE, NV>(out_row ? out_row[e] : e, i, j, gl, h4, rowq, acc1, acc2, xcn1, xcn2, xij,
                      !out_row || has1, !out_row || has1 || has2);
#ifdef OCN_X_POOL_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  if (gl == 0 && slot < (1 << 17)) {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_pool_stamps[4 * slot + 0] = t_start;
    g_pool_stamps[4 * slot + 1] = __builtin_amdgcn_s_memtime();
    g_pool_stamps[4 * slot + 2] = (unsigned long long)da;
    g_pool_stamps[4 * slot + 3] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
  }
#endif
}

// The sequential sum of ranks [0, nr) of a compacted round: acc += w[r] * x[r], one multiply and one add per entry and
// accumulator, in rank order.  Eight entries' operands are read from LDS ahead of their adds — a rolled loop would pay
// the LDS latency once per entry (the hub rows of the citation2 shape: 0.45 ms -> see DESIGN.md).
template <int FPL, int HF>
__device__ __forceinline__ void chain_rows(const float2* __restrict__ w, const float* __restrict__ xs, int nr,
                                           float (&acc1)[FPL], float (&acc2)[FPL]) {
  constexpr int BLK = FPL >= 4 ? 2 : 8 / FPL;       // entries per block: 2 * BLK * FPL operand registers per lane
  f32x2 acc[FPL];
#pragma unroll
  for (int q = 0; q < FPL; ++q) acc[q] = f32x2{acc1[q], acc2[q]};
  auto load = [&](int r, float2 (&w8)[BLK], float (&x8)[BLK][FPL]) {
#pragma unroll
    for (int u = 0; u < BLK; ++u) {
      w8[u] = w[r + u];
#pragma unroll
      for (int q = 0; q < FPL; ++q) x8[u][q] = xs[(r + u) * HF + q];
    }
  };
  auto add = [&](const float2 (&w8)[BLK], const float (&x8)[BLK][FPL]) {
#pragma unroll
    for (int u = 0; u < BLK; ++u)
#pragma unroll
      for (int q = 0; q < FPL; ++q) axpy_pair_lds(acc[q], f32x2{w8[u].x, w8[u].y}, x8[u][q]);
  };
  const int nb = nr / BLK;                   // blocks of BLK entries, two register sets: block b + 1 is read while b is added
  if (nb > 0) {
    float2 wa[BLK], wb[BLK];
    float xa[BLK][FPL], xb[BLK][FPL];
    load(0, wa, xa);
    int b = 0;
    for (; b + 2 <= nb; b += 2) {
      load((b + 1) * BLK, wb, xb);
      add(wa, xa);
      if (b + 2 < nb) load((b + 2) * BLK, wa, xa);
      add(wb, xb);
    }
    if (b < nb) add(wa, xa);
  }
  for (int r = nb * BLK; r < nr; ++r) {
    const float2 wr = w[r];
#pragma unroll
    for (int q = 0; q < FPL; ++q) axpy_pair_lds(acc[q], f32x2{wr.x, wr.y}, xs[r * HF + q]);
  }
#pragma unroll
  for (int q = 0; q < FPL; ++q) { acc1[q] = acc[q].x; acc2[q] = acc[q].y; }
}

// Small batches of narrow embeddings (ppa / citation2: B = 2048, H = 32..64) leave the packed kernel
// above with a few hundred waves, each lane group walking its row 4 gathers at a time.  Here ONE WAVE
// takes one batch row: per round of 64 positions the live entries are compacted (rank = position among
// the live ones), the 64/LPE lane groups fetch all their embedding rows at once (up to 64 gathers in
// flight per wave) into the wave's LDS slab, and lane group 0 accumulates them in rank order — the
// same sequential ascending-column fp32 sum as the packed kernel, bit for bit.
// LONG: the same kernel over the batch rows whose source row is LONGER than LONG_ROW only (large batches of narrow
// embeddings on a dense graph — ogbl-ddi: a third of the candidates have such a source; a lane group of the packed
// kernel would walk 2 000 positions four gathers at a time).
#ifdef OCN_X_WAVE_CHECK
__device__ unsigned g_wave_chk_n;
__device__ float g_wave_chk[4096 * 6];
extern "C" int ocn_debug_wave_check(float* out, unsigned* n) {
  const int e = (int)hipMemcpyFromSymbol(n, HIP_SYMBOL(g_wave_chk_n), sizeof(unsigned));
  return e ? e : (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_chk), sizeof(float) * 4096 * 6);
}
#endif
// (waves per workgroup: the slabs of four waves at H = 64 are 70 KiB of static LDS — above the 64 KiB a workgroup gets without an
// opt-in, and measured to go wrong exactly then: with workgroups of ANOTHER kernel on the same CU (a scoring loop's heads beside
// this pooling on a second stream) about one batch in a hundred came back with 64 bytes of one xcn1 row wrong; two waves there)
template <int LPE, int NV, bool LONG>
#ifdef OCN_X_WAVE_WPB   /* diagnostic build only (tools/dbg_two_stream.py) */
constexpr int gather_wave_wpb() { return LONG ? 1 : OCN_X_WAVE_WPB; }
#else
constexpr int gather_wave_wpb() { return LONG ? 1 : ((size_t)OCN_WPB * OCN_WAVE * (LPE * NV * 16 + 24) > 65536 ? 2 : OCN_WPB); }
#endif

template <int LPE, int NV, bool LONG = false, int WPB = gather_wave_wpb<LPE, NV, LONG>()>
__global__ __launch_bounds__(WPB * OCN_WAVE) void cn_gather_wave_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    const i64* __restrict__ off, const uint8_t* __restrict__ flags, const int32_t* __restrict__ wc,
    const float4* __restrict__ weights, const float* __restrict__ h, int H,
    float* __restrict__ xcn1, float* __restrict__ xcn2, float* __restrict__ xij,
    const i64* __restrict__ out_row,     // out_row[batch row] = output row (class-major heads), or NULL
    const int32_t* __restrict__ cnt2, const float* __restrict__ rowsum) {   // see cn_gather_kernel
  // WPB waves per workgroup: ONE for the LONG pass — most batch rows are not its and leave at once, and a wave that
  // has left frees its LDS slab only when it is a workgroup of its own.
  constexpr int G = OCN_WAVE / LPE;
  constexpr int UNR = LPE;                  // G * UNR = 64 rows: a whole round in flight
  constexpr int HF = LPE * NV * 4;          // features: lane l < HF accumulates feature l (all lanes busy at H = 64,
  static_assert(HF <= OCN_WAVE, "");        // a quarter of the VALU work per entry of a float4-per-lane layout)
  __shared__ float4 s_x[WPB][OCN_WAVE][LPE * NV];
  __shared__ int32_t s_k[WPB][2][OCN_WAVE];
  __shared__ float2 s_w[WPB][2][OCN_WAVE];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gl = lane % LPE, g = lane / LPE;
  const i64 slot = (i64)blockIdx.x * WPB + wv;
  if (slot >= B) return;                    // whole wave leaves together (no workgroup barriers below)
  const i64 e = order ? order[slot] : slot;
  const i64 i = src[e], j = dst[e];
  const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0;
  const bool full2 = rowsum && cnt2 && da > 0 && (i64)cnt2[e] == da;
  if (LONG ? (da <= LONG_ROW || full2) : (da > LONG_ROW && !full2)) return;       // the other launch's rows
  const i64 base = off[e];
  const float4* h4 = reinterpret_cast<const float4*>(h);
  const i64 rowq = H >> 2;
  float acc1[1] = {0.f}, acc2[1] = {0.f};
  // Software pipeline over the 64-position rounds: a round needs column id -> column weights -> rows, three dependent
  // trips to memory, then the sequential sum.  The ids of round r+2 and the weights of round r+1 are requested before
  // round r's rows are; round r+1 is compacted and its rows requested (into registers) before round r is summed.
  int32_t k_n = 0, k_nn = 0, cv_n = 1, cv_nn = 1;
  unsigned f_n = 0, f_nn = 0;
  const unsigned fmask = full2 ? ~OCN_F_CN2 : ~0u;          // (full2: the row's cn2 pool is rowsum[i])
  if (lane < da) { k_n = colA[a0 + lane]; f_n = flags[base + lane] & fmask; if (wc) cv_n = wc[base + lane]; }
  if (OCN_WAVE + lane < da) {
    k_nn = colA[a0 + OCN_WAVE + lane]; f_nn = flags[base + OCN_WAVE + lane] & fmask;
    if (wc) cv_nn = wc[base + OCN_WAVE + lane];
  }
  float4 w_n = make_float4(0.f, 0.f, 0.f, 0.f);
  if (f_n) w_n = weights[k_n];
  f32x4 x[UNR][NV];
  // compact the round at positions [p0, p0 + 64) into half `b` of s_k / s_w (rank = position among the live entries),
  // advance the id / weight prefetch, request the round's rows; returns its number of live entries
  auto stage = [&](i64 p0, int b) -> int {
    const int32_t k = k_n, cv = cv_n;
    const unsigned f = f_n;
    const float4 wk = w_n;
    k_n = k_nn; f_n = f_nn; cv_n = cv_nn;
    w_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f_n) w_n = weights[k_n];
    k_nn = 0; f_nn = 0; cv_nn = 1;
    if (p0 + 2 * OCN_WAVE + lane < da) {
      k_nn = colA[a0 + p0 + 2 * OCN_WAVE + lane]; f_nn = flags[base + p0 + 2 * OCN_WAVE + lane] & fmask;
      if (wc) cv_nn = wc[base + p0 + 2 * OCN_WAVE + lane];
    }
    float wa = 0.f, wb = 0.f;
    if (f) entry_weights(f, wk, (float)cv, wa, wb);
    const bool need = (wa != 0.f) | (wb != 0.f);
    const unsigned long long m = __ballot(need);
    const int n = __popcll(m);
    if (need) {
      const int rank = __popcll(m & ((1ull << lane) - 1ull));
      s_k[wv][b][rank] = k;
      s_w[wv][b][rank] = make_float2(wa, wb);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int u = 0; u < UNR; ++u) {          // unconditional loads (a predicated one is followed by a wait for it): ranks past
      const int r = u * G + g;               // the last live entry read row 0 and are never stored
      const int32_t kr = r < n ? s_k[wv][b][r] : 0;
      const f32x4* row = reinterpret_cast<const f32x4*>(h4 + (i64)kr * rowq + gl);
#pragma unroll
      for (int v = 0; v < NV; ++v) x[u][v] = row[v * LPE];
    }
    return n;
  };
  int n = stage(0, 0);
  int b = 0;
  for (i64 p0 = 0; p0 < da; p0 += OCN_WAVE, b ^= 1) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {          // the rows requested a round ago -> this wave's slab
      const int r = u * G + g;
      if (r < n) {
#pragma unroll
        for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(&s_x[wv][r][gl + v * LPE]) = x[u][v];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int n_next = p0 + OCN_WAVE < da ? stage(p0 + OCN_WAVE, b ^ 1) : 0;
    if (lane < HF)                           // rank order = ascending column, one feature per lane
      chain_rows<1, HF>(&s_w[wv][b][0], reinterpret_cast<const float*>(&s_x[wv][0][0]) + lane, n, acc1, acc2);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    n = n_next;
  }
#ifdef OCN_X_WAVE_CHECK   /* diagnostic build only (tools/dbg_two_stream.py): the same sums once more, straight from memory */
  if (!LONG && lane < HF) {
    float c1 = 0.f, c2 = 0.f;
    for (i64 p = 0; p < da; ++p) {
      const unsigned f = flags[base + p] & fmask;
      if (!f) continue;
      const int32_t k = colA[a0 + p];
      float wa = 0.f, wb = 0.f;
      entry_weights(f, weights[k], wc ? (float)wc[base + p] : 1.f, wa, wb);
      if (wa == 0.f && wb == 0.f) continue;
      const float xv = h[(i64)k * H + lane];
      c1 = __fadd_rn(c1, __fmul_rn(wa, xv));
      c2 = __fadd_rn(c2, __fmul_rn(wb, xv));
    }
    const bool bad1 = __float_as_uint(c1) != __float_as_uint(acc1[0]), bad2 = !full2 && __float_as_uint(c2) != __float_as_uint(acc2[0]);
    if (bad1 || bad2) {
      const unsigned q = atomicAdd(&g_wave_chk_n, 1u);
      if (q < 4096) {
        g_wave_chk[q * 6 + 0] = (float)e; g_wave_chk[q * 6 + 1] = (float)lane; g_wave_chk[q * 6 + 2] = acc1[0];
        g_wave_chk[q * 6 + 3] = c1; g_wave_chk[q * 6 + 4] = acc2[0]; g_wave_chk[q * 6 + 5] = c2;
      }
    }
  }
#endif
  if (lane < HF) {
    const i64 o = (out_row ? out_row[e] : e) * H + lane;
    xcn1[o] = acc1[0];
    xcn2[o] = full2 ? rowsum[i * H + lane] : acc2[0];
    xij[o] = __fmul_rn(h[i * H + lane], h[j * H + lane]);
  }
}

// cn_gather_long_kernel's fetching lane groups: request the rows of sub-round u into registers / store them into a slab
// half.  The loads are unconditional (a predicated load is followed by a wait for it): slots past the sub-round's rows
// read row 0 and are never stored.
template <int LPE, int NV, int RPG, int FG, int SLAB>
__device__ __forceinline__ void long_request(f32x4 (&x)[RPG][NV], const int32_t* __restrict__ s_k, int u, int n, int fg,
                                             int gl, const float4* __restrict__ h4, i64 rowq) {
  const int r0 = u * SLAB;
  const int nr = n - r0;                                        // <= 0 past the last sub-round
#pragma unroll
  for (int q = 0; q < RPG; ++q) {
    const int r = fg + q * FG;
    const bool ok = r < SLAB && r < nr;
    const int32_t kr = s_k[ok ? r0 + r : 0];
    const f32x4* row = reinterpret_cast<const f32x4*>(h4 + (i64)(ok ? kr : 0) * rowq + gl);
#pragma unroll
    for (int v = 0; v < NV; ++v) x[q][v] = row[v * LPE];
  }
}

template <int LPE, int NV, int RPG, int FG, int SLAB>
__device__ __forceinline__ void long_store(const f32x4 (&x)[RPG][NV], float4* __restrict__ half, int nr, int fg, int gl) {
#pragma unroll
  for (int q = 0; q < RPG; ++q) {
    const int r = fg + q * FG;
    if (r < SLAB && r < nr) {
#pragma unroll
      for (int v = 0; v < NV; ++v) *reinterpret_cast<f32x4*>(half + r * (LPE * NV) + gl + v * LPE) = x[q][v];
    }
  }
}

// One workgroup per batch row whose source row is longer than LONG_ROW (hub sources; 8 403 neighbours at the citation2
// shape).  The sum stays the reference's: strictly sequential in ascending column order.  What a workgroup adds is
// the memory parallelism: per round of LONG_THREADS positions the live entries are compacted by rank; waves 1.. fetch
// their embedding rows into one half of a double-buffered LDS slab while wave 0 accumulates the other half in rank
// order — bit for bit the sum cn_gather_kernel forms for a short row.  The fetch is itself pipelined: a step writes
// the rows requested two steps earlier into the slab and requests a later sub-round's into the registers they
// leave, so a barrier never waits for a load issued in its own step.  Wave 0 holds the H features spread over its
// lanes (one per lane at H <= 64, H/64 from there): the sequential chain costs a multiply and an add per lane and
// entry and accumulator, not a float4's worth of them on a quarter of the lanes.
// LONG_THREADS: 1024 for small batches (few hub rows, each as parallel as a workgroup gets), 256 for large ones (one
// workgroup is launched per batch row and all but the hub rows' leave at once).
#ifndef LONG_SMALL_THREADS
#define LONG_SMALL_THREADS 1024             /* hub-row workgroup of a small batch (B <= 4096); 512 (two per CU) measured 5 % slower */
#endif
#define LONG_SLAB_BYTES(threads) ((threads) >= 512 ? 32768 : 16384)   /* per half; dynamic LDS = two halves */
#ifdef OCN_X_LONG_STAMPS   /* diagnostic build: where a hub row's time goes (wave 0 of every hub-row workgroup; tools/longstamps.py) */
__device__ unsigned long long g_long_stamps[8];
extern "C" int ocn_debug_long_stamps(unsigned long long* out, int reset) {
  if (reset) { static unsigned long long z[8]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_long_stamps), z, sizeof(z)); }
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_long_stamps), sizeof(unsigned long long) * 8);
}
#define LSTAMP(k) do { if (threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); const unsigned long long tn = __builtin_amdgcn_s_memtime(); lt[k] += tn - lprev; lprev = tn; } } while (0)
#else
#define LSTAMP(k) do {} while (0)
#endif
template <int LPE, int NV, int LONG_THREADS>
__global__ __launch_bounds__(LONG_THREADS) void cn_gather_long_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ src, const i64* __restrict__ dst, i64 B,
    const i64* __restrict__ off, const uint8_t* __restrict__ flags, const int32_t* __restrict__ wc,
    const float4* __restrict__ weights, const float* __restrict__ h, int H,
    float* __restrict__ xcn1, float* __restrict__ xcn2, float* __restrict__ xij,
    const i64* __restrict__ out_row,     // out_row[batch row] = output row (class-major heads), or NULL
    const int32_t* __restrict__ cnt2, const float* __restrict__ rowsum) {   // see cn_gather_kernel
  constexpr int FG = (LONG_THREADS - OCN_WAVE) / LPE;          // fetching lane groups (waves 1..)
  constexpr int ROWQ = LPE * NV;                               // float4 per embedding row
  constexpr int HF = ROWQ * 4;                                 // features
  constexpr int FPL = HF >= OCN_WAVE ? HF / OCN_WAVE : 1;      // features per lane of wave 0
  constexpr int AL = HF / FPL;                                 // its active lanes
  constexpr int SLAB = LONG_SLAB_BYTES(LONG_THREADS) / (16 * ROWQ);   // rows per sub-round
  constexpr int RPG = (SLAB + FG - 1) / FG;                    // rows a fetching lane group requests per sub-round
  constexpr int WAVES = LONG_THREADS / OCN_WAVE;
  static_assert(FG >= 1 && SLAB >= 1, "");
  extern __shared__ __attribute__((aligned(16))) float4 s_x[];          // [2][SLAB][ROWQ]
  __shared__ int32_t s_k[LONG_THREADS];
  __shared__ float2 s_w[LONG_THREADS];
  __shared__ int s_wcnt[WAVES];
  const i64 e = blockIdx.x;
  const i64 i = src[e], j = dst[e];
  const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0;
  if (da <= LONG_ROW) return;               // whole workgroup leaves together
  if (rowsum && cnt2 && (i64)cnt2[e] == da) return;            // ... also for a row pooled from rowsum by the other launch
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int gl = threadIdx.x % LPE, fg = ((int)threadIdx.x - OCN_WAVE) / LPE;
  const i64 base = off[e];
  const float4* h4 = reinterpret_cast<const float4*>(h);
  const i64 rowq = H >> 2;
  float acc1[FPL], acc2[FPL];
#pragma unroll
  for (int q = 0; q < FPL; ++q) acc1[q] = acc2[q] = 0.f;
#ifdef OCN_X_LONG_STAMPS
  unsigned long long lt[6] = {0, 0, 0, 0, 0, 0}, lprev = __builtin_amdgcn_s_memtime();
#endif
  for (i64 p0 = 0; p0 < da; p0 += LONG_THREADS) {
    const i64 p = p0 + threadIdx.x;
    int32_t k = 0, cv = 1;
    unsigned f = 0;
    if (p < da) { k = colA[a0 + p]; f = flags[base + p]; if (wc) cv = wc[base + p]; }
    LSTAMP(0);                                                   // ids / flags / cn2 values
    float wa = 0.f, wb = 0.f;
    if (f) entry_weights(f, weights[k], (float)cv, wa, wb);
    LSTAMP(1);                                                   // column weights
    const bool need = (wa != 0.f) | (wb != 0.f);
    const unsigned long long m = __ballot(need);
    if (lane == 0) s_wcnt[wv] = __popcll(m);
    __syncthreads();
    int before = 0, n = 0;
#pragma unroll
    for (int q = 0; q < WAVES; ++q) {
      const int c = s_wcnt[q];
      if (q < wv) before += c;
      n += c;
    }
    if (need) {                               // compaction by rank: ascending position = ascending column
      const int rank = before + __popcll(m & ((1ull << lane) - 1ull));
      s_k[rank] = k;
      s_w[rank] = make_float2(wa, wb);
    }
    __syncthreads();
    const int nsr = (n + SLAB - 1) / SLAB;
    LSTAMP(2);                                                   // compaction (two barriers)
    // Sub-round u lives in register set u & 1 of the fetching waves from its request until it is stored two steps
    // later.  Step t: waves 1.. store sub-round t + 1 and request t + 3 into the set it leaves; wave 0 sums sub-round t
    // from the slab half t & 1; barrier.
    f32x4 xa[RPG][NV], xb[RPG][NV];
    if (wv > 0) {
      long_request<LPE, NV, RPG, FG, SLAB>(xa, s_k, 0, n, fg, gl, h4, rowq);
      long_request<LPE, NV, RPG, FG, SLAB>(xb, s_k, 1, n, fg, gl, h4, rowq);
    }
    for (int t = -1; t < nsr; t += 2) {
      // t + 1 is even: set a
      if (wv > 0) {
        long_store<LPE, NV, RPG, FG, SLAB>(xa, s_x + (size_t)((t + 1) & 1) * SLAB * ROWQ, n - (t + 1) * SLAB, fg, gl);
        long_request<LPE, NV, RPG, FG, SLAB>(xa, s_k, t + 3, n, fg, gl, h4, rowq);
      } else if (t >= 0 && lane < AL) {
        const int r0 = t * SLAB;
        chain_rows<FPL, HF>(s_w + r0, reinterpret_cast<const float*>(s_x + (size_t)(t & 1) * SLAB * ROWQ) + lane * FPL,
                            n - r0 < SLAB ? n - r0 : SLAB, acc1, acc2);
      }
      __syncthreads();
      if (t + 1 < nsr) {                                        // uniform: every thread takes the same barriers
        if (wv > 0) {
          long_store<LPE, NV, RPG, FG, SLAB>(xb, s_x + (size_t)((t + 2) & 1) * SLAB * ROWQ, n - (t + 2) * SLAB, fg, gl);
          long_request<LPE, NV, RPG, FG, SLAB>(xb, s_k, t + 4, n, fg, gl, h4, rowq);
        } else if (lane < AL) {
          const int r0 = (t + 1) * SLAB;
          chain_rows<FPL, HF>(s_w + r0, reinterpret_cast<const float*>(s_x + (size_t)((t + 1) & 1) * SLAB * ROWQ) + lane * FPL,
                              n - r0 < SLAB ? n - r0 : SLAB, acc1, acc2);
        }
        __syncthreads();
      }
    }
    LSTAMP(3);                                                   // the round's sub-rounds: fetch / store / sum, a barrier each
#ifdef OCN_X_LONG_STAMPS
    if (threadIdx.x == 0) { lt[4] += (unsigned long long)n; lt[5] += 1; }
#endif
  }
#ifdef OCN_X_LONG_STAMPS
  if (threadIdx.x == 0) {
    for (int q = 0; q < 6; ++q) atomicAdd(&g_long_stamps[q], lt[q]);
    atomicAdd(&g_long_stamps[6], 1ull);
    atomicMax(&g_long_stamps[7], lt[0] + lt[1] + lt[2] + lt[3]);
  }
#endif
  if (wv == 0 && lane < AL) {
    const i64 o = (out_row ? out_row[e] : e) * H + lane * FPL;
#pragma unroll
    for (int q = 0; q < FPL; ++q) {
      xcn1[o + q] = acc1[q];
      xcn2[o + q] = acc2[q];
      xij[o + q] = __fmul_rn(h[i * H + lane * FPL + q], h[j * H + lane * FPL + q]);
    }
  }
}

// cn6 pooling: three pooled vectors.  flagsA carries the cn1 / cn2 bits, flagsB's bit 0 the cn3 bit (two
// intersection passes over the same source rows, so the same `off`); per entry
//   w1 = [cn1]*inv1,  w2 = ([cn2] - t*[cn1])*inv2,  w3 = (([cn3] - t*[cn1]) - nip*w2)*inv3,
// each product / difference rounded separately, pooled in ascending column order.  LPE lanes per
// batch row, like cn_gather_kernel (no hub-row split: every row keeps the sequential order).
template <int LPE, int NV>
__global__ __launch_bounds__(OCN_BLOCK) void cn_gather3_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    const i64* __restrict__ off, const uint8_t* __restrict__ flagsA, const uint8_t* __restrict__ flagsB,
    const float4* __restrict__ wA, const float4* __restrict__ wB, const float* __restrict__ nip_p,
    const float* __restrict__ h, int H, float* __restrict__ xcn1, float* __restrict__ xcn2,
    float* __restrict__ xcn3, float* __restrict__ xij) {
  constexpr int GPW = OCN_WAVE / LPE;
  constexpr int UNR = 4;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPE;
  const int gbase = lane - gl;
  const i64 slot = ((i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6)) * GPW + lane / LPE;
  if (slot >= B) return;
  const i64 e = order ? order[slot] : slot;
  const i64 i = src[e], j = dst[e];
  const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0;
  const i64 base = off[e];
  const float nip = nip_p[0];
  const float4* h4 = reinterpret_cast<const float4*>(h);
  const i64 rowq = H >> 2;
  float4 acc1[NV], acc2[NV], acc3[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc1[v] = acc2[v] = acc3[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (i64 p0 = 0; p0 < da; p0 += LPE) {
    const i64 p = p0 + gl;
    int32_t k = 0;
    unsigned fa = 0, fb = 0;
    if (p < da) { k = colA[a0 + p]; fa = flagsA[base + p]; fb = flagsB[base + p] & OCN_F_CN1; }
    float w1 = 0.f, w2 = 0.f, w3 = 0.f;
    if (fa | fb) {
      const float4 a = wA[k];
      const float inv3 = wB[k].x;
      const float tt = (fa & OCN_F_CN1) ? a.y : 0.f;
      w1 = (fa & OCN_F_CN1) ? a.x : 0.f;
      w2 = __fmul_rn(__fsub_rn((fa & OCN_F_CN2) ? 1.0f : 0.f, tt), a.z);
      w3 = __fmul_rn(__fsub_rn(__fsub_rn(fb ? 1.0f : 0.f, tt), __fmul_rn(nip, w2)), inv3);
    }
    const bool need = (w1 != 0.f) | (w2 != 0.f) | (w3 != 0.f);
    unsigned long long m = __ballot(need);
    if (LPE < 64) m = (m >> gbase) & ((1ull << (LPE & 63)) - 1ull);
    while (m) {
      int bsel[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        bsel[u] = m ? (__ffsll((long long)m) - 1) : -1;
        m &= m - 1;
      }
      float ww1[UNR], ww2[UNR], ww3[UNR];
      float4 x[UNR][NV];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int sl = gbase + (bsel[u] < 0 ? 0 : bsel[u]);
        const int32_t kk = __shfl(k, sl, OCN_WAVE);
        ww1[u] = __shfl(w1, sl, OCN_WAVE);
        ww2[u] = __shfl(w2, sl, OCN_WAVE);
        ww3[u] = __shfl(w3, sl, OCN_WAVE);
        if (bsel[u] >= 0) {
          const float4* row = h4 + (i64)kk * rowq + gl;
#pragma unroll
          for (int v = 0; v < NV; ++v) x[u][v] = row[v * LPE];
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (bsel[u] >= 0) {
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            axpy4(acc1[v], ww1[u], x[u][v]);
            axpy4(acc2[v], ww2[u], x[u][v]);
            axpy4(acc3[v], ww3[u], x[u][v]);
          }
        }
      }
    }
  }
  pool_store<LPE, NV>(e, i, j, gl, h4, rowq, acc1, acc2, xcn1, xcn2, xij);
  float4* o3 = reinterpret_cast<float4*>(xcn3) + e * rowq + gl;
#pragma unroll
  for (int v = 0; v < NV; ++v) o3[v * LPE] = acc3[v];
}

// any H: one wave per edge, one feature per lane per 64-wide chunk (re-walks the flags per chunk)
__global__ __launch_bounds__(OCN_BLOCK) void cn_gather_generic(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    const i64* __restrict__ off, const uint8_t* __restrict__ flags, const int32_t* __restrict__ wc,
    const float4* __restrict__ weights, const float* __restrict__ h, int H,
    float* __restrict__ xcn1, float* __restrict__ xcn2, float* __restrict__ xij,
    const i64* __restrict__ out_row) {   // out_row[batch row] = output row (class-major heads), or NULL
  const int lane = threadIdx.x & 63;
  const i64 slot = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6);
  if (slot >= B) return;
  const i64 e = order ? order[slot] : slot;
  const i64 i = src[e], j = dst[e];
  const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0;
  const i64 base = off[e];
  for (int f0 = 0; f0 < H; f0 += OCN_WAVE) {
    const int ft = f0 + lane;
    const bool fin = ft < H;
    float acc1 = 0.f, acc2 = 0.f;
    for (i64 p0 = 0; p0 < da; p0 += OCN_WAVE) {
      const i64 p = p0 + lane;
      int32_t k = 0;
      unsigned f = 0;
      if (p < da) { k = colA[a0 + p]; f = flags[base + p]; }
      float wa = 0.f, wb = 0.f;
      if (f) entry_weights(f, weights[k], wc ? (float)wc[base + p] : 1.0f, wa, wb);
      unsigned long long m = __ballot((wa != 0.f) | (wb != 0.f));
      while (m) {
        const int b = __ffsll((long long)m) - 1;
        m &= m - 1;
        const int32_t kk = __shfl(k, b, OCN_WAVE);
        const float a = __shfl(wa, b, OCN_WAVE), bb = __shfl(wb, b, OCN_WAVE);
        if (fin) {
          const float x = h[(i64)kk * H + ft];
          acc1 = __fadd_rn(acc1, __fmul_rn(a, x));
          acc2 = __fadd_rn(acc2, __fmul_rn(bb, x));
        }
      }
    }
    if (fin) {
      const i64 oe = out_row ? out_row[e] : e;
      xcn1[oe * H + ft] = acc1;
      xcn2[oe * H + ft] = acc2;
      xij[oe * H + ft] = __fmul_rn(h[i * H + ft], h[j * H + ft]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// K3 backward: dh[k] += wa·g1[e] + wb·g2[e] over the flagged neighbours (the transposed pooling),
// dh[i] += g3[e] ⊙ h[j], dh[j] += g3[e] ⊙ h[i].  fp32 atomics, one 1-KiB row segment per
// wave-instruction (the shape the memory-side atomic units take at full rate).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add4(float* p, const float4& v) {
  atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}

template <int LPE, int NV>
__global__ __launch_bounds__(OCN_BLOCK) void cn_scatter_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    const i64* __restrict__ off, const uint8_t* __restrict__ flags, const int32_t* __restrict__ wc,
    const float4* __restrict__ weights, const float* __restrict__ h, int H,
    const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ g3,
    float* __restrict__ dh) {
  constexpr int GPW = OCN_WAVE / LPE;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPE;
  const int gbase = lane - gl;
  const i64 slot = ((i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6)) * GPW + lane / LPE;
  if (slot >= B) return;
  const i64 e = order ? order[slot] : slot;
  const i64 i = src[e], j = dst[e];
  const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0;
  const i64 base = off[e];
  const i64 rowq = H >> 2;
  const float4* h4 = reinterpret_cast<const float4*>(h);
  float4 v1[NV], v2[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    v1[v] = reinterpret_cast<const float4*>(g1)[e * rowq + gl + v * LPE];
    v2[v] = reinterpret_cast<const float4*>(g2)[e * rowq + gl + v * LPE];
  }
  for (i64 p0 = 0; p0 < da; p0 += LPE) {
    const i64 p = p0 + gl;
    int32_t k = 0;
    unsigned f = 0;
    if (p < da) { k = colA[a0 + p]; f = flags[base + p]; }
    float wa = 0.f, wb = 0.f;
    if (f) entry_weights(f, weights[k], wc ? (float)wc[base + p] : 1.0f, wa, wb);
    unsigned long long m = __ballot((wa != 0.f) | (wb != 0.f));
    if (LPE < 64) m = (m >> gbase) & ((1ull << (LPE & 63)) - 1ull);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int32_t kk = __shfl(k, gbase + b, OCN_WAVE);
      const float a = __shfl(wa, gbase + b, OCN_WAVE), bb = __shfl(wb, gbase + b, OCN_WAVE);
      float* row = dh + (i64)kk * H + 4 * gl;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float4 c;
        c.x = a * v1[v].x + bb * v2[v].x; c.y = a * v1[v].y + bb * v2[v].y;
        c.z = a * v1[v].z + bb * v2[v].z; c.w = a * v1[v].w + bb * v2[v].w;
        atomic_add4(row + 4 * v * LPE, c);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float4 g = reinterpret_cast<const float4*>(g3)[e * rowq + gl + v * LPE];
    const float4 hi = h4[i * rowq + gl + v * LPE], hj = h4[j * rowq + gl + v * LPE];
    atomic_add4(dh + i * H + 4 * (gl + v * LPE), make_float4(g.x * hj.x, g.y * hj.y, g.z * hj.z, g.w * hj.w));
    atomic_add4(dh + j * H + 4 * (gl + v * LPE), make_float4(g.x * hi.x, g.y * hi.y, g.z * hi.z, g.w * hi.w));
  }
}

// cn6: the transposed pooling of cn_gather3_kernel, dh[k] += w1 g1[e] + w2 g2[e] + w3 g3[e] over the union entries with the
// weights formed exactly as the forward forms them, and the Hadamard term's two ends from g4.  fp32 atomics.
template <int LPE, int NV>
__global__ __launch_bounds__(OCN_BLOCK) void cn_scatter3_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA,
    const i64* __restrict__ src, const i64* __restrict__ dst, const i64* __restrict__ order, i64 B,
    const i64* __restrict__ off, const uint8_t* __restrict__ flagsA, const uint8_t* __restrict__ flagsB,
    const float4* __restrict__ wA, const float4* __restrict__ wB, const float* __restrict__ nip_p,
    const float* __restrict__ h, int H, const float* __restrict__ g1, const float* __restrict__ g2,
    const float* __restrict__ g3, const float* __restrict__ g4, float* __restrict__ dh) {
  constexpr int GPW = OCN_WAVE / LPE;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPE;
  const int gbase = lane - gl;
  const i64 slot = ((i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6)) * GPW + lane / LPE;
  if (slot >= B) return;
  const i64 e = order ? order[slot] : slot;
  const i64 i = src[e], j = dst[e];
  const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0;
  const i64 base = off[e];
  const float nip = nip_p[0];
  const i64 rowq = H >> 2;
  const float4* h4 = reinterpret_cast<const float4*>(h);
  float4 v1[NV], v2[NV], v3[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    v1[v] = reinterpret_cast<const float4*>(g1)[e * rowq + gl + v * LPE];
    v2[v] = reinterpret_cast<const float4*>(g2)[e * rowq + gl + v * LPE];
    v3[v] = reinterpret_cast<const float4*>(g3)[e * rowq + gl + v * LPE];
  }
  for (i64 p0 = 0; p0 < da; p0 += LPE) {
    const i64 p = p0 + gl;
    int32_t k = 0;
    unsigned fa = 0, fb = 0;
    if (p < da) { k = colA[a0 + p]; fa = flagsA[base + p]; fb = flagsB[base + p] & OCN_F_CN1; }
    float w1 = 0.f, w2 = 0.f, w3 = 0.f;
    if (fa | fb) {                           // (the weights of cn_gather3_kernel, term for term)
      const float4 a = wA[k];
      const float inv3 = wB[k].x;
      const float tt = (fa & OCN_F_CN1) ? a.y : 0.f;
      w1 = (fa & OCN_F_CN1) ? a.x : 0.f;
      w2 = __fmul_rn(__fsub_rn((fa & OCN_F_CN2) ? 1.0f : 0.f, tt), a.z);
      w3 = __fmul_rn(__fsub_rn(__fsub_rn(fb ? 1.0f : 0.f, tt), __fmul_rn(nip, w2)), inv3);
    }
    unsigned long long m = __ballot((w1 != 0.f) | (w2 != 0.f) | (w3 != 0.f));
    if (LPE < 64) m = (m >> gbase) & ((1ull << (LPE & 63)) - 1ull);
    while (m) {
      const int b = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int32_t kk = __shfl(k, gbase + b, OCN_WAVE);
      const float a = __shfl(w1, gbase + b, OCN_WAVE), bb = __shfl(w2, gbase + b, OCN_WAVE), cc = __shfl(w3, gbase + b, OCN_WAVE);
      float* row = dh + (i64)kk * H + 4 * gl;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        float4 c;
        c.x = a * v1[v].x + bb * v2[v].x + cc * v3[v].x; c.y = a * v1[v].y + bb * v2[v].y + cc * v3[v].y;
        c.z = a * v1[v].z + bb * v2[v].z + cc * v3[v].z; c.w = a * v1[v].w + bb * v2[v].w + cc * v3[v].w;
        atomic_add4(row + 4 * v * LPE, c);
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const float4 g = reinterpret_cast<const float4*>(g4)[e * rowq + gl + v * LPE];
    const float4 hi = h4[i * rowq + gl + v * LPE], hj = h4[j * rowq + gl + v * LPE];
    atomic_add4(dh + i * H + 4 * (gl + v * LPE), make_float4(g.x * hj.x, g.y * hj.y, g.z * hj.z, g.w * hj.w));
    atomic_add4(dh + j * H + 4 * (gl + v * LPE), make_float4(g.x * hi.x, g.y * hi.y, g.z * hi.z, g.w * hi.w));
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
template <int LPE, int NV>
static void launch_gather(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                          const int64_t* order, int64_t B, const int64_t* off, const uint8_t* flags,
                          const int32_t* wc, const float* weights, const float* h, int32_t H, int64_t max_row_len,
                          float* xcn1, float* xcn2, float* xij, const int64_t* out_row, const int32_t* cnt1,
                          const int32_t* cnt2, const uint64_t* rec, const int32_t* perm, const float* rowsum, hipStream_t st) {
  const i64 epb = (i64)OCN_WPB * (OCN_WAVE / LPE);
  // the schedule's groups are the workgroups of the intersection pass: usable where the pooling's workgroups are the same
  const bool sched = perm && rec && epb == SCHED_GROUP * POOL_FOLD && ((B + epb - 1) / epb) % 8 == 0;
  bool packed = true;
  if constexpr (LPE <= 16) {
    if (B * LPE < 262144) {                  // the packed form would not fill the SIMDs
      constexpr int WW = gather_wave_wpb<LPE, NV, false>();
      hipLaunchKernelGGL((cn_gather_wave_kernel<LPE, NV>), dim3((unsigned)((B + WW - 1) / WW)),
                         dim3(WW * OCN_WAVE), 0, st, (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst,
                         (const i64*)order, (i64)B, (const i64*)off, flags, wc, (const float4*)weights, h, (int)H,
                         xcn1, xcn2, xij, (const i64*)out_row, cnt2, rowsum);
      packed = false;
    }
  }
#define PACKED_ARGS (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst, (const i64*)order, (i64)B, \
                    (const i64*)off, flags, wc, (const float4*)weights, h, (int)H, xcn1, xcn2, xij,           \
                    (const i64*)out_row, cnt1, cnt2, (const u64*)rec
  if (packed) {
    bool sliced = false;
#ifdef OCN_X_SLICE
    // EXPERIMENT, off in the product (pooling): one feature slice per XCD.  Bit-identical scores, but 8
    // candidates share a wave and run in lockstep to the longest of them: 0.76 ms against 0.22 ms at the collab shape.
    if constexpr (LPE * NV >= 64 && (LPE * NV) % 8 == 0) {
      constexpr int SL = LPE * NV / 8;                           // float4 per slice
      const i64 spb = (i64)OCN_WPB * (OCN_WAVE / SL);
      if (B >= GATHER_SLICE_MIN_BATCH) {
        hipLaunchKernelGGL((cn_gather_kernel<SL, 1, true>), dim3((unsigned)(8 * ((B + spb - 1) / spb))), dim3(OCN_BLOCK),
                           0, st, PACKED_ARGS, (const int32_t*)nullptr, rowsum);
        sliced = true;
      }
    }
#endif
    if (!sliced)
      hipLaunchKernelGGL((cn_gather_kernel<LPE, NV>), dim3((unsigned)((B + epb - 1) / epb)), dim3(OCN_BLOCK), 0, st,
                         PACKED_ARGS, sched ? perm : (const int32_t*)nullptr, rowsum);
  }
#undef PACKED_ARGS
#define LONG_ARGS (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst, (i64)B, (const i64*)off, flags, wc, \
                  (const float4*)weights, h, (int)H, xcn1, xcn2, xij, (const i64*)out_row, cnt2, rowsum
  if (max_row_len > LONG_ROW) {
    // (H = 512 — two float4 per lane — does not fit the 128 registers a 1024-thread workgroup leaves a lane: 104 spilled
    // VGPRs in round 3; it takes the 256-thread form at every batch size)
    bool small = false;
    if constexpr (NV == 1) {
      if (B <= 4096) {
        static bool raised_dev[64] = {};        // 2 x 64 KiB of slab: above the default dynamic-LDS limit (attribute is per device)
        int devid = 0;
        if (hipGetDevice(&devid) == hipSuccess && devid >= 0 && devid < 64 && !raised_dev[devid]) {
          if (hipFuncSetAttribute((const void*)cn_gather_long_kernel<LPE, NV, LONG_SMALL_THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * LONG_SLAB_BYTES(LONG_SMALL_THREADS)) == hipSuccess) raised_dev[devid] = true;
        }
        hipLaunchKernelGGL((cn_gather_long_kernel<LPE, NV, LONG_SMALL_THREADS>), dim3((unsigned)B), dim3(LONG_SMALL_THREADS), 2 * LONG_SLAB_BYTES(LONG_SMALL_THREADS), st, LONG_ARGS);
        small = true;
      }
    }
    if (!small) {
      bool by_wave = false;
      if constexpr (LPE <= 16) {             // narrow embeddings: a wave per hub row, 64 gathers in flight each
        hipLaunchKernelGGL((cn_gather_wave_kernel<LPE, NV, true>), dim3((unsigned)B), dim3(OCN_WAVE), 0, st, (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst,
                           (const i64*)order, (i64)B, (const i64*)off, flags, wc, (const float4*)weights, h, (int)H,
                           xcn1, xcn2, xij, (const i64*)out_row, cnt2, rowsum);
        by_wave = true;
      }
      if (!by_wave) hipLaunchKernelGGL((cn_gather_long_kernel<LPE, NV, 256>), dim3((unsigned)B), dim3(256), 2 * LONG_SLAB_BYTES(256), st, LONG_ARGS);
    }
  }
#undef LONG_ARGS
}

extern "C" {

int32_t ocn_cn_flags_small_graph_cols(void) { return LH_MAX_COLS; }

int ocn_cn_flags(const int64_t* rowptrA, const int32_t* colA, const int64_t* rowptrT1,
                 const int32_t* colT1, const int64_t* rowptrT2, const int32_t* colT2,
                 const uint32_t* bitmapT1, int64_t bm1_stride_words, const uint32_t* bitmapT2, int64_t bm_stride_words,
                 const int64_t* src, const int64_t* dst, const int64_t* order, int64_t B,
                 int64_t n_cols, const int64_t* off, uint8_t* flags, int64_t flags_cap, uint64_t* hist,
                 int32_t* cnt1, int32_t* cnt2, int32_t* status, uint64_t* rec, int32_t* gcost, void* stream) {
  if (B < 0 || flags_cap < 0 || B > (int64_t)HF_MASK) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!rowptrA || (!rowptrT1 && !bitmapT1) || !src || !dst || !off || !hist || !cnt1 || !status) return OCN_EINVAL;
  if ((bitmapT1 && bm1_stride_words * 32 < n_cols) || (bitmapT2 && bm_stride_words * 32 < n_cols)) return OCN_EINVAL;
  // col pointers may legitimately be NULL for an adjacency with no entries
  constexpr int GPB = OCN_BLOCK / OCN_X_G;
  hipStream_t st = (hipStream_t)stream;
  const bool lh = n_cols > 0 && n_cols <= LH_MAX_COLS;
  const size_t lds = lh ? (size_t)n_cols * sizeof(u64) : 0;
  // LH: a persistent grid (a few workgroups per CU) so that each LDS histogram absorbs many edges
  const int grid = lh ? grid_for((B + GPB - 1) / GPB, 256 * 3) : grid_for((B + GPB - 1) / GPB);
  if (lh) {                                   // static (target rows) + dynamic (histogram) LDS can pass 64 KiB
    static bool raised_dev[64] = {};          // the attribute is per device (function objects are per device)
    int devid = 0;
    if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return OCN_EINVAL;
    bool& raised = raised_dev[devid];
    if (!raised) {
      hipError_t e1 = hipFuncSetAttribute((const void*)cn_flags_kernel<OCN_X_G, true, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          LH_MAX_COLS * (int)sizeof(u64));
      hipError_t e2 = hipFuncSetAttribute((const void*)cn_flags_kernel<OCN_X_G, false, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize,
                                          LH_MAX_COLS * (int)sizeof(u64));
      if (e1 != hipSuccess) return (int)e1;
      if (e2 != hipSuccess) return (int)e2;
      raised = true;
    }
  }
#define CN_FLAGS_ARGS(T2P, T2C)                                                                      \
  (const i64*)rowptrA, colA, (const i64*)rowptrT1, colT1, (const i64*)(T2P), (T2C),                   \
      (const unsigned*)bitmapT1, (i64)bm1_stride_words, (const unsigned*)bitmapT2, (i64)bm_stride_words, (const i64*)src, \
      (const i64*)dst, (const i64*)order, (i64)B, (i64)n_cols, (const i64*)off, flags, (i64)flags_cap, \
      (u64*)hist, cnt1, cnt2, status, (u64*)rec, gcost
  if (!rowptrT2 && bitmapT2 && lh) return OCN_EINVAL;      // (small graphs read T2's row lengths beside its bit rows)
  if (rowptrT2 || bitmapT2) {                              // T2 by its bit rows alone: a product whose rows are built on demand
    if (lh) hipLaunchKernelGGL((cn_flags_kernel<OCN_X_G, true, true>), dim3(grid), dim3(OCN_BLOCK), lds, st,
                               CN_FLAGS_ARGS(rowptrT2, colT2));
    else hipLaunchKernelGGL((cn_flags_kernel<OCN_X_G, true, false>), dim3(grid), dim3(OCN_BLOCK), 0, st,
                            CN_FLAGS_ARGS(rowptrT2, colT2));
  } else {
    if (lh) hipLaunchKernelGGL((cn_flags_kernel<OCN_X_G, false, true>), dim3(grid), dim3(OCN_BLOCK), lds, st,
                               CN_FLAGS_ARGS(nullptr, (const int32_t*)nullptr));
    else hipLaunchKernelGGL((cn_flags_kernel<OCN_X_G, false, false>), dim3(grid), dim3(OCN_BLOCK), 0, st,
                            CN_FLAGS_ARGS(nullptr, (const int32_t*)nullptr));
  }
  return launch_status();
}

int ocn_cn_walk_flags(const int64_t* rowptrA, const int32_t* colA, const int64_t* nds, const int64_t* src,
                      const int64_t* dst, const int64_t* order, int64_t B, const int64_t* chunk_off,
                      const int64_t* rev_off, const int64_t* off, int64_t max_row_len, uint8_t* flags, int32_t* wc,
                      int64_t flags_cap, uint64_t* hist, int32_t* cnt1, int32_t* cnt2, int32_t* status,
                      void* stream) {
  if (B < 0 || flags_cap < 0 || B > (int64_t)HF_MASK) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!rowptrA || !src || !dst || !chunk_off || !off || !hist || !cnt1 || !cnt2 || !status) return OCN_EINVAL;
  if (flags_cap > 0 && (!flags || !wc)) return OCN_EINVAL;
  if ((nds == nullptr) != (rev_off == nullptr)) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  (void)max_row_len;
  // the item counts live on the device (chunk_off[B], rev_off[B]); fixed grids stride over them
  const int grid = grid_for(4 * B, 256 * 4);   // what can be resident; the items are drawn from a ticket counter
  if (nds && flags_cap > 0) {
    hipLaunchKernelGGL(walk_zero_kernel, dim3(grid_for((flags_cap + OCN_BLOCK - 1) / OCN_BLOCK, 2048)),
                       dim3(OCN_BLOCK), 0, st, (const i64*)off, (i64)B, (i64)flags_cap, wc);
    hipLaunchKernelGGL(cn_walk_rev_kernel, dim3(grid), dim3(WALK_THREADS), 0, st,
                       (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst, (const i64*)order, (i64)B,
                       (const i64*)rev_off, (const i64*)off, wc, (i64)flags_cap, status + 2);
  }
  hipLaunchKernelGGL(cn_walk_kernel, dim3(grid), dim3(WALK_THREADS), 0, st,
                     (const i64*)rowptrA, colA, (const i64*)nds, (const i64*)src,
                     (const i64*)dst, (const i64*)order, (i64)B, (const i64*)chunk_off, (const i64*)off, flags, wc,
                     (i64)flags_cap, (u64*)hist, cnt1, cnt2, status);
  return launch_status();
}

int ocn_neighbor_degree_sum(const int64_t* rowptr, const int32_t* col, int64_t n_rows, int64_t* out, void* stream) {
  if (n_rows < 0 || (n_rows > 0 && (!rowptr || !out))) return OCN_EINVAL;
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(neighbor_degree_sum_kernel, dim3(grid_for((n_rows + OCN_WPB - 1) / OCN_WPB, 1 << 16)),
                     dim3(OCN_BLOCK), 0, (hipStream_t)stream, (const i64*)rowptr, col, (i64)n_rows, (i64*)out);
  return launch_status();
}

int32_t ocn_walk_chunk(void) { return WALK_CHUNK; }

int ocn_cn5_column_stats(const uint64_t* hist, int64_t N, int32_t* scalars, void* stream) {
  if (N < 0 || (N > 0 && (!hist || !scalars))) return OCN_EINVAL;
  if (N == 0) return 0;
  const int grid = grid_for((N + OCN_BLOCK - 1) / OCN_BLOCK, 512);
  hipLaunchKernelGGL(cn5_column_stats, dim3(grid), dim3(OCN_BLOCK), 0, (hipStream_t)stream, (const u64*)hist, (i64)N,
                     scalars);
  return launch_status();
}

int ocn_cn_weights_cn5(uint64_t* hist, int64_t N, const float* innerprod, int32_t* scalars,
                       int32_t valued, const float* s2_exact, void* stream) {
  if (N < 0 || (N > 0 && (!hist || !innerprod || !scalars))) return OCN_EINVAL;
  if (N == 0) return 0;
  const int grid = grid_for((N + OCN_BLOCK - 1) / OCN_BLOCK, 2048);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cn5_column_weights, dim3(grid), dim3(OCN_BLOCK), 0, st, (u64*)hist, (i64)N,
                     innerprod, (const int32_t*)scalars, (int)valued, s2_exact);
  return launch_status();
}

int ocn_cn_weights_cn6(uint64_t* histA, uint64_t* histB, int64_t N, const float* innerprod, int32_t* scalars,
                       float* nip_out, const float* s2_exact, const float* s3_exact, void* stream) {
  if (N < 0 || (N > 0 && (!histA || !histB || !innerprod || !scalars || !nip_out))) return OCN_EINVAL;
  if (N == 0) return 0;
  const int grid = grid_for((N + OCN_BLOCK - 1) / OCN_BLOCK, 2048);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(cn6_column_weights, dim3(grid), dim3(OCN_BLOCK), 0, st, (u64*)histA, (u64*)histB, (i64)N,
                     innerprod, (const int32_t*)scalars, nip_out, s2_exact, s3_exact);
  return launch_status();
}

int ocn_cn_weights_cn7(uint64_t* hist, int64_t N, float sum_fill, const float* diag1, const float* diag2, void* stream) {
  if (N < 0 || (N > 0 && !hist)) return OCN_EINVAL;
  if (N == 0) return 0;
  const int grid = grid_for((N + OCN_BLOCK - 1) / OCN_BLOCK, 2048);
  hipLaunchKernelGGL(cn7_column_weights, dim3(grid), dim3(OCN_BLOCK), 0, (hipStream_t)stream,
                     (u64*)hist, (i64)N, sum_fill, diag1, diag2);
  return launch_status();
}

#define GATHER_ARGS (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst, (const i64*)order, (i64)B, (const i64*)off, \
                    flags, wc, (const float4*)weights, h, (int)H, xcn1, xcn2, xij, (const i64*)out_row
#define LAUNCH_GATHER(LPE, NV)                                                                      \
  launch_gather<LPE, NV>(rowptrA, colA, src, dst, order, B, off, flags, wc, weights, h, H, max_row_len, xcn1, xcn2, xij, out_row, cnt1, cnt2, rec, perm, rowsum, st)

// The visiting order of the pooling's slot groups: each XCD's contiguous eighth of the groups, stable-sorted by
// descending cost (64 buckets) — one workgroup per eighth, a counting sort in LDS: thread t counts the
// buckets of its contiguous share, the bucket-major table of counts is scanned, and the second pass writes every group
// to its rank.  Stable: groups of one source (one cost) stay neighbours.
#define SCHED_BUCKETS 64
// `per`: groups per sorted range — an XCD's whole eighth, or a SEGMENT of it (ocn_gather_schedule `segment`): longest first inside
// segments keeps the sources in flight on an XCD within a narrow id range (their rows then meet in its L2) and still starts
// every segment with its long jobs.
__global__ __launch_bounds__(OCN_BLOCK) void gather_schedule_kernel(const int32_t* __restrict__ gcost, i64 per,
                                                                    int32_t* __restrict__ perm) {
  __shared__ unsigned short tc[SCHED_BUCKETS * OCN_BLOCK];
  __shared__ i64 s_scan[2 * OCN_WPB];
  const int t = threadIdx.x;
  const i64 lo = (i64)blockIdx.x * per;
  const i64 ipt = (per + OCN_BLOCK - 1) / OCN_BLOCK;                          // groups per thread, contiguous
  for (int b = 0; b < SCHED_BUCKETS; ++b) tc[b * OCN_BLOCK + t] = 0;
  auto bucket = [&](i64 q) -> int {
    int c = 0;
#pragma unroll
    for (int f = 0; f < POOL_FOLD; ++f) c = SCHED_COST(c, gcost[(lo + q) * POOL_FOLD + f]);
    c >>= OCN_X_SCHED_SHIFT;
    return SCHED_BUCKETS - 1 - (c < SCHED_BUCKETS - 1 ? c : SCHED_BUCKETS - 1);
  };
  for (i64 k = 0; k < ipt; ++k) {
    const i64 q = (i64)t * ipt + k;
    if (q < per) tc[bucket(q) * OCN_BLOCK + t] += 1;
  }
  __syncthreads();
  // exclusive scan of the flattened table (bucket-major, then thread): thread t owns entries [64 t, 64 t + 64)
  i64 mine = 0;
  for (int q = 0; q < SCHED_BUCKETS; ++q) mine += tc[t * SCHED_BUCKETS + q];
  i64 tot;
  i64 run = block_excl_scan(mine, s_scan, &tot);
  for (int q = 0; q < SCHED_BUCKETS; ++q) {
    const int v = tc[t * SCHED_BUCKETS + q];
    tc[t * SCHED_BUCKETS + q] = (unsigned short)run;
    run += v;
  }
  __syncthreads();
  for (i64 k = 0; k < ipt; ++k) {
    const i64 q = (i64)t * ipt + k;
    if (q < per) {
      const int b = bucket(q);
      perm[lo + tc[b * OCN_BLOCK + t]++] = (int32_t)(lo + q);
    }
  }
}

int ocn_gather_schedule(const int32_t* gcost, int64_t n_groups, int64_t segment, int32_t* perm, void* stream) {
  if (n_groups < 0 || (n_groups & 7) || (n_groups >> 3) > 65535 || segment < 0) return OCN_EINVAL;      // eighths; ranks are 16-bit
  if (n_groups == 0) return 0;
  if (!gcost || !perm) return OCN_EINVAL;
  if (n_groups % (8 * POOL_FOLD)) return OCN_EINVAL;
  n_groups /= POOL_FOLD;
  i64 per = n_groups >> 3;
  if (segment > 0 && segment < per && per % segment == 0) per = segment;      // (a segment that does not divide the eighth: whole eighths)
  hipLaunchKernelGGL(gather_schedule_kernel, dim3((unsigned)(n_groups / per)), dim3(OCN_BLOCK), 0, (hipStream_t)stream, gcost, per, perm);
  return launch_status();
}

int ocn_cn_gather(const int64_t* rowptrA, const int32_t* colA, const int64_t* src,
                  const int64_t* dst, const int64_t* order, int64_t B, const int64_t* off, const uint8_t* flags,
                  const int32_t* wc, const float* weights, const float* h, int32_t H,
                  int64_t max_row_len, float* xcn1, float* xcn2, float* xij, const int64_t* out_row,
                  const int32_t* cnt1, const int32_t* cnt2, const uint64_t* rec, const int32_t* perm, const float* rowsum,
                  void* stream) {
  if (B < 0 || H <= 0) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!rowptrA || !src || !dst || !off || !weights || !h || !xcn1 || !xcn2 || !xij) return OCN_EINVAL;
  if (rowsum && (wc || !cnt2)) return OCN_EINVAL;          // the shortcut is the pattern route's, and needs the per-row counts
  hipStream_t st = (hipStream_t)stream;
  switch (H) {
    case 16:  LAUNCH_GATHER(4, 1); break;
    case 32:  LAUNCH_GATHER(8, 1); break;
    case 64:  LAUNCH_GATHER(16, 1); break;
    case 128: LAUNCH_GATHER(32, 1); break;
    case 256: LAUNCH_GATHER(OCN_X_POOL_LPE, (64 / OCN_X_POOL_LPE)); break;
    case 512: LAUNCH_GATHER(64, 2); break;
    default:   /* generic widths: every row by one wave, no long-row split */
      hipLaunchKernelGGL(cn_gather_generic, dim3((unsigned)((B + OCN_WPB - 1) / OCN_WPB)),
                         dim3(OCN_BLOCK), 0, st, GATHER_ARGS);
  }
  return launch_status();
}

#define LAUNCH_GATHER3(LPE, NV)                                                                      \
  hipLaunchKernelGGL((cn_gather3_kernel<LPE, NV>),                                                   \
                     dim3((unsigned)((B + (i64)OCN_WPB * (OCN_WAVE / (LPE)) - 1) / ((i64)OCN_WPB * (OCN_WAVE / (LPE))))), \
                     dim3(OCN_BLOCK), 0, (hipStream_t)stream, (const i64*)rowptrA, colA, (const i64*)src,      \
                     (const i64*)dst, (const i64*)order, (i64)B, (const i64*)off, flagsA, flagsB,             \
                     (const float4*)weightsA, (const float4*)weightsB, nip, h, (int)H, xcn1, xcn2, xcn3, xij)

int ocn_cn_gather3(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                   const int64_t* order, int64_t B, const int64_t* off, const uint8_t* flagsA,
                   const uint8_t* flagsB, const float* weightsA, const float* weightsB, const float* nip,
                   const float* h, int32_t H, float* xcn1, float* xcn2, float* xcn3, float* xij, void* stream) {
  if (B < 0 || H <= 0) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!rowptrA || !src || !dst || !off || !weightsA || !weightsB || !nip || !h || !xcn1 || !xcn2 || !xcn3 || !xij)
    return OCN_EINVAL;
  switch (H) {
    case 16:  LAUNCH_GATHER3(4, 1); break;
    case 32:  LAUNCH_GATHER3(8, 1); break;
    case 64:  LAUNCH_GATHER3(16, 1); break;
    case 128: LAUNCH_GATHER3(32, 1); break;
    case 256: LAUNCH_GATHER3(64, 1); break;
    case 512: LAUNCH_GATHER3(64, 2); break;
    default: return OCN_EINVAL;
  }
  return launch_status();
}

#define SCATTER_ARGS (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst, (const i64*)order, (i64)B, \
                     (const i64*)off, flags, wc, (const float4*)weights, h, (int)H, g1, g2, g3, dh
#define LAUNCH_SCATTER(LPE, NV)                                                                     \
  do {                                                                                              \
    const i64 epb = (i64)OCN_WPB * (OCN_WAVE / (LPE));                                              \
    hipLaunchKernelGGL((cn_scatter_kernel<LPE, NV>), dim3((unsigned)((B + epb - 1) / epb)),         \
                       dim3(OCN_BLOCK), 0, (hipStream_t)stream, SCATTER_ARGS);                      \
  } while (0)

int ocn_cn_gather_backward(const int64_t* rowptrA, const int32_t* colA, const int64_t* src,
                           const int64_t* dst, const int64_t* order, int64_t B, const int64_t* off,
                           const uint8_t* flags, const int32_t* wc, const float* weights,
                           const float* h, int32_t H, const float* g1, const float* g2,
                           const float* g3, float* dh, void* stream) {
  if (B < 0 || H <= 0) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!rowptrA || !src || !dst || !off || !weights || !h || !g1 || !g2 || !g3 || !dh) return OCN_EINVAL;
  switch (H) {
    case 16:  LAUNCH_SCATTER(4, 1); break;
    case 32:  LAUNCH_SCATTER(8, 1); break;
    case 64:  LAUNCH_SCATTER(16, 1); break;
    case 128: LAUNCH_SCATTER(32, 1); break;
    case 256: LAUNCH_SCATTER(64, 1); break;
    case 512: LAUNCH_SCATTER(64, 2); break;
    default: return OCN_EINVAL;
  }
  return launch_status();
}

#define LAUNCH_SCATTER3(LPE, NV)                                                                    \
  do {                                                                                              \
    const i64 epb = (i64)OCN_WPB * (OCN_WAVE / (LPE));                                              \
    hipLaunchKernelGGL((cn_scatter3_kernel<LPE, NV>), dim3((unsigned)((B + epb - 1) / epb)),        \
                       dim3(OCN_BLOCK), 0, (hipStream_t)stream, (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst, \
                       (const i64*)order, (i64)B, (const i64*)off, flagsA, flagsB, (const float4*)weightsA,                 \
                       (const float4*)weightsB, nip, h, (int)H, g1, g2, g3, g4, dh);                                     \
  } while (0)

int ocn_cn_gather3_backward(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                            const int64_t* order, int64_t B, const int64_t* off, const uint8_t* flagsA,
                            const uint8_t* flagsB, const float* weightsA, const float* weightsB, const float* nip,
                            const float* h, int32_t H, const float* g1, const float* g2, const float* g3,
                            const float* g4, float* dh, void* stream) {
  if (B < 0 || H <= 0) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!rowptrA || !src || !dst || !off || !flagsA || !flagsB || !weightsA || !weightsB || !nip || !h || !g1 || !g2 || !g3 || !g4 || !dh)
    return OCN_EINVAL;
  switch (H) {
    case 16:  LAUNCH_SCATTER3(4, 1); break;
    case 32:  LAUNCH_SCATTER3(8, 1); break;
    case 64:  LAUNCH_SCATTER3(16, 1); break;
    case 128: LAUNCH_SCATTER3(32, 1); break;
    case 256: LAUNCH_SCATTER3(64, 1); break;
    case 512: LAUNCH_SCATTER3(64, 2); break;
    default: return OCN_EINVAL;
  }
  return launch_status();
}

}  // extern "C"
