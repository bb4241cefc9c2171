// Walk-count route for candidates that SHARE their source node (the MRR test layout of the citation2 driver scores
// 1000 negatives per source: NeighborOverlapCitation2.py:248-254).
//
// cn2[e, k] = |N(k) ∩ N(j_e)| for k in N(i): for a fixed source i the rows N(k), k in N(i), that a candidate sweeps are
// the same for every candidate (i, j_c).  cn_walk_group_kernel sweeps them ONCE per group of up to 64 candidates with
// the same source: the neighbour lists of the group's targets j_c go into one hash table in LDS that maps a node m to
// the 64-bit mask of the targets it neighbours, and every swept element m of a row N(k) bumps the counters of the
// targets in its mask.  cn1 comes out of the same table (the mask of k itself).  A Bloom bitmap in front of the table
// keeps the common case (m neighbours none of the targets) at one LDS read.
//
// ocn_walk_prep replaces the launch chain in front of the walk kernels for small batches (B <= 4096: the drivers
// use 2048): ONE single-workgroup launch sorts the candidates by source (LDS bitonic sort), scans the flag offsets,
// forms the groups and decides per group whether it goes to the shared sweep (>= 2 members, target rows that fit
// the table) or to the per-candidate two-sided kernels of cn_stage.hip, scans the three work-item lists and clears
// the per-candidate counters.
#include "common.h"

#define WG_MAX_B 4096                 /* ocn_walk_prep: candidates per batch */
#define WG_THREADS 1024               /* prep kernel */
#define WG_GROUP 64                   /* candidates per group (one bit of the mask each) */
#define WG_ROWS 64                    /* neighbours of the source per work item */
#define WG_SLOTS 8192                 /* hash table slots */
#define WG_KEYCAP 4096                /* sum of deg(target) a group may bring: load factor <= 1/2 */
#define WG_BM_BITS 17
#define WG_BM_WORDS (1 << (WG_BM_BITS - 5))
#define WG_SWEEP_THREADS 512
#define WG_CPI_MAX 8                  /* chunks of WG_ROWS rows per work item (the table is built once per item): chosen per batch */

// ---------------------------------------------------------------------------------------------
// block-wide helpers for the single-workgroup prep kernel (1024 threads)
// ---------------------------------------------------------------------------------------------
// exclusive scan of v[0..n) (LDS, n <= IPT * WG_THREADS) in place; returns the total.  `tmp` = 32 i64 of LDS.
template <int IPT, typename T>
__device__ __forceinline__ i64 wgp_scan(T* v, int n, i64* tmp) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  i64 x[IPT], s = 0;
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const int idx = threadIdx.x * IPT + q;
    x[q] = idx < n ? (i64)v[idx] : 0;
    s += x[q];
  }
  const i64 inc = wave_incl_scan(s, lane);
  if (lane == 63) tmp[w] = inc;
  __syncthreads();
  i64 base = 0, tot = 0;
#pragma unroll
  for (int q = 0; q < WG_THREADS / 64; ++q) {
    const i64 t = tmp[q];
    if (q < w) base += t;
    tot += t;
  }
  i64 ex = base + inc - s;
#pragma unroll
  for (int q = 0; q < IPT; ++q) {
    const int idx = threadIdx.x * IPT + q;
    if (idx < n) v[idx] = (T)ex;
    ex += x[q];
  }
  __syncthreads();
  return tot;
}

#define WG_TAB (2 * WG_MAX_B)         /* source hash table of the prep kernel */
#define WG_SHARE_OVERHEAD 32768        /* fixed cost of a shared group (table build, counters), in swept elements */
#define WG_OWN_OVERHEAD 16384          /* fixed cost of a candidate's own sweep (its work items' set-up), likewise */
#define WG_BIG 512                    /* a target with more neighbours than this always keeps its own sweep */

__global__ __launch_bounds__(WG_THREADS) void walk_prep_kernel(
    const i64* __restrict__ rowptrA, const i64* __restrict__ nds, const i64* __restrict__ src, const i64* __restrict__ dst,
    int B, int min_share, i64* __restrict__ order, i64* __restrict__ off, i64* __restrict__ chunk_off,
    i64* __restrict__ rev_off, int32_t* __restrict__ g_head, i64* __restrict__ g_item_off, int32_t* __restrict__ g_active,
    int32_t* __restrict__ meta, int32_t* __restrict__ cnt1, int32_t* __restrict__ cnt2, int32_t* __restrict__ status,
    int32_t* __restrict__ scal) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  i64* s_a = reinterpret_cast<i64*>(wg_smem);                                   // [WG_MAX_B] scan scratch
  int32_t* s_tkey = reinterpret_cast<int32_t*>(wg_smem + 8 * WG_MAX_B);         // [WG_TAB] source of a table slot
  int32_t* s_tcnt = s_tkey + WG_TAB;                                            // [WG_TAB] candidates of that source -> first slot
  int32_t* s_ord = s_tcnt + WG_TAB;                                             // [WG_MAX_B] processing slot -> batch row
  int32_t* s_pos = s_ord + WG_MAX_B;                                            // [WG_MAX_B] batch row -> position among its source's
  int32_t* s_gid = s_pos + WG_MAX_B;                                            // [WG_MAX_B] group of a slot
  int32_t* s_act = s_tkey;                       // [WG_MAX_B] slot goes to the shared sweep (the table is done with by then)
  int32_t* s_gel = s_pos;                        // [WG_MAX_B] group is shared (positions are done with by then)
  i64* s_tmp = reinterpret_cast<i64*>(s_gid + WG_MAX_B);                        // [32]
  const int t = threadIdx.x;
  for (int q = t; q < WG_TAB; q += WG_THREADS) { s_tkey[q] = -1; s_tcnt[q] = 0; }
  for (int e = t; e < B; e += WG_THREADS) {
    s_a[e] = rowptrA[src[e] + 1] - rowptrA[src[e]];
    cnt1[e] = 0;
    cnt2[e] = 0;
  }
  if (t < 4) { if (t < 3) status[t] = 0; scal[t] = 0; }      // (status[3] is the caller's sticky error word)
  __syncthreads();
  // flag offsets: exclusive scan of deg(src) in BATCH order
  const i64 total = wgp_scan<WG_MAX_B / WG_THREADS>(s_a, B, s_tmp);
  for (int e = t; e < B; e += WG_THREADS) off[e] = s_a[e];
  if (t == 0) off[B] = total;
  // processing order: candidates grouped by source (hash grouping: which candidate of a source comes first is
  // whatever the atomics decide — the order only fixes bit positions and locality, never a result)
  for (int e = t; e < B; e += WG_THREADS) {
    const int32_t v = (int32_t)src[e];
    unsigned h = ((unsigned)v * 2654435761u) >> (32 - 13);
    for (int probe = 0; probe < WG_TAB; ++probe) {
      const int32_t prev = atomicCAS(&s_tkey[h], -1, v);
      if (prev == -1 || prev == v) break;
      h = (h + 1) & (WG_TAB - 1);
    }
    s_gid[e] = (int)h;                                        // (table slot of the batch row, until the groups are formed)
    s_pos[e] = atomicAdd(&s_tcnt[h], 1);
  }
  __syncthreads();
  wgp_scan<WG_TAB / WG_THREADS>(s_tcnt, WG_TAB, s_tmp);       // s_tcnt[h] = first processing slot of the source in table slot h
  for (int e = t; e < B; e += WG_THREADS) {
    const int s = s_tcnt[s_gid[e]] + s_pos[e];
    s_ord[s] = e;
    order[s] = e;
  }
  __syncthreads();
  // groups: a source's candidates cut into pieces of WG_GROUP by their position in the run (all in parallel).  Members
  // whose target has at most WG_BIG neighbours go to the shared sweep if there are at least min_share of them and their
  // targets' rows fit the sweep's table (sum of degrees <= WG_KEYCAP); everybody else keeps a sweep of their own.
  int32_t* s_dj = s_tcnt;                        // [WG_MAX_B] degree of the slot's target (the run starts are done with)
  int32_t* s_gcnt = s_tcnt + WG_MAX_B;           // [WG_MAX_B] small members of a group
  int32_t* s_gsum = s_tkey + WG_MAX_B;           // [WG_MAX_B] sum of their degrees; later: group is shared
  for (int s = t; s < B; s += WG_THREADS) s_a[s] = (s_pos[s_ord[s]] % WG_GROUP == 0) ? 1 : 0;
  __syncthreads();
  const i64 ng = wgp_scan<WG_MAX_B / WG_THREADS>(s_a, B, s_tmp);        // s_a[s] = number of heads before s
  for (int s = t; s < B; s += WG_THREADS) {
    const bool head = (s + 1 < B ? s_a[s + 1] : ng) != s_a[s];
    const int g = (int)s_a[s] - (head ? 0 : 1);
    s_gid[s] = g;
    if (head) g_head[g] = s;
    const i64 j = dst[s_ord[s]];
    s_dj[s] = (int)min((i64)(WG_KEYCAP + 1), rowptrA[j + 1] - rowptrA[j]);
    s_gcnt[s] = 0;
    s_gsum[s] = 0;
  }
  if (t == 0) g_head[ng] = B;
  __syncthreads();
  // ... and the sweep shared by the group (the rows of N(i) once: nds[i] elements, plus building the table) must be
  // cheaper than its members' own sweeps, each from its cheaper endpoint: random low-degree targets are swept from
  // the target's side for next to nothing, and a group of them gains little from sharing.
  int32_t* s_gcost = reinterpret_cast<int32_t*>(s_a);                  // [WG_MAX_B] members' own cost, in units of 64 elements
  for (int g = t; g < (int)ng; g += WG_THREADS) s_gcost[g] = 0;
  __syncthreads();
  for (int s = t; s < B; s += WG_THREADS)
    if (s_dj[s] <= WG_BIG) {
      atomicAdd(&s_gsum[s_gid[s]], s_dj[s]);
      atomicAdd(&s_gcnt[s_gid[s]], 1);
      if (nds) {
        const i64 e = s_ord[s];
        const i64 i = src[e], j = dst[e];
        const i64 di = rowptrA[i + 1] - rowptrA[i], dj = s_dj[s];
        const i64 own = walk_reverse(nds, i, j, di, dj) ? 2 * nds[j] + di * ((dj + WALK_REV_CHUNK - 1) / WALK_REV_CHUNK) + 2 * di : nds[i];
        atomicAdd(&s_gcost[s_gid[s]], (int)min((i64)(1 << 24), ((own + WG_OWN_OVERHEAD) >> 6) + 1));
      }
    }
  __syncthreads();
  for (int g = t; g < (int)ng; g += WG_THREADS) {
    bool ok = s_gcnt[g] >= min_share && s_gsum[g] <= WG_KEYCAP;
    if (ok && nds) {
      const i64 i = src[s_ord[g_head[g]]];
      ok = ((nds[i] + WG_SHARE_OVERHEAD) >> 6) < (i64)s_gcost[g];
    }
    s_gel[g] = ok ? 1 : 0;
  }
  __syncthreads();
  for (int s = t; s < B; s += WG_THREADS) s_act[s] = (s_dj[s] <= WG_BIG && s_gel[s_gid[s]]) ? 1 : 0;
  __syncthreads();
  for (int s = t; s < B; s += WG_THREADS) g_active[s] = s_act[s];
  // the per-candidate kernels' work items (0 for the candidates of the shared sweep) ...
  for (int s = t; s < B; s += WG_THREADS) {
    const i64 i = src[s_ord[s]];
    const i64 di = rowptrA[i + 1] - rowptrA[i];
    const i64 chunks = (di + WALK_CHUNK - 1) / WALK_CHUNK;
    const i64 cg = walk_group(nds, i, di);
    s_a[s] = s_act[s] ? 0 : (chunks + cg - 1) / cg;
  }
  __syncthreads();
  i64 tot = wgp_scan<WG_MAX_B / WG_THREADS>(s_a, B, s_tmp);
  for (int s = t; s < B; s += WG_THREADS) chunk_off[s] = s_a[s];
  if (t == 0) chunk_off[B] = tot;
  __syncthreads();
  for (int s = t; s < B; s += WG_THREADS) {
    const i64 e = s_ord[s];
    const i64 i = src[e], j = dst[e];
    const i64 di = rowptrA[i + 1] - rowptrA[i], dj = rowptrA[j + 1] - rowptrA[j];
    s_a[s] = (!s_act[s] && walk_reverse(nds, i, j, di, dj)) ? (dj + WALK_REV_CHUNK - 1) / WALK_REV_CHUNK : 0;
  }
  __syncthreads();
  tot = wgp_scan<WG_MAX_B / WG_THREADS>(s_a, B, s_tmp);
  if (rev_off) {
    for (int s = t; s < B; s += WG_THREADS) rev_off[s] = s_a[s];
    if (t == 0) rev_off[B] = tot;
  }
  __syncthreads();
  // ... and the shared sweep's: one item per WG_CPI chunks of WG_ROWS neighbours of the source, per group with members in it
  // (chunks per item: as many as keep about three items per CU in the batch — every item builds the group's table anew)
  for (int g = t; g < (int)ng; g += WG_THREADS) {
    i64 v = 0;
    if (s_gel[g]) {
      const i64 i = src[s_ord[g_head[g]]];
      v = (rowptrA[i + 1] - rowptrA[i] + WG_ROWS - 1) / WG_ROWS;
    }
    s_a[g] = v;
  }
  __syncthreads();
  const i64 all_chunks = wgp_scan<WG_MAX_B / WG_THREADS>(s_a, (int)ng, s_tmp);
  int cpi = (int)(all_chunks / 768);
  cpi = cpi < 1 ? 1 : (cpi > WG_CPI_MAX ? WG_CPI_MAX : cpi);
  for (int g = t; g < (int)ng; g += WG_THREADS) {
    const i64 chunks = (g + 1 < (int)ng ? s_a[g + 1] : all_chunks) - s_a[g];
    s_gid[g] = (int)((chunks + cpi - 1) / cpi);               // (group ids are done with)
  }
  __syncthreads();
  for (int g = t; g < (int)ng; g += WG_THREADS) s_a[g] = s_gid[g];
  __syncthreads();
  tot = wgp_scan<WG_MAX_B / WG_THREADS>(s_a, (int)ng, s_tmp);
  for (int g = t; g < (int)ng; g += WG_THREADS) g_item_off[g] = s_a[g];
  if (t == 0) { g_item_off[ng] = tot; meta[0] = (int)ng; meta[1] = 0; meta[2] = cpi; meta[3] = 0; }
}

// ---------------------------------------------------------------------------------------------
// the shared sweep
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned wg_hash(int32_t v) { return ((unsigned)v * 2654435761u) >> (32 - 13); }
__device__ __forceinline__ unsigned wg_bit(int32_t v) { return ((unsigned)v * 2246822519u) >> (32 - WG_BM_BITS); }

// mask of the group's targets that neighbour m (0 if none)
__device__ __forceinline__ u64 wg_lookup(const int32_t* keys, const u64* masks, int32_t m) {
  unsigned h = wg_hash(m);
  for (int probe = 0; probe < WG_SLOTS; ++probe) {
    const int32_t k = keys[h];
    if (k == m) return masks[h];
    if (k < 0) return 0ull;
    h = (h + 1) & (WG_SLOTS - 1);
  }
  return 0ull;
}

__global__ __launch_bounds__(WG_SWEEP_THREADS) void cn_walk_group_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA, const i64* __restrict__ src,
    const i64* __restrict__ dst, const i64* __restrict__ order, int B, const int32_t* __restrict__ g_head,
    const i64* __restrict__ g_item_off, const int32_t* __restrict__ g_active, int32_t* __restrict__ meta,
    const i64* __restrict__ off,
    uint8_t* __restrict__ flags, int32_t* __restrict__ wc, i64 cap, u64* __restrict__ hist,
    int32_t* __restrict__ cnt1, int32_t* __restrict__ cnt2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
  u64* s_masks = reinterpret_cast<u64*>(wg_smem);                                   // [WG_SLOTS]
  i64* s_e = reinterpret_cast<i64*>(wg_smem + 8 * WG_SLOTS);                        // [WG_GROUP]
  i64* s_base = s_e + WG_GROUP;                                                     // [WG_GROUP]
  i64* s_r0 = s_base + WG_GROUP;                                                    // [WG_ROWS]
  u64* s_kmask = reinterpret_cast<u64*>(s_r0 + WG_ROWS);                            // [WG_ROWS]
  int32_t* s_keys = reinterpret_cast<int32_t*>(s_kmask + WG_ROWS);                  // [WG_SLOTS]
  unsigned* s_bm = reinterpret_cast<unsigned*>(s_keys + WG_SLOTS);                  // [WG_BM_WORDS]
  int* s_cnt = reinterpret_cast<int*>(s_bm + WG_BM_WORDS);                          // walks[candidate][row]
  int32_t* s_j = s_cnt + WG_GROUP * WG_ROWS;                                        // [WG_GROUP]
  int32_t* s_r = s_j + WG_GROUP;                                                    // [WG_ROWS]
  int* s_pre = s_r + WG_ROWS;                                                       // [WG_ROWS + 1]
  __shared__ i64 s_item;
  __shared__ int s_g;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ng = meta[0];
  const i64 n_items = g_item_off[ng];
  for (;;) {
    // draw an item and find its group (largest g with g_item_off[g] <= item); meanwhile clear the tables
    if (w == 0) {
      i64 it = 0;
      if (lane == 0) it = (i64)atomicAdd(meta + 1, 1);
      it = __shfl(it, 0, 64);
      int lo = 0, hi = ng;                       // g_item_off[lo] <= it < g_item_off[hi]
      if (it < n_items) {
        while (hi - lo > 1) {
          const int step = (hi - lo + 63) / 64;
          const int idx = lo + (lane + 1) * step;
          const bool le = idx < hi && g_item_off[idx] <= it;
          const int c = __popcll(__ballot(le));
          lo += c * step;
          hi = lo + step < hi ? lo + step : hi;
        }
      }
      if (lane == 0) { s_item = it; s_g = lo; }
    } else {
      for (int q = threadIdx.x - 64; q < WG_BM_WORDS; q += WG_SWEEP_THREADS - 64) s_bm[q] = 0u;
      for (int q = threadIdx.x - 64; q < WG_SLOTS; q += WG_SWEEP_THREADS - 64) { s_keys[q] = -1; s_masks[q] = 0ull; }
    }
    __syncthreads();
    const i64 item = s_item;
    if (item >= n_items) break;                  // every wave gets here: the grid drains
    const int g = s_g;
    const int head = g_head[g];
    int gs = g_head[g + 1] - head;
    gs = gs > WG_GROUP ? WG_GROUP : gs;
    if (threadIdx.x < WG_GROUP) {
      const int c = threadIdx.x;
      const bool on = c < gs && g_active[head + c] != 0;       // (the other members keep their own sweeps)
      const i64 e = c < gs ? order[head + c] : 0;
      s_e[c] = e;
      s_j[c] = on ? (int32_t)dst[e] : -1;
      s_base[c] = on ? off[e] : 0;
    }
    __syncthreads();
    const i64 i = src[s_e[0]];
    const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0;
    // the targets' neighbour lists -> Bloom bitmap + hash table (node -> mask of targets): once per item.  The lists are
    // FLATTENED (one round of loads for all of them: target by target it was a chain of dependent loads per target)
    if (w == 0) {
      const i64 j = s_j[lane];
      i64 b0 = 0, db = 0;
      if (j >= 0) { b0 = rowptrA[j]; db = rowptrA[j + 1] - b0; }
      const i64 incl = wave_incl_scan(db, lane);
      s_r0[lane] = b0;
      s_pre[lane] = (int)(incl - db);
      if (lane == 63) s_pre[WG_ROWS] = (int)incl;
    }
    __syncthreads();
    {
      const int nkeys = s_pre[WG_ROWS];
      for (int x = threadIdx.x; x < nkeys; x += WG_SWEEP_THREADS) {
        int lo = 0, hi = WG_GROUP;                 // target c of element x: last c with s_pre[c] <= x
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_pre[mid] <= x) lo = mid; else hi = mid; }
        const int c = lo;
        const int32_t m = colA[s_r0[c] + (x - s_pre[c])];
        const unsigned bb = wg_bit(m);
        atomicOr(&s_bm[bb >> 5], 1u << (bb & 31));
        unsigned h = wg_hash(m);
        for (int probe = 0; probe < WG_SLOTS; ++probe) {
          const int32_t prev = atomicCAS(&s_keys[h], -1, m);
          if (prev == -1 || prev == m) { atomicOr(&s_masks[h], 1ull << c); break; }
          h = (h + 1) & (WG_SLOTS - 1);
        }
      }
    }
    u64 on_mask = 0ull;                          // members of the group that take part
    for (int c = 0; c < gs; ++c) on_mask |= (u64)(s_j[c] >= 0) << c;
    const int cpi = meta[2];
    const i64 chunk0 = (item - g_item_off[g]) * cpi;
    for (int ch = 0; ch < cpi; ++ch) {
      const i64 p_lo = (chunk0 + ch) * WG_ROWS;
      if (p_lo >= da) break;                     // workgroup-uniform
      const int nk = (int)((p_lo + WG_ROWS < da ? p_lo + WG_ROWS : da) - p_lo);
      __syncthreads();                           // table complete (first chunk) / previous chunk's counters read
      for (int q = threadIdx.x; q < WG_GROUP * WG_ROWS; q += WG_SWEEP_THREADS) s_cnt[q] = 0;
      // the chunk's rows N(k), flattened
      if (w == 0) {
        int32_t k = 0;
        i64 r0 = 0, dr = 0;
        if (lane < nk) { k = colA[a0 + p_lo + lane]; r0 = rowptrA[k]; dr = rowptrA[k + 1] - r0; }
        const i64 incl = wave_incl_scan(dr, lane);
        s_r[lane] = k; s_r0[lane] = r0;
        s_pre[lane] = lane < nk ? (int)(incl - dr) : 0x7fffffff;
        if (lane == 63) s_pre[WG_ROWS] = 0x7fffffff;
        if (lane == nk - 1) s_item = incl;       // (s_item is free again: elements of the chunk)
        if (lane < nk) s_kmask[lane] = wg_lookup(s_keys, s_masks, k) & on_mask;      // cn1: is k itself a neighbour of target c?
      }
      __syncthreads();
      const int total = (int)s_item;
      {
        constexpr int WU = 8;
        int lo = 0;
        for (int x0 = 0; x0 < total; x0 += WU * WG_SWEEP_THREADS) {
          int row[WU];
          int32_t m[WU];
#pragma unroll
          for (int u = 0; u < WU; ++u) {
            const int x = x0 + u * WG_SWEEP_THREADS + threadIdx.x;
            m[u] = -1;
            if (x < total) {
              while (s_pre[lo + 1] <= x) ++lo;
              m[u] = colA[s_r0[lo] + (x - s_pre[lo])];
            }
            row[u] = lo;
          }
#pragma unroll
          for (int u = 0; u < WU; ++u) {
            if (m[u] < 0) continue;
            const unsigned bb = wg_bit(m[u]);
            if (!((s_bm[bb >> 5] >> (bb & 31)) & 1u)) continue;
            u64 mask = wg_lookup(s_keys, s_masks, m[u]);
            while (mask) {
              const int c = __ffsll((long long)mask) - 1;
              mask &= mask - 1;
              atomicAdd(&s_cnt[c * WG_ROWS + row[u]], 1);
            }
          }
        }
      }
      __syncthreads();
      // finalise: flags / walk counts of every (target, row); one histogram atomic per row; per-target counts
      for (int idx = threadIdx.x; idx < WG_GROUP * WG_ROWS; idx += WG_SWEEP_THREADS) {
        const int c = idx / WG_ROWS, tt = idx % WG_ROWS;
        if (tt < nk && ((on_mask >> c) & 1ull)) {
          const i64 base = s_base[c];
          if (base + da <= cap) {
            const int walks = s_cnt[idx];
            const unsigned f1 = (unsigned)((s_kmask[tt] >> c) & 1ull);
            flags[base + p_lo + tt] = (uint8_t)(f1 * OCN_F_CN1 | (walks > 0 ? OCN_F_CN2 : 0u));
            wc[base + p_lo + tt] = walks;
          }
        }
      }
      if (threadIdx.x < WG_ROWS && threadIdx.x < nk) {
        const int tt = threadIdx.x;
        const u64 km = s_kmask[tt];
        int n2 = 0, nu = 0;
        i64 ws = 0;
        for (int c = 0; c < gs; ++c) {
          const int walks = s_cnt[c * WG_ROWS + tt];
          n2 += walks > 0;
          nu += (walks > 0) | (int)((km >> c) & 1ull);
          ws += walks;
        }
        const int n1 = __popcll(km);
        if (nu) {
          const i64 k = s_r[tt];
          atomicAdd(hist + 2 * k, (u64)n1 | ((u64)n2 << HF_BITS) | ((u64)nu << (2 * HF_BITS)));
          if (ws) atomicAdd(hist + 2 * k + 1, (u64)ws);
        }
      } else if (threadIdx.x >= 64 && threadIdx.x < 64 + WG_GROUP && ((on_mask >> (threadIdx.x - 64)) & 1ull)) {
        const int c = threadIdx.x - 64;
        int c1 = 0, c2 = 0;
        for (int tt = 0; tt < nk; ++tt) {
          c1 += (int)((s_kmask[tt] >> c) & 1ull);
          c2 += s_cnt[c * WG_ROWS + tt] > 0;
        }
        if (c1) atomicAdd(cnt1 + s_e[c], c1);
        if (c2) atomicAdd(cnt2 + s_e[c], c2);
      }
    }
    __syncthreads();
  }
}

#define WG_PREP_LDS (8 * WG_MAX_B + 2 * 4 * WG_TAB + 3 * 4 * WG_MAX_B + 32 * 8)
#define WG_SWEEP_LDS (8 * WG_SLOTS + 8 * (2 * WG_GROUP + 2 * WG_ROWS) + 4 * WG_SLOTS + 4 * WG_BM_WORDS + 4 * WG_GROUP * WG_ROWS + \
                      4 * (WG_GROUP + WG_ROWS + WG_ROWS + 1) + 16)

static int wg_raise(const void* fn, int bytes) {
  // (per call: the attribute is per device and cheap to set; no process-wide flag to go stale on a second GPU)
  return (int)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

extern "C" {

int32_t ocn_walk_prep_max_batch(void) { return WG_MAX_B; }

int ocn_walk_prep(const int64_t* rowptrA, const int64_t* nds, const int64_t* src, const int64_t* dst, int64_t B,
                  int32_t min_share, int64_t* order, int64_t* off, int64_t* chunk_off, int64_t* rev_off, int32_t* g_head,
                  int64_t* g_item_off, int32_t* g_active, int32_t* meta, int32_t* cnt1, int32_t* cnt2, int32_t* status,
                  int32_t* scal, void* stream) {
  if (B < 0 || B > WG_MAX_B) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!rowptrA || !src || !dst || !order || !off || !chunk_off || !g_head || !g_item_off || !g_active || !meta || !cnt1 ||
      !cnt2 || !status || !scal)
    return OCN_EINVAL;
  if ((nds == nullptr) != (rev_off == nullptr)) return OCN_EINVAL;
  if (int rc = wg_raise((const void*)walk_prep_kernel, WG_PREP_LDS)) return rc;
  hipLaunchKernelGGL(walk_prep_kernel, dim3(1), dim3(WG_THREADS), WG_PREP_LDS, (hipStream_t)stream, (const i64*)rowptrA,
                     (const i64*)nds, (const i64*)src, (const i64*)dst, (int)B, (int)min_share, (i64*)order, (i64*)off,
                     (i64*)chunk_off, (i64*)rev_off, g_head, (i64*)g_item_off, g_active, meta, cnt1, cnt2, status, scal);
  return launch_status();
}

int ocn_cn_walk_group(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                      const int64_t* order, int64_t B, const int32_t* g_head, const int64_t* g_item_off,
                      const int32_t* g_active, int32_t* meta, const int64_t* off, uint8_t* flags, int32_t* wc,
                      int64_t flags_cap, uint64_t* hist, int32_t* cnt1, int32_t* cnt2, void* stream) {
  if (B < 0 || B > WG_MAX_B || flags_cap < 0) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!rowptrA || !src || !dst || !order || !g_head || !g_item_off || !g_active || !meta || !off || !hist || !cnt1 || !cnt2)
    return OCN_EINVAL;
  if (flags_cap > 0 && (!flags || !wc)) return OCN_EINVAL;
  if (int rc = wg_raise((const void*)cn_walk_group_kernel, WG_SWEEP_LDS)) return rc;
  hipLaunchKernelGGL(cn_walk_group_kernel, dim3(256), dim3(WG_SWEEP_THREADS), WG_SWEEP_LDS, (hipStream_t)stream,
                     (const i64*)rowptrA, colA, (const i64*)src, (const i64*)dst, (const i64*)order, (int)B, g_head,
                     (const i64*)g_item_off, g_active, meta, (const i64*)off, flags, wc, (i64)flags_cap, (u64*)hist, cnt1, cnt2);
  return launch_status();
}

}  // extern "C"
