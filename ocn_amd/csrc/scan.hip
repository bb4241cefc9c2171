// Exclusive scans: batch-row offsets (deg of src per edge) and int32 counts -> int64 offsets.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// scans (edge offsets, row offsets)
// ---------------------------------------------------------------------------------------------
#define SCAN_IPT 8
#define SCAN_TILE (OCN_BLOCK * SCAN_IPT)

struct DegOfSrc {
  const i64* rowptr;
  const i64* src;
  __device__ __forceinline__ i64 operator()(i64 e) const {
    i64 i = src[e];
    return rowptr[i + 1] - rowptr[i];
  }
};
struct ChunksOfSlot {      // forward work items of a batch row, in processing order: groups of walk_group() 64-row chunks
  const i64* rowptr;
  const i64* nds;
  const i64* src;
  const i64* order;
  __device__ __forceinline__ i64 operator()(i64 slot) const {
    const i64 i = src[order ? order[slot] : slot];
    const i64 di = rowptr[i + 1] - rowptr[i];
    const i64 chunks = (di + WALK_CHUNK - 1) / WALK_CHUNK;
    const i64 cg = walk_group(nds, i, di);
    return (chunks + cg - 1) / cg;
  }
};
struct RevChunksOfSlot {   // reverse-sweep work items of a batch row: ceil(deg(dst) / chunk) where walk_reverse(), else 0
  const i64* rowptr;
  const i64* nds;
  const i64* src;
  const i64* dst;
  const i64* order;
  __device__ __forceinline__ i64 operator()(i64 slot) const {
    const i64 e = order ? order[slot] : slot;
    const i64 i = src[e], j = dst[e];
    const i64 di = rowptr[i + 1] - rowptr[i], dj = rowptr[j + 1] - rowptr[j];
    return walk_reverse(nds, i, j, di, dj) ? (dj + WALK_REV_CHUNK - 1) / WALK_REV_CHUNK : 0;
  }
};
// class of a batch row for the heads: 3 = cn1 and cn2 entries, 2 = cn1 only, 1 = cn2 only, 0 = none; the
// scan carries the three non-zero classes' running counts in 21-bit fields of one word
#define CLS_BITS 21
#define CLS_MASK ((1ll << CLS_BITS) - 1)
struct ClassOfSlot {
  const int32_t* cnt1;
  const int32_t* cnt2;
  const i64* order;
  __device__ __forceinline__ int cls(i64 slot) const {
    const i64 e = order ? order[slot] : slot;
    return (cnt1[e] > 0 ? 2 : 0) | ((cnt2 && cnt2[e] > 0) ? 1 : 0);
  }
  __device__ __forceinline__ i64 operator()(i64 slot) const {
    const int c = cls(slot);
    return c == 3 ? 1ll : (c == 2 ? (1ll << CLS_BITS) : (c == 1 ? (1ll << (2 * CLS_BITS)) : 0ll));
  }
};

__global__ __launch_bounds__(OCN_BLOCK) void class_scatter(ClassOfSlot op, i64 B, const i64* __restrict__ prefix,
                                                           i64* __restrict__ order_out, i64* __restrict__ inv_out,
                                                           i64* __restrict__ ranges) {
  const i64 tot = prefix[B];
  const bool poisoned = tot < 0;             // the scan gave up (scan_chained): every row "has both kinds of entry", nothing is skipped
  const i64 n3 = poisoned ? B : (tot & CLS_MASK), n2 = poisoned ? 0 : ((tot >> CLS_BITS) & CLS_MASK),
            n1 = poisoned ? 0 : ((tot >> (2 * CLS_BITS)) & CLS_MASK);
  for (i64 slot = (i64)blockIdx.x * blockDim.x + threadIdx.x; slot < B; slot += (i64)gridDim.x * blockDim.x) {
    const i64 p = prefix[slot];
    const i64 b3 = p & CLS_MASK, b2 = (p >> CLS_BITS) & CLS_MASK, b1 = (p >> (2 * CLS_BITS)) & CLS_MASK;
    const int c = op.cls(slot);
    const i64 pos = poisoned ? slot : (c == 3 ? b3 : (c == 2 ? n3 + b2 : (c == 1 ? n3 + n2 + b1 : n3 + n2 + n1 + (slot - b3 - b2 - b1))));
    const i64 e = op.order ? op.order[slot] : slot;
    order_out[pos] = e;
    inv_out[e] = pos;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const i64 m3 = n3, m32 = n3 + n2, m321 = n3 + n2 + n1;
    const i64 r[OCN_CLASS_RANGES][2] = {{0, m32}, {0, m3}, {m32, m321}, {0, m321}, {m321, B}, {m3, m32}, {0, B}};
    for (int q = 0; q < OCN_CLASS_RANGES; ++q) { ranges[2 * q] = r[q][0]; ranges[2 * q + 1] = r[q][1]; }
  }
}

struct I32In {
  const int32_t* in;
  __device__ __forceinline__ i64 operator()(i64 e) const { return (i64)in[e]; }
};

// small inputs (a ppa / citation2 batch has 2048 rows): the whole scan in one workgroup, one launch
#define SCAN_SINGLE_MAX (8 * SCAN_TILE)
template <typename Op>
__global__ __launch_bounds__(OCN_BLOCK) void scan_single(Op op, i64 n, i64* out) {
  __shared__ i64 sh[2 * OCN_WPB];
  i64 carry = 0;
  for (i64 t0 = 0; t0 < n; t0 += SCAN_TILE) {
    const i64 base = t0 + (i64)threadIdx.x * SCAN_IPT;
    i64 v[SCAN_IPT];
    i64 s = 0;
#pragma unroll
    for (int t = 0; t < SCAN_IPT; ++t) {
      const i64 e = base + t;
      v[t] = e < n ? op(e) : 0;
      s += v[t];
    }
    i64 tot;
    i64 ex = carry + block_excl_scan(s, sh, &tot);
#pragma unroll
    for (int t = 0; t < SCAN_IPT; ++t) {
      const i64 e = base + t;
      if (e < n) out[e] = ex;
      ex += v[t];
    }
    carry += tot;
  }
  if (threadIdx.x == 0) out[n] = carry;
}

// several small scratch arrays zeroed by ONE launch (a candidate batch resets its histogram, counters and status
// words: as separate fills that was four launches of a few microseconds each)
#define ZERO_MAX_REGIONS 8
struct ZeroArgs { int n; uint32_t* p[ZERO_MAX_REGIONS]; i64 words[ZERO_MAX_REGIONS]; };
// (vb, vg): this workgroup's index among the vg workgroups that share the job
__device__ __forceinline__ void zero_body(const ZeroArgs& a, i64 vb, i64 vg) {
  i64 total = 0;
  for (int r = 0; r < a.n; ++r) total += (a.words[r] + 3) >> 2;          // in 16-byte quads
  for (i64 q = vb * blockDim.x + threadIdx.x; q < total; q += vg * blockDim.x) {
    i64 o = q;
    int r = 0;
    while (o >= ((a.words[r] + 3) >> 2)) { o -= (a.words[r] + 3) >> 2; ++r; }
    uint32_t* base = a.p[r] + 4 * o;
    const i64 left = a.words[r] - 4 * o;
    if (left >= 4 && ((uintptr_t)base & 15) == 0) *reinterpret_cast<uint4*>(base) = make_uint4(0u, 0u, 0u, 0u);
    else for (int k = 0; k < 4 && k < left; ++k) base[k] = 0u;
  }
}
__device__ __forceinline__ void order_count_body(const i64* __restrict__ node, i64 B, int32_t* __restrict__ counts, i64 vb, i64 vg) {
  for (i64 e = vb * blockDim.x + threadIdx.x; e < B; e += vg * blockDim.x) atomicAdd(counts + node[e], 1);
}
// Work that needs nothing from the scan, carried by extra workgroups of the scan's launch (blockIdx >= number of
// tiles): a batch's resets and the counting phase of its processing order cost no launch of their own.
struct NoExtra { __device__ __forceinline__ void operator()(i64, i64) const {} };
struct PrepExtra {
  ZeroArgs z;
  i64 zero_blocks;
  const i64* node; i64 B; int32_t* counts;       // counts == NULL: no processing order wanted
  __device__ __forceinline__ void operator()(i64 vb, i64 vg) const {
    if (vb < zero_blocks) zero_body(z, vb, zero_blocks);
    else if (counts) order_count_body(node, B, counts, vb - zero_blocks, vg - zero_blocks);
  }
};

// Large inputs: ONE launch, tiles chained through device memory.  A workgroup draws its tile from a ticket
// (so every predecessor tile has started, whatever the dispatch order), publishes its tile total as one 8-byte
// granule {ready bit | total} (relaxed agent-scope store: the granule carries its own tag, no fence), and wave 0
// sums the granules of ALL earlier tiles, 64 per round, polling the ones not yet published.  The last workgroup
// to finish clears the state, so the workspace is left as it was found: ZERO (ocn_scan_workspace_bytes; the
// caller zeroes it once, when it allocates it).  state[0] = ticket, state[1] = finished tiles, state[2 + t] = tile t.
//
// A workspace that was NOT zero on entry (the caller's contract broken: a workspace shared by two launches in flight, a
// buffer that was never cleared) cannot be scanned: tickets start beyond the launch or a granule never turns ready.
// The launch then neither hangs nor traps (a trap aborts the caller's process): the poll is bounded, a tile that gives
// up publishes SCAN_ERR so that its successors give up at once, writes ZERO offsets for its own items (every consumer
// indexes with them: zero and the good tiles' prefixes stay inside the true total) and the grand total out[n] becomes
// OCN_SCAN_POISON (-1).  Consumers test out[n] < 0: the intersection kernels raise bit OCN_ST_SCAN of their status words and
// leave, the processing / class orders fall back to batch order, the host raises where it reads a total.
#define SCAN_READY (1ull << 63)
#define SCAN_ERR (1ull << 62)
#define SCAN_SPIN_MAX (1 << 20)     /* polls of one predecessor granule (~0.5 us each) before the tile gives up */
template <typename Op, typename Extra = NoExtra>
__global__ __launch_bounds__(OCN_BLOCK) void scan_chained(Op op, i64 n, i64* __restrict__ out, u64* __restrict__ state,
                                                          i64 nt, const Extra extra = Extra()) {
  __shared__ i64 sh[2 * OCN_WPB];
  __shared__ i64 s_tile, s_prefix;
  if ((i64)blockIdx.x >= nt) {               // (uniform per workgroup; these draw no tile ticket)
    extra((i64)blockIdx.x - nt, (i64)gridDim.x - nt);
    return;
  }
  __shared__ int s_bad;
  if (threadIdx.x == 0) { s_tile = (i64)atomicAdd(&state[0], 1ull); s_bad = 0; }
  __syncthreads();
  const i64 t = s_tile;
  if (t < 0 || t >= nt) {                     // a ticket outside the launch: the workspace was not zero
    if (threadIdx.x == 0) out[n] = OCN_SCAN_POISON;
    for (int q = 0; q < SCAN_IPT; ++q) {      // (best effort: the tile this workgroup would have had in dispatch order reads as zeros)
      const i64 e = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_IPT + q;
      if (e < n) out[e] = 0;
    }
    return;
  }
  const i64 base = t * SCAN_TILE + (i64)threadIdx.x * SCAN_IPT;      // thread-contiguous items: prefix order = item order
  i64 v[SCAN_IPT];
  i64 s = 0;
#pragma unroll
  for (int q = 0; q < SCAN_IPT; ++q) {
    const i64 e = base + q;
    v[q] = e < n ? op(e) : 0;
    s += v[q];
  }
  i64 tot;
  i64 ex = block_excl_scan(s, sh, &tot);
  if (threadIdx.x == 0)
    __hip_atomic_store(&state[2 + t], SCAN_READY | (u64)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < OCN_WAVE) {
    i64 prefix = 0;
    bool bad = false;
    for (i64 p0 = 0; p0 < t && !bad; p0 += OCN_WAVE) {
      const i64 p = p0 + threadIdx.x;
      u64 w = SCAN_READY;
      if (p < t) {
        // Tickets are drawn in order, so every tile p < t belongs to a workgroup that is already running and publishes its
        // granule before it waits for anything itself (tile 0 waits for nobody, tile p only for tiles < p): with a ZERO
        // workspace (the caller's contract) the wait always ends, after microseconds.
        int spins = 0;
        do {
          w = __hip_atomic_load(&state[2 + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (!(w & (SCAN_READY | SCAN_ERR))) __builtin_amdgcn_s_sleep(1);
        } while (!(w & (SCAN_READY | SCAN_ERR)) && ++spins < SCAN_SPIN_MAX);
      }
      bad = __ballot(!(w & SCAN_READY) || (w & SCAN_ERR)) != 0ull;
      prefix += (i64)(w & ~(SCAN_READY | SCAN_ERR));
    }
    prefix = wave_sum(prefix);
    if (threadIdx.x == 0) {
      s_prefix = prefix;
      s_bad = bad;
      if (bad)                                 // successors still polling give up at once instead of timing out themselves
        __hip_atomic_store(&state[2 + t], SCAN_READY | SCAN_ERR, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  __syncthreads();
  const bool failed = s_bad != 0;
  ex += s_prefix;
#pragma unroll
  for (int q = 0; q < SCAN_IPT; ++q) {
    const i64 e = base + q;
    if (e < n) out[e] = failed ? 0 : ex;
    ex += v[q];
  }
  if (t == nt - 1 && threadIdx.x == 0) out[n] = failed ? OCN_SCAN_POISON : s_prefix + tot;
  // the last workgroup to get here has no reader left behind it: leave the state zero for the next call
  __syncthreads();
  if (threadIdx.x == 0) s_tile = (i64)atomicAdd(&state[1], 1ull);
  __syncthreads();
  if (s_tile == nt - 1)
    for (i64 q = threadIdx.x; q < nt + 2; q += OCN_BLOCK)
      __hip_atomic_store(&state[q], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename Op>
static int run_scan(Op op, i64 n, i64* out, void* ws, hipStream_t st) {
  if (n < 0 || !out || !ws) return OCN_EINVAL;
  if (n <= SCAN_SINGLE_MAX) {
    hipLaunchKernelGGL(scan_single<Op>, dim3(1), dim3(OCN_BLOCK), 0, st, op, n, out);
    return launch_status();
  }
  const i64 nt = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL((scan_chained<Op, NoExtra>), dim3((unsigned)nt), dim3(OCN_BLOCK), 0, st, op, n, out, (u64*)ws, nt, NoExtra());
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// processing order of a candidate batch: counting sort of the batch rows by source node (arbitrary
// order among rows with the same source), so that rows that gather the same neighbourhood are
// visited back to back
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OCN_BLOCK) void order_count(const i64* __restrict__ node, i64 B,
                                                         int32_t* __restrict__ counts) {
  order_count_body(node, B, counts, (i64)blockIdx.x, (i64)gridDim.x);
}

// (the counters were consumed by the scan: each row clears the one it raised, so the workspace is left zero)
__global__ __launch_bounds__(OCN_BLOCK) void order_scatter(const i64* __restrict__ node, i64 B,
                                                           unsigned long long* __restrict__ cursor, const i64* __restrict__ total,
                                                           i64* __restrict__ order, int32_t* __restrict__ counts) {
  const bool poisoned = total[0] < 0;        // the scan of the counters gave up (scan_chained): batch order — any order is a correct one
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < B; e += (i64)gridDim.x * blockDim.x) {
    const i64 v = node[e];
    if (poisoned) order[e] = e;
    else order[atomicAdd(cursor + v, 1ull)] = e;
    counts[v] = 0;
  }
}

__global__ __launch_bounds__(OCN_BLOCK) void check_edges_kernel(const i64* __restrict__ src, const i64* __restrict__ dst,
                                                               i64 B, i64 n_src, i64 n_dst, int32_t* __restrict__ bad) {
  bool b = false;
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < B; e += (i64)gridDim.x * blockDim.x) {
    const i64 i = src[e], j = dst[e];
    b |= (i < 0) | (i >= n_src) | (j < 0) | (j >= n_dst);
  }
  if (__ballot(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

__global__ __launch_bounds__(OCN_BLOCK) void zero_regions_kernel(const ZeroArgs a) {
  zero_body(a, (i64)blockIdx.x, (i64)gridDim.x);
}

// (ptrs, bytes) -> ZeroArgs; returns the number of 16-byte quads, or -1 for bad arguments
static i64 zero_args(void* const* ptrs, const int64_t* bytes, int32_t n, ZeroArgs& a) {
  if (n < 0 || n > ZERO_MAX_REGIONS || (n > 0 && (!ptrs || !bytes))) return -1;
  a.n = 0;
  i64 quads = 0;
  for (int r = 0; r < n; ++r) {
    if (bytes[r] < 0 || (bytes[r] & 3) || (bytes[r] > 0 && (!ptrs[r] || ((uintptr_t)ptrs[r] & 3)))) return -1;
    if (bytes[r] == 0) continue;
    a.p[a.n] = (uint32_t*)ptrs[r];
    a.words[a.n] = bytes[r] >> 2;
    quads += (a.words[a.n] + 3) >> 2;
    ++a.n;
  }
  return quads;
}

extern "C" {

int64_t ocn_scan_workspace_bytes(int64_t n);
int ocn_order_by_node_finish(const int64_t* node, int64_t B, int64_t n_nodes, int64_t* order, void* workspace, void* stream);

int ocn_zero_regions(void* const* ptrs, const int64_t* bytes, int32_t n, void* stream) {
  ZeroArgs a;
  const i64 quads = zero_args(ptrs, bytes, n, a);
  if (quads < 0) return OCN_EINVAL;
  if (a.n == 0) return 0;
  hipLaunchKernelGGL(zero_regions_kernel, dim3(grid_for((quads + OCN_BLOCK - 1) / OCN_BLOCK, 2048)), dim3(OCN_BLOCK), 0,
                     (hipStream_t)stream, a);
  return launch_status();
}

int ocn_check_edges(const int64_t* src, const int64_t* dst, int64_t B, int64_t n_src, int64_t n_dst, int32_t* bad,
                    void* stream) {
  if (B < 0 || !bad || (B > 0 && (!src || !dst))) return OCN_EINVAL;
  if (B == 0) return 0;
  hipLaunchKernelGGL(check_edges_kernel, dim3(grid_for((B + OCN_BLOCK - 1) / OCN_BLOCK, 1024)), dim3(OCN_BLOCK), 0,
                     (hipStream_t)stream, (const i64*)src, (const i64*)dst, (i64)B, (i64)n_src, (i64)n_dst, bad);
  return launch_status();
}

static inline i64 order_counts_bytes(i64 n_nodes) { return ((n_nodes * 4 + 15) / 16) * 16; }

int64_t ocn_order_workspace_bytes(int64_t n_nodes) {
  return order_counts_bytes(n_nodes) + (n_nodes + 1) * 8 + ocn_scan_workspace_bytes(n_nodes);
}

int ocn_order_by_node(const int64_t* node, int64_t B, int64_t n_nodes, int64_t* order, void* workspace,
                      void* stream) {
  if (B < 0 || n_nodes <= 0) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!node || !order || !workspace) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int32_t* counts = (int32_t*)workspace;
  i64* offs = (i64*)((char*)workspace + order_counts_bytes(n_nodes));
  void* scan_ws = (void*)(offs + n_nodes + 1);
  const int grid = grid_for((B + OCN_BLOCK - 1) / OCN_BLOCK, 1024);
  hipLaunchKernelGGL(order_count, dim3(grid), dim3(OCN_BLOCK), 0, st, (const i64*)node, (i64)B, counts);
  return ocn_order_by_node_finish(node, B, n_nodes, order, workspace, stream);
}

int ocn_order_by_node_finish(const int64_t* node, int64_t B, int64_t n_nodes, int64_t* order, void* workspace,
                             void* stream) {
  if (B < 0 || n_nodes <= 0) return OCN_EINVAL;
  if (B == 0) return 0;
  if (!node || !order || !workspace) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int32_t* counts = (int32_t*)workspace;
  i64* offs = (i64*)((char*)workspace + order_counts_bytes(n_nodes));
  void* scan_ws = (void*)(offs + n_nodes + 1);
  const int grid = grid_for((B + OCN_BLOCK - 1) / OCN_BLOCK, 1024);
  I32In op{counts};
  int rc = run_scan(op, (i64)n_nodes, offs, scan_ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(order_scatter, dim3(grid), dim3(OCN_BLOCK), 0, st, (const i64*)node, (i64)B,
                     (unsigned long long*)offs, (const i64*)(offs + n_nodes), (i64*)order, counts);
  return launch_status();
}

int ocn_batch_prep(const int64_t* rowptrA, const int64_t* src, int64_t B, int64_t* off, void* scan_workspace,
                   int64_t n_nodes, void* order_workspace, void* const* zero_ptrs, const int64_t* zero_bytes,
                   int32_t n_zero, void* stream) {
  if (B < 0 || !rowptrA || (!src && B > 0) || !off || !scan_workspace || (order_workspace && n_nodes <= 0)) return OCN_EINVAL;
  ZeroArgs z;
  const i64 quads = zero_args(zero_ptrs, zero_bytes, n_zero, z);
  if (quads < 0) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  DegOfSrc op{(const i64*)rowptrA, (const i64*)src};
  int32_t* counts = (int32_t*)order_workspace;
  if (B <= SCAN_SINGLE_MAX) {                // small batch: the single-workgroup scan has no extra workgroups to lend
    int rc = ocn_zero_regions(zero_ptrs, zero_bytes, n_zero, stream);
    if (rc) return rc;
    if (counts && B > 0)
      hipLaunchKernelGGL(order_count, dim3(grid_for((B + OCN_BLOCK - 1) / OCN_BLOCK, 1024)), dim3(OCN_BLOCK), 0, st,
                         (const i64*)src, (i64)B, counts);
    return run_scan(op, (i64)B, (i64*)off, scan_workspace, st);
  }
  const i64 nt = (B + SCAN_TILE - 1) / SCAN_TILE;
  PrepExtra x;
  x.z = z;
  x.zero_blocks = z.n ? grid_for((quads + OCN_BLOCK - 1) / OCN_BLOCK, 1024) : 0;
  x.node = (const i64*)src; x.B = (i64)B; x.counts = counts;
  const i64 count_blocks = counts ? grid_for((B + OCN_BLOCK - 1) / OCN_BLOCK, 512) : 0;
  hipLaunchKernelGGL((scan_chained<DegOfSrc, PrepExtra>), dim3((unsigned)(nt + x.zero_blocks + count_blocks)), dim3(OCN_BLOCK), 0, st,
                     op, (i64)B, (i64*)off, (u64*)scan_workspace, nt, x);
  return launch_status();
}


int ocn_abi_version(void) { return OCN_ABI_VERSION; }

int64_t ocn_scan_workspace_bytes(int64_t n) {
  i64 nt = (n + SCAN_TILE - 1) / SCAN_TILE;
  return (nt + 4) * (int64_t)sizeof(i64);
}

int ocn_edge_offsets(const int64_t* rowptrA, const int64_t* src, int64_t B, int64_t* off,
                     void* workspace, void* stream) {
  if (!rowptrA || (!src && B > 0)) return OCN_EINVAL;
  DegOfSrc op{(const i64*)rowptrA, (const i64*)src};
  return run_scan(op, B, (i64*)off, workspace, (hipStream_t)stream);
}

int ocn_chunk_offsets(const int64_t* rowptrA, const int64_t* nds, const int64_t* src, const int64_t* order,
                      int64_t B, int64_t* out, void* workspace, void* stream) {
  if (!rowptrA || (!src && B > 0)) return OCN_EINVAL;
  ChunksOfSlot op{(const i64*)rowptrA, (const i64*)nds, (const i64*)src, (const i64*)order};
  return run_scan(op, B, (i64*)out, workspace, (hipStream_t)stream);
}

int ocn_walk_rev_offsets(const int64_t* rowptrA, const int64_t* nds, const int64_t* src, const int64_t* dst,
                         const int64_t* order, int64_t B, int64_t* out, void* workspace, void* stream) {
  if (!rowptrA || !nds || ((!src || !dst) && B > 0)) return OCN_EINVAL;
  RevChunksOfSlot op{(const i64*)rowptrA, (const i64*)nds, (const i64*)src, (const i64*)dst, (const i64*)order};
  return run_scan(op, B, (i64*)out, workspace, (hipStream_t)stream);
}

int ocn_class_order(const int32_t* cnt1, const int32_t* cnt2, const int64_t* order_in, int64_t B,
                    int64_t* order_out, int64_t* inv_out, int64_t* ranges, int64_t* prefix, void* workspace,
                    void* stream) {
  if (B < 0 || B > (int64_t)CLS_MASK || !ranges || (B > 0 && (!cnt1 || !order_out || !inv_out || !prefix))) return OCN_EINVAL;
  ClassOfSlot op{cnt1, cnt2, (const i64*)order_in};
  const int rc = run_scan(op, (i64)B, (i64*)prefix, workspace, (hipStream_t)stream);
  if (rc) return rc;
  hipLaunchKernelGGL(class_scatter, dim3(grid_for((B + OCN_BLOCK - 1) / OCN_BLOCK, 1024)), dim3(OCN_BLOCK), 0,
                     (hipStream_t)stream, op, (i64)B, (const i64*)prefix, (i64*)order_out, (i64*)inv_out, (i64*)ranges);
  return launch_status();
}

int ocn_scan_i32(const int32_t* in, int64_t n, int64_t* out, void* workspace, void* stream) {
  if (!in && n > 0) return OCN_EINVAL;
  I32In op{in};
  return run_scan(op, n, (i64*)out, workspace, (hipStream_t)stream);
}

}  // extern "C"
