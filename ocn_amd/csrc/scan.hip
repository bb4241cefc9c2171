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
  const i64 n3 = tot & CLS_MASK, n2 = (tot >> CLS_BITS) & CLS_MASK, n1 = (tot >> (2 * CLS_BITS)) & CLS_MASK;
  for (i64 slot = (i64)blockIdx.x * blockDim.x + threadIdx.x; slot < B; slot += (i64)gridDim.x * blockDim.x) {
    const i64 p = prefix[slot];
    const i64 b3 = p & CLS_MASK, b2 = (p >> CLS_BITS) & CLS_MASK, b1 = (p >> (2 * CLS_BITS)) & CLS_MASK;
    const int c = op.cls(slot);
    const i64 pos = c == 3 ? b3 : (c == 2 ? n3 + b2 : (c == 1 ? n3 + n2 + b1 : n3 + n2 + n1 + (slot - b3 - b2 - b1)));
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

template <typename Op>
__global__ __launch_bounds__(OCN_BLOCK) void scan_tile_sums(Op op, i64 n, i64* tile_sum) {
  __shared__ i64 sh[OCN_WPB];
  const i64 base = (i64)blockIdx.x * SCAN_TILE;
  i64 s = 0;
#pragma unroll
  for (int t = 0; t < SCAN_IPT; ++t) {
    i64 e = base + (i64)t * OCN_BLOCK + threadIdx.x;
    if (e < n) s += op(e);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    i64 t = 0;
    for (int i = 0; i < OCN_WPB; ++i) t += sh[i];
    tile_sum[blockIdx.x] = t;
  }
}

// one block: exclusive scan of tile_sum[0..nt) in place, tile_sum[nt] = total
__global__ __launch_bounds__(OCN_BLOCK) void scan_spine(i64* tile_sum, i64 nt) {
  __shared__ i64 sh[2 * OCN_WPB];
  i64 carry = 0;
  for (i64 c0 = 0; c0 < nt; c0 += OCN_BLOCK) {
    i64 idx = c0 + threadIdx.x;
    i64 v = idx < nt ? tile_sum[idx] : 0;
    i64 tot;
    i64 ex = block_excl_scan(v, sh, &tot);
    if (idx < nt) tile_sum[idx] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) tile_sum[nt] = carry;
}

template <typename Op>
__global__ __launch_bounds__(OCN_BLOCK) void scan_apply(Op op, i64 n, const i64* tile_sum, i64 nt,
                                                        i64* out) {
  __shared__ i64 sh[2 * OCN_WPB];
  // thread-contiguous items so that the prefix order is the item order
  const i64 base = (i64)blockIdx.x * SCAN_TILE + (i64)threadIdx.x * SCAN_IPT;
  i64 v[SCAN_IPT];
  i64 s = 0;
#pragma unroll
  for (int t = 0; t < SCAN_IPT; ++t) {
    i64 e = base + t;
    v[t] = e < n ? op(e) : 0;
    s += v[t];
  }
  i64 tot;
  i64 ex = block_excl_scan(s, sh, &tot) + tile_sum[blockIdx.x];
#pragma unroll
  for (int t = 0; t < SCAN_IPT; ++t) {
    i64 e = base + t;
    if (e < n) out[e] = ex;
    ex += v[t];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = tile_sum[nt];
}

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

template <typename Op>
static int run_scan(Op op, i64 n, i64* out, void* ws, hipStream_t st) {
  if (n < 0 || !out || !ws) return OCN_EINVAL;
  if (n <= SCAN_SINGLE_MAX) {
    hipLaunchKernelGGL(scan_single<Op>, dim3(1), dim3(OCN_BLOCK), 0, st, op, n, out);
    return launch_status();
  }
  i64 nt = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nt == 0) nt = 1;
  i64* tile_sum = (i64*)ws;
  hipLaunchKernelGGL(scan_tile_sums<Op>, dim3((unsigned)nt), dim3(OCN_BLOCK), 0, st, op, n, tile_sum);
  hipLaunchKernelGGL(scan_spine, dim3(1), dim3(OCN_BLOCK), 0, st, tile_sum, nt);
  hipLaunchKernelGGL(scan_apply<Op>, dim3((unsigned)nt), dim3(OCN_BLOCK), 0, st, op, n, tile_sum, nt, out);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------
// processing order of a candidate batch: counting sort of the batch rows by source node (arbitrary
// order among rows with the same source), so that rows that gather the same neighbourhood are
// visited back to back
// ---------------------------------------------------------------------------------------------
// (a kernel rather than hipMemsetAsync: the library then enqueues nothing but kernel nodes, so a captured
// candidate batch replays as a pure kernel graph)
__global__ __launch_bounds__(OCN_BLOCK) void zero_i32_kernel(int32_t* __restrict__ p, i64 n) {
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (i64)gridDim.x * blockDim.x) p[q] = 0;
}

__global__ __launch_bounds__(OCN_BLOCK) void order_count(const i64* __restrict__ node, i64 B,
                                                         int32_t* __restrict__ counts) {
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < B; e += (i64)gridDim.x * blockDim.x)
    atomicAdd(counts + node[e], 1);
}

__global__ __launch_bounds__(OCN_BLOCK) void order_scatter(const i64* __restrict__ node, i64 B,
                                                           unsigned long long* __restrict__ cursor,
                                                           i64* __restrict__ order) {
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < B; e += (i64)gridDim.x * blockDim.x)
    order[atomicAdd(cursor + node[e], 1ull)] = e;
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

extern "C" {

int64_t ocn_scan_workspace_bytes(int64_t n);

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
#ifdef OCN_X_ORDER_MEMSET   /* experiment of tools/graph_fault_ab.py only: the round-1 form, a memset node in a captured batch */
  if (hipMemsetAsync(counts, 0, (size_t)n_nodes * 4, st) != hipSuccess) return OCN_EINVAL;
#else
  hipLaunchKernelGGL(zero_i32_kernel, dim3(grid_for((n_nodes + OCN_BLOCK - 1) / OCN_BLOCK, 1024)), dim3(OCN_BLOCK), 0, st,
                     counts, (i64)n_nodes);
#endif
  const int grid = grid_for((B + OCN_BLOCK - 1) / OCN_BLOCK, 1024);
  hipLaunchKernelGGL(order_count, dim3(grid), dim3(OCN_BLOCK), 0, st, (const i64*)node, (i64)B, counts);
  I32In op{counts};
  int rc = run_scan(op, (i64)n_nodes, offs, scan_ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(order_scatter, dim3(grid), dim3(OCN_BLOCK), 0, st, (const i64*)node, (i64)B,
                     (unsigned long long*)offs, (i64*)order);
  return launch_status();
}


int ocn_abi_version(void) { return OCN_ABI_VERSION; }

int64_t ocn_scan_workspace_bytes(int64_t n) {
  i64 nt = (n + SCAN_TILE - 1) / SCAN_TILE;
  return (nt + 2) * (int64_t)sizeof(i64);
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
