// COO -> CSR on the device: the per-batch masked adjacency of the training loop
// (NeighborOverlap_large.py:56-63: SparseTensor.from_edge_index(tei, sparse_sizes).to_symmetric() once per batch — in
// torch_sparse a sort by (row, col), then cat + coalesce).  Pattern only (the drivers build it without values):
//
//   count   one thread per input entry: bounds check, cnt[r] += 1 (and cnt[c] += 1 when the transposed entry joins)
//   scan    the library's chained scan -> start of every row in the staging array
//   fill    entry -> next free slot of its row (one atomic cursor per row; any order: the sort below is canonical)
//   sort    per row, ascending column: <= 64 entries one wave (rank by readlane), longer rows one workgroup (LDS bitonic
//           network up to CC_LDS entries, in place in memory beyond); duplicates marked while the row is still in
//           registers / LDS -> unique count per row
//   scan + compact (dedupe only): final rowptr, unique columns moved to their final place
//
// No comparison sort over the whole edge list, no host round trip inside; the caller reads {nnz, status} once.
#include "rowsort.h"

__global__ __launch_bounds__(OCN_BLOCK) void cc_zero_kernel(int32_t* __restrict__ a, i64 n, i64* __restrict__ res) {
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (i64)gridDim.x * blockDim.x) a[q] = 0;
  if (res && blockIdx.x == 0 && threadIdx.x < 2) res[threadIdx.x] = 0;
}

template <bool FILL>
__global__ __launch_bounds__(OCN_BLOCK) void cc_entries_kernel(const i64* __restrict__ row, const i64* __restrict__ col, i64 nnz,
                                                               i64 n_rows, i64 n_cols, int sym, int32_t* __restrict__ cursor,
                                                               const i64* __restrict__ start, int32_t* __restrict__ stage,
                                                               i64* __restrict__ res) {
  bool bad = false;
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < nnz; q += (i64)gridDim.x * blockDim.x) {
    const i64 r = row[q], c = col[q];
    if (r < 0 || r >= n_rows || c < 0 || c >= n_cols || (sym && (c >= n_rows || r >= n_cols))) { bad = true; continue; }
    if (FILL) {
      stage[start[r] + atomicAdd(cursor + r, 1)] = (int32_t)c;
      if (sym) stage[start[c] + atomicAdd(cursor + c, 1)] = (int32_t)r;
    } else {
      atomicAdd(cursor + r, 1);
      if (sym) atomicAdd(cursor + c, 1);
    }
  }
  if (!FILL && bad) res[1] = 1;               // any writer, same value
}

// One wave per row: the distinct columns of the sorted staging row go to their final place.
__global__ __launch_bounds__(OCN_BLOCK) void cc_compact_kernel(const i64* __restrict__ start, i64 n_rows, const int32_t* __restrict__ stage,
                                                               const i64* __restrict__ rowptr, int32_t* __restrict__ col_out,
                                                               i64* __restrict__ res) {
  const int lane = threadIdx.x & 63;
  for (i64 r = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6); r < n_rows; r += (i64)gridDim.x * OCN_WPB) {
    const i64 b = start[r];
    const i64 n = start[r + 1] - b;
    i64 o = rowptr[r];
    for (i64 q0 = 0; q0 < n; q0 += OCN_WAVE) {
      const i64 q = q0 + lane;
      int32_t v = 0;
      bool first = false;
      if (q < n) { v = stage[b + q]; first = q == 0 || stage[b + q - 1] != v; }
      const u64 m = __ballot(first);
      if (first) col_out[o + __popcll(m & ((1ull << lane) - 1ull))] = v;
      o += __popcll(m);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) res[0] = rowptr[n_rows];
}

__global__ void cc_result_kernel(const i64* __restrict__ rowptr, i64 n_rows, i64* __restrict__ res) { res[0] = rowptr[n_rows]; }

extern "C" {

static inline int64_t cc_align(int64_t b) { return (b + 15) / 16 * 16; }

int64_t ocn_coo_to_csr_workspace_bytes(int64_t nnz, int64_t n_rows, int32_t symmetrize, int32_t dedupe) {
  // cursor int32[n_rows] | ucount int32[n_rows] | long_list int32[n_rows] | tickets int32[4] | scan state |
  // (dedupe:) start int64[n_rows + 1] | stage int32[m]
  const int64_t m = nnz * (symmetrize ? 2 : 1);
  int64_t t = 3 * cc_align(n_rows * 4) + 16 + cc_align(ocn_scan_workspace_bytes(n_rows));
  if (dedupe) t += cc_align((n_rows + 1) * 8) + cc_align(m * 4);
  return t + 64;
}

int ocn_coo_to_csr(const int64_t* row, const int64_t* col, int64_t nnz, int64_t n_rows, int64_t n_cols,
                   int32_t symmetrize, int32_t dedupe, int64_t* rowptr, int32_t* col_out, void* workspace,
                   int64_t* result, void* stream) {
  if (nnz < 0 || n_rows < 0 || n_cols < 0 || n_cols > 0x7fffffffll || n_rows > 0x7fffffffll) return OCN_EINVAL;
  if (!rowptr || !workspace || !result || (nnz > 0 && (!row || !col || !col_out))) return OCN_EINVAL;
  if (symmetrize && n_rows != n_cols) return OCN_EINVAL;
  const int64_t m = nnz * (symmetrize ? 2 : 1);
  if (m > 0x7fffffffll) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  char* ws = (char*)workspace;
  const int64_t a = cc_align(n_rows * 4), sw = cc_align(ocn_scan_workspace_bytes(n_rows));
  int32_t* cursor = (int32_t*)ws;
  int32_t* ucount = (int32_t*)(ws + a);
  int32_t* long_list = (int32_t*)(ws + 2 * a);
  int32_t* tickets = (int32_t*)(ws + 3 * a);            // [0] number of long rows, [1] work ticket; the scan state follows
  void* scan_ws = (void*)(ws + 3 * a + 16);
  i64* start = dedupe ? (i64*)(ws + 3 * a + 16 + sw) : (i64*)rowptr;
  int32_t* stage = dedupe ? (int32_t*)(ws + 3 * a + 16 + sw + cc_align((n_rows + 1) * 8)) : col_out;
  const int gridR = grid_for((n_rows + OCN_BLOCK - 1) / OCN_BLOCK, 2048);
  const int gridE = grid_for((nnz + OCN_BLOCK - 1) / OCN_BLOCK, 4096);
  const int gridW = grid_for((n_rows + OCN_WPB - 1) / OCN_WPB, 1 << 15);
  hipLaunchKernelGGL(cc_zero_kernel, dim3(gridR), dim3(OCN_BLOCK), 0, st, cursor, (i64)n_rows, (i64*)result);
  hipLaunchKernelGGL(cc_zero_kernel, dim3(1), dim3(OCN_BLOCK), 0, st, tickets, (i64)(4 + sw / 4), (i64*)nullptr);
  if (nnz > 0)
    hipLaunchKernelGGL((cc_entries_kernel<false>), dim3(gridE), dim3(OCN_BLOCK), 0, st, (const i64*)row, (const i64*)col, (i64)nnz,
                       (i64)n_rows, (i64)n_cols, (int)symmetrize, cursor, (const i64*)nullptr, (int32_t*)nullptr, (i64*)result);
  int rc = ocn_scan_i32(cursor, n_rows, (int64_t*)start, scan_ws, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(cc_zero_kernel, dim3(gridR), dim3(OCN_BLOCK), 0, st, cursor, (i64)n_rows, (i64*)nullptr);
  if (nnz > 0) {
    hipLaunchKernelGGL((cc_entries_kernel<true>), dim3(gridE), dim3(OCN_BLOCK), 0, st, (const i64*)row, (const i64*)col, (i64)nnz,
                       (i64)n_rows, (i64)n_cols, (int)symmetrize, cursor, (const i64*)start, stage, (i64*)result);
    hipLaunchKernelGGL(cc_sort_short_kernel, dim3(gridW), dim3(OCN_BLOCK), 0, st, (const i64*)start, (i64)n_rows, stage,
                       dedupe ? ucount : (int32_t*)nullptr, long_list, tickets);
    hipLaunchKernelGGL(cc_sort_long_kernel, dim3(256), dim3(OCN_BLOCK), 0, st, (const i64*)start, stage,
                       dedupe ? ucount : (int32_t*)nullptr, (const int32_t*)long_list, (const int32_t*)tickets, tickets + 1);
  }
  if (!dedupe) {
    hipLaunchKernelGGL(cc_result_kernel, dim3(1), dim3(1), 0, st, (const i64*)rowptr, (i64)n_rows, (i64*)result);
    return launch_status();
  }
  if (nnz == 0) hipLaunchKernelGGL(cc_zero_kernel, dim3(gridR), dim3(OCN_BLOCK), 0, st, ucount, (i64)n_rows, (i64*)nullptr);
  rc = ocn_scan_i32(ucount, n_rows, rowptr, scan_ws, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(cc_compact_kernel, dim3(gridW), dim3(OCN_BLOCK), 0, st, (const i64*)start, (i64)n_rows, (const int32_t*)stage,
                     (const i64*)rowptr, col_out, (i64*)result);
  return launch_status();
}

}  // extern "C"
