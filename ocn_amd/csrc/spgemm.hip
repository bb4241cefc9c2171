// Pattern of A*B: one workgroup per output row, the row's column set as a bitmap in LDS.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// A*A pattern: one workgroup per output row, the row's column set as a bitmap in LDS
// ---------------------------------------------------------------------------------------------
#define SPGEMM_MAX_LDS (160 * 1024 - 2048)

template <bool FILL>
__global__ __launch_bounds__(OCN_BLOCK) void spgemm_pattern_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA, i64 n_rows,
    const i64* __restrict__ rowptrB, const int32_t* __restrict__ colB, i64 n_colsB,
    int32_t* __restrict__ row_count, const i64* __restrict__ rowptrC, int32_t* __restrict__ colC,
    unsigned* __restrict__ bitmap_out, i64 bm_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned bm[];
  __shared__ i64 sh[2 * OCN_WPB];
  const int words = (int)((n_colsB + 31) >> 5);
  const int wpt = (words + OCN_BLOCK - 1) / OCN_BLOCK;       // contiguous words per thread
  const int w0 = threadIdx.x * wpt;
  const int w1 = (w0 + wpt) < words ? (w0 + wpt) : words;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w = threadIdx.x; w < words; w += OCN_BLOCK) bm[w] = 0u;
  __syncthreads();
  for (i64 r = blockIdx.x; r < n_rows; r += gridDim.x) {
    const i64 a0 = rowptrA[r], da = rowptrA[r + 1] - a0;
    for (i64 q = wave; q < da; q += OCN_WPB) {
      const i64 m = colA[a0 + q];
      const i64 b0 = rowptrB[m], db = rowptrB[m + 1] - b0;
      for (i64 t = lane; t < db; t += OCN_WAVE) {
        const unsigned k = (unsigned)colB[b0 + t];
        atomicOr(&bm[k >> 5], 1u << (k & 31u));
      }
    }
    __syncthreads();
    i64 c = 0;
    for (int w = w0; w < w1; ++w) c += __popc(bm[w]);
    i64 tot;
    i64 ex = block_excl_scan(c, sh, &tot);
    if (!FILL) {
      if (threadIdx.x == 0) row_count[r] = (int32_t)tot;
      if (bitmap_out) {                       // the row as a dense bit row: one probe answers k in row r
        for (int w = threadIdx.x; w < words; w += OCN_BLOCK) bitmap_out[r * bm_stride + w] = bm[w];
        __syncthreads();
      }
      for (int w = w0; w < w1; ++w) bm[w] = 0u;
    } else {
      int32_t* out = colC + rowptrC[r] + ex;
      for (int w = w0; w < w1; ++w) {
        unsigned bits = bm[w];
        bm[w] = 0u;
        while (bits) {
          const int b = __ffs((int)bits) - 1;
          bits &= bits - 1;
          *out++ = (w << 5) + b;
        }
      }
    }
    __syncthreads();
  }
}

extern "C" int64_t ocn_spgemm_max_cols(void);

// Bit rows of A*B for the rows somebody asks for (a training step's per-batch A² is probed at the candidates' target rows
// only — two fifths of the rows at the collab shape, and a 29 KiB dense row is what the counting pass pays for each):
// request i names row rows[i]; whoever turns done[row] from 0 to 1 builds it, duplicates and rows of earlier requests skip.
__global__ __launch_bounds__(OCN_BLOCK) void spgemm_rows_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA, i64 n_rows,
    const i64* __restrict__ rowptrB, const int32_t* __restrict__ colB, i64 n_colsB,
    const i64* __restrict__ rows, i64 n_req, int32_t* __restrict__ done,
    unsigned* __restrict__ bitmap_out, i64 bm_stride) {
  extern __shared__ __attribute__((aligned(16))) unsigned bm[];
  __shared__ int s_mine;
  const int words = (int)((n_colsB + 31) >> 5);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int w = threadIdx.x; w < words; w += OCN_BLOCK) bm[w] = 0u;
  for (i64 i = blockIdx.x; i < n_req; i += gridDim.x) {
    const i64 r = rows[i];
    if (threadIdx.x == 0) s_mine = (r >= 0 && r < n_rows && atomicExch(&done[r], 1) == 0) ? 1 : 0;
    __syncthreads();                                        // (also: the bitmap is zero)
    const int mine = s_mine;
    __syncthreads();
    if (!mine) continue;
    const i64 a0 = rowptrA[r], da = rowptrA[r + 1] - a0;
    for (i64 q = wave; q < da; q += OCN_WPB) {
      const i64 m = colA[a0 + q];
      const i64 b0 = rowptrB[m], db = rowptrB[m + 1] - b0;
      for (i64 t = lane; t < db; t += OCN_WAVE) {
        const unsigned k = (unsigned)colB[b0 + t];
        atomicOr(&bm[k >> 5], 1u << (k & 31u));
      }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < words; w += OCN_BLOCK) {
      bitmap_out[r * bm_stride + w] = bm[w];
      bm[w] = 0u;
    }
  }
}

extern "C" {

int ocn_spgemm_bit_rows(const int64_t* rowptrA, const int32_t* colA, int64_t n_rows, const int64_t* rowptrB, const int32_t* colB,
                        int64_t n_colsB, const int64_t* rows, int64_t n_req, int32_t* done, uint32_t* bitmap,
                        int64_t bm_stride_words, void* stream) {
  if (n_rows < 0 || n_req < 0 || n_colsB <= 0 || n_colsB > ocn_spgemm_max_cols()) return OCN_EINVAL;
  if (n_req == 0 || n_rows == 0) return 0;
  if (!rowptrA || !rowptrB || !rows || !done || !bitmap || bm_stride_words < (n_colsB + 31) / 32) return OCN_EINVAL;
  const size_t lds = (size_t)(((n_colsB + 31) >> 5) * 4);
  const int per_cu = (int)((160 * 1024) / (lds + 256));
  int64_t grid = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
  if (grid > n_req) grid = n_req;
  const hipError_t err = hipFuncSetAttribute((const void*)spgemm_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return (int)err;
  hipLaunchKernelGGL(spgemm_rows_kernel, dim3((unsigned)grid), dim3(OCN_BLOCK), lds, (hipStream_t)stream,
                     (const i64*)rowptrA, colA, (i64)n_rows, (const i64*)rowptrB, colB, (i64)n_colsB, (const i64*)rows, (i64)n_req,
                     done, (unsigned*)bitmap, (i64)bm_stride_words);
  return launch_status();
}

int64_t ocn_spgemm_max_cols(void) { return (int64_t)SPGEMM_MAX_LDS * 8; }

static int spgemm_launch(bool fill, const int64_t* rowptrA, const int32_t* colA, int64_t n_rows,
                         const int64_t* rowptrB, const int32_t* colB, int64_t n_colsB,
                         int32_t* row_count, const int64_t* rowptrC, int32_t* colC, uint32_t* bitmap,
                         int64_t bm_stride, void* stream) {
  if (n_rows < 0 || n_colsB <= 0 || n_colsB > ocn_spgemm_max_cols()) return OCN_EINVAL;
  if (n_rows == 0) return 0;
  if (!rowptrA || !rowptrB) return OCN_EINVAL;
  const size_t lds = (size_t)(((n_colsB + 31) >> 5) * 4);
  const int per_cu = (int)((160 * 1024) / (lds + 256));
  int grid = 256 * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
  if (grid > n_rows) grid = (int)n_rows;
  hipStream_t st = (hipStream_t)stream;
  hipError_t err;
  if (fill) {
    err = hipFuncSetAttribute((const void*)spgemm_pattern_kernel<true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL(spgemm_pattern_kernel<true>, dim3(grid), dim3(OCN_BLOCK), lds, st,
                       (const i64*)rowptrA, colA, (i64)n_rows, (const i64*)rowptrB, colB, (i64)n_colsB,
                       row_count, (const i64*)rowptrC, colC, (unsigned*)nullptr, (i64)0);
  } else {
    err = hipFuncSetAttribute((const void*)spgemm_pattern_kernel<false>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL(spgemm_pattern_kernel<false>, dim3(grid), dim3(OCN_BLOCK), lds, st,
                       (const i64*)rowptrA, colA, (i64)n_rows, (const i64*)rowptrB, colB, (i64)n_colsB,
                       row_count, (const i64*)rowptrC, colC, (unsigned*)bitmap, (i64)bm_stride);
  }
  return launch_status();
}

int ocn_spgemm_pattern_count(const int64_t* rowptrA, const int32_t* colA, int64_t n_rows,
                             const int64_t* rowptrB, const int32_t* colB, int64_t n_colsB,
                             int32_t* row_count, uint32_t* bitmap, int64_t bm_stride_words, void* stream) {
  if (!row_count && n_rows > 0) return OCN_EINVAL;
  if (bitmap && bm_stride_words < (n_colsB + 31) / 32) return OCN_EINVAL;
  return spgemm_launch(false, rowptrA, colA, n_rows, rowptrB, colB, n_colsB, row_count, nullptr,
                       nullptr, bitmap, bm_stride_words, stream);
}

int ocn_spgemm_pattern_fill(const int64_t* rowptrA, const int32_t* colA, int64_t n_rows,
                            const int64_t* rowptrB, const int32_t* colB, int64_t n_colsB,
                            const int64_t* rowptrC, int32_t* colC, void* stream) {
  if ((!rowptrC || !colC) && n_rows > 0) return OCN_EINVAL;
  return spgemm_launch(true, rowptrA, colA, n_rows, rowptrB, colB, n_colsB, nullptr, rowptrC, colC,
                       nullptr, 0, stream);
}

}  // extern "C"
