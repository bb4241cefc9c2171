// A·A for dense graphs (ogbl-ddi: 4 267 nodes, 11.7 % dense; A² is full) on the integer matrix cores — the block route of
// utils.block_matrix_multiply (utils.py:287-323): A as a dense 0/1 int8 matrix, every (row block, column block) of
// `block_size` multiplied with v_mfma_i32_32x32x32_i8 (fp32-exact: the products are walk counts < 2^31), the non-zero
// pattern written as bit rows.  The reference adds each block's SparseTensor.from_dense(result) — whose indices are
// block-LOCAL — into one matrix without the block's offset: `fold` reproduces that (every block lands on the
// top-left corner), fold = 0 places the blocks where they belong (the intended A²).
#include "common.h"

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

#define DA_TILE 128                   /* output tile of a workgroup: 4 waves x (64 x 64) */

// dense[r][c] = 1 for every stored (r, c); denseT likewise transposed.  Both zero on entry, row stride ld bytes.
__global__ __launch_bounds__(OCN_BLOCK) void densify_kernel(const i64* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            i64 n, i64 ld, int8_t* __restrict__ dense, int8_t* __restrict__ denseT) {
  const int lane = threadIdx.x & 63;
  for (i64 r = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n; r += ((i64)gridDim.x * blockDim.x) >> 6)
    for (i64 p = rowptr[r] + lane; p < rowptr[r + 1]; p += OCN_WAVE) {
      const i64 c = col[p];
      dense[r * ld + c] = 1;
      denseT[c * ld + r] = 1;
    }
}

// C = A . B on the block rows [r0, r1) x cols [c0, c1), K = kpad (multiple of 64, zero padded); A row-major, Bt = B
// transposed row-major.  bits[(row - ro)][(col - co) / 32] |= (C > 0) — atomically: folded blocks overlap.
__global__ __launch_bounds__(OCN_BLOCK) void dense_block_mm_kernel(const int8_t* __restrict__ A, const int8_t* __restrict__ Bt,
                                                                   i64 ld, int kpad, int r0, int r1, int c0, int c1, int ro, int co,
                                                                   unsigned* __restrict__ bits, i64 bm_stride) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int rr = lane & 31, hh = lane >> 5;
  const int tiles_c = (c1 - c0 + DA_TILE - 1) / DA_TILE;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const int wr = r0 + tr * DA_TILE + 64 * (w >> 1), wc = c0 + tc * DA_TILE + 64 * (w & 1);     // this wave's 64 x 64 corner
  i32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0;
  // operand rows beyond the block (ragged last block) are clamped: their products land in rows / columns never written
  const int8_t* ap[2];
  const int8_t* bp[2];
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int row = wr + 32 * a + rr, colr = wc + 32 * a + rr;
    ap[a] = A + (i64)(row < r1 ? row : r1 - 1) * ld + 16 * hh;
    bp[a] = Bt + (i64)(colr < c1 ? colr : c1 - 1) * ld + 16 * hh;
  }
  for (int k = 0; k < kpad; k += 64) {          // two 32-deep MFMA steps per trip: 32 contiguous bytes per lane and operand row
    i32x4 af[2][2], bf[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        af[a][s] = *reinterpret_cast<const i32x4*>(ap[a] + k + 32 * s);
        bf[a][s] = *reinterpret_cast<const i32x4*>(bp[a] + k + 32 * s);
      }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af[a][s], bf[b][s], acc[a][b], 0, 0, 0);
  }
  // lane holds column (lane & 31) of rows (q & 3) + 8 (q >> 2) + 4 hh: one ballot per register = the 32-bit words of two rows
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = wr + 32 * a + (q & 3) + 8 * (q >> 2) + 4 * hh;
        const int colb = wc + 32 * b;
        const bool on = acc[a][b][q] > 0 && row < r1 && colb + rr < c1;
        const unsigned long long m = __ballot(on);
        const unsigned word = hh ? (unsigned)(m >> 32) : (unsigned)m;
        if (rr == 0 && word && row < r1) {
          const int orow = row - ro, ocol = colb - co;            // ocol is a multiple of 32 (blocks start on multiples of 32)
          atomicOr(bits + (i64)orow * bm_stride + (ocol >> 5), word);
        }
      }
}

// bit rows -> per-row counts / CSR columns (one wave per row)
template <bool FILL>
__global__ __launch_bounds__(OCN_BLOCK) void bitrows_kernel(const unsigned* __restrict__ bits, i64 bm_stride, i64 n_rows, int words,
                                                            int32_t* __restrict__ row_count, const i64* __restrict__ rowptr,
                                                            int32_t* __restrict__ col) {
  const int lane = threadIdx.x & 63;
  for (i64 r = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n_rows; r += ((i64)gridDim.x * blockDim.x) >> 6) {
    const unsigned* row = bits + r * bm_stride;
    i64 run = 0;
    for (int w0 = 0; w0 < words; w0 += OCN_WAVE) {
      const int w = w0 + lane;
      unsigned b = w < words ? row[w] : 0u;
      const i64 c = __popc(b);
      const i64 incl = wave_incl_scan(c, lane);
      if (FILL) {
        int32_t* out = col + rowptr[r] + run + (incl - c);
        while (b) {
          const int q = __ffs((int)b) - 1;
          b &= b - 1;
          *out++ = (w << 5) + q;
        }
      }
      run += __shfl(incl, 63, OCN_WAVE);
    }
    if (!FILL && lane == 0) row_count[r] = (int32_t)run;
  }
}

// A CSR pattern as dense bit rows (bits ZERO on entry): one wave per row, a lane sets the bits of its entries.
__global__ __launch_bounds__(OCN_BLOCK) void csr_bitrows_kernel(const i64* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                i64 n_rows, unsigned* __restrict__ bits, i64 bm_stride) {
  const int lane = threadIdx.x & 63;
  for (i64 r = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6); r < n_rows; r += (i64)gridDim.x * OCN_WPB) {
    const i64 a = rowptr[r], b = rowptr[r + 1];
    for (i64 p = a + lane; p < b; p += OCN_WAVE) {
      const int32_t c = col[p];
      atomicOr(bits + r * bm_stride + (c >> 5), 1u << (c & 31));
    }
  }
}

extern "C" {

int ocn_dense_from_csr(const int64_t* rowptr, const int32_t* col, int64_t n, int64_t ld, int8_t* dense, int8_t* denseT,
                       void* stream) {
  if (n < 0 || ld < n || (ld & 63)) return OCN_EINVAL;
  if (n == 0) return 0;
  if (!rowptr || !dense || !denseT) return OCN_EINVAL;
  hipLaunchKernelGGL(densify_kernel, dim3(grid_for((n + OCN_WPB - 1) / OCN_WPB, 1 << 16)), dim3(OCN_BLOCK), 0,
                     (hipStream_t)stream, (const i64*)rowptr, col, (i64)n, (i64)ld, dense, denseT);
  return launch_status();
}

int ocn_dense_block_mm_bits(const int8_t* A, const int8_t* Bt, int64_t ld, int64_t K, int32_t r0, int32_t r1, int32_t c0,
                            int32_t c1, int32_t fold, uint32_t* bits, int64_t bm_stride_words, void* stream) {
  if (!A || !Bt || !bits || ld < K || (ld & 63) || K <= 0 || r0 < 0 || c0 < 0 || r1 < r0 || c1 < c0 || (r0 & 31) || (c0 & 31))
    return OCN_EINVAL;
  if (r1 == r0 || c1 == c0) return 0;
  const int kpad = (int)((K + 63) / 64 * 64);
  if (kpad > ld) return OCN_EINVAL;
  const int tiles = ((r1 - r0 + DA_TILE - 1) / DA_TILE) * ((c1 - c0 + DA_TILE - 1) / DA_TILE);
  hipLaunchKernelGGL(dense_block_mm_kernel, dim3(tiles), dim3(OCN_BLOCK), 0, (hipStream_t)stream, A, Bt, (i64)ld, kpad,
                     (int)r0, (int)r1, (int)c0, (int)c1, fold ? (int)r0 : 0, fold ? (int)c0 : 0, (unsigned*)bits,
                     (i64)bm_stride_words);
  return launch_status();
}

int ocn_bitrows_from_csr(const int64_t* rowptr, const int32_t* col, int64_t n_rows, uint32_t* bits, int64_t bm_stride_words,
                         void* stream) {
  if (n_rows < 0 || bm_stride_words < 0) return OCN_EINVAL;
  if (n_rows == 0) return 0;
  if (!rowptr || !bits) return OCN_EINVAL;
  hipLaunchKernelGGL(csr_bitrows_kernel, dim3(grid_for((n_rows + OCN_WPB - 1) / OCN_WPB, 1 << 16)), dim3(OCN_BLOCK), 0,
                     (hipStream_t)stream, (const i64*)rowptr, col, (i64)n_rows, (unsigned*)bits, (i64)bm_stride_words);
  return launch_status();
}

int ocn_bitrows_count(const uint32_t* bits, int64_t bm_stride_words, int64_t n_rows, int64_t n_cols, int32_t* row_count,
                      void* stream) {
  if (n_rows < 0 || n_cols < 0 || bm_stride_words * 32 < n_cols) return OCN_EINVAL;
  if (n_rows == 0) return 0;
  if (!bits || !row_count) return OCN_EINVAL;
  hipLaunchKernelGGL((bitrows_kernel<false>), dim3(grid_for((n_rows + OCN_WPB - 1) / OCN_WPB, 1 << 16)), dim3(OCN_BLOCK), 0,
                     (hipStream_t)stream, (const unsigned*)bits, (i64)bm_stride_words, (i64)n_rows, (int)((n_cols + 31) / 32),
                     row_count, (const i64*)nullptr, (int32_t*)nullptr);
  return launch_status();
}

int ocn_bitrows_fill(const uint32_t* bits, int64_t bm_stride_words, int64_t n_rows, int64_t n_cols, const int64_t* rowptr,
                     int32_t* col, void* stream) {
  if (n_rows < 0 || n_cols < 0 || bm_stride_words * 32 < n_cols) return OCN_EINVAL;
  if (n_rows == 0) return 0;
  if (!bits || !rowptr || !col) return OCN_EINVAL;
  hipLaunchKernelGGL((bitrows_kernel<true>), dim3(grid_for((n_rows + OCN_WPB - 1) / OCN_WPB, 1 << 16)), dim3(OCN_BLOCK), 0,
                     (hipStream_t)stream, (const unsigned*)bits, (i64)bm_stride_words, (i64)n_rows, (int)((n_cols + 31) / 32),
                     (int32_t*)nullptr, (const i64*)rowptr, col);
  return launch_status();
}

}  // extern "C"
