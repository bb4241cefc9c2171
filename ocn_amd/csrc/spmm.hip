// Encoder SpMM: one lane group per output row, fused row scales / self term / sum-mean-max.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// encoder SpMM
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(OCN_BLOCK) void deg_rsqrt_kernel(const i64* __restrict__ rowptr,
                                                              const float* __restrict__ val, i64 n,
                                                              float add, float* __restrict__ out) {
  for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (i64)gridDim.x * blockDim.x) {
    float deg;
    if (val) {                               // valued adjacency (DropAdj rescale): adj.sum(dim=-1)
      deg = 0.f;
      for (i64 p = rowptr[r]; p < rowptr[r + 1]; ++p) deg += val[p];
    } else {
      deg = (float)(rowptr[r + 1] - rowptr[r]);
    }
    const float d = add + deg;
    out[r] = d > 0.f ? 1.0f / sqrtf(d) : 0.f;
  }
}

enum { SPMM_SUM = 0, SPMM_MEAN = 1, SPMM_MAX = 2 };

template <int MODE>
__device__ __forceinline__ void red4(float4& acc, float w, bool weighted, const float4& x) {
  if (MODE == SPMM_MAX) {
    acc.x = fmaxf(acc.x, x.x); acc.y = fmaxf(acc.y, x.y);
    acc.z = fmaxf(acc.z, x.z); acc.w = fmaxf(acc.w, x.w);
  } else if (weighted) {
    axpy4(acc, w, x);
  } else {
    acc.x = __fadd_rn(acc.x, x.x); acc.y = __fadd_rn(acc.y, x.y);
    acc.z = __fadd_rn(acc.z, x.z); acc.w = __fadd_rn(acc.w, x.w);
  }
}

template <int LPE, int NV, int MODE>
__global__ __launch_bounds__(OCN_BLOCK) void spmm_csr_kernel(
    const i64* __restrict__ rowptr, const int32_t* __restrict__ col, i64 n_rows,
    const float* __restrict__ val, const float* __restrict__ x, int F, const float* __restrict__ pre,
    const float* __restrict__ post, int edge_scale, int self_mode, float* __restrict__ y) {
  constexpr int GPW = OCN_WAVE / LPE;
  constexpr int UNR = 4;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPE;
  const int gbase = lane - gl;
  const i64 r = ((i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6)) * GPW + lane / LPE;
  if (r >= n_rows) return;
  const i64 a0 = rowptr[r], da = rowptr[r + 1] - a0;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const i64 rowq = F >> 2;
  const bool weighted = pre != nullptr || val != nullptr;
  const float pr = pre ? pre[r] : 1.0f;

  float4 acc[NV];
  const float init = MODE == SPMM_MAX ? -INFINITY : 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = make_float4(init, init, init, init);

  // the row's own term: weight pre[r] (edge_scale 0) or fl(pre[r]*pre[r]) (edge_scale 1)
  const float wself = pre ? (edge_scale ? __fmul_rn(pr, pr) : pr) : 1.0f;
  bool self_done = self_mode != 2;
  i64 seen = 0;

  for (i64 p0 = 0; p0 < da; p0 += LPE) {
    const i64 p = p0 + gl;
    int32_t k = 0;
    float wk = 1.0f;
    if (p < da) {
      k = col[a0 + p];
      if (pre) wk = edge_scale ? __fmul_rn(pr, pre[k]) : pre[k];
      if (val) wk = __fmul_rn(val[a0 + p], wk);
    }
    const int cnt = (int)((da - p0) < LPE ? (da - p0) : LPE);
    for (int b0 = 0; b0 < cnt; b0 += UNR) {
      int32_t kk[UNR];
      float ww[UNR];
      float4 xv[UNR][NV];
#pragma unroll
      for (int t = 0; t < UNR; ++t) {
        const int b = b0 + t;
        const int sl = gbase + (b < cnt ? b : 0);
        kk[t] = __shfl(k, sl, OCN_WAVE);
        ww[t] = __shfl(wk, sl, OCN_WAVE);
        if (b < cnt) {
          const float4* row = x4 + (i64)kk[t] * rowq + gl;
#pragma unroll
          for (int v = 0; v < NV; ++v) xv[t][v] = row[v * LPE];
        }
      }
#pragma unroll
      for (int t = 0; t < UNR; ++t) {
        if (b0 + t < cnt) {
          if (!self_done && (i64)kk[t] >= r) {
            // sorted position of the diagonal (fill_diag): add it before the first column >= r
            const float4* row = x4 + r * rowq + gl;
#pragma unroll
            for (int v = 0; v < NV; ++v) red4<MODE>(acc[v], wself, weighted, row[v * LPE]);
            self_done = true;
            ++seen;
            if ((i64)kk[t] == r) continue;   // an explicit self loop is replaced, not doubled
          }
#pragma unroll
          for (int v = 0; v < NV; ++v) red4<MODE>(acc[v], ww[t], weighted, xv[t][v]);
          ++seen;
        }
      }
    }
  }
  if (self_mode == 1 || !self_done) {
    const float4* row = x4 + r * rowq + gl;
#pragma unroll
    for (int v = 0; v < NV; ++v) red4<MODE>(acc[v], wself, weighted, row[v * LPE]);
    ++seen;
  }
  const float po = post ? post[r] : 1.0f;
  float4* o = reinterpret_cast<float4*>(y) + r * rowq + gl;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    float4 a = acc[v];
    if (MODE == SPMM_MEAN) {
      const float d = (float)(seen > 0 ? seen : 1);
      a.x /= d; a.y /= d; a.z /= d; a.w /= d;
    }
    if (MODE == SPMM_MAX && seen == 0) a = make_float4(0.f, 0.f, 0.f, 0.f);
    if (post) { a.x = __fmul_rn(po, a.x); a.y = __fmul_rn(po, a.y); a.z = __fmul_rn(po, a.z); a.w = __fmul_rn(po, a.w); }
    o[v * LPE] = a;
  }
}

extern "C" {

#define SPMM_ARGS (const i64*)rowptr, col, (i64)n_rows, val, x, (int)F, pre, post, (int)edge_scale, \
                  (int)self_mode, y
#define LAUNCH_SPMM(LPE, NV, MODE)                                                                  \
  do {                                                                                              \
    const i64 rpb = (i64)OCN_WPB * (OCN_WAVE / (LPE));                                              \
    hipLaunchKernelGGL((spmm_csr_kernel<LPE, NV, MODE>), dim3((unsigned)((n_rows + rpb - 1) / rpb)),\
                       dim3(OCN_BLOCK), 0, st, SPMM_ARGS);                                          \
  } while (0)
#define DISPATCH_SPMM(MODE)                                                                         \
  do {                                                                                              \
    if (F == 16) LAUNCH_SPMM(4, 1, MODE);                                                           \
    else if (F == 32) LAUNCH_SPMM(8, 1, MODE);                                                      \
    else if (F == 64) LAUNCH_SPMM(16, 1, MODE);                                                     \
    else if (F == 128) LAUNCH_SPMM(32, 1, MODE);                                                    \
    else if (F == 256) LAUNCH_SPMM(64, 1, MODE);                                                    \
    else if (F == 512) LAUNCH_SPMM(64, 2, MODE);                                                    \
    else return OCN_EINVAL; /* feature widths of the reference configs only (16..512, pow2) */      \
  } while (0)

int ocn_spmm_csr(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows, const float* x,
                 int32_t F, const float* pre, const float* post, int32_t mode, int32_t edge_scale,
                 int32_t self_mode, float* y, void* stream) {
  if (n_rows < 0 || F <= 0 || mode < 0 || mode > 2 || self_mode < 0 || self_mode > 2) return OCN_EINVAL;
  if (n_rows == 0) return 0;
  if (!rowptr || !x || !y) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (mode == SPMM_SUM) DISPATCH_SPMM(SPMM_SUM);
  else if (mode == SPMM_MEAN) DISPATCH_SPMM(SPMM_MEAN);
  else DISPATCH_SPMM(SPMM_MAX);
  return launch_status();
}

int ocn_deg_rsqrt(const int64_t* rowptr, const float* val, int64_t n_rows, float add, float* out,
                  void* stream) {
  if (n_rows < 0 || (n_rows > 0 && (!rowptr || !out))) return OCN_EINVAL;
  if (n_rows == 0) return 0;
  const int grid = grid_for((n_rows + OCN_BLOCK - 1) / OCN_BLOCK, 2048);
  hipLaunchKernelGGL(deg_rsqrt_kernel, dim3(grid), dim3(OCN_BLOCK), 0, (hipStream_t)stream,
                     (const i64*)rowptr, val, (i64)n_rows, add, out);
  return launch_status();
}

}  // extern "C"
