// Weight gradient of the heads' Linear layers (autograd of model.py:2203-2235 in the training loop,
// NeighborOverlap_large.py:76-90):   dW[N][K] = dYᵀ · X = Σ_b dY[b][:]ᵀ X[b][:],   db[N] = Σ_b dY[b][:]
//
// A contraction over the whole batch (65 536 rows for a 256 x 256 result): split over the batch.  Every WAVE owns one
// 64 x 64 tile of dW for one slice of rows — no LDS, no barrier: both operands have the contraction index as their
// ROW index, and a lane's MFMA fragment is 8 consecutive rows of ONE column (lanes = 32 consecutive columns), so the
// fragments come straight from memory as 8 dword loads of 2 x 128 contiguous bytes per wave.  fp32 operands are split
// into three bf16 terms and multiplied as the six leading cross terms on v_mfma_f32_32x32x16_bf16 (as linear.hip).
// The slices' partial tiles go to a workspace and a second kernel adds them in slice order: no float atomics, the same
// bits on every run.  db rides along in the waves of the first column tile.
#include "common.h"

#define WG_TILE 64

__device__ __forceinline__ void wg_split(const float (&x)[8], bf16x8& a1, bf16x8& a2, bf16x8& a3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 p, q, u;
    split3(x[j], p, q, u);
    a1[j] = p; a2[j] = q; a3[j] = u;
  }
}

// rows [b, b + 16) of two 32-column tiles: lane (r, hh) takes rows b + 8 hh + j of column c[t]
__device__ __forceinline__ void wg_load(const float* __restrict__ base, i64 ld, const int (&c)[2], i64 b, int hh, i64 B,
                                        float (&out)[2][8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const i64 row = b + 8 * hh + j;
    const i64 rr = row < B ? row : B - 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float v = base[rr * ld + c[t]];
      out[t][j] = row < B ? v : 0.0f;
    }
  }
}

__global__ __launch_bounds__(OCN_BLOCK) void wgrad_bf16x6_kernel(const float* __restrict__ dY, i64 ldY, const float* __restrict__ X, i64 ldX,
                                                                 i64 B, int N, int K, int tn, int tk, int S, i64 R,
                                                                 float* __restrict__ part, float* __restrict__ bpart) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const i64 gw = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6);
  const i64 tiles = (i64)tn * tk;
  if (gw >= tiles * S) return;
  const int s = (int)(gw / tiles), t = (int)(gw % tiles);
  const int n0 = WG_TILE * (t / tk), k0 = WG_TILE * (t % tk);
  const i64 b0 = (i64)s * R;
  const i64 b1 = b0 + R < B ? b0 + R : B;
  int nc[2], kc[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    nc[q] = n0 + 32 * q + r < N ? n0 + 32 * q + r : N - 1;        // columns past the edge: a valid address, never stored
    kc[q] = k0 + 32 * q + r < K ? k0 + 32 * q + r : K - 1;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
  float bs[2] = {0.f, 0.f};
  const bool with_bias = bpart && k0 == 0;

  float ra[2][8], rb[2][8];
  wg_load(dY, ldY, nc, b0, hh, B, ra);
  wg_load(X, ldX, kc, b0, hh, B, rb);
#pragma unroll 1
  for (i64 b = b0; b < b1; b += 16) {
    float na[2][8], nb[2][8];
    const i64 bn = b + 16 < b1 ? b + 16 : b;                      // the last step re-reads its own rows (cached) and drops them
    wg_load(dY, ldY, nc, bn, hh, B, na);
    wg_load(X, ldX, kc, bn, hh, B, nb);
    bf16x8 a1[2], a2[2], a3[2], w1[2], w2[2], w3[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      wg_split(ra[q], a1[q], a2[q], a3[q]);
      wg_split(rb[q], w1[q], w2[q], w3[q]);
    }
    if (with_bias) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) bs[q] += ra[q][j];
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ki = 0; ki < 2; ++ki) {
        f32x16 c = acc[mi][ki];                                   // smallest cross terms first
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[mi], w2[ki], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[mi], w3[ki], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[mi], w1[ki], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[mi], w2[ki], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[mi], w1[ki], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[mi], w1[ki], c, 0, 0, 0);
        acc[mi][ki] = c;
      }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) { ra[q][j] = na[q][j]; rb[q][j] = nb[q][j]; }
  }
  // lane holds column k0 + 32 ki + r of rows n0 + 32 mi + (i & 3) + 8 (i >> 2) + 4 hh
  float* P = part + (i64)s * N * K;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) {
      const int k = k0 + 32 * ki + r;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int n = n0 + 32 * mi + (i & 3) + 8 * (i >> 2) + 4 * hh;
        if (n < N && k < K) P[(i64)n * K + k] = acc[mi][ki][i];
      }
    }
  if (with_bias) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float tot = bs[q] + __shfl_xor(bs[q], 32, OCN_WAVE);  // rows 8hh.. of both halves: a fixed order
      const int n = n0 + 32 * q + r;
      if (hh == 0 && n < N) bpart[(i64)s * N + n] = tot;
    }
  }
}

// out[e] = part[0][e] + part[1][e] + ... in slice order
__global__ __launch_bounds__(OCN_BLOCK) void wgrad_reduce_kernel(const float* __restrict__ part, int S, i64 n, float* __restrict__ out) {
  for (i64 e = (i64)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (i64)gridDim.x * blockDim.x) {
    float a = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = part[(i64)(s + q) * n + e];
#pragma unroll
      for (int q = 0; q < 8; ++q) a = __fadd_rn(a, v[q]);
    }
    for (; s < S; ++s) a = __fadd_rn(a, part[(i64)s * n + e]);
    out[e] = a;
  }
}

extern "C" {

// slices of the batch: enough waves to fill the SIMDs twice, at least 64 rows each, a multiple of 16 rows
static void wgrad_plan(int64_t B, int32_t N, int32_t K, int& tn, int& tk, int& S, int64_t& R) {
  tn = (N + WG_TILE - 1) / WG_TILE;
  tk = (K + WG_TILE - 1) / WG_TILE;
  const int64_t tiles = (int64_t)tn * tk;
  int64_t want = (2048 + tiles - 1) / tiles;
  const int64_t most = (B + 63) / 64;
  if (want > most) want = most;
  if (want < 1) want = 1;
  R = ((B + want - 1) / want + 15) / 16 * 16;
  if (R < 16) R = 16;
  S = (int)((B + R - 1) / R);
  if (S < 1) S = 1;
}

int64_t ocn_wgrad_workspace_bytes(int64_t B, int32_t N, int32_t K) {
  if (B < 0 || N <= 0 || K <= 0) return 0;
  int tn, tk, S; int64_t R;
  wgrad_plan(B, N, K, tn, tk, S, R);
  return (int64_t)S * ((int64_t)N * K + N) * 4 + 64;
}

int ocn_wgrad(const float* dY, int64_t ldY, const float* X, int64_t ldX, int64_t B, int32_t N, int32_t K,
              float* dW, float* db, void* workspace, void* stream) {
  if (B < 0 || N <= 0 || K <= 0 || !dW || !workspace || ldY < N || ldX < K) return OCN_EINVAL;
  if (B > 0 && (!dY || !X)) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {                                // an empty batch: zeros, from the reduce kernel over no slices (the library issues no memset)
    const int64_t nk0 = (int64_t)N * K;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for((nk0 + OCN_BLOCK - 1) / OCN_BLOCK, 4096)), dim3(OCN_BLOCK), 0, st,
                       (const float*)workspace, 0, (i64)nk0, dW);
    if (db)
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for((N + OCN_BLOCK - 1) / OCN_BLOCK, 64)), dim3(OCN_BLOCK), 0, st,
                         (const float*)workspace, 0, (i64)N, db);
    return launch_status();
  }
  int tn, tk, S; int64_t R;
  wgrad_plan(B, N, K, tn, tk, S, R);
  float* part = (float*)workspace;
  float* bpart = part + (int64_t)S * N * K;
  const int64_t waves = (int64_t)tn * tk * S;
  hipLaunchKernelGGL(wgrad_bf16x6_kernel, dim3((unsigned)((waves + OCN_WPB - 1) / OCN_WPB)), dim3(OCN_BLOCK), 0, st, dY, (i64)ldY, X,
                     (i64)ldX, (i64)B, (int)N, (int)K, tn, tk, S, (i64)R, part, db ? bpart : (float*)nullptr);
  const int64_t nk = (int64_t)N * K;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for((nk + OCN_BLOCK - 1) / OCN_BLOCK, 4096)), dim3(OCN_BLOCK), 0, st,
                     (const float*)part, S, (i64)nk, dW);
  if (db)
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for((N + OCN_BLOCK - 1) / OCN_BLOCK, 64)), dim3(OCN_BLOCK), 0, st,
                       (const float*)bpart, S, (i64)N, db);
  return launch_status();
}

}  // extern "C"
