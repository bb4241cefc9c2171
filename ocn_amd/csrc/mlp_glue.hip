// Glue kernels between the library GEMMs of the MLP heads.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// MLP-head glue: LayerNorm(+ReLU) over rows and the three-way branch combine
// ---------------------------------------------------------------------------------------------
template <int LPE, int NV>
__global__ __launch_bounds__(OCN_BLOCK) void rows_ln_relu_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    float eps, int relu, i64 rows, int H, float* __restrict__ y) {
  constexpr int GPW = OCN_WAVE / LPE;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPE;
  const i64 r = ((i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6)) * GPW + lane / LPE;
  if (r >= rows) return;
  const i64 rowq = H >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x) + r * rowq + gl;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    v[t] = xr[t * LPE];
    s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
  }
#pragma unroll
  for (int o = LPE / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, OCN_WAVE);
  const float mean = s / (float)H;
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const float a = v[t].x - mean, b = v[t].y - mean, c = v[t].z - mean, d = v[t].w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
#pragma unroll
  for (int o = LPE / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, OCN_WAVE);
  const float rstd = 1.0f / sqrtf(q / (float)H + eps);
  const float4* g4 = reinterpret_cast<const float4*>(gamma) + gl;
  const float4* b4 = reinterpret_cast<const float4*>(beta) + gl;
  float4* yr = reinterpret_cast<float4*>(y) + r * rowq + gl;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    const float4 g = g4[t * LPE], b = b4[t * LPE];
    float4 o;
    o.x = (v[t].x - mean) * rstd * g.x + b.x;
    o.y = (v[t].y - mean) * rstd * g.y + b.y;
    o.z = (v[t].z - mean) * rstd * g.z + b.z;
    o.w = (v[t].w - mean) * rstd * g.w + b.w;
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    yr[t * LPE] = o;
  }
}

// out = c[0]*x1 + c[1]*x2 + c[2]*x3, evaluated left to right like the reference's expression
// (model.py:2436 / 3222); c lives on the device (sigmoid/cumprod of the alpha parameter, beta).
__global__ __launch_bounds__(OCN_BLOCK) void combine3_kernel(const float* __restrict__ c,
                                                             const float4* __restrict__ x1,
                                                             const float4* __restrict__ x2,
                                                             const float4* __restrict__ x3, i64 n4,
                                                             float4* __restrict__ out) {
  const float c0 = c[0], c1 = c[1], c2 = c[2];
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (i64)gridDim.x * blockDim.x) {
    const float4 a = x1[i], b = x2[i], d = x3[i];
    float4 o;
    o.x = (c0 * a.x + c1 * b.x) + c2 * d.x;
    o.y = (c0 * a.y + c1 * b.y) + c2 * d.y;
    o.z = (c0 * a.z + c1 * b.z) + c2 * d.z;
    o.w = (c0 * a.w + c1 * b.w) + c2 * d.w;
    out[i] = o;
  }
}

// dst[row][0 .. 4q) = vec for the rows of a device-side range (the constant activations of the candidates
// whose pooled input is all zero)
__global__ __launch_bounds__(OCN_BLOCK) void fill_rows_kernel(float* __restrict__ dst, i64 ld, int q,
                                                              const float4* __restrict__ vec,
                                                              const i64* __restrict__ row_range, i64 max_rows) {
  i64 rb = row_range[0], re = row_range[1];
  if (rb < 0) rb = 0;
  if (re > max_rows) re = max_rows;
  const i64 n = (re - rb) * q;
  for (i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (i64)gridDim.x * blockDim.x) {
    const i64 row = rb + t / q;
    const int c = (int)(t % q);
    reinterpret_cast<float4*>(dst + row * ld)[c] = vec[c];
  }
}

extern "C" {

#define LAUNCH_LN(LPE, NV)                                                                          \
  do {                                                                                              \
    const i64 rpb = (i64)OCN_WPB * (OCN_WAVE / (LPE));                                              \
    hipLaunchKernelGGL((rows_ln_relu_kernel<LPE, NV>), dim3((unsigned)((rows + rpb - 1) / rpb)),    \
                       dim3(OCN_BLOCK), 0, (hipStream_t)stream, x, gamma, beta, eps, (int)relu,     \
                       (i64)rows, (int)H, y);                                                       \
  } while (0)

int ocn_rows_ln_relu(const float* x, const float* gamma, const float* beta, float eps, int32_t relu,
                     int64_t rows, int32_t H, float* y, void* stream) {
  if (rows < 0 || H <= 0) return OCN_EINVAL;
  if (rows == 0) return 0;
  if (!x || !gamma || !beta || !y) return OCN_EINVAL;
  switch (H) {
    case 16:  LAUNCH_LN(4, 1); break;
    case 32:  LAUNCH_LN(8, 1); break;
    case 64:  LAUNCH_LN(16, 1); break;
    case 128: LAUNCH_LN(32, 1); break;
    case 256: LAUNCH_LN(64, 1); break;
    case 512: LAUNCH_LN(64, 2); break;
    default: return OCN_EINVAL;
  }
  return launch_status();
}

int ocn_fill_rows(float* dst, int64_t ld, int32_t n_cols, const float* vec, const int64_t* row_range,
                  int64_t max_rows, void* stream) {
  if (max_rows < 0 || n_cols <= 0 || (n_cols & 3) || ld < n_cols || (ld & 3)) return OCN_EINVAL;
  if (max_rows == 0) return 0;
  if (!dst || !vec || !row_range) return OCN_EINVAL;
  const i64 q = n_cols >> 2;
  hipLaunchKernelGGL(fill_rows_kernel, dim3(grid_for((max_rows * q + OCN_BLOCK - 1) / OCN_BLOCK, 2048)),
                     dim3(OCN_BLOCK), 0, (hipStream_t)stream, dst, (i64)ld, (int)q, (const float4*)vec,
                     (const i64*)row_range, (i64)max_rows);
  return launch_status();
}

int ocn_combine3(const float* coef, const float* x1, const float* x2, const float* x3, int64_t n,
                 float* out, void* stream) {
  if (n < 0 || (n & 3)) return OCN_EINVAL;
  if (n == 0) return 0;
  if (!coef || !x1 || !x2 || !x3 || !out) return OCN_EINVAL;
  const i64 n4 = n >> 2;
  hipLaunchKernelGGL(combine3_kernel, dim3(grid_for((n4 + OCN_BLOCK - 1) / OCN_BLOCK, 4096)),
                     dim3(OCN_BLOCK), 0, (hipStream_t)stream, coef, (const float4*)x1, (const float4*)x2,
                     (const float4*)x3, n4, (float4*)out);
  return launch_status();
}

}  // extern "C"
