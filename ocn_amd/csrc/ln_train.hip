// The heads' LayerNorm -> Dropout -> ReLU tails under autograd (training drop-in; the nn.Sequential layouts
// of model.py:2203-2235 between two Linear layers): ONE forward launch and TWO backward launches instead of torch's
// LayerNorm / dropout / relu kernels and their backward (about ten elementwise and reduction launches per tail, 4 tails per
// predictor call, 2 calls per training step: 25 % of a collab-shaped training step's kernel time, profiles/r03_train_step_kernel_stats.csv).
//
//   forward   u = LN(x) = (x - mean) rstd gamma + beta   (gamma == NULL: u = x)
//             d = u * keep / (1 - p)                      keep(seed, element) from a counter-based hash: no mask is stored
//             y = max(d, 0)                               (relu == 0: y = d)
//   backward  g' = g [y > 0] keep / (1 - p);  dx = rstd (g' gamma - mean_row(g' gamma) - xhat mean_row(g' gamma xhat))
//             dgamma = sum_rows g' xhat, dbeta = sum_rows g': every lane group walks a contiguous block of rows and keeps its own
//             partial sums, a second launch adds the partials in block order — the same bits on every run (no float atomics).
//
// Dropout here is THIS library's random stream, not torch's Philox: the reference's results under dropout are stochastic
// too (nn.Dropout), nothing pins a particular stream; eval and p = 0 are exact.
#include "common.h"

__device__ __forceinline__ bool drop_keep(u64 seed, u64 idx, unsigned thresh) {
  u64 z = idx + seed * 0x9E3779B97F4A7C15ull;            // splitmix64 finaliser over (seed, element index)
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (unsigned)(z >> 32) >= thresh;                  // thresh = p * 2^32: dropped with probability p
}

template <int LPE, int NV>
__global__ __launch_bounds__(OCN_BLOCK) void ln_drop_relu_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
    unsigned thresh, float scale, u64 seed, int relu, i64 rows, int H, float* __restrict__ y, float* __restrict__ stats) {
  constexpr int GPW = OCN_WAVE / LPE;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPE;
  const i64 r = ((i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6)) * GPW + lane / LPE;
  if (r >= rows) return;
  const i64 rowq = H >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x) + r * rowq + gl;
  float4 v[NV];
  float mean = 0.f, rstd = 1.f;
#pragma unroll
  for (int t = 0; t < NV; ++t) v[t] = xr[t * LPE];
  if (gamma) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) s += (v[t].x + v[t].y) + (v[t].z + v[t].w);
#pragma unroll
    for (int o = LPE / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, OCN_WAVE);
    mean = s / (float)H;
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const float a = v[t].x - mean, b = v[t].y - mean, c = v[t].z - mean, d = v[t].w - mean;
      q += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
    for (int o = LPE / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, OCN_WAVE);
    rstd = 1.0f / sqrtf(q / (float)H + eps);
    if (gl == 0 && stats) { stats[2 * r] = mean; stats[2 * r + 1] = rstd; }
  }
  float4* yr = reinterpret_cast<float4*>(y) + r * rowq + gl;
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    float o[4] = {v[t].x, v[t].y, v[t].z, v[t].w};
    if (gamma) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[gl + t * LPE], b = reinterpret_cast<const float4*>(beta)[gl + t * LPE];
      o[0] = (o[0] - mean) * rstd * g.x + b.x; o[1] = (o[1] - mean) * rstd * g.y + b.y;
      o[2] = (o[2] - mean) * rstd * g.z + b.z; o[3] = (o[3] - mean) * rstd * g.w + b.w;
    }
    if (thresh) {
      const u64 e0 = (u64)r * (u64)H + 4ull * (u64)(gl + t * LPE);
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = drop_keep(seed, e0 + c, thresh) ? o[c] * scale : 0.f;
    }
    if (relu) {
#pragma unroll
      for (int c = 0; c < 4; ++c) o[c] = fmaxf(o[c], 0.f);
    }
    yr[t * LPE] = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// Every lane group owns the contiguous rows [gidx * chunk, (gidx + 1) * chunk): dx row by row, the partial column sums of
// its block in registers, written once to part[gidx][2][H].
template <int LPE, int NV>
__global__ __launch_bounds__(OCN_BLOCK) void ln_drop_relu_bwd_kernel(
    const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ stats,
    const float* __restrict__ gamma, unsigned thresh, float scale, u64 seed, int relu, i64 rows, int H, i64 chunk,
    float* __restrict__ dx, float* __restrict__ part) {
  constexpr int GPW = OCN_WAVE / LPE;
  const int lane = threadIdx.x & 63;
  const int gl = lane % LPE;
  const i64 gidx = ((i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6)) * GPW + lane / LPE;
  const i64 rowq = H >> 2;
  float4 gm[NV], pg[NV], pb[NV];
#pragma unroll
  for (int t = 0; t < NV; ++t) {
    gm[t] = gamma ? reinterpret_cast<const float4*>(gamma)[gl + t * LPE] : make_float4(1.f, 1.f, 1.f, 1.f);
    pg[t] = pb[t] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const i64 r_lo = gidx * chunk, r_hi = r_lo + chunk < rows ? r_lo + chunk : rows;
  for (i64 r = r_lo; r < r_hi; ++r) {
    float gp[NV][4], xh[NV][4];
    float mean = 0.f, rstd = 1.f;
    if (gamma) { mean = stats[2 * r]; rstd = stats[2 * r + 1]; }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      const float4 gv = reinterpret_cast<const float4*>(g)[r * rowq + gl + t * LPE];
      const float4 yv = reinterpret_cast<const float4*>(y)[r * rowq + gl + t * LPE];
      float gg[4] = {gv.x, gv.y, gv.z, gv.w};
      const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
      const u64 e0 = (u64)r * (u64)H + 4ull * (u64)(gl + t * LPE);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (relu && !(yy[c] > 0.f)) gg[c] = 0.f;
        if (thresh) gg[c] = drop_keep(seed, e0 + c, thresh) ? gg[c] * scale : 0.f;
        gp[t][c] = gg[c];
      }
      if (gamma) {
        const float4 xv = reinterpret_cast<const float4*>(x)[r * rowq + gl + t * LPE];
        const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
        const float gc[4] = {gm[t].x, gm[t].y, gm[t].z, gm[t].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          xh[t][c] = (xx[c] - mean) * rstd;
          s1 += gp[t][c] * gc[c];
          s2 += gp[t][c] * gc[c] * xh[t][c];
        }
      }
    }
    if (gamma) {
#pragma unroll
      for (int o = LPE / 2; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, OCN_WAVE); s2 += __shfl_xor(s2, o, OCN_WAVE); }
      const float m1 = s1 / (float)H, m2 = s2 / (float)H;
#pragma unroll
      for (int t = 0; t < NV; ++t) {
        const float gc[4] = {gm[t].x, gm[t].y, gm[t].z, gm[t].w};
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = rstd * (gp[t][c] * gc[c] - m1 - xh[t][c] * m2);
        reinterpret_cast<float4*>(dx)[r * rowq + gl + t * LPE] = make_float4(o[0], o[1], o[2], o[3]);
        pg[t].x += gp[t][0] * xh[t][0]; pg[t].y += gp[t][1] * xh[t][1]; pg[t].z += gp[t][2] * xh[t][2]; pg[t].w += gp[t][3] * xh[t][3];
        pb[t].x += gp[t][0]; pb[t].y += gp[t][1]; pb[t].z += gp[t][2]; pb[t].w += gp[t][3];
      }
    } else {
#pragma unroll
      for (int t = 0; t < NV; ++t)
        reinterpret_cast<float4*>(dx)[r * rowq + gl + t * LPE] = make_float4(gp[t][0], gp[t][1], gp[t][2], gp[t][3]);
    }
  }
  if (gamma && part) {
#pragma unroll
    for (int t = 0; t < NV; ++t) {
      reinterpret_cast<float4*>(part + (2 * gidx) * H)[gl + t * LPE] = pg[t];
      reinterpret_cast<float4*>(part + (2 * gidx + 1) * H)[gl + t * LPE] = pb[t];
    }
  }
}

// dgamma[c] = sum over blocks of part[b][0][c], dbeta likewise, in a FIXED order: a workgroup owns 64 columns of one of the two
// vectors; its wave w of 16 adds the blocks b = w, w + 16, w + 32, ... (64 coalesced floats per block, eight loads in flight),
// the sixteen wave sums are added in wave order.
#define LN_RED_WAVES 16
__global__ __launch_bounds__(LN_RED_WAVES * OCN_WAVE) void ln_partials_reduce_kernel(const float* __restrict__ part, i64 n_blocks, int H,
                                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float sh[LN_RED_WAVES][OCN_WAVE];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int G = (H + OCN_WAVE - 1) / OCN_WAVE;   // column groups of 64 per vector; the grid is 2 G: [dgamma groups | dbeta groups]
  const int which = blockIdx.x / G, col = (blockIdx.x % G) * OCN_WAVE + lane;
  const bool ok = col < H;
  float s = 0.f;
  i64 b = w;
  for (; b + 7 * LN_RED_WAVES < n_blocks; b += 8 * LN_RED_WAVES) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ok ? part[(2 * (b + u * LN_RED_WAVES) + which) * H + col] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; b < n_blocks; b += LN_RED_WAVES) s += ok ? part[(2 * b + which) * H + col] : 0.f;
  sh[w][lane] = s;
  __syncthreads();
  if (w == 0 && ok) {
    float t = sh[0][lane];
#pragma unroll
    for (int q = 1; q < LN_RED_WAVES; ++q) t += sh[q][lane];
    (which ? dbeta : dgamma)[col] = t;
  }
}

// The branch mix z = c0 x1 + c1 x2 + c2 x3 (model.py:2436 / 3222) under autograd: dx_k = c_k g in one pass, and the three dot
// products <g, x_k> (the gradients of the coefficients, which torch's autograd carries on through sigmoid / cumprod) as
// per-workgroup partial sums over contiguous chunks, added in workgroup order by one wave.
#define MIX_GROUPS 1024
__global__ __launch_bounds__(OCN_BLOCK) void mix3_bwd_kernel(const float* __restrict__ coef, const float4* __restrict__ g,
                                                             const float4* __restrict__ x1, const float4* __restrict__ x2,
                                                             const float4* __restrict__ x3, i64 n4, float4* __restrict__ d1,
                                                             float4* __restrict__ d2, float4* __restrict__ d3, float* __restrict__ part) {
  __shared__ float sh[3][OCN_WPB];
  const float c0 = coef[0], c1 = coef[1], c2 = coef[2];
  const i64 chunk = (n4 + MIX_GROUPS - 1) / MIX_GROUPS;
  const i64 lo = (i64)blockIdx.x * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
  float s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (i64 q = lo + threadIdx.x; q < hi; q += OCN_BLOCK) {
    const float4 gv = g[q], a = x1[q], b = x2[q], c = x3[q];
    d1[q] = make_float4(c0 * gv.x, c0 * gv.y, c0 * gv.z, c0 * gv.w);
    d2[q] = make_float4(c1 * gv.x, c1 * gv.y, c1 * gv.z, c1 * gv.w);
    d3[q] = make_float4(c2 * gv.x, c2 * gv.y, c2 * gv.z, c2 * gv.w);
    s1 += (gv.x * a.x + gv.y * a.y) + (gv.z * a.z + gv.w * a.w);
    s2 += (gv.x * b.x + gv.y * b.y) + (gv.z * b.z + gv.w * b.w);
    s3 += (gv.x * c.x + gv.y * c.y) + (gv.z * c.z + gv.w * c.w);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, OCN_WAVE); s2 += __shfl_xor(s2, o, OCN_WAVE); s3 += __shfl_xor(s3, o, OCN_WAVE); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s1; sh[1][threadIdx.x >> 6] = s2; sh[2][threadIdx.x >> 6] = s3; }
  __syncthreads();
  if (threadIdx.x < 3) part[3 * (i64)blockIdx.x + threadIdx.x] = ((sh[threadIdx.x][0] + sh[threadIdx.x][1]) + sh[threadIdx.x][2]) + sh[threadIdx.x][3];
}
__global__ __launch_bounds__(OCN_WAVE) void mix3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dcoef) {
  const int lane = threadIdx.x;
  for (int k = 0; k < 3; ++k) {
    float s = 0.f;
    for (int b = lane; b < MIX_GROUPS; b += OCN_WAVE) s += part[3 * b + k];      // lane l: groups l, l + 64, ... in order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, OCN_WAVE);
    if (lane == 0) dcoef[k] = s;
  }
}

__global__ __launch_bounds__(OCN_BLOCK) void drop_mask_kernel(u64 seed, unsigned thresh, i64 n, uint8_t* __restrict__ out) {
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (i64)gridDim.x * blockDim.x)
    out[q] = (uint8_t)drop_keep(seed, (u64)q, thresh);
}

extern "C" {

static inline unsigned ln_thresh(float p) {
  if (!(p > 0.f)) return 0u;
  const double t = (double)p * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : (unsigned)t;
}

// blocks of rows of the backward: a fixed number of lane groups, whatever the batch (the partial sums' order — hence the bits of
// dgamma / dbeta — depends on it, so it is part of the contract)
#define LN_BWD_GROUPS 4096
int64_t ocn_ln_drop_relu_workspace_bytes(int32_t H) { return (int64_t)LN_BWD_GROUPS * 2 * H * (int64_t)sizeof(float); }

#define LN_DISPATCH(M)                   \
  switch (H) {                           \
    case 16:  M(4, 1); break;            \
    case 32:  M(8, 1); break;            \
    case 64:  M(16, 1); break;           \
    case 128: M(32, 1); break;           \
    case 256: M(64, 1); break;           \
    case 512: M(64, 2); break;           \
    default: return OCN_EINVAL;          \
  }

int ocn_ln_drop_relu_forward(const float* x, const float* gamma, const float* beta, float eps, float p, uint64_t seed, int32_t relu,
                             int64_t rows, int32_t H, float* y, float* stats, void* stream) {
  if (rows < 0 || H <= 0 || !(p >= 0.f) || p >= 1.f || (gamma != nullptr) != (beta != nullptr)) return OCN_EINVAL;
  if (rows == 0) return 0;
  if (!x || !y || (gamma && !stats)) return OCN_EINVAL;
  const unsigned th = ln_thresh(p);
  const float scale = 1.0f / (1.0f - p);
#define LN_FWD(LPE, NV)                                                                                             \
  do {                                                                                                              \
    const i64 rpb = (i64)OCN_WPB * (OCN_WAVE / (LPE));                                                              \
    hipLaunchKernelGGL((ln_drop_relu_fwd_kernel<LPE, NV>), dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(OCN_BLOCK), 0, \
                       (hipStream_t)stream, x, gamma, beta, eps, th, scale, (u64)seed, (int)relu, (i64)rows, (int)H, y, stats); \
  } while (0)
  LN_DISPATCH(LN_FWD)
#undef LN_FWD
  return launch_status();
}

int ocn_ln_drop_relu_backward(const float* g, const float* x, const float* y, const float* stats, const float* gamma, float p,
                              uint64_t seed, int32_t relu, int64_t rows, int32_t H, float* dx, float* dgamma, float* dbeta,
                              void* workspace, void* stream) {
  if (rows < 0 || H <= 0 || !(p >= 0.f) || p >= 1.f) return OCN_EINVAL;
  if (rows == 0) return 0;
  if (!g || !y || !dx || (gamma && (!x || !stats || !dgamma || !dbeta || !workspace))) return OCN_EINVAL;
  const unsigned th = ln_thresh(p);
  const float scale = 1.0f / (1.0f - p);
  const i64 chunk = (rows + LN_BWD_GROUPS - 1) / LN_BWD_GROUPS;
  float* part = (float*)workspace;
#define LN_BWD(LPE, NV)                                                                                             \
  do {                                                                                                              \
    const i64 gpb = (i64)OCN_WPB * (OCN_WAVE / (LPE));                                                              \
    hipLaunchKernelGGL((ln_drop_relu_bwd_kernel<LPE, NV>), dim3((unsigned)(LN_BWD_GROUPS / gpb)), dim3(OCN_BLOCK), 0,    \
                       (hipStream_t)stream, g, x, y, stats, gamma, th, scale, (u64)seed, (int)relu, (i64)rows, (int)H, chunk, dx, part); \
  } while (0)
  LN_DISPATCH(LN_BWD)
#undef LN_BWD
  if (gamma) {
    // column groups of 64 over [dgamma | dbeta]; widths below 64 take one group per vector (lanes beyond H idle)
    const int groups_per_vec = (H + OCN_WAVE - 1) / OCN_WAVE;
    hipLaunchKernelGGL(ln_partials_reduce_kernel, dim3((unsigned)(2 * groups_per_vec)), dim3(LN_RED_WAVES * OCN_WAVE), 0,
                       (hipStream_t)stream, (const float*)part, (i64)LN_BWD_GROUPS, (int)H, dgamma, dbeta);
  }
  return launch_status();
}

int64_t ocn_mix3_workspace_bytes(void) { return (int64_t)MIX_GROUPS * 3 * (int64_t)sizeof(float); }
int ocn_mix3_backward(const float* coef, const float* g, const float* x1, const float* x2, const float* x3, int64_t n,
                      float* d1, float* d2, float* d3, float* dcoef, void* workspace, void* stream) {
  if (n < 0 || (n & 3)) return OCN_EINVAL;
  if (!coef || !dcoef || !workspace || (n > 0 && (!g || !x1 || !x2 || !x3 || !d1 || !d2 || !d3))) return OCN_EINVAL;
  hipLaunchKernelGGL(mix3_bwd_kernel, dim3(MIX_GROUPS), dim3(OCN_BLOCK), 0, (hipStream_t)stream, coef, (const float4*)g, (const float4*)x1,
                     (const float4*)x2, (const float4*)x3, (i64)(n >> 2), (float4*)d1, (float4*)d2, (float4*)d3, (float*)workspace);
  hipLaunchKernelGGL(mix3_reduce_kernel, dim3(1), dim3(OCN_WAVE), 0, (hipStream_t)stream, (const float*)workspace, dcoef);
  return launch_status();
}

// the keep decisions of elements 0 .. n-1 (1 = kept): what tests rebuild the dropout mask from
int ocn_dropout_keep_mask(uint64_t seed, float p, int64_t n, uint8_t* out, void* stream) {
  if (n < 0 || !(p >= 0.f) || p >= 1.f) return OCN_EINVAL;
  if (n == 0) return 0;
  if (!out) return OCN_EINVAL;
  hipLaunchKernelGGL(drop_mask_kernel, dim3(grid_for((n + OCN_BLOCK - 1) / OCN_BLOCK, 4096)), dim3(OCN_BLOCK), 0, (hipStream_t)stream,
                     (u64)seed, ln_thresh(p), (i64)n, out);
  return launch_status();
}

}  // extern "C"
