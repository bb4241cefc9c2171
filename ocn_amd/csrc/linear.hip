// Dense Linear layers of the MLP heads (model.py:2203-2235) on the gfx950 matrix cores.
//
//   Y[M,N] = epilogue( X[M,K] · Wᵀ + bias )        X, W, Y fp32; W is nn.Linear's [N][K]
//
// fp32 operands are split into three bf16 terms each (x = x1 + x2 + x3 to 2^-27 relative) and the
// product is formed from the six leading cross terms on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation ("bf16x6"): the dropped terms are below 2^-26 of |x||w|, i.e. under fp32's own
// rounding, at 6/16 of the cost of the f32-input MFMA.  This is the one MFMA-shaped piece of the
// path.
//
// Tiling: a workgroup = 4 waves = 128 rows; each wave owns 32 full rows x all N columns, so the
// LayerNorm / ReLU / final-dot epilogues are row-local (in-wave shuffles, no LDS).  X fragments go
// straight from memory to registers (each element is used by exactly one wave); the pre-split,
// fragment-ordered weight panel of each 16-deep k-step is staged in LDS and shared by the 4 waves.
#include "common.h"

#define LIN_KS 16                    // k per MFMA step
#define LIN_ROWS 128                 // rows per workgroup
#ifndef LIN_PF
#define LIN_PF 4                     // X fragments in flight per lane (k-steps of look-ahead)
#endif

__device__ __forceinline__ void split_frag(const float4& xa, const float4& xb, bf16x8& a1, bf16x8& a2, bf16x8& a3) {
  const float xs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 p, q, u;
    split3(xs[j], p, q, u);
    a1[j] = p; a2[j] = q; a3[j] = u;
  }
}

// Wp[s][t][split][lane][8]: the B fragment (k = 16s + 8(lane>>5) + j, n = 32t + (lane&31)) of
// split `split`, so that one k-step's panel is a contiguous, lane-linear LDS image.
__global__ __launch_bounds__(OCN_BLOCK) void split_weight_kernel(const float* __restrict__ W, int N, int K,
                                                                 __bf16* __restrict__ Wp) {
  const int NT = N >> 5;
  const i64 total = (i64)(K / LIN_KS) * NT * 64;           // fragments
  for (i64 f = (i64)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (i64)gridDim.x * blockDim.x) {
    const int lane = (int)(f & 63);
    const int t = (int)((f >> 6) % NT);
    const int s = (int)((f >> 6) / NT);
    const float* src = W + (i64)(32 * t + (lane & 31)) * K + LIN_KS * s + 8 * (lane >> 5);
    bf16x8 p[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 a, b, c;
      split3(src[j], a, b, c);
      p[0][j] = a; p[1][j] = b; p[2][j] = c;
    }
    bf16x8* dst = reinterpret_cast<bf16x8*>(Wp) + ((i64)(s * NT + t) * 3) * 64 + lane;
    dst[0] = p[0];
    dst[64] = p[1];
    dst[128] = p[2];
  }
}

// One launch can carry up to LIN_MAX_GROUPS independent Linear layers of the same (K, N): each
// group has its own rows, weights and epilogue; workgroup tiles are numbered group after group.
#define LIN_MAX_GROUPS 5
struct LinGroup {
  const float* X; i64 ldX; i64 M;            // input rows [M][K], row stride ldX floats
  const __bf16* Wp;                          // pre-split weight panel
  const float *bias, *gamma, *beta;          // + bias; LayerNorm (gamma/beta or both NULL)
  const float* scale;                        // device scalar multiplied in after ReLU (or NULL)
  const float* addend; i64 ldAdd;            // [M][N] tensor added last (or NULL)
  const float *dotw, *dotb;                  // trailing Linear(N -> 1): Y becomes [M]
  float* Y; i64 ldY;                         // output rows, row stride ldY floats
  const i64* row_range;                      // device {begin, end} within [0, M], or NULL
  const i64* y_row_map;                      // dot epilogue: destination row, or NULL
  float eps; int relu; int tiles;            // tiles = ceil(M / LIN_ROWS)
  int add_bcast;
};
struct LinArgs { int n_groups; int K; LinGroup g[LIN_MAX_GROUPS]; };

template <int NT>
__global__ __launch_bounds__(OCN_BLOCK, 2) void linear_bf16x6_kernel(const LinArgs args) {
  constexpr int N = NT * 32;
  constexpr int PANEL = NT * 3 * 64;                      // bf16x8 fragments per k-step panel
  __shared__ __attribute__((aligned(16))) bf16x8 wbuf[2][PANEL];
  __shared__ float s_ep[4][N];                            // bias, gamma, beta, dotw: read once, up front
  int tile = blockIdx.x, gi = 0;
  while (gi + 1 < args.n_groups && tile >= args.g[gi].tiles) { tile -= args.g[gi].tiles; ++gi; }
  const LinGroup& G = args.g[gi];
  const float* __restrict__ X = G.X;
  i64 r_begin = 0, M = G.M;                               // rows [r_begin, M) of this group
  if (G.row_range) {
    const i64 rb = G.row_range[0], re = G.row_range[1];
    r_begin = rb < 0 ? 0 : rb;
    M = re < M ? re : M;
  }
  if (r_begin + (i64)tile * LIN_ROWS >= M) return;        // whole workgroup: nothing in this tile
  const int K = args.K;
  const float* __restrict__ bias = G.bias;
  const float* __restrict__ gamma = G.gamma;
  const float* __restrict__ beta = G.beta;
  const float* __restrict__ dotw = G.dotw;
  const float* __restrict__ dotb = G.dotb;
  float* __restrict__ Y = G.Y;
  const float eps = G.eps;
  const int relu = G.relu;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const i64 row0 = r_begin + (i64)tile * LIN_ROWS + 32 * w;
  i64 arow = row0 + r;
  if (arow >= M) arow = M - 1;                            // tail rows: load something valid, never store
  const float4* xrow = reinterpret_cast<const float4*>(X + arow * G.ldX + 8 * hh);
  const bf16x8* wp8 = reinterpret_cast<const bf16x8*>(G.Wp);
  const int nks = K / LIN_KS;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  for (int c = threadIdx.x; c < N; c += OCN_BLOCK) {
    s_ep[0][c] = bias ? bias[c] : 0.f;
    s_ep[1][c] = gamma ? gamma[c] : 1.f;
    s_ep[2][c] = beta ? beta[c] : 0.f;
    s_ep[3][c] = dotw ? dotw[c] : 0.f;
  }
  // prologue: panel 0 -> LDS, the first LIN_PF X fragments -> a register ring
  for (int f = threadIdx.x; f < PANEL; f += OCN_BLOCK) wbuf[0][f] = wp8[f];
  float4 xr[LIN_PF][2];
#pragma unroll
  for (int d = 0; d < LIN_PF; ++d) {
    const int sd = d < nks ? d : nks - 1;
    xr[d][0] = xrow[sd * (LIN_KS / 4)];
    xr[d][1] = xrow[sd * (LIN_KS / 4) + 1];
  }
  __syncthreads();

  // X streams from HBM exactly once (latency ~1-2 us under load), so its fragments are requested
  // LIN_PF k-steps ahead; the weight panel comes from L2 and is staged one step ahead.  The bf16
  // split of step s+1's fragment is issued in the shadow of step s's MFMAs (VALU and MFMA pipes
  // run side by side); done at the top of the step it would idle the matrix pipe ~25 % of the time.
  bf16x8 a1, a2, a3;
  split_frag(xr[0][0], xr[0][1], a1, a2, a3);
#pragma unroll 1
  for (int s0 = 0; s0 < nks; s0 += LIN_PF) {
#pragma unroll
    for (int d = 0; d < LIN_PF; ++d) {
      const int s = s0 + d;
      if (s < nks) {
        const int cur = s & 1;
        bf16x8 stage[(PANEL + OCN_BLOCK - 1) / OCN_BLOCK];
#ifndef OCN_X_LIN_NOW    /* timing experiments: no weight-panel reload / no X reload */
        if (s + 1 < nks) {
#pragma unroll
          for (int q = 0; q < (PANEL + OCN_BLOCK - 1) / OCN_BLOCK; ++q) {
            const int f = threadIdx.x + q * OCN_BLOCK;
            if (f < PANEL) stage[q] = wp8[(i64)(s + 1) * PANEL + f];
          }
        }
#endif
#ifndef OCN_X_LIN_NOX
        if (s + LIN_PF < nks) {                            // slot d was consumed when step s was split
          xr[d][0] = xrow[(s + LIN_PF) * (LIN_KS / 4)];
          xr[d][1] = xrow[(s + LIN_PF) * (LIN_KS / 4) + 1];
        }
#endif
        bf16x8 n1 = a1, n2 = a2, n3 = a3;
        // B fragments are fetched from LDS one column tile ahead of the MFMAs that consume them
        bf16x8 bq[2][3];
#pragma unroll
        for (int u = 0; u < 3; ++u) bq[0][u] = wbuf[cur][u * 64 + lane];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (t + 1 < NT) {
#pragma unroll
            for (int u = 0; u < 3; ++u) bq[(t + 1) & 1][u] = wbuf[cur][((t + 1) * 3 + u) * 64 + lane];
          }
          __builtin_amdgcn_sched_barrier(0);               // keep the next tile's LDS reads ahead of these MFMAs
          const bf16x8 b1 = bq[t & 1][0], b2 = bq[t & 1][1], b3 = bq[t & 1][2];
          // smallest cross terms first
#ifndef OCN_X_LIN_T3    /* timing experiment: three cross terms only (what a 2-way split would issue) */
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[t], 0, 0, 0);
#endif
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[t], 0, 0, 0);
          if (t == 0 && s + 1 < nks) {
            const int dn = (d + 1) % LIN_PF;
            split_frag(xr[dn][0], xr[dn][1], n1, n2, n3);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#ifndef OCN_X_LIN_NOW
        if (s + 1 < nks) {
#pragma unroll
          for (int q = 0; q < (PANEL + OCN_BLOCK - 1) / OCN_BLOCK; ++q) {
            const int f = threadIdx.x + q * OCN_BLOCK;
            if (f < PANEL) wbuf[cur ^ 1][f] = stage[q];
          }
        }
#endif
        a1 = n1; a2 = n2; a3 = n3;
#ifndef OCN_X_LIN_NOBAR
        __syncthreads();
#endif
      }
    }
  }

  // ---- epilogue: lane holds column c = 32t + r of rows (i&3) + 8(i>>2) + 4hh, i = 0..15 ----
  if (bias) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float bv = s_ep[0][32 * t + r];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] += bv;
    }
  }
  if (gamma) {
    // Row statistics for the 16 rows a lane touches, butterflied together: the 16 independent
    // shuffles of each step pipeline, where 16 separate 5-step chains would each wait on LDS.
    float sum[16], sq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float a = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) a += acc[t][i];
      sum[i] = a;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < 16; ++i) sum[i] += __shfl_xor(sum[i], o, OCN_WAVE);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      sum[i] = sum[i] * (1.0f / (float)N);                 // mean (N is a power of two: exact)
      float q = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) { const float d = acc[t][i] - sum[i]; q += d * d; }
      sq[i] = q;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1)
#pragma unroll
      for (int i = 0; i < 16; ++i) sq[i] += __shfl_xor(sq[i], o, OCN_WAVE);
#pragma unroll
    for (int i = 0; i < 16; ++i) sq[i] = rsqrtf(sq[i] * (1.0f / (float)N) + eps);   // rstd, as torch's LayerNorm kernel
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const float g = s_ep[1][32 * t + r], be = s_ep[2][32 * t + r];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = (acc[t][i] - sum[i]) * sq[i] * g + be;
    }
  }
  if (relu) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = fmaxf(acc[t][i], 0.f);
  }
  if (G.scale) {
    const float sc = G.scale[0];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] *= sc;
  }
  if (dotw) {
    // trailing Linear(N -> 1): y[row] = <row, dotw> + dotb, one float per row
    float wv[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wv[t] = s_ep[3][32 * t + r];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float d = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) d += acc[t][i] * wv[t];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) d += __shfl_xor(d, o, OCN_WAVE);
      const i64 row = row0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
      if (r == 0 && row < M) Y[G.y_row_map ? G.y_row_map[row] : row] = d + (dotb ? dotb[0] : 0.f);
    }
    return;
  }
#ifdef OCN_X_LIN_NOSTORE   /* timing experiment: keep the accumulators live, store one word */
  { float z = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) z += acc[t][i];
    if (z == 12345.678f) Y[0] = z;
    return; }
#endif
#ifdef OCN_X_LIN_DIRECTSTORE   /* timing experiment: 4-byte stores straight from the accumulator layout */
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const i64 row = row0 + (i & 3) + 8 * (i >> 2) + 4 * hh;
    if (row < M) {
      float* yr = Y + row * G.ldY + r;
#pragma unroll
      for (int t = 0; t < NT; ++t)
        yr[32 * t] = acc[t][i] + (G.addend ? G.addend[(G.add_bcast ? 0 : row * G.ldAdd) + 32 * t + r] : 0.f);
    }
  }
#else
  // The accumulator layout gives each lane one column of 16 rows: 128 four-byte stores per lane.
  // Instead each wave transposes 8 rows at a time through its slice of the (now idle) weight
  // buffers and writes them back as whole rows, 16 bytes per lane, 1 KiB per wave-instruction.
  float* slab = reinterpret_cast<float*>(&wbuf[0][0]) + w * (8 * N);     // 8 rows x N floats per wave
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii)
#pragma unroll
      for (int t = 0; t < NT; ++t) slab[(ii + 4 * hh) * N + 32 * t + r] = acc[t][4 * gq + ii];
    __syncthreads();
    const float4* t4 = reinterpret_cast<const float4*>(slab);
#pragma unroll
    for (int u = 0; u < (8 * N / 4) / OCN_WAVE; ++u) {
      const int q = lane + OCN_WAVE * u;                     // float4 index inside the 8 x N slab
      const i64 row = row0 + 8 * gq + q / (N / 4);
      if (row < M) {
        float4 v = t4[q];
        if (G.addend) {
          const float4 ad = reinterpret_cast<const float4*>(G.addend + (G.add_bcast ? 0 : row * G.ldAdd))[q % (N / 4)];
          v.x += ad.x; v.y += ad.y; v.z += ad.z; v.w += ad.w;
        }
        reinterpret_cast<float4*>(Y + row * G.ldY)[q % (N / 4)] = v;
      }
    }
    __syncthreads();
  }
#endif
}

extern "C" {

int64_t ocn_linear_panel_bytes(int32_t N, int32_t K) { return (int64_t)N * K * 3 * 2; }

int ocn_linear_split_weight(const float* W, int32_t N, int32_t K, void* Wp, void* stream) {
  if (!W || !Wp || N <= 0 || K <= 0 || (N & 31) || (K % LIN_KS)) return OCN_EINVAL;
  const i64 frags = (i64)(K / LIN_KS) * (N >> 5) * 64;
  hipLaunchKernelGGL(split_weight_kernel, dim3(grid_for((frags + OCN_BLOCK - 1) / OCN_BLOCK, 1024)),
                     dim3(OCN_BLOCK), 0, (hipStream_t)stream, W, (int)N, (int)K, (__bf16*)Wp);
  return launch_status();
}

static int launch_linear(const LinArgs& a, int N, hipStream_t st) {
  int tiles = 0;
  for (int g = 0; g < a.n_groups; ++g) tiles += a.g[g].tiles;
  if (tiles == 0) return 0;
  switch (N) {
    case 32:  hipLaunchKernelGGL((linear_bf16x6_kernel<1>), dim3(tiles), dim3(OCN_BLOCK), 0, st, a); break;
    case 64:  hipLaunchKernelGGL((linear_bf16x6_kernel<2>), dim3(tiles), dim3(OCN_BLOCK), 0, st, a); break;
    case 128: hipLaunchKernelGGL((linear_bf16x6_kernel<4>), dim3(tiles), dim3(OCN_BLOCK), 0, st, a); break;
    case 256: hipLaunchKernelGGL((linear_bf16x6_kernel<8>), dim3(tiles), dim3(OCN_BLOCK), 0, st, a); break;
    default: return OCN_EINVAL;
  }
  return launch_status();
}

static int fill_group(LinGroup& g, const OcnLinearGroup& s, int K, int N) {
  if (s.M < 0 || (s.M > 0 && (!s.X || !s.Wp || !s.Y))) return OCN_EINVAL;
  if ((s.gamma == nullptr) != (s.beta == nullptr)) return OCN_EINVAL;
  const int64_t ldX = s.ldX ? s.ldX : K, ldY = s.ldY ? s.ldY : (s.dotw ? 1 : N), ldA = s.ldAdd ? s.ldAdd : N;
  if (ldX < K || (ldX & 3) || (!s.dotw && (ldY < N || (ldY & 3))) || (s.addend && (ldA < N || (ldA & 3)))) return OCN_EINVAL;
  g.X = s.X; g.ldX = ldX; g.M = s.M; g.Wp = (const __bf16*)s.Wp;
  g.bias = s.bias; g.gamma = s.gamma; g.beta = s.beta; g.scale = s.scale;
  g.addend = s.addend; g.ldAdd = ldA; g.dotw = s.dotw; g.dotb = s.dotb;
  g.Y = s.Y; g.ldY = ldY; g.eps = s.eps; g.relu = s.relu;
  g.row_range = (const i64*)s.row_range; g.y_row_map = (const i64*)s.y_row_map; g.add_bcast = s.add_bcast;
  g.tiles = (int)((s.M + LIN_ROWS - 1) / LIN_ROWS);
  return 0;
}

int ocn_linear_grouped(const OcnLinearGroup* groups, int32_t n_groups, int32_t K, int32_t N, void* stream) {
  if (!groups || n_groups < 1 || n_groups > LIN_MAX_GROUPS || K <= 0 || (K % LIN_KS)) return OCN_EINVAL;
  LinArgs a;
  a.n_groups = n_groups;
  a.K = K;
  for (int g = 0; g < n_groups; ++g) {
    const int rc = fill_group(a.g[g], groups[g], K, N);
    if (rc) return rc;
  }
  return launch_linear(a, N, (hipStream_t)stream);
}

int ocn_linear_bf16x6(const float* X, int64_t M, int32_t K, const void* Wp, int32_t N,
                      const float* bias, const float* gamma, const float* beta, float eps,
                      int32_t relu, const float* dotw, const float* dotb, float* Y, void* stream) {
  OcnLinearGroup g = {};
  g.X = X; g.M = M; g.Wp = Wp; g.bias = bias; g.gamma = gamma; g.beta = beta; g.eps = eps; g.relu = relu;
  g.dotw = dotw; g.dotb = dotb; g.Y = Y;
  return ocn_linear_grouped(&g, 1, K, N, stream);
}

}  // extern "C"
