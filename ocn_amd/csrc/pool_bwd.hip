// Deterministic backward of the pooling with respect to h (training drop-in): the transposed pooling
//   dh[k] += Σ_e  wa(e,k) g1[e] + wb(e,k) g2[e]   over the CN entries (e, k)
//   dh[i] += g3[e] ⊙ h[j],   dh[j] += g3[e] ⊙ h[i]
// summed COLUMN by column in a fixed order instead of with fp32 atomics (cn_scatter_kernel): the batch's entries are
// transposed into per-node lists — count -> chained scan -> fill -> per-list sort by key (rowsort.h) — where the key of a
// CN entry is its flag position (ascending with the batch row) and the keys of a candidate's two endpoint terms follow
// all flag positions; one wave per node then adds its list's terms in ascending key order, every product and sum rounded
// separately.  The same bits on every run; no reference order exists for this sum (torch's own backward of
// spmm / index_select is atomic), so the order is this file's choice.
#include "rowsort.h"

template <bool FILL>
__global__ __launch_bounds__(OCN_BLOCK) void pb_entries_kernel(
    const i64* __restrict__ rowptrA, const int32_t* __restrict__ colA, const i64* __restrict__ src, const i64* __restrict__ dst,
    i64 B, const i64* __restrict__ off, const uint8_t* __restrict__ flags, i64 cap, const i64* __restrict__ col_off,
    int32_t* __restrict__ cursor, int32_t* __restrict__ keys) {
  const int lane = threadIdx.x & 63;
  for (i64 e = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6); e < B; e += (i64)gridDim.x * OCN_WPB) {
    const i64 i = src[e], j = dst[e];
    if (lane < 2) {                                           // the Hadamard term's two ends
      const i64 k = lane ? j : i;
      if (FILL) keys[col_off[k] + atomicAdd(cursor + k, 1)] = (int32_t)(cap + 2 * e + lane);
      else atomicAdd(cursor + k, 1);
    }
    const i64 a0 = rowptrA[i], da = rowptrA[i + 1] - a0, base = off[e];
    if (base + da > cap) continue;                            // row beyond the flag capacity: nothing was written for it
    for (i64 p = lane; p < da; p += OCN_WAVE) {
      if (!flags[base + p]) continue;
      const int32_t k = colA[a0 + p];
      if (FILL) keys[col_off[k] + atomicAdd(cursor + k, 1)] = (int32_t)(base + p);
      else atomicAdd(cursor + k, 1);
    }
  }
}

// One wave per node: lane l owns float4 l (+ 64 v) of the H features.  Per round of 64 keys every lane decodes one
// (batch row by binary search in `off`, weights, the two rows to read), then the terms are added one after the other —
// their rows requested four ahead.
template <int NV>
__global__ __launch_bounds__(OCN_BLOCK) void pb_accumulate_kernel(
    const i64* __restrict__ col_off, const int32_t* __restrict__ keys, i64 N, const i64* __restrict__ src,
    const i64* __restrict__ dst, i64 B, const i64* __restrict__ off, const uint8_t* __restrict__ flags,
    const int32_t* __restrict__ wc, i64 cap, const float4* __restrict__ weights, const float* __restrict__ h, int H,
    const float* __restrict__ g1, const float* __restrict__ g2, const float* __restrict__ g3, float* __restrict__ dh) {
  constexpr int UNR = 4;
  const int lane = threadIdx.x & 63;
  const int q4 = H >> 2;                                      // float4 per row
  for (i64 k = (i64)blockIdx.x * OCN_WPB + (threadIdx.x >> 6); k < N; k += (i64)gridDim.x * OCN_WPB) {
    const i64 b = col_off[k];
    const i64 n = col_off[k + 1] - b;
    if (n == 0) continue;
    const float4 wk = weights[k];
    float4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v)
      acc[v] = lane + 64 * v < q4 ? reinterpret_cast<const float4*>(dh + k * H)[lane + 64 * v] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (i64 q0 = 0; q0 < n; q0 += OCN_WAVE) {
      const int m = (int)(n - q0 < OCN_WAVE ? n - q0 : OCN_WAVE);
      int ra = 0, rb = 0, mode = 0;
      float ca = 0.f, cb = 0.f;
      if (lane < m) {
        const i64 key = keys[b + q0 + lane];
        if (key < cap) {
          i64 lo = 0, hi = B - 1;                             // the batch row whose flag range holds position `key`
          while (lo < hi) {
            const i64 mid = (lo + hi + 1) >> 1;
            if (off[mid] <= key) lo = mid; else hi = mid - 1;
          }
          entry_weights(flags[key], wk, wc ? (float)wc[key] : 1.0f, ca, cb);
          ra = rb = (int)lo;
        } else {
          const i64 t = key - cap;
          const i64 e = t >> 1;
          ra = (int)e;
          rb = (int)((t & 1) ? src[e] : dst[e]);
          mode = 1;
        }
      }
      for (int r0 = 0; r0 < m; r0 += UNR) {
        float4 xa[UNR][NV], xb[UNR][NV];
        float fa[UNR], fb[UNR];
        int md[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          // The look-ahead loads below are issued for all UNR terms, also past the round's end (only their ADDS are skipped):
          // their row indices must come from a lane that decoded a key in THIS round.  A lane >= m decodes nothing; round 3's
          // first form read such lanes here and formed g2 + (a stale endpoint term's NODE id) * H — a row index up to N - 1
          // into a [B, H] buffer: the GPU memory fault of DESIGN.md §6.  Hence the clamp (and the zero-initialised state above).
          const int r = r0 + u < m ? r0 + u : m - 1;
          const int ea = __builtin_amdgcn_readlane(ra, r), eb = __builtin_amdgcn_readlane(rb, r);
          md[u] = __builtin_amdgcn_readlane(mode, r);        // (a round's tail re-reads its last term's rows: valid addresses)
          fa[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ca), r));
          fb[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cb), r));
          const float4* pa = reinterpret_cast<const float4*>((md[u] == 1 ? g3 : g1) + (i64)ea * H);
          const float4* pb = reinterpret_cast<const float4*>((md[u] == 1 ? h : g2) + (i64)eb * H);
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            const int f4 = lane + 64 * v < q4 ? lane + 64 * v : 0;
            xa[u][v] = pa[f4];
            xb[u][v] = pb[f4];
          }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
          if (r0 + u >= m) continue;
#pragma unroll
          for (int v = 0; v < NV; ++v) {
            float4 c;
            if (md[u] == 1) {
              c.x = __fmul_rn(xa[u][v].x, xb[u][v].x); c.y = __fmul_rn(xa[u][v].y, xb[u][v].y);
              c.z = __fmul_rn(xa[u][v].z, xb[u][v].z); c.w = __fmul_rn(xa[u][v].w, xb[u][v].w);
            } else {
              c.x = __fadd_rn(__fmul_rn(fa[u], xa[u][v].x), __fmul_rn(fb[u], xb[u][v].x));
              c.y = __fadd_rn(__fmul_rn(fa[u], xa[u][v].y), __fmul_rn(fb[u], xb[u][v].y));
              c.z = __fadd_rn(__fmul_rn(fa[u], xa[u][v].z), __fmul_rn(fb[u], xb[u][v].z));
              c.w = __fadd_rn(__fmul_rn(fa[u], xa[u][v].w), __fmul_rn(fb[u], xb[u][v].w));
            }
            acc[v].x = __fadd_rn(acc[v].x, c.x); acc[v].y = __fadd_rn(acc[v].y, c.y);
            acc[v].z = __fadd_rn(acc[v].z, c.z); acc[v].w = __fadd_rn(acc[v].w, c.w);
          }
        }
      }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v)
      if (lane + 64 * v < q4) reinterpret_cast<float4*>(dh + k * H)[lane + 64 * v] = acc[v];
  }
}

__global__ __launch_bounds__(OCN_BLOCK) void pb_zero_kernel(int32_t* __restrict__ p, i64 n) {
  for (i64 q = (i64)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += (i64)gridDim.x * blockDim.x) p[q] = 0;
}

extern "C" {

static inline int64_t pb_align(int64_t b) { return (b + 15) / 16 * 16; }

int64_t ocn_cn_gather_backward_det_workspace_bytes(int64_t N, int64_t B, int64_t flags_cap) {
  // col_off int64[N+1] | cursor int32[N] | long_list int32[N] | tickets int32[4] | scan state | keys int32[cap + 2B]
  return pb_align((N + 1) * 8) + 2 * pb_align(N * 4) + 16 + pb_align(ocn_scan_workspace_bytes(N)) + pb_align((flags_cap + 2 * B) * 4) + 64;
}

// workspace layout (ocn_cn_gather_backward_det_workspace_bytes)
struct PbLayout { i64* col_off; int32_t* cursor; int32_t* long_list; int32_t* tickets; void* scan_ws; int32_t* keys; int64_t sw; };
static PbLayout pb_layout(void* workspace, int64_t N) {
  char* ws = (char*)workspace;
  const int64_t a = pb_align((N + 1) * 8), b = pb_align(N * 4), sw = pb_align(ocn_scan_workspace_bytes(N));
  return PbLayout{(i64*)ws, (int32_t*)(ws + a), (int32_t*)(ws + a + b), (int32_t*)(ws + a + 2 * b), (void*)(ws + a + 2 * b + 16),
                  (int32_t*)(ws + a + 2 * b + 16 + sw), sw};
}

// The per-node key lists alone: count -> chained scan -> fill -> per-list sort.  col_off = workspace (int64[N + 1]); the keys
// sit at ocn_cn_gather_backward_det_keys_offset(N) bytes.
static int pb_build_lists(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst, int64_t B,
                          const int64_t* off, const uint8_t* flags, int64_t flags_cap, int64_t N, void* workspace, hipStream_t st) {
  const PbLayout L = pb_layout(workspace, N);
  const int gridN = grid_for((N + OCN_BLOCK - 1) / OCN_BLOCK, 2048);
  const int gridB = grid_for((B + OCN_WPB - 1) / OCN_WPB, 1 << 16);
  const int gridW = grid_for((N + OCN_WPB - 1) / OCN_WPB, 1 << 15);
  hipLaunchKernelGGL(pb_zero_kernel, dim3(gridN), dim3(OCN_BLOCK), 0, st, L.cursor, (i64)N);
  hipLaunchKernelGGL(pb_zero_kernel, dim3(1), dim3(OCN_BLOCK), 0, st, L.tickets, (i64)(4 + L.sw / 4));
#define PB_ENTRIES(FILL)                                                                                              \
  hipLaunchKernelGGL((pb_entries_kernel<FILL>), dim3(gridB), dim3(OCN_BLOCK), 0, st, (const i64*)rowptrA, colA,      \
                     (const i64*)src, (const i64*)dst, (i64)B, (const i64*)off, flags, (i64)flags_cap,               \
                     (const i64*)L.col_off, L.cursor, L.keys)
  PB_ENTRIES(false);
  int rc = ocn_scan_i32(L.cursor, N, (int64_t*)L.col_off, L.scan_ws, (void*)st);
  if (rc) return rc;
  hipLaunchKernelGGL(pb_zero_kernel, dim3(gridN), dim3(OCN_BLOCK), 0, st, L.cursor, (i64)N);
  PB_ENTRIES(true);
#undef PB_ENTRIES
  hipLaunchKernelGGL(cc_sort_short_kernel, dim3(gridW), dim3(OCN_BLOCK), 0, st, (const i64*)L.col_off, (i64)N, L.keys,
                     (int32_t*)nullptr, L.long_list, L.tickets);
  hipLaunchKernelGGL(cc_sort_long_kernel, dim3(256), dim3(OCN_BLOCK), 0, st, (const i64*)L.col_off, L.keys, (int32_t*)nullptr,
                     (const int32_t*)L.long_list, (const int32_t*)L.tickets, L.tickets + 1);
  return launch_status();
}

int64_t ocn_cn_gather_backward_det_keys_offset(int64_t N) {
  return pb_align((N + 1) * 8) + 2 * pb_align(N * 4) + 16 + pb_align(ocn_scan_workspace_bytes(N));
}

int ocn_cn_gather_backward_det_lists(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                                     int64_t B, const int64_t* off, const uint8_t* flags, int64_t flags_cap, int64_t N,
                                     void* workspace, void* stream) {
  if (B < 0 || N < 0 || flags_cap < 0 || flags_cap + 2 * B > 0x7fffffffll) return OCN_EINVAL;
  if (B == 0 || N == 0) return 0;
  if (!rowptrA || !src || !dst || !off || !flags || !workspace) return OCN_EINVAL;
  return pb_build_lists(rowptrA, colA, src, dst, B, off, flags, flags_cap, N, workspace, (hipStream_t)stream);
}

int ocn_cn_gather_backward_det(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                               int64_t B, const int64_t* off, const uint8_t* flags, const int32_t* wc, int64_t flags_cap,
                               const float* weights, const float* h, int64_t N, int32_t H, const float* g1,
                               const float* g2, const float* g3, float* dh, void* workspace, void* stream) {
  if (B < 0 || N < 0 || H <= 0 || (H & 3) || H > 512 || flags_cap < 0 || flags_cap + 2 * B > 0x7fffffffll) return OCN_EINVAL;
  if (B == 0 || N == 0) return 0;
  if (!rowptrA || !src || !dst || !off || !flags || !weights || !h || !g1 || !g2 || !g3 || !dh || !workspace) return OCN_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int rc = pb_build_lists(rowptrA, colA, src, dst, B, off, flags, flags_cap, N, workspace, st);
  if (rc) return rc;
  const PbLayout L = pb_layout(workspace, N);
  const int gridW = grid_for((N + OCN_WPB - 1) / OCN_WPB, 1 << 15);
#define PB_ACC(NV)                                                                                                   \
  hipLaunchKernelGGL((pb_accumulate_kernel<NV>), dim3(gridW), dim3(OCN_BLOCK), 0, st, (const i64*)L.col_off,        \
                     (const int32_t*)L.keys, (i64)N, (const i64*)src, (const i64*)dst, (i64)B, (const i64*)off, flags, wc, \
                     (i64)flags_cap, (const float4*)weights, h, (int)H, g1, g2, g3, dh)
  if (H <= 256) PB_ACC(1); else PB_ACC(2);
#undef PB_ACC
  return launch_status();
}

}  // extern "C"
