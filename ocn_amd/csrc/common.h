// Shared primitives of the libocn_hip.so translation units (gfx950, wave64, 256-thread workgroups).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ocn_hip.h"

#define OCN_WAVE 64
#define OCN_BLOCK 256
#define OCN_WPB (OCN_BLOCK / OCN_WAVE)

typedef long long i64;
typedef unsigned long long u64;

// hist[c] = {packed, walks}: packed = n1 | n2 << 21 | n_union << 42 (one 64-bit atomic per CN entry
// instead of three 32-bit ones), walks = sum of the walk counts of column c (valued cn2 only).
#define HF_BITS 21
#define HF_MASK ((1ull << HF_BITS) - 1ull)
__device__ __forceinline__ int hf_n1(u64 w) { return (int)(w & HF_MASK); }
__device__ __forceinline__ int hf_n2(u64 w) { return (int)((w >> HF_BITS) & HF_MASK); }
__device__ __forceinline__ int hf_nu(u64 w) { return (int)((w >> (2 * HF_BITS)) & HF_MASK); }

// cn5 / cn6: nip = innerprod / scale from the column statistics word (cn5_column_stats: 0 = no union entry,
// -1 = union entries but no column with n1 >= 2, else min{n1 >= 2} - INT_MAX - 1); model.py:2370-2376
__device__ __forceinline__ float cn5_nip(int sc, float ip) {
  float scale;
  if (sc == 0) scale = 1.0f;                                        // empty union vector
  else if (sc == -1) scale = 0.0f;                                  // only singleton columns: every ncn1 value is 0
  else scale = 1.0f / (float)(sc + 0x7fffffff + 1);                 // largest 1/S1 among columns with S1 >= 2
  return scale > 0.0f ? ip / scale : ip;
}

static inline int launch_status() { return (int)hipGetLastError(); }

// status words of a candidate batch (ocn_hip.h): [0] = this batch's error bits, [3] = the same bits, sticky — the library
// only ever ORs into it, the caller clears it when it has read it (a scoring loop reads it once per split, not per batch)
__device__ __forceinline__ void status_raise(int32_t* status, i64 total, i64 cap) {
  const int bits = total < 0 ? OCN_ST_SCAN : (total > cap ? OCN_ST_CAP : 0);
  if (bits) { atomicOr(status, bits); atomicOr(status + 3, bits); }
}

static inline int grid_for(i64 items_per_block_units, i64 cap = (1 << 20)) {
  i64 g = items_per_block_units < 1 ? 1 : items_per_block_units;
  return (int)(g > cap ? cap : g);
}

// ---------------------------------------------------------------------------------------------
// wave / block primitives
// ---------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, OCN_WAVE);
  return v;
}

// inclusive scan across the 64 lanes of a wave
__device__ __forceinline__ i64 wave_incl_scan(i64 v, int lane) {
#pragma unroll
  for (int o = 1; o < OCN_WAVE; o <<= 1) {
    i64 t = __shfl_up(v, o, OCN_WAVE);
    if (lane >= o) v += t;
  }
  return v;
}

// exclusive scan of one value per thread over a 256-thread block; returns the thread's prefix and
// the block total.  `sh` is 2*OCN_WPB i64 of LDS.
__device__ __forceinline__ i64 block_excl_scan(i64 v, i64* sh, i64* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  i64 inc = wave_incl_scan(v, lane);
  if (lane == 63) sh[w] = inc;
  __syncthreads();
  i64 base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < OCN_WPB; ++i) {
    i64 s = sh[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

// ---------------------------------------------------------------------------------------------
// walk-count route: which endpoint enumerates the 2-walks of a candidate edge (i, j)
// ---------------------------------------------------------------------------------------------
// cn2[e,k] = |N(k) ∩ N(j)| for k in N(i) is the number of 2-walks j -> m -> k.  "Forward" sweeps the
// rows N(k), k in N(i) (nds[i] = Σ_{k∈N(i)} d_k probes against N(j)); "reverse" sweeps the rows N(m),
// m in N(j) (nds[j] probes against N(i); its hits land through global atomics, each 64-chunk of N(j)
// sets up N(i)'s bitmap again, and the row is finalised by the forward kernel's items afterwards).
// A hub source with a modest target is 10-100x cheaper from the target's side.
#define WALK_CHUNK 64        /* rows (neighbours of i) per forward work item */
#define WALK_REV_CHUNK 16    /* rows (neighbours of j) per reverse work item: few items, so finer ones */
// Forward work items take `walk_group()` consecutive 64-row chunks of the source row, so that an item
// sweeps about WALK_ITEM_ELEMS elements: a light row (most of a batch) is one item instead of one per
// 64 neighbours — the per-item set-up (ticket, slot search, row pointers, N(j) bitmap) is what such
// items cost — while a hub row still spreads over many workgroups.
#ifndef OCN_X_WALK_ITEM_ELEMS
#define OCN_X_WALK_ITEM_ELEMS 16384
#endif
#define WALK_ITEM_ELEMS OCN_X_WALK_ITEM_ELEMS
#define WALK_GROUP_MAX 8      /* chunks per forward item: one per wave of the 512-thread workgroup */
__device__ __forceinline__ i64 walk_group(const i64* __restrict__ nds, i64 i, i64 di) {
  const i64 chunks = (di + WALK_CHUNK - 1) / WALK_CHUNK;
  if (!nds || chunks <= 1) return 1;
  const i64 per_chunk = nds[i] / chunks + 1;
  i64 cg = WALK_ITEM_ELEMS / per_chunk;
  if (cg > WALK_GROUP_MAX) cg = WALK_GROUP_MAX;
  return cg < 1 ? 1 : (cg > chunks ? chunks : cg);
}

__device__ __forceinline__ bool walk_reverse(const i64* __restrict__ nds, i64 i, i64 j, i64 di, i64 dj) {
  if (!nds || dj == 0 || di == 0) return false;
  const i64 chunks_j = (dj + WALK_REV_CHUNK - 1) / WALK_REV_CHUNK;
  const i64 rev = 2 * nds[j] + di * chunks_j + 2 * di;
  return rev < nds[i];
}

// Two accumulators that take the same x: (a.x, a.y) += (w.x, w.y) * x on the packed-f32 pipe (v_pk_mul_f32 +
// v_pk_add_f32: one issue slot for both) — product and sum rounded separately, component for component what the two
// scalar statements a1 = a1 + w1 * x; a2 = a2 + w2 * x compute.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));      // register arrays carried across loop iterations: the native
                                                              // vector type is kept in registers where an array of the
                                                              // float4 struct ends up in scratch
__device__ __forceinline__ void axpy_pair(f32x2& a, const f32x2 w, const float x) {
#pragma clang fp contract(off)
  const f32x2 xx = {x, x};
  const f32x2 p = w * xx;
  a = a + p;
}
// The same for sums whose x comes out of LDS one float at a time (chain_rows).  There the form above compiles to
// `v_pk_mul_f32 v[w:w+1], v[x:x+1] op_sel_hi:[1,0]` — the pair's high register is never written, the instruction only names it — and
// the H = 64 wave kernel (all 64 lanes summing) returned wrong .x sums in lanes 48..63 about once in a hundred batches whenever
// another kernel ran beside it (a scoring loop's heads on a second stream): DESIGN.md section 6, tools/loop_race_check.py.  With both
// halves of the pair real registers (one v_mov per entry) the in-kernel check (-DOCN_X_WAVE_CHECK) found 0 disagreements in 3 600
// batches where the form above had ~50; two scalar chains are as clean and 5 % slower on the ppa / citation2 steps.
__device__ __forceinline__ void axpy_pair_lds(f32x2& a, const f32x2 w, const float x) {
#if defined(OCN_X_PACKED_CHAIN)     /* A/B builds only (tools/ab_flags.sh): the form that failed ... */
  axpy_pair(a, w, x);
#elif defined(OCN_X_SCALAR_CHAIN)   /* ... and two scalar chains */
  a[0] = __fadd_rn(a[0], __fmul_rn(w[0], x));
  a[1] = __fadd_rn(a[1], __fmul_rn(w[1], x));
#else
#pragma clang fp contract(off)
  f32x2 xx = {x, x};
  asm volatile("" : "+v"(xx));
  const f32x2 p = w * xx;
  a = a + p;
#endif
}

__device__ __forceinline__ void axpy4(float4& acc, float w, const float4& x) {
  acc.x = __fadd_rn(acc.x, __fmul_rn(w, x.x));
  acc.y = __fadd_rn(acc.y, __fmul_rn(w, x.y));
  acc.z = __fadd_rn(acc.z, __fmul_rn(w, x.z));
  acc.w = __fadd_rn(acc.w, __fmul_rn(w, x.w));
}

// per-entry pooling weights from the flag byte, the column's {w1, t, inv2} and the cn2 value c (model.py:2380-2427)
__device__ __forceinline__ void entry_weights(unsigned f, const float4& w, float c, float& wa, float& wb) {
  wa = (f & OCN_F_CN1) ? w.x : 0.f;
  const float v = __fsub_rn((f & OCN_F_CN2) ? c : 0.f, (f & OCN_F_CN1) ? w.y : 0.f);
  wb = __fmul_rn(v, w.z);
}

// ---------------------------------------------------------------------------------------------
// fp32 operands on the bf16 matrix pipe (linear.hip, wgrad.hip)
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ float bf16_to_f32(__bf16 v) {
  return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, v) << 16);
}

// x = p1 + p2 + p3 (+ O(2^-27 |x|)), each term a bf16 (round to nearest even)
__device__ __forceinline__ void split3(float x, __bf16& p1, __bf16& p2, __bf16& p3) {
  p1 = (__bf16)x;
  const float r1 = x - bf16_to_f32(p1);
  p2 = (__bf16)r1;
  const float r2 = r1 - bf16_to_f32(p2);
  p3 = (__bf16)r2;
}

