// The MLP heads of cn5 / cn7 (model.py:2203-2235, 2429-2437 == 3216-3223) as ONE kernel per candidate batch.
//
//   a = ReLU(LN(W3a ReLU(W0a xcn1 + b0a) + b3a))        xcn1lin, layers 0 and 3        (b: the same for xcn2)
//   c = ReLU(LN(W0x (x_i * x_j) + b0x))                 xijlin, layer 0
//   y = LN/ReLU/dot( Ma a + Mb b + Mc c + bf )          Ma = s(a0) W0l W7a, Mb = s(a0)s(a1) W0l W7b, Mc = beta W0l W4x
//
// The reference's last three products — the third layers of the pooled branches, the second layer of xijlin, the
// branch mix alpha0*xcn1 + alpha1*xcn2 + beta*xij and lin[0] — have no non-linearity between them; the host folds
// them (fp64, rounded once) into the three H x H matrices above, so a candidate costs 5 + 3 = 8 Linear(H,H)
// instead of 9 and, more importantly, nothing but the three pooled inputs is ever read from HBM and nothing but
// the score written: every intermediate activation stays in the registers of the wave that owns its 32 rows.
//
// Transposed formulation.  A wave owns 32 candidates and computes Y^T = W X^T with the weights as the MFMA's A
// operand: an accumulator tile then holds 32 features (registers) x 32 candidates (lanes), which is — without any
// lane movement — the B operand of the NEXT layer's MFMAs (the k index of the next product is this layer's
// feature index, i.e. the accumulator's register index; cdna_hip_programming.md §3 "An accumulator tile as the
// next MFMA's operand").  The next layer's weight panel is pre-permuted on the host to the k order in which the
// accumulator registers come ("chained" panel).  fp32 operands are split into three bf16 terms and the product
// formed from the six leading cross terms with fp32 accumulation, as in linear.hip (bf16x6).
//
// Workgroup = 4 waves = 128 candidates, one wave per SIMD (three accumulator sets of 128 registers are live);
// weight panels stream through LDS two k-steps at a time, double-buffered, by LDS-DMA (global_load_lds), one
// barrier per 96 MFMAs.  Candidates come in class-major order (ocn_class_order): a workgroup none of whose rows
// has cn1 (cn2) entries adds the branch's constant instead of running it — that constant is the branch's output on
// an all-zero row computed BY THIS KERNEL (dump mode), so skipping changes no bit of any score.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define HD_ROWS 128
#define HD_KC 2                       /* k-steps per staged panel chunk */
#ifndef HD_SCHED_VALU
#define HD_SCHED_VALU 3                   /* VALU / SALU instructions placed behind each MFMA of a k-step */
#endif
#ifndef HD_SCHED_SALU
#define HD_SCHED_SALU 2
#endif
#ifndef HD_VMEM_SLOTS
#define HD_VMEM_SLOTS 20                  /* MFMA gaps of a k-step that may carry an LDS-DMA piece */
#endif
#define HD_NVEC 17                    /* epilogue vectors of H floats, then one scalar (dot bias) */
enum { V_B0A = 0, V_B3A, V_G3A, V_E3A, V_B0B, V_B3B, V_G3B, V_E3B, V_B0X, V_GX, V_EX, V_BF, V_GL, V_EL, V_DOTW, V_CA, V_CB };

struct HeadsArgs {
  const float* x[3];                  // pooled xcn1, xcn2, x_i*x_j: [B][ldx]
  i64 ldx, B;
  const bf16x8* p_first[3];           // natural panels: xcn1lin.0, xcn2lin.0, xijlin.0
  const bf16x8* p_mid[2];             // chained panels: xcn1lin.3, xcn2lin.3
  const bf16x8* p_out[3];             // chained panels: Ma, Mb, Mc
  const float* vec;                   // HD_NVEC * H floats + dot bias
  const i64* ranges;                  // ocn_class_order's range table, or NULL (every row runs every branch)
  const i64* y_row_map;               // destination row of a score, or NULL
  float* y;
  float* dump;                        // constants mode: [2][H] <- the branch outputs Ma a, Mb b of row 0
  float* scratch;                     // ocn_heads_scratch_bytes(): two parked accumulator sets per resident wave
  float eps;
  int ln, b_on_union;
};

__device__ __forceinline__ float hd_bf(__bf16 v) {
  return __builtin_bit_cast(float, (unsigned)__builtin_bit_cast(unsigned short, v) << 16);
}
__device__ __forceinline__ void hd_split8(const float (&xs)[8], bf16x8& a1, bf16x8& a2, bf16x8& a3) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 p = (__bf16)xs[j];
    const float r1 = xs[j] - hd_bf(p);
    const __bf16 q = (__bf16)r1;
    const float r2 = r1 - hd_bf(q);
    a1[j] = p; a2[j] = q; a3[j] = (__bf16)r2;
  }
}

// feature of accumulator register i of tile t for lane half hh
#define HD_FEAT(t, i, hh) (32 * (t) + ((i) & 3) + 8 * ((i) >> 2) + 4 * (hh))

template <int NT>
struct Heads {
  static constexpr int H = 32 * NT;
  static constexpr int PANEL = NT * 3 * 64;              // fragments (16 B) per k-step
  static constexpr int CHUNK = HD_KC * PANEL;            // fragments per staged chunk
  static constexpr int NCH = (H / 16) / HD_KC;           // chunks per panel (== NT)
  static constexpr int VEC_FLOATS = 16 + (HD_NVEC * H + 4 + 3) / 4 * 4;   // 8 panel pointers, then the vectors
  static constexpr int XBUF = 4 * HD_KC * 2 * 64;        // float4 per landing buffer: 4 waves x k-steps x two 16-byte pieces x 64 lanes
  static constexpr size_t LDS_BYTES = (size_t)VEC_FLOATS * 4 + 2 * (size_t)CHUNK * 16 + 2 * (size_t)XBUF * 16;

  // six cross terms of one k-step of one output tile (smallest first)
  static __device__ __forceinline__ void mma6(f32x16& acc, const bf16x8 (&wf)[3], const bf16x8 (&xf)[3]) {
#ifdef OCN_X_HD_NOMFMA   /* timing experiment (tools/headsbench.py): everything but the matrix instructions */
    acc[0] += hd_bf(wf[0][0]) * hd_bf(xf[0][0]) + hd_bf(wf[1][1]) * hd_bf(xf[1][1]) + hd_bf(wf[2][2]) * hd_bf(xf[2][2]);
    return;
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], xf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[2], xf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1], xf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0], xf[0], acc, 0, 0, 0);
  }

  // one k-step: all NT output tiles, weight fragments from the staged chunk (A operand), xf = B operand.
  // `between(t)` runs right after tile t's six MFMAs have been ISSUED — they then execute for ~190 cycles on their
  // own: that is where the wave issues its LDS-DMA pieces and splits the next k-step's operand instead of idling
  // the matrix pipe with them at the chunk boundary (one wave per SIMD: nobody else would fill it).
  template <typename F>
  static __device__ __forceinline__ void kstep(f32x16 (&acc)[NT], const bf16x8* wl, const bf16x8 (&xf)[3], int lane, F&& between) {
    __builtin_amdgcn_sched_barrier(0);
    bf16x8 wq[2][3];
#pragma unroll
    for (int u = 0; u < 3; ++u) wq[0][u] = wl[u * 64 + lane];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t + 1 < NT) {
#pragma unroll
        for (int u = 0; u < 3; ++u) wq[(t + 1) & 1][u] = wl[((t + 1) * 3 + u) * 64 + lane];
      }
      mma6(acc[t], wq[t & 1], xf);
      between(t);
    }
    // The schedule of this region, spelled out: an MFMA occupies the matrix pipe for 32 cycles but the issue port for
    // 8 only, so every MFMA is followed by the next tile's fragment reads (one per two MFMAs), a slice of the operand
    // splitting / address arithmetic (VALU), and now and then an LDS-DMA piece.  Left to itself the scheduler issues the
    // six MFMAs of a tile back to back and everything else in blocks between tiles, where nothing overlaps it.
    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);               // the first tile's fragments
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int m = 0; m < 6; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);           // one MFMA
        if (m < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // one fragment read of the next tile
        __builtin_amdgcn_sched_group_barrier(0x002, HD_SCHED_VALU, 0);   // VALU
        // the LDS-DMA pieces of the next chunk: one behind each of the FIRST MFMAs of the k-step, so that they have the
        // rest of this k-step and all of the next to land before the chunk boundary waits for them
        if (t * 6 + m < HD_VMEM_SLOTS) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x004, HD_SCHED_SALU, 0);   // SALU (M0, addresses)
      }
    __builtin_amdgcn_sched_barrier(0);
  }

  // split elements [j0, j1) of an operand fragment
  static __device__ __forceinline__ void split_part(const float (&xs)[8], bf16x8 (&xf)[3], int j0, int j1) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j >= j0 && j < j1) {
        const __bf16 p = (__bf16)xs[j];
        const float r1 = xs[j] - hd_bf(p);
        const __bf16 q = (__bf16)r1;
        xf[0][j] = p; xf[1][j] = q; xf[2][j] = (__bf16)(r1 - hd_bf(q));
      }
  }

  // Materialise the accumulators HERE.  Left alone, the optimiser sinks an epilogue (bias, LayerNorm, ReLU) into the
  // next layer's k-steps, where its values are consumed — and keeps the raw accumulators AND every epilogue
  // vector (128 registers each) live across that whole layer.
  static __device__ __forceinline__ void pin(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(acc[t]));
  }

  static __device__ __forceinline__ void zero(f32x16 (&acc)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  }

  // acc[t][i] += v[feature]
  static __device__ __forceinline__ void add_vec(f32x16 (&acc)[NT], const float* v, int hh) {
#ifdef OCN_X_HD_NOEPI   /* timing experiment: the cost of the register epilogues */
    pin(acc); return;
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(v + 32 * t + 8 * g + 4 * hh);
        acc[t][4 * g + 0] += b.x; acc[t][4 * g + 1] += b.y; acc[t][4 * g + 2] += b.z; acc[t][4 * g + 3] += b.w;
      }
    pin(acc);
  }

  // LayerNorm over the H features of every candidate (a candidate's features: the 16 NT registers of lanes r, r+32)
  static __device__ __forceinline__ void layer_norm(f32x16 (&acc)[NT], const float* g, const float* b, float eps, int hh) {
#ifdef OCN_X_HD_NOEPI
    pin(acc); return;
#endif
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[t][i];
    s += __shfl_xor(s, 32, OCN_WAVE);
    const float mean = s * (1.0f / (float)H);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) { const float d = acc[t][i] - mean; q += d * d; }
    q += __shfl_xor(q, 32, OCN_WAVE);
    const float rstd = rsqrtf(q * (1.0f / (float)H) + eps);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 gg = *reinterpret_cast<const float4*>(g + 32 * t + 8 * gq + 4 * hh);
        const float4 bb = *reinterpret_cast<const float4*>(b + 32 * t + 8 * gq + 4 * hh);
        acc[t][4 * gq + 0] = (acc[t][4 * gq + 0] - mean) * rstd * gg.x + bb.x;
        acc[t][4 * gq + 1] = (acc[t][4 * gq + 1] - mean) * rstd * gg.y + bb.y;
        acc[t][4 * gq + 2] = (acc[t][4 * gq + 2] - mean) * rstd * gg.z + bb.z;
        acc[t][4 * gq + 3] = (acc[t][4 * gq + 3] - mean) * rstd * gg.w + bb.w;
      }
    pin(acc);
  }

  static __device__ __forceinline__ void relu(f32x16 (&acc)[NT]) {
#ifdef OCN_X_HD_NOEPI
    pin(acc); return;
#endif
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = fmaxf(acc[t][i], 0.f);
    pin(acc);
  }
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
// Every load of the main loops is an LDS-DMA (weights AND the candidates' input rows): beside LDS-DMA the compiler's
// wait-count bookkeeping drains the whole queue (vmcnt(0)) at the first use of an ordinary load's result
// (cdna_hip_programming.md §5, trap 4b), and asynchronous inline-asm loads into registers do not survive a loop
// back edge (a compiler-inserted copy reads the register before the data lands).  A lane's 32 bytes of a k-step
// land in its own slots of a per-wave LDS buffer and are read back by the same lane.

template <int NT>
__global__ __launch_bounds__(OCN_BLOCK, 1) void heads_fused_kernel(const HeadsArgs a) {
  using HD = Heads<NT>;
  constexpr int H = HD::H, PANEL = HD::PANEL, CHUNK = HD::CHUNK, NCH = HD::NCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // ONE array: panel list | vectors | two panel buffers
  const bf16x8** s_list = reinterpret_cast<const bf16x8**>(smem);
  float* s_vec = reinterpret_cast<float*>(smem) + 16;
  bf16x8* s_w = reinterpret_cast<bf16x8*>(smem + (size_t)HD::VEC_FLOATS * 4);
  float4* s_x = reinterpret_cast<float4*>(smem + (size_t)HD::VEC_FLOATS * 4 + 2 * (size_t)CHUNK * 16);
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);          // scalar: everything decided per wave stays on the scalar unit
  const int r = lane & 31, hh = lane >> 5;
  for (int q = threadIdx.x; q < HD_NVEC * H + 1; q += OCN_BLOCK) s_vec[q] = a.vec[q];
  // where this wave parks a finished branch's share of the output while the next branch needs the registers
  float4* park = reinterpret_cast<float4*>(a.scratch) + ((size_t)(blockIdx.x * 4 + w) * 2) * (NT * 4) * 64 + lane;
  const i64 n_tiles = a.dump ? 1 : (a.B + HD_ROWS - 1) / HD_ROWS;
#ifdef OCN_X_HD_CLOCK                /* diagnostic build only: the clock the chip holds under this kernel (guide, DVFS give-back item 6) */
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif

#pragma unroll 1
  for (i64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const i64 slot = tile * HD_ROWS + 32 * w + r;
    const bool live = slot < a.B;
    const i64 arow = live ? slot : a.B - 1;
    // class of this candidate (class-major order: both | cn1 only | cn2 only | none)
    bool has1 = live, hasB = live;
    if (a.ranges) {
      const i64 m3 = a.ranges[2 * 1 + 1], m32 = a.ranges[2 * 0 + 1], m321 = a.ranges[2 * 3 + 1];
      has1 = live && slot < m32;
      const bool has2 = slot < m3 || (slot >= m32 && slot < m321);
      hasB = live && (a.b_on_union ? slot < m321 : has2);
    }
    const int wgA = __syncthreads_or(has1), wgB = __syncthreads_or(hasB);     // (also: the previous tile's panel reads are over)
    const int np = 2 + (wgA ? 3 : 0) + (wgB ? 3 : 0);
    if (threadIdx.x == 0) {                  // the panels this workgroup streams, in order
      int q = 0;
      if (wgA) { s_list[q++] = a.p_first[0]; s_list[q++] = a.p_mid[0]; s_list[q++] = a.p_out[0]; }
      if (wgB) { s_list[q++] = a.p_first[1]; s_list[q++] = a.p_mid[1]; s_list[q++] = a.p_out[1]; }
      s_list[q++] = a.p_first[2]; s_list[q++] = a.p_out[2];
    }
    __syncthreads();                                                           // publishes s_list (and, first time, s_vec)
    const int total_chunks = np * NCH;
    int gi = 0;                                                                // chunks consumed so far

    // Stage chunk ci of the panel stream into buffer ci & 1 by LDS-DMA, 1 KiB per wave-instruction: this wave's piece
    // number q of NPW (pieces are dealt to the four waves round robin).
    constexpr int NPW = (HD_KC * NT * 3 + 3) / 4;
    const unsigned lane16 = (unsigned)lane * 16u;
    // scalar base of chunk ci in memory (the panel list is read once per chunk, not once per piece)
    // (past the end of the stream the last chunk is staged once more, into the buffer nobody reads any more: the main
    // loops stay free of branches, i.e. one scheduling region per k-step)
    auto chunk_src = [&](int ci) -> const char* {
      const int cc = ci < total_chunks ? ci : total_chunks - 1;
      const bf16x8* p = s_list[cc / NCH] + (size_t)(cc % NCH) * CHUNK;
      const unsigned long long v = (unsigned long long)p;
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
      return (const char*)(((unsigned long long)hi << 32) | lo);
    };
    auto issue_piece = [&](const char* src, int ci, int q) {
#ifndef OCN_X_HD_NOGLDS
      int i = w + 4 * q;
      if ((HD_KC * NT * 3) % 4 != 0) i = i < HD_KC * NT * 3 ? i : HD_KC * NT * 3 - 1;     // (the last piece twice: same bytes)
      bf16x8* dst = s_w + (size_t)(ci & 1) * CHUNK;
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + (size_t)i * 1024 + lane16), (lds_ptr_t)(dst + i * 64), 16, 0, 0);
#endif
    };
    // the pieces of tile slot t of the chunk's FIRST k-step (so they have the second k-step's time to land)
    auto issue_slot = [&](const char* src, int ci, int t) {
#pragma unroll
      for (int q = 0; q < NPW; ++q)
        if (q % NT == t) issue_piece(src, ci, q);
    };
    {
      const char* src0 = chunk_src(0);
#pragma unroll
      for (int q = 0; q < NPW; ++q) issue_piece(src0, 0, q);
    }

    // ---- first layer of a branch from global memory: acc = W0 . X^T ------------------------------------------
    // rows of chunk c (k-steps 2c, 2c+1) -> this wave's slots of landing buffer `buf`: piece pc of k-step ks
    auto issue_x = [&](const float* xrow, int c, int buf, int ks, int pc) {
#ifndef OCN_X_HD_NOGLDS
      float4* dst = s_x + (size_t)buf * HD::XBUF + (size_t)w * (HD_KC * 2 * 64);
      __builtin_amdgcn_global_load_lds((gbl_ptr_t)(xrow + (c * HD_KC + ks) * 16 + 4 * pc),
                                       (lds_ptr_t)(dst + (ks * 2 + pc) * 64), 16, 0, 0);
#endif
    };
    auto first_layer = [&](f32x16 (&acc)[NT], const float* xb, bool rowmask) {
      const float* xrow = xb + arow * a.ldx + 8 * hh;
      HD::zero(acc);
#pragma unroll
      for (int q = 0; q < HD_KC * 2; ++q) issue_x(xrow, 0, gi & 1, q >> 1, q & 1);
#pragma unroll 1
      for (int c = 0; c < NCH; ++c) {
        // chunk boundary: everything this wave issued has landed (vmcnt), everybody's has (barrier), and nobody still
        // reads the buffers chunk gi + 1 is about to overwrite
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bf16x8* wl = s_w + (size_t)(gi & 1) * CHUNK;
        const float4* xl = s_x + (size_t)(gi & 1) * HD::XBUF + (size_t)w * (HD_KC * 2 * 64) + lane;
        float xs[HD_KC][8];
#pragma unroll
        for (int ks = 0; ks < HD_KC; ++ks) {
          const float4 x0 = xl[(ks * 2) * 64], x1 = xl[(ks * 2 + 1) * 64];
          const float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) xs[ks][j] = rowmask ? v[j] : 0.f;      // rows the pooling never wrote count as zero rows
        }
        bf16x8 xf[3], xg[3];
        HD::split_part(xs[0], xf, 0, 8);
        const int cn = c + 1 < NCH ? c + 1 : c;           // (the last chunk's rows once more, into the idle buffer)
        const char* nsrc = chunk_src(gi + 1);
        HD::kstep(acc, wl, xf, lane, [&](int t) {         // k-step 0: issue the next chunk, split k-step 1's operand
          issue_slot(nsrc, gi + 1, t);
          if (t < HD_KC * 2) issue_x(xrow, cn, (gi + 1) & 1, t >> 1, t & 1);
          if (NT >= 8) HD::split_part(xs[1], xg, t, t + 1);
          else HD::split_part(xs[1], xg, t * (8 / NT), (t + 1) * (8 / NT));
        });
        if (NT < HD_KC * 2) {
#pragma unroll
          for (int q = NT; q < HD_KC * 2; ++q) issue_x(xrow, cn, (gi + 1) & 1, q >> 1, q & 1);
        }
        HD::kstep(acc, wl + PANEL, xg, lane, [](int) {});
        ++gi;
      }
    };

    // ---- a layer whose input is the previous layer's accumulators: acc = W . in (chained panel) --------------
    auto chained_layer = [&](f32x16 (&acc)[NT], const f32x16 (&in)[NT]) {
      HD::zero(acc);
      bf16x8 xf[3], xg[3];
      {
        float xs[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) xs[j] = in[0][j];
        HD::split_part(xs, xf, 0, 8);
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {                    // chunk c = k-steps 2c, 2c+1 = input tile c
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const bf16x8* wl = s_w + (size_t)(gi & 1) * CHUNK;
        float xs1[8], xs2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { xs1[j] = in[c][8 + j]; xs2[j] = in[c + 1 < NCH ? c + 1 : c][j]; }
        const char* nsrc = chunk_src(gi + 1);
        HD::kstep(acc, wl, xf, lane, [&](int t) {          // k-step 0: issue the next chunk, split k-step 1's operand
          issue_slot(nsrc, gi + 1, t);
          if (NT >= 8) HD::split_part(xs1, xg, t, t + 1);
          else HD::split_part(xs1, xg, t * (8 / NT), (t + 1) * (8 / NT));
        });
        HD::kstep(acc, wl + PANEL, xg, lane, [&](int t) {  // k-step 1: split the next chunk's first operand
          if (c + 1 < NCH) {
            if (NT >= 8) HD::split_part(xs2, xf, t, t + 1);
            else HD::split_part(xs2, xf, t * (8 / NT), (t + 1) * (8 / NT));
          }
        });
        ++gi;
      }
    };

    f32x16 l1[NT], l2[NT];
    // ---- pooled branches a (xcn1lin) and b (xcn2lin): their share M . act of the output is parked in memory ------
#pragma unroll 1
    for (int br = 0; br < 2; ++br) {
      if (!(br == 0 ? wgA : wgB)) continue;
      const float* vb = s_vec + (br == 0 ? V_B0A : V_B0B) * H;                // b0, b3, gamma3, beta3 of this branch
      first_layer(l1, br == 0 ? a.x[0] : a.x[1], br == 0 ? has1 : hasB);
      HD::add_vec(l1, vb, hh);
      HD::relu(l1);
      chained_layer(l2, l1);
      HD::add_vec(l2, vb + H, hh);
      if (a.ln) HD::layer_norm(l2, vb + 2 * H, vb + 3 * H, a.eps, hh);
      HD::relu(l2);
      chained_layer(l1, l2);
      float4* pk = park + (size_t)br * (NT * 4) * 64;
      asm volatile("" : "+v"(pk));           // (or every one of the 2 x 4 NT addresses is precomputed outside the tile loop and kept)
#ifdef OCN_X_HD_NOPARK  /* timing experiment: the cost of parking a branch's share in memory */
      HD::pin(l1);
      if (false)
#endif
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          pk[g * 64] = make_float4(l1[t][4 * g], l1[t][4 * g + 1], l1[t][4 * g + 2], l1[t][4 * g + 3]);
        pk += 4 * 64;
        asm volatile("" : "+v"(pk));
      }
      if (a.dump && w == 0 && r == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) a.dump[br * H + HD_FEAT(t, i, hh)] = l1[t][i];
      }
    }
    // ---- xijlin ---------------------------------------------------------------------------------------------
    first_layer(l1, a.x[2], live);
    HD::add_vec(l1, s_vec + V_B0X * H, hh);
    if (a.ln) HD::layer_norm(l1, s_vec + V_GX * H, s_vec + V_EX * H, a.eps, hh);
    HD::relu(l1);
    chained_layer(l2, l1);
    // ---- out = ((share a + share b) + share c) + folded bias; a skipped branch's share is its constant ---------
    {
      const float4* pa = park;
      const float4* pb = park + (size_t)(NT * 4) * 64;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        asm volatile("" : "+v"(pa), "+v"(pb));               // a tile at a time: addresses and values of all tiles at once is 400 registers
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int fo = 32 * t + 8 * g + 4 * hh;
#ifdef OCN_X_HD_NOPARK
          const float4 sa = *reinterpret_cast<const float4*>(s_vec + V_CA * H + fo);
          const float4 sb = *reinterpret_cast<const float4*>(s_vec + V_CB * H + fo);
#else
          const float4 sa = wgA ? pa[g * 64] : *reinterpret_cast<const float4*>(s_vec + V_CA * H + fo);
          const float4 sb = wgB ? pb[g * 64] : *reinterpret_cast<const float4*>(s_vec + V_CB * H + fo);
#endif
          const float4 bf = *reinterpret_cast<const float4*>(s_vec + V_BF * H + fo);
          l2[t][4 * g + 0] = ((sa.x + sb.x) + l2[t][4 * g + 0]) + bf.x;
          l2[t][4 * g + 1] = ((sa.y + sb.y) + l2[t][4 * g + 1]) + bf.y;
          l2[t][4 * g + 2] = ((sa.z + sb.z) + l2[t][4 * g + 2]) + bf.z;
          l2[t][4 * g + 3] = ((sa.w + sb.w) + l2[t][4 * g + 3]) + bf.w;
        }
        asm volatile("" : "+v"(l2[t]));
        pa += 4 * 64;
        pb += 4 * 64;
      }
    }
    // ---- lin: LayerNorm, ReLU, Linear(H, 1) -------------------------------------------------------------------
    if (a.ln) HD::layer_norm(l2, s_vec + V_GL * H, s_vec + V_EL * H, a.eps, hh);
    HD::relu(l2);
    float d = 0.f;
    const float* dw = s_vec + V_DOTW * H;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 ww = *reinterpret_cast<const float4*>(dw + 32 * t + 8 * gq + 4 * hh);
        d += l2[t][4 * gq + 0] * ww.x + l2[t][4 * gq + 1] * ww.y + l2[t][4 * gq + 2] * ww.z + l2[t][4 * gq + 3] * ww.w;
      }
    d += __shfl_xor(d, 32, OCN_WAVE);
    if (live && hh == 0 && !a.dump) a.y[a.y_row_map ? a.y_row_map[slot] : slot] = d + s_vec[HD_NVEC * H];
  }
#ifdef OCN_X_HD_CLOCK
  if (threadIdx.x == 0) {            // into this workgroup's own (now dead) park area: nothing reads it
    unsigned long long* o = reinterpret_cast<unsigned long long*>(reinterpret_cast<float4*>(a.scratch) + ((size_t)(blockIdx.x * 4) * 2) * (NT * 4) * 64);
    o[0] = __builtin_amdgcn_s_memtime() - clk0;
    o[1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}

// Wp[s][t][split][lane][8] of a CHAINED layer: k-step s = 2 tt + ss consumes accumulator tile tt, registers
// 8 ss .. 8 ss + 7, of the previous layer: element j of lane (r, hh) is W[32 t + r][32 tt + 16 ss + 8 (j >> 2) + 4 hh + (j & 3)]
__global__ __launch_bounds__(OCN_BLOCK) void split_weight_chained_kernel(const float* __restrict__ W, int N, int K,
                                                                         __bf16* __restrict__ Wp) {
  const int NT = N >> 5;
  const i64 total = (i64)(K / 16) * NT * 64;
  for (i64 f = (i64)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (i64)gridDim.x * blockDim.x) {
    const int lane = (int)(f & 63);
    const int t = (int)((f >> 6) % NT);
    const int s = (int)((f >> 6) / NT);
    const int tt = s >> 1, ss = s & 1, hh = lane >> 5;
    const float* src = W + (i64)(32 * t + (lane & 31)) * K + 32 * tt + 16 * ss + 4 * hh;
    bf16x8 p[3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = src[8 * (j >> 2) + (j & 3)];
      const __bf16 p1 = (__bf16)x;
      const float r1 = x - hd_bf(p1);
      const __bf16 p2 = (__bf16)r1;
      p[0][j] = p1; p[1][j] = p2; p[2][j] = (__bf16)(r1 - hd_bf(p2));
    }
    bf16x8* dst = reinterpret_cast<bf16x8*>(Wp) + ((i64)(s * NT + t) * 3) * 64 + lane;
    dst[0] = p[0];
    dst[64] = p[1];
    dst[128] = p[2];
  }
}

#define HD_MAX_GRID 256               /* one workgroup per CU (LDS and registers admit no second one): a persistent grid */
template <int NT>
static int heads_launch(const HeadsArgs& a, i64 tiles, hipStream_t st) {
  static bool raised_dev[64] = {};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return OCN_EINVAL;
  if (!raised_dev[devid]) {
    const hipError_t e = hipFuncSetAttribute((const void*)heads_fused_kernel<NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)Heads<NT>::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    raised_dev[devid] = true;
  }
  hipLaunchKernelGGL((heads_fused_kernel<NT>), dim3((unsigned)(tiles < HD_MAX_GRID ? tiles : HD_MAX_GRID)), dim3(OCN_BLOCK),
                     Heads<NT>::LDS_BYTES, st, a);
  return launch_status();
}

extern "C" {

int ocn_linear_split_weight_chained(const float* W, int32_t N, int32_t K, void* Wp, void* stream) {
  if (!W || !Wp || N <= 0 || K <= 0 || (N & 31) || (K & 31)) return OCN_EINVAL;
  const i64 frags = (i64)(K / 16) * (N >> 5) * 64;
  hipLaunchKernelGGL(split_weight_chained_kernel, dim3(grid_for((frags + OCN_BLOCK - 1) / OCN_BLOCK, 1024)),
                     dim3(OCN_BLOCK), 0, (hipStream_t)stream, W, (int)N, (int)K, (__bf16*)Wp);
  return launch_status();
}

int32_t ocn_heads_nvec(void) { return HD_NVEC; }

int64_t ocn_heads_scratch_bytes(int32_t H) { return (int64_t)HD_MAX_GRID * 4 * 2 * H * 32 * 4; }

int ocn_heads_fused(const OcnHeadsArgs* h, void* stream) {
  if (!h || h->B < 0 || h->H <= 0) return OCN_EINVAL;
  if (h->B == 0) return 0;
  for (int i = 0; i < 3; ++i)
    if (!h->x[i] || !h->p_first[i] || !h->p_out[i]) return OCN_EINVAL;
  if (!h->p_mid[0] || !h->p_mid[1] || !h->vec || !h->scratch || (!h->y && !h->dump)) return OCN_EINVAL;
  const int64_t ldx = h->ldx ? h->ldx : h->H;
  if (ldx < h->H || (ldx & 3)) return OCN_EINVAL;
  HeadsArgs a;
  for (int i = 0; i < 3; ++i) { a.x[i] = h->x[i]; a.p_first[i] = (const bf16x8*)h->p_first[i]; a.p_out[i] = (const bf16x8*)h->p_out[i]; }
  a.p_mid[0] = (const bf16x8*)h->p_mid[0]; a.p_mid[1] = (const bf16x8*)h->p_mid[1];
  a.ldx = ldx; a.B = h->B; a.vec = h->vec; a.ranges = (const i64*)h->ranges; a.y_row_map = (const i64*)h->y_row_map;
  a.y = h->y; a.dump = h->dump; a.scratch = h->scratch; a.eps = h->eps; a.ln = h->ln; a.b_on_union = h->b_on_union;
  const i64 tiles = h->dump ? 1 : (h->B + HD_ROWS - 1) / HD_ROWS;
  hipStream_t st = (hipStream_t)stream;
  switch (h->H) {
    case 32:  return heads_launch<1>(a, tiles, st);
    case 64:  return heads_launch<2>(a, tiles, st);
    case 128: return heads_launch<4>(a, tiles, st);
    case 256: return heads_launch<8>(a, tiles, st);
    default: return OCN_EINVAL;
  }
}

}  // extern "C"
