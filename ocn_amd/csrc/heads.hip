// The MLP heads of cn5 / cn7 (model.py:2203-2235, 2429-2437 == 3216-3223) as ONE kernel per candidate batch.
//
//   a = ReLU(LN(W3a ReLU(W0a xcn1 + b0a) + b3a))        xcn1lin, layers 0 and 3        (b: the same for xcn2)
//   c = ReLU(LN(W0x (x_i * x_j) + b0x))                 xijlin, layer 0
//   y = LN/ReLU/dot( Ma a + Mb b + Mc c + bf )          Ma = s(a0) W0l W7a, Mb = s(a0)s(a1) W0l W7b, Mc = beta W0l W4x
//
// The reference's last three products — the third layers of the pooled branches, the second layer of xijlin, the
// branch mix alpha0*xcn1 + alpha1*xcn2 + beta*xij and lin[0] — have no non-linearity between them; the host folds
// them (fp64, rounded once) into the three H x H matrices above, so a candidate costs 5 + 3 = 8 Linear(H,H)
// instead of 9, nothing but the three pooled inputs is read from HBM and nothing but the score written: every
// intermediate activation stays in the registers of the wave that owns its 32 rows.
//
// Transposed formulation.  A wave owns 32 candidates and computes Y^T = W X^T with the weights as the MFMA's A
// operand: an accumulator tile then holds 32 features (registers) x 32 candidates (lanes), which is — without any
// lane movement — the B operand of the NEXT layer's MFMAs (cdna_hip_programming.md §3 "An accumulator tile as the
// next MFMA's operand").  The raw inputs are loaded into the same register layout, so all eight layers are "chained"
// layers and all eight weight panels have one format, pre-permuted on the host to the k order in which the
// accumulator registers come.
//
// Arithmetic: an f32 product as THREE f16 MFMAs.  x = xh + xl, w = wh + wl with xh = f16(x), xl = f16(x - xh)
// (22 significant bits; the dropped wl xl term is below 2^-22 |w x|) and w x ~ wh xl + wl xh + wh xh accumulated in
// fp32 — half the matrix instructions of the bf16x6 split (six cross terms of three 8-bit terms) at the same fp64
// error (tests/test_parity_gpu.py::test_heads_product_accuracy; the fp32 accumulation dominates both).  f16 has a
// 5-bit exponent, so both operands are scaled by powers of two (exact): a weight panel once on the host so that
// max |w| is in [2^13, 2^14), an activation row per layer by the exponent of its own largest element (a candidate is
// a lane: the row maximum is a register reduction and one lane exchange).  Elements more than 2^27 below their
// row's maximum lose low bits or flush — an absolute error of 2^-38 of the row maximum.  The accumulator is
// multiplied back by the inverse powers of two inside the bias fma of the epilogue (exact).
//
// Schedule.  One wave per SIMD (two activation sets of 128 registers + operands), 4 waves = 128 candidates per
// workgroup, one workgroup per CU, persistent over 128-row tiles.  Weight panels stream through a three-slot LDS
// ring by LDS-DMA, two chunks (of two k-steps) ahead; the k-step itself is hand-placed inline asm, one statement per
// instruction group: per output tile three MFMAs, the two fragment reads of the tile THREE tiles ahead (counted
// lgkmcnt waits: the compiler's own schedule waited for each read right behind its issue — 23 % of the wave cycles in
// s_waitcnt, matrix pipe 50 % busy, profiles/r03a_heads_pmc_before.json), a quarter of the next k-step's operand
// split (8 VALU) and one LDS-DMA piece, all inside the 96 cycles the tile's MFMAs occupy the matrix pipe.
//
// Candidates come in class-major order (ocn_class_order): a workgroup none of whose rows has cn1 (cn2) entries adds
// the branch's constant instead of running it — that constant is the branch's output on an all-zero row computed BY
// THIS KERNEL (dump mode), so skipping changes no bit of any score.
#include "common.h"
#include <type_traits>
#include <utility>

typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(3))) unsigned char* lds_bytes_t;

#define HD_ROWS 128
#define HD_NVEC 17                    /* epilogue vectors of H floats; then 16 scalars: dot bias, 8 inverse panel scales */
#define HD_NSCAL 16
enum { V_B0A = 0, V_B3A, V_G3A, V_E3A, V_B0B, V_B3B, V_G3B, V_E3B, V_B0X, V_GX, V_EX, V_BF, V_GL, V_EL, V_DOTW, V_CA, V_CB };
enum { P_A0 = 0, P_A3, P_MA, P_B0, P_B3, P_MB, P_X0, P_MC, HD_NPANEL };

struct HeadsArgs {
  const float* x[3];                  // pooled xcn1, xcn2, x_i*x_j: [B][ldx]
  i64 ldx, B;
  const char* panel[HD_NPANEL];       // f16 hi/lo panels (ocn_heads_split_weight) in the order of the P_ enum
  const float* vec;                   // HD_NVEC * H floats + HD_NSCAL scalars
  const i64* ranges;                  // ocn_class_order's range table, or NULL (every row runs every branch)
  const i64* y_row_map;               // destination row of a score, or NULL
  float* y;
  float* dump;                        // constants mode: [2][H] <- the branch outputs Ma a, Mb b of row 0
  float* scratch;                     // ocn_heads_scratch_bytes(): two parked branch shares per resident wave
  float eps;
  int ln, b_on_union;
};

template <typename F, int... I>
__device__ __forceinline__ void hd_unroll_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void hd_unroll(F&& f) { hd_unroll_impl(f, std::make_integer_sequence<int, N>{}); }

// ---- the instruction groups of a k-step (cdna_hip_programming.md §5.7: the compiler neither counts the memory
// operations of an asm statement nor pads its hazards; every wait below is counted by hand, see hd_layer) -----------

// acc (+)= A . B after at most N LDS reads are still outstanding (N < 0: no wait); CZ: the accumulator starts at 0
template <bool ACC_A, bool CZ, int N>
__device__ __forceinline__ void hd_mfma(f32x16& c, const h16x8& a, const h16x8& b) {
  if constexpr (ACC_A) {
    if constexpr (CZ) {
      if constexpr (N >= 0) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b), "i"(N));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
    } else {
      if constexpr (N >= 0) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b), "i"(N));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    }
  } else {
    if constexpr (CZ) {
      if constexpr (N >= 0) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b), "i"(N));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(b));
    } else {
      if constexpr (N >= 0) asm volatile("s_waitcnt lgkmcnt(%3)\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b), "i"(N));
      else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
    }
  }
}

// one weight fragment (1 KiB per wave) from the ring; `f` is in flight until the hd_mfma that waits for it
template <int OFF>
__device__ __forceinline__ void hd_dsread(h16x8& f, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF));
}

// operand split of two activations, first half: t = x * sc, H = f16x2(t0, t1), t2 = f32(H.lo)
template <bool IN_A>
__device__ __forceinline__ void hd_split1(float x0, float x1, float sc, unsigned& H, float& t0, float& t1, float& t2) {
  if constexpr (IN_A)
    asm volatile("v_accvgpr_read_b32 %1, %4\n\tv_accvgpr_read_b32 %2, %5\n\tv_mul_f32 %1, %1, %6\n\tv_mul_f32 %2, %2, %6\n\t"
                 "v_cvt_pk_f16_f32 %0, %1, %2\n\tv_cvt_f32_f16 %3, %0"
                 : "=&v"(H), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "a"(x0), "a"(x1), "v"(sc));
  else
    asm volatile("v_mul_f32 %1, %4, %6\n\tv_mul_f32 %2, %5, %6\n\tv_cvt_pk_f16_f32 %0, %1, %2\n\tv_cvt_f32_f16 %3, %0"
                 : "=&v"(H), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(x0), "v"(x1), "v"(sc));
}
// second half: L = f16x2(t0 - f32(H.lo), t1 - f32(H.hi))   (the differences are exact)
__device__ __forceinline__ void hd_split2(unsigned H, float t0, float t1, float t2, unsigned& L) {
  float t3;
  asm volatile("v_cvt_f32_f16_sdwa %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
               "v_sub_f32 %2, %2, %5\n\tv_sub_f32 %3, %3, %1\n\tv_cvt_pk_f16_f32 %0, %2, %3"
               : "=&v"(L), "=&v"(t3), "+v"(t0), "+v"(t1) : "v"(H), "v"(t2));
}

// one LDS-DMA piece: 64 lanes x 16 bytes from base + voff to the LDS byte address `lds` (+ 16 lane).  M0 is written
// and read in ONE statement and handed back (§5.7: M0 is compiler-reserved); the s_nop covers SALU -> VMEM SGPR reads.
__device__ __forceinline__ void hd_dma(unsigned voff, const char* base, unsigned lds) {
#ifndef OCN_X_HD_NOGLDS   /* timing experiment (tools/headsbench.py): no weight traffic */
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds) : "memory");
#endif
}

// my pieces of the next chunk have landed (all but the N youngest vector-memory operations are done), then everybody's
template <int N>
__device__ __forceinline__ void hd_sync() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"i"(N) : "memory");
}

__device__ __forceinline__ h16x8 hd_frag(const unsigned (&v)[4]) {
  const u32x4 u = {v[0], v[1], v[2], v[3]};
  return __builtin_bit_cast(h16x8, u);
}

// per-row power-of-two scale: m = the row's largest |x|, pinv = the panel's inverse scale.  x * sc has its largest
// element in [2^13, 2^14); inv = 1 / (sc * panel scale).  Rows below 2^-63 are scaled as if they were 2^-63.
__device__ __forceinline__ void hd_row_scale(float m, float pinv, float& sc, float& inv) {
  m = fmaxf(m, __shfl_xor(m, 32, OCN_WAVE));
  unsigned e = (__float_as_uint(m) >> 23) & 0xffu;
  e = e < 64u ? 64u : (e > 254u ? 254u : e);
  sc = __uint_as_float((267u - e) << 23);
  inv = __uint_as_float((e - 13u) << 23) * pinv;
}

template <int NT>
struct Heads {
  static constexpr int H = 32 * NT;
  static constexpr int KS = 2 * NT;                        // k-steps per layer
  static constexpr int TPC = 2 * NT;                       // output tiles per chunk (two k-steps)
  static constexpr int NCH = NT;                           // chunks per layer
  static constexpr int G = KS * NT;                        // output-tile steps per layer
  static constexpr int CHB = TPC * 2048;                   // bytes per chunk: per tile step an f16 hi and an f16 lo fragment
  static constexpr int NPW = TPC / 2;                      // LDS-DMA pieces per wave and chunk
  static constexpr int VEC_FLOATS = HD_NVEC * H + HD_NSCAL;
  static constexpr int LDS_W = VEC_FLOATS * 4;
  static constexpr size_t LDS_BYTES = (size_t)LDS_W + 3 * (size_t)CHB;
  static_assert(NT == 4 || NT == 8, "tiles per k-step");
  // which tile steps of a chunk issue an LDS-DMA piece (NPW of the TPC), and how many pieces come before step q.
  // NT = 8: the steps whose shadow holds no operand split (tiles 4..7 of both k-steps); NT = 4: every other step.
  static constexpr bool has_piece(int q) { return NT == 8 ? (q % NT) >= 4 : (q & 1) != 0; }
  static constexpr int piece_of(int q) {
    int n = 0;
    for (int i = 0; i < q; ++i) n += has_piece(i) ? 1 : 0;
    return n;
  }
  static_assert(piece_of(TPC) == NPW, "pieces per chunk");

  struct Ring {                       // the three ring slots in the roles {this chunk, next, the one after}
    unsigned pa[3];                   // per lane: byte address of the lane's 16 bytes of the slot's first fragment
    unsigned ld[3];                   // per wave: byte address of the wave's first LDS-DMA piece in the slot
  };

  static __device__ __forceinline__ void pin_a(f32x16 (&v)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+a"(v[t]));
  }
  static __device__ __forceinline__ void pin_v(f32x16 (&v)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+v"(v[t]));
  }

  // acc = Wp . in on the 32 candidates of this wave.  `in` holds the layer's input in accumulator layout (tile tt,
  // register i = feature 32 tt + (i & 3) + 8 (i >> 2) + 4 hh), `sc` the row's scale.  The weight stream: this layer's
  // panel p_cur (its chunk c is in ring role c % 3), then p_nxt; LAST: the stream ends with this layer.
  template <bool ACC_A, bool IN_A, bool LAST>
  static __device__ __forceinline__ void layer(f32x16 (&acc)[NT], f32x16 (&in)[NT], float sc, Ring& rg, const char* p_cur,
                                               const char* p_nxt, unsigned lane16) {
    h16x8 fh[4], fl[4];
    unsigned xb[2][2][4];             // [k-step parity][hi, lo][4 registers]: the B operand
#pragma unroll
    for (int p = 0; p < 4; ++p) {     // operand of k-step 0
      float t0, t1, t2;
      hd_split1<IN_A>(in[0][2 * p], in[0][2 * p + 1], sc, xb[0][0][p], t0, t1, t2);
      hd_split2(xb[0][0][p], t0, t1, t2, xb[0][1][p]);
    }
    hd_dsread<0>(fh[0], rg.pa[0]);
    hd_dsread<1024>(fl[0], rg.pa[0]);
    hd_dsread<2048>(fh[1], rg.pa[0]);
    hd_dsread<3072>(fl[1], rg.pa[0]);
    hd_dsread<4096>(fh[2], rg.pa[0]);
    hd_dsread<5120>(fl[2], rg.pa[0]);
    float st0[4], st1[4], st2[4];     // split state between the two halves of a tile step
    hd_unroll<G>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int s = g / NT, t = g % NT, c = s / 2, q = g % TPC, f = g % 4;
      constexpr int g3 = g + 3, c3 = g3 / TPC, q3 = g3 % TPC, f3 = g3 % 4;
      constexpr bool pf = g3 < G;                                      // fragments of the tile three steps ahead
      constexpr bool sp = s + 1 < KS && t < 4;                         // a quarter of the next k-step's operand
      constexpr int sn = s + 1;
      // the chunk boundary, three tile steps early (the first read of the next chunk is this step's prefetch): the
      // pieces of the next chunk were issued a chunk ago; younger are only the pieces this chunk has issued so far
      if constexpr (q == TPC - 3 && !(LAST && c == NCH - 1)) hd_sync<(LAST && c + 2 >= NCH) ? 0 : piece_of(TPC - 3)>();
      // LDS reads are issued in the order hi(0) lo(0) hi(1) lo(1) ...: read 2g must be back before the first MFMA
      constexpr int issued1 = 2 * (g + 3) < 2 * G ? 2 * (g + 3) : 2 * G;
      hd_mfma<ACC_A, s == 0, issued1 - (2 * g + 1)>(acc[t], fh[f], hd_frag(xb[s & 1][1]));          // wh . xl
      if constexpr (pf) hd_dsread<q3 * 2048>(fh[f3], rg.pa[c3 % 3]);
      if constexpr (sp) hd_split1<IN_A>(in[sn / 2][8 * (sn & 1) + 2 * t], in[sn / 2][8 * (sn & 1) + 2 * t + 1], sc,
                                        xb[sn & 1][0][t], st0[t], st1[t], st2[t]);
      constexpr int issued2 = pf ? 2 * g3 + 1 : 2 * G;
      hd_mfma<ACC_A, false, issued2 - (2 * g + 2)>(acc[t], fl[f], hd_frag(xb[s & 1][0]));           // wl . xh
      if constexpr (pf) hd_dsread<q3 * 2048 + 1024>(fl[f3], rg.pa[c3 % 3]);
      if constexpr (sp) hd_split2(xb[sn & 1][0][t], st0[t], st1[t], st2[t], xb[sn & 1][1][t]);
      hd_mfma<ACC_A, false, -1>(acc[t], fh[f], hd_frag(xb[s & 1][0]));                               // wh . xh
      // one piece of the chunk two ahead behind the tile steps that carry no operand split
      if constexpr (has_piece(q) && !(LAST && c + 2 >= NCH)) {
        constexpr int c2 = c + 2, j = piece_of(q);
        const char* src = (c2 < NCH ? p_cur + (size_t)c2 * CHB : p_nxt + (size_t)(c2 - NCH) * CHB) + (size_t)j * 4096;
        hd_dma(lane16, src, rg.ld[c2 % 3] + j * 4096);
      }
    });
    // the last MFMAs' results must not be read by the epilogue's VALU for 12 wait states (§5.7 item 2); every tile
    // is an operand of the fence, or the compiler hoists the epilogue's first reads above the last k-step's MFMAs
    if constexpr (ACC_A) {
      asm volatile("s_nop 15" : "+a"(acc[0])::"memory");
#pragma unroll
      for (int t = 1; t < NT; ++t) asm volatile("" : "+a"(acc[t]));
    } else {
      asm volatile("s_nop 15" : "+v"(acc[0])::"memory");
#pragma unroll
      for (int t = 1; t < NT; ++t) asm volatile("" : "+v"(acc[t]));
    }
    const Ring o = rg;                // the roles after NCH chunks
#pragma unroll
    for (int k = 0; k < 3; ++k) { rg.pa[k] = o.pa[(NCH + k) % 3]; rg.ld[k] = o.ld[(NCH + k) % 3]; }
  }

  // acc = acc * inv + v[feature]   (inv is a power of two: the fma rounds exactly as the add alone would)
  template <bool RELU>
  static __device__ __forceinline__ float bias(f32x16 (&acc)[NT], float inv, const float* v, int hh) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(v + 32 * t + 8 * g + 4 * hh);
        const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float y = __builtin_fmaf(acc[t][4 * g + j], inv, bb[j]);
          if (RELU) { y = fmaxf(y, 0.f); m = fmaxf(m, y); }
          acc[t][4 * g + j] = y;
        }
      }
    return m;
  }

  // LayerNorm over the H features of every candidate (a candidate's features: the 16 NT registers of lanes r, r+32),
  // then ReLU; returns the lane's largest result
  static __device__ __forceinline__ float layer_norm_relu(f32x16 (&acc)[NT], const float* g, const float* b, float eps, int hh) {
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
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 gg = *reinterpret_cast<const float4*>(g + 32 * t + 8 * gq + 4 * hh);
        const float4 bb = *reinterpret_cast<const float4*>(b + 32 * t + 8 * gq + 4 * hh);
        const float ga[4] = {gg.x, gg.y, gg.z, gg.w}, be[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float y = fmaxf((acc[t][4 * gq + j] - mean) * rstd * ga[j] + be[j], 0.f);
          m = fmaxf(m, y);
          acc[t][4 * gq + j] = y;
        }
      }
    return m;
  }

  static __device__ __forceinline__ float relu_max(f32x16 (&acc)[NT]) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[t][i] = fmaxf(acc[t][i], 0.f); m = fmaxf(m, acc[t][i]); }
    return m;
  }
};

template <int NT>
__global__ __launch_bounds__(OCN_BLOCK, 1) void heads_fused_kernel(const HeadsArgs a) {
  using HD = Heads<NT>;
  using Ring = typename HD::Ring;
  constexpr int H = HD::H, NPW = HD::NPW, CHB = HD::CHB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // vectors | three ring slots
  float* s_vec = reinterpret_cast<float*>(smem);
  const float* s_scal = s_vec + HD_NVEC * H;                               // [0] dot bias, [1 + P] inverse scale of panel P
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned lds0 = (unsigned)(size_t)(lds_bytes_t)smem;
  for (int q = threadIdx.x; q < HD::VEC_FLOATS; q += OCN_BLOCK) s_vec[q] = a.vec[q];
  // where this wave parks a finished branch's share of the output while the next branch needs the registers
  float4* park = reinterpret_cast<float4*>(a.scratch) + ((size_t)(blockIdx.x * 4 + w) * 2) * (NT * 4) * 64 + lane;
  const i64 n_tiles = a.dump ? 1 : (a.B + HD_ROWS - 1) / HD_ROWS;
#ifdef OCN_X_HD_CLOCK                /* diagnostic build only: the clock the chip holds under this kernel (guide, DVFS give-back item 6) */
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // uniform pointers as scalar pairs (the "s" operands of hd_dma)
  auto uni = [](const char* p) -> const char* {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
  };

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
    // (the barriers also mean: s_vec is written, and nobody reads the previous tile's ring slots any more)
    const int wgA = __syncthreads_or(has1), wgB = __syncthreads_or(hasB);

    Ring rg;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rg.pa[k] = lds0 + (unsigned)HD::LDS_W + (unsigned)(k * CHB) + lane16;
      rg.ld[k] = lds0 + (unsigned)HD::LDS_W + (unsigned)(k * CHB) + (unsigned)w * 1024u;
    }
    // the first two chunks of this tile's weight stream
    {
      const char* p0 = uni(a.panel[wgA ? P_A0 : (wgB ? P_B0 : P_X0)]) + (size_t)w * 1024;
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < NPW; ++q) hd_dma(lane16, p0 + (size_t)k * CHB + (size_t)q * 4096, rg.ld[k] + q * 4096);
      hd_sync<NPW>();                                                          // chunk 0 is there, chunk 1 on its way
    }

    f32x16 rA[NT], rB[NT];            // rA lives in VGPRs, rB in the accumulator file
    // the raw input rows of a branch, in accumulator layout; rows the pooling never wrote count as zero rows
    auto load_x = [&](const float* xb, bool rowmask) -> float {
      const float* xrow = xb + arow * a.ldx + 4 * hh;
      float m = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 v = *reinterpret_cast<const float4*>(xrow + 32 * t + 8 * g);
          const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float y = rowmask ? vv[j] : 0.f;
            m = fmaxf(m, fabsf(y));
            rB[t][4 * g + j] = y;
          }
        }
      HD::pin_a(rB);
      return m;
    };

    // ---- pooled branches a (xcn1lin) and b (xcn2lin): their share M . act of the output is parked in memory ------
#pragma unroll 1
    for (int br = 0; br < 2; ++br) {
      if (!(br == 0 ? wgA : wgB)) continue;
      const float* vb = s_vec + (br == 0 ? V_B0A : V_B0B) * H;                // b0, b3, gamma3, beta3 of this branch
      const char* p0 = uni(a.panel[3 * br]) + (size_t)w * 1024;
      const char* p1 = uni(a.panel[3 * br + 1]) + (size_t)w * 1024;
      const char* p2 = uni(a.panel[3 * br + 2]) + (size_t)w * 1024;
      const char* pn = uni(a.panel[(br == 0 && wgB) ? P_B0 : P_X0]) + (size_t)w * 1024;
      float sc, inv;
      hd_row_scale(load_x(br == 0 ? a.x[0] : a.x[1], br == 0 ? has1 : hasB), s_scal[1 + 3 * br], sc, inv);
      HD::template layer<false, true, false>(rA, rB, sc, rg, p0, p1, lane16);
      float m = HD::template bias<true>(rA, inv, vb, hh);
      HD::pin_v(rA);
      hd_row_scale(m, s_scal[2 + 3 * br], sc, inv);
      HD::template layer<true, false, false>(rB, rA, sc, rg, p1, p2, lane16);
      m = HD::template bias<false>(rB, inv, vb + H, hh);
      m = a.ln ? HD::layer_norm_relu(rB, vb + 2 * H, vb + 3 * H, a.eps, hh) : HD::relu_max(rB);
      HD::pin_a(rB);
      hd_row_scale(m, s_scal[3 + 3 * br], sc, inv);
      HD::template layer<false, true, false>(rA, rB, sc, rg, p2, pn, lane16);
      float4* pk = park + (size_t)br * (NT * 4) * 64;
      asm volatile("" : "+v"(pk));           // (or every one of the 2 x 4 NT addresses is precomputed outside the tile loop and kept)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i) rA[t][i] *= inv;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          pk[g * 64] = make_float4(rA[t][4 * g], rA[t][4 * g + 1], rA[t][4 * g + 2], rA[t][4 * g + 3]);
        pk += 4 * 64;
        asm volatile("" : "+v"(pk));
      }
      if (a.dump && w == 0 && r == 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) a.dump[br * H + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh] = rA[t][i];
      }
    }
    // ---- xijlin ---------------------------------------------------------------------------------------------
    {
      const char* px = uni(a.panel[P_X0]) + (size_t)w * 1024;
      const char* pc = uni(a.panel[P_MC]) + (size_t)w * 1024;
      float sc, inv;
      hd_row_scale(load_x(a.x[2], live), s_scal[1 + P_X0], sc, inv);
      HD::template layer<false, true, false>(rA, rB, sc, rg, px, pc, lane16);
      float m = HD::template bias<false>(rA, inv, s_vec + V_B0X * H, hh);
      m = a.ln ? HD::layer_norm_relu(rA, s_vec + V_GX * H, s_vec + V_EX * H, a.eps, hh) : HD::relu_max(rA);
      HD::pin_v(rA);
      hd_row_scale(m, s_scal[1 + P_MC], sc, inv);
      HD::template layer<true, false, true>(rB, rA, sc, rg, pc, pc, lane16);
      // ---- out = ((share a + share b) + share c) + folded bias; a skipped branch's share is its constant -------
      const float4* pa = park;
      const float4* pb = park + (size_t)(NT * 4) * 64;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        asm volatile("" : "+v"(pa), "+v"(pb));               // a tile at a time: addresses and values of all tiles at once is 400 registers
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int fo = 32 * t + 8 * g + 4 * hh;
          const float4 sa = wgA ? pa[g * 64] : *reinterpret_cast<const float4*>(s_vec + V_CA * H + fo);
          const float4 sb = wgB ? pb[g * 64] : *reinterpret_cast<const float4*>(s_vec + V_CB * H + fo);
          const float4 bf = *reinterpret_cast<const float4*>(s_vec + V_BF * H + fo);
          rB[t][4 * g + 0] = ((sa.x + sb.x) + rB[t][4 * g + 0] * inv) + bf.x;
          rB[t][4 * g + 1] = ((sa.y + sb.y) + rB[t][4 * g + 1] * inv) + bf.y;
          rB[t][4 * g + 2] = ((sa.z + sb.z) + rB[t][4 * g + 2] * inv) + bf.z;
          rB[t][4 * g + 3] = ((sa.w + sb.w) + rB[t][4 * g + 3] * inv) + bf.w;
        }
        asm volatile("" : "+a"(rB[t]));
        pa += 4 * 64;
        pb += 4 * 64;
      }
    }
    // ---- lin: LayerNorm, ReLU, Linear(H, 1) -------------------------------------------------------------------
    if (a.ln) HD::layer_norm_relu(rB, s_vec + V_GL * H, s_vec + V_EL * H, a.eps, hh);
    else HD::relu_max(rB);
    float d = 0.f;
    const float* dw = s_vec + V_DOTW * H;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 ww = *reinterpret_cast<const float4*>(dw + 32 * t + 8 * gq + 4 * hh);
        d += rB[t][4 * gq + 0] * ww.x + rB[t][4 * gq + 1] * ww.y + rB[t][4 * gq + 2] * ww.z + rB[t][4 * gq + 3] * ww.w;
      }
    d += __shfl_xor(d, 32, OCN_WAVE);
    if (live && hh == 0 && !a.dump) a.y[a.y_row_map ? a.y_row_map[slot] : slot] = d + s_scal[0];
  }
#ifdef OCN_X_HD_CLOCK
  if (threadIdx.x == 0) {            // into this workgroup's own (now dead) park area: nothing reads it
    unsigned long long* o = reinterpret_cast<unsigned long long*>(reinterpret_cast<float4*>(a.scratch) + ((size_t)(blockIdx.x * 4) * 2) * (NT * 4) * 64);
    o[0] = __builtin_amdgcn_s_memtime() - clk0;
    o[1] = __builtin_amdgcn_s_memrealtime() - rt0;
  }
#endif
}

// Wp[s][t][hi, lo][lane][8 halves]: k-step s = 2 tt + ss consumes accumulator tile tt, registers 8 ss .. 8 ss + 7, of
// the previous layer: element j of lane (r, hh) is scale * W[32 t + r][32 tt + 16 ss + 8 (j >> 2) + 4 hh + (j & 3)]
__global__ __launch_bounds__(OCN_BLOCK) void split_weight_f16_kernel(const float* __restrict__ W, int N, int K, float scale,
                                                                     h16x8* __restrict__ Wp) {
  const int NT = N >> 5;
  const i64 total = (i64)(K / 16) * NT * 64;
  for (i64 f = (i64)blockIdx.x * blockDim.x + threadIdx.x; f < total; f += (i64)gridDim.x * blockDim.x) {
    const int lane = (int)(f & 63);
    const int t = (int)((f >> 6) % NT);
    const int s = (int)((f >> 6) / NT);
    const int tt = s >> 1, ss = s & 1, hh = lane >> 5;
    const float* src = W + (i64)(32 * t + (lane & 31)) * K + 32 * tt + 16 * ss + 4 * hh;
    h16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = src[8 * (j >> 2) + (j & 3)] * scale;
      const _Float16 h = (_Float16)x;
      hi[j] = h;
      lo[j] = (_Float16)(x - (float)h);
    }
    h16x8* dst = Wp + ((i64)(s * NT + t) * 2) * 64 + lane;
    dst[0] = hi;
    dst[64] = lo;
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

int64_t ocn_heads_panel_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0 || (N & 31) || (K & 31)) return 0;
  return (int64_t)N * K * 4;
}

int ocn_heads_split_weight(const float* W, int32_t N, int32_t K, float scale, void* Wp, void* stream) {
  if (!W || !Wp || N <= 0 || K <= 0 || (N & 31) || (K & 31) || !(scale > 0.f)) return OCN_EINVAL;
  const i64 frags = (i64)(K / 16) * (N >> 5) * 64;
  hipLaunchKernelGGL(split_weight_f16_kernel, dim3(grid_for((frags + OCN_BLOCK - 1) / OCN_BLOCK, 1024)),
                     dim3(OCN_BLOCK), 0, (hipStream_t)stream, W, (int)N, (int)K, scale, (h16x8*)Wp);
  return launch_status();
}

int32_t ocn_heads_nvec(void) { return HD_NVEC; }
int32_t ocn_heads_nscal(void) { return HD_NSCAL; }

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
  for (int i = 0; i < 3; ++i) a.x[i] = h->x[i];
  a.panel[P_A0] = (const char*)h->p_first[0]; a.panel[P_A3] = (const char*)h->p_mid[0]; a.panel[P_MA] = (const char*)h->p_out[0];
  a.panel[P_B0] = (const char*)h->p_first[1]; a.panel[P_B3] = (const char*)h->p_mid[1]; a.panel[P_MB] = (const char*)h->p_out[1];
  a.panel[P_X0] = (const char*)h->p_first[2]; a.panel[P_MC] = (const char*)h->p_out[2];
  a.ldx = ldx; a.B = h->B; a.vec = h->vec; a.ranges = (const i64*)h->ranges; a.y_row_map = (const i64*)h->y_row_map;
  a.y = h->y; a.dump = h->dump; a.scratch = h->scratch; a.eps = h->eps; a.ln = h->ln; a.b_on_union = h->b_on_union;
  const i64 tiles = h->dump ? 1 : (h->B + HD_ROWS - 1) / HD_ROWS;
  hipStream_t st = (hipStream_t)stream;
  switch (h->H) {
    case 128: return heads_launch<4>(a, tiles, st);
    case 256: return heads_launch<8>(a, tiles, st);
    default: return OCN_EINVAL;
  }
}

}  // extern "C"
