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
  float* dump;                        // constants mode: [2][4 NT][64] float4 <- the shares Ma a, Mb b of row 0, in the park layout
  const float* cpark;                 // ... which come back here: what a skipped branch contributes
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
// operations of an asm statement nor pads its hazards; every wait below is counted by hand, see Heads::layer; the
// ISA is audited by tools/check_heads_asm.py) --------------------------------------------------------------------

// A tile step = three statements, each one MFMA and what is issued in its shadow.  Operands by name; an instruction
// group that a variant does not have is an empty string, its operands are simply not named.
#define HD_WAIT "s_waitcnt lgkmcnt(%[n])\n\t"
#define HD_MF0 "v_mfma_f32_32x32x16_f16 %[c], %[a], %[b], 0\n\t"
#define HD_MFA "v_mfma_f32_32x32x16_f16 %[c], %[a], %[b], %[c]\n\t"
#define HD_RD "ds_read_b128 %[nf], %[ad] offset:%[off]\n\t"
#define HD_SP1 "v_mul_f32 %[t0], %[x0], %[sc]\n\tv_mul_f32 %[t1], %[x1], %[sc]\n\tv_cvt_pk_f16_f32 %[H], %[t0], %[t1]\n\tv_cvt_f32_f16 %[t2], %[H]\n\t"
#define HD_SP2 "v_cvt_f32_f16_sdwa %[t3], %[H] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" \
               "v_sub_f32 %[t0], %[t0], %[t2]\n\tv_sub_f32 %[t1], %[t1], %[t3]\n\tv_cvt_pk_f16_f32 %[L], %[t0], %[t1]\n\t"
#ifndef OCN_X_HD_NOGLDS   /* timing experiment (tools/headsbench.py): no weight traffic */
#define HD_DMA "s_add_u32 m0, %[ld], %[imm]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vo], %[gb]\n\t"
#else
#define HD_DMA "s_add_u32 m0, %[ld], %[imm]\n\t"
#endif

// statement 1: (wait until at most N LDS reads are outstanding) acc (+)= wh . xl; the hi fragment of the tile three
// steps ahead; first half of an operand split: t = x * sc, H = f16x2(t0, t1), t2 = f32(H.lo)
template <bool CZ, int N, bool PF, int OFF, bool SP>
__device__ __forceinline__ void hd_s1(f32x16& c, const h16x8& a, const h16x8& b, h16x8& nf, unsigned ad, float x0, float x1, float sc,
                                      unsigned& H, float& t0, float& t1, float& t2) {
  static_assert(OFF >= 0 && OFF < 65536 && N >= 0 && N < 16, "immediates");
#define HD_S1(MF, CC, RD, SP1, OUTS, INS) asm volatile(HD_WAIT MF RD SP1 : [c] CC(c) OUTS : [a] "v"(a), [b] "v"(b), [n] "i"(N) INS)
#define HD_O_RD , [nf] "=&v"(nf)      /* early clobber: the read lands while the statement still reads its inputs */
#define HD_I_RD , [ad] "v"(ad), [off] "i"(OFF)
#define HD_O_SP , [H] "=&v"(H), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
#define HD_I_SP , [x0] "v"(x0), [x1] "v"(x1), [sc] "v"(sc)
  if constexpr (CZ) {
    if constexpr (PF && SP) HD_S1(HD_MF0, "=a", HD_RD, HD_SP1, HD_O_RD HD_O_SP, HD_I_RD HD_I_SP);
    else if constexpr (PF) HD_S1(HD_MF0, "=a", HD_RD, "", HD_O_RD, HD_I_RD);
    else if constexpr (SP) HD_S1(HD_MF0, "=a", "", HD_SP1, HD_O_SP, HD_I_SP);
    else HD_S1(HD_MF0, "=a", "", "", , );
  } else {
    if constexpr (PF && SP) HD_S1(HD_MFA, "+a", HD_RD, HD_SP1, HD_O_RD HD_O_SP, HD_I_RD HD_I_SP);
    else if constexpr (PF) HD_S1(HD_MFA, "+a", HD_RD, "", HD_O_RD, HD_I_RD);
    else if constexpr (SP) HD_S1(HD_MFA, "+a", "", HD_SP1, HD_O_SP, HD_I_SP);
    else HD_S1(HD_MFA, "+a", "", "", , );
  }
#undef HD_S1
}

// statement 2: acc += wl . xh; the lo fragment of the tile three steps ahead; second half of the split:
// L = f16x2(t0 - f32(H.lo), t1 - f32(H.hi))   (the differences are exact)
template <bool PF, int OFF, bool SP>
__device__ __forceinline__ void hd_s2(f32x16& c, const h16x8& a, const h16x8& b, h16x8& nf, unsigned ad, unsigned H, float& t0, float& t1,
                                      float t2, unsigned& L) {
  static_assert(OFF >= 0 && OFF < 65536, "immediates");
  float t3;
#define HD_S2(RD, SP2, OUTS, INS) asm volatile(HD_MFA RD SP2 : [c] "+a"(c) OUTS : [a] "v"(a), [b] "v"(b) INS)
#define HD_O_SP2 , [L] "=&v"(L), [t3] "=&v"(t3), [t0] "+v"(t0), [t1] "+v"(t1)
#define HD_I_SP2 , [H] "v"(H), [t2] "v"(t2)
  if constexpr (PF && SP) HD_S2(HD_RD, HD_SP2, HD_O_RD HD_O_SP2, HD_I_RD HD_I_SP2);
  else if constexpr (PF) HD_S2(HD_RD, "", HD_O_RD, HD_I_RD);
  else if constexpr (SP) HD_S2("", HD_SP2, HD_O_SP2, HD_I_SP2);
  else HD_S2("", "", , );
#undef HD_S2
}

// statement 3: acc += wh . xh; one LDS-DMA piece: 64 lanes x 16 bytes from gb + vo to the LDS byte address ld + IMM
// (+ 16 lane).  M0 is written and read in ONE statement; the compiler keeps nothing in M0 in this kernel (audited).
template <bool DMA, int IMM>
__device__ __forceinline__ void hd_s3(f32x16& c, const h16x8& a, const h16x8& b, unsigned vo, const char* gb, unsigned ld) {
  if constexpr (DMA) asm volatile(HD_MFA HD_DMA : [c] "+a"(c) : [a] "v"(a), [b] "v"(b), [vo] "v"(vo), [gb] "s"(gb), [ld] "s"(ld), [imm] "i"(IMM) : "memory", "scc");
  else asm volatile(HD_MFA : [c] "+a"(c) : [a] "v"(a), [b] "v"(b));
}

// one weight fragment (1 KiB per wave) from the ring; `f` is in flight until the hd_s1 that waits for it
template <int OFF>
__device__ __forceinline__ void hd_dsread(h16x8& f, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds offset");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF));
}

// operand split of two activations outside a k-step (the first k-step's operand)
__device__ __forceinline__ void hd_split(float x0, float x1, float sc, unsigned& H, unsigned& L) {
  float t0, t1, t2, t3;
  asm volatile(HD_SP1 HD_SP2 : [H] "=&v"(H), [L] "=&v"(L), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
               : [x0] "v"(x0), [x1] "v"(x1), [sc] "v"(sc));
}

// one LDS-DMA piece outside a k-step
__device__ __forceinline__ void hd_dma(unsigned vo, const char* gb, unsigned ld) {
  asm volatile(HD_DMA : : [vo] "v"(vo), [gb] "s"(gb), [ld] "s"(ld), [imm] "i"(0) : "memory", "scc");
}

// 16 bytes of a lane's input row; `v` is in flight until hd_xwait
template <int OFF>
__device__ __forceinline__ void hd_xload(f32x4& v, const float* row) {
  static_assert(OFF >= 0 && OFF < 4096, "global offset");
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(v) : "v"(row), "i"(OFF) : "memory");
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

__device__ __forceinline__ f32x4 hd_lds4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ float hd_max4(float m, f32x4 v) { return fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3])); }
__device__ __forceinline__ f32x4 hd_grp(const f32x16& v, int g) { return f32x4{v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]}; }
#define HD_GRP(v, g) hd_grp(v, g)

// A sum over a candidate's H features is formed as four quarter sums — tiles [w NT/4, (w + 1) NT/4), the features one wave of
// heads_nsplit_kernel owns — each a sequential f32x4 chain folded (p0 + p1) + (p2 + p3) and joined with the other half-lane,
// then (P0 + P1) + (P2 + P3): the order both forms of the head share.
__device__ __forceinline__ float hd_quarter(const f32x4& p) {
  const float P = (p[0] + p[1]) + (p[2] + p[3]);
  return P + __shfl_xor(P, 32, OCN_WAVE);
}
__device__ __forceinline__ float hd_quad(const float (&P)[4]) { return (P[0] + P[1]) + (P[2] + P[3]); }

template <int NT>
struct Heads {
  static constexpr int H = 32 * NT;
  static constexpr int TQ = NT / 4;                        // tiles per quarter sum
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

  // The three register sets of a wave.  in[4 t + g][j] = feature 32 t + 8 g + 4 hh + j of the lane's candidate: the
  // layer's input (arch VGPRs; the accumulator layout, register 4 g + j of tile t, in groups of four).  acc: the layer's
  // output; sum: the branches' shares of the last layer's input — both in the accumulator file.
  typedef f32x4 In[4 * NT];
  typedef f32x16 Acc[NT];

  static __device__ __forceinline__ void pin_in(In& v) {
#pragma unroll
    for (int t = 0; t < 4 * NT; ++t) asm volatile("" : "+v"(v[t]));
  }
  static __device__ __forceinline__ void pin_acc(Acc& v) {
#pragma unroll
    for (int t = 0; t < NT; ++t) asm volatile("" : "+a"(v[t]));
  }

  // the prefetched rows have landed
  static __device__ __forceinline__ void xwait(In& v) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
                 : : "memory");
#pragma unroll
    for (int t = 8; t < 4 * NT; t += 8)
      asm volatile("" : "+v"(v[t]), "+v"(v[t + 1]), "+v"(v[t + 2]), "+v"(v[t + 3]), "+v"(v[t + 4]), "+v"(v[t + 5]), "+v"(v[t + 6]), "+v"(v[t + 7]));
  }

  // acc = Wp . in on the 32 candidates of this wave; `sc` = the row's scale.  The weight stream: this layer's panel
  // p_cur (its chunk c is in ring role c % 3), then p_nxt; LAST: the stream ends with this layer.  XP: the 16 registers
  // of input tile c are dead once chunk c has split its second k-step — they receive tile c of the NEXT input rows
  // (xnext: the lane's row + 4 hh floats), so that the next first layer finds its input in registers.
  template <bool LAST, bool XP>
  static __device__ __forceinline__ void layer(Acc& acc, In& in, float sc, Ring& rg, const char* p_cur, const char* p_nxt,
                                               unsigned lane16, const float* xnext) {
    h16x8 fh[4], fl[4];
    unsigned xb[2][2][4];             // [k-step parity][hi, lo][4 registers]: the B operand
#pragma unroll
    for (int p = 0; p < 4; ++p)       // operand of k-step 0
      hd_split(in[p >> 1][2 * (p & 1)], in[p >> 1][2 * (p & 1) + 1], sc, xb[0][0][p], xb[0][1][p]);
    hd_dsread<0>(fh[0], rg.pa[0]);
    hd_dsread<1024>(fl[0], rg.pa[0]);
    hd_dsread<2048>(fh[1], rg.pa[0]);
    hd_dsread<3072>(fl[1], rg.pa[0]);
    hd_dsread<4096>(fh[2], rg.pa[0]);
    hd_dsread<5120>(fl[2], rg.pa[0]);
    float st0[4], st1[4], st2[4];     // split state between the two halves of a tile step
    h16x8 nof;                        // (operands a variant does not name)
    float nox = 0.f;
    unsigned nou = 0;
    hd_unroll<G>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int s = g / NT, t = g % NT, c = s / 2, q = g % TPC, f = g % 4;
      constexpr int g3 = g + 3, c3 = g3 / TPC, q3 = g3 % TPC, f3 = g3 % 4;
      constexpr bool pf = g3 < G;                                      // fragments of the tile three steps ahead
      constexpr bool sp = s + 1 < KS && t < 4;                         // a quarter of the next k-step's operand
      constexpr int sn = s + 1, si = sp ? 4 * (sn / 2) + 2 * (sn & 1) + (t >> 1) : 0, sj = 2 * (t & 1), tq = sp ? t : 0;
      // the chunk boundary, three tile steps early (the first read of the next chunk is this step's prefetch): the
      // pieces of the next chunk were issued a chunk ago; younger are only the pieces this chunk has issued so far
      if constexpr (q == TPC - 3 && !(LAST && c == NCH - 1)) hd_sync<(LAST && c + 2 >= NCH) ? 0 : piece_of(TPC - 3)>();
      // LDS reads are issued in the order hi(0) lo(0) hi(1) lo(1) ...: reads 2g and 2g + 1 must be back
      constexpr int issued = 2 * (g + 3) < 2 * G ? 2 * (g + 3) : 2 * G;
      hd_s1<s == 0, issued - (2 * g + 2), pf, q3 * 2048, sp>(acc[t], fh[f], hd_frag(xb[s & 1][1]), pf ? fh[f3] : nof, rg.pa[c3 % 3],
                                                             sp ? in[si][sj] : nox, sp ? in[si][sj + 1] : nox, sc,
                                                             sp ? xb[sn & 1][0][tq] : nou, st0[tq], st1[tq], st2[tq]);        // wh . xl
      hd_s2<pf, q3 * 2048 + 1024, sp>(acc[t], fl[f], hd_frag(xb[s & 1][0]), pf ? fl[f3] : nof, rg.pa[c3 % 3], xb[sn & 1][0][tq],
                                      st0[tq], st1[tq], st2[tq], sp ? xb[sn & 1][1][tq] : nou);                                 // wl . xh
      // one piece of the chunk two ahead behind the tile steps that carry no operand split
      constexpr bool dma = has_piece(q) && !(LAST && c + 2 >= NCH);
      constexpr int c2 = c + 2, jp = piece_of(q);
      const char* src = c2 < NCH ? p_cur + (size_t)c2 * CHB : p_nxt + (size_t)(c2 - NCH) * CHB;
      hd_s3<dma, jp * 4096>(acc[t], fh[f], hd_frag(xb[s & 1][0]), lane16 + jp * 4096, src, rg.ld[c2 % 3]);                    // wh . xh
      // the next rows' tile c into the registers of this layer's input tile c (last read: step 3 of chunk c)
      if constexpr (XP && q >= 4 && q < 8) hd_xload<(32 * c + 8 * (q - 4)) * 4>(in[4 * c + (q - 4)], xnext);
    });
    // the last MFMAs' results must not be read by the epilogue's VALU for 12 wait states (§5.7 item 2); every tile
    // is an operand of the fence, or the compiler hoists the epilogue's first reads above the last k-step's MFMAs
    asm volatile("s_nop 15" : "+a"(acc[0])::"memory");
#pragma unroll
    for (int t = 1; t < NT; ++t) asm volatile("" : "+a"(acc[t]));
    if constexpr (XP) xwait(in);
    const Ring o = rg;                // the roles after NCH chunks
#pragma unroll
    for (int k = 0; k < 3; ++k) { rg.pa[k] = o.pa[(NCH + k) % 3]; rg.ld[k] = o.ld[(NCH + k) % 3]; }
  }

  // ---- epilogues: accumulators (accumulator file) -> the next layer's input (VGPRs), on packed f32 pairs -----------
  // One tile at a time: left alone, the scheduler reads all 128 accumulator registers into VGPRs at once — beside the
  // 128 registers of the input set that is scratch traffic.  Tile t's results and tile t + 1's accumulators pass
  // through one empty statement, so no read of tile t + 1 is scheduled above the end of tile t.
  static __device__ __forceinline__ void next_tile(In& in, Acc& acc, int t) {
    if (t + 1 < NT) asm volatile("" : "+v"(in[4 * t]), "+v"(in[4 * t + 1]), "+v"(in[4 * t + 2]), "+v"(in[4 * t + 3]), "+a"(acc[t + 1]));
    else asm volatile("" : "+v"(in[4 * t]), "+v"(in[4 * t + 1]), "+v"(in[4 * t + 2]), "+v"(in[4 * t + 3]));
  }

  // in = ReLU(acc * inv + v[feature]); returns the lane's largest result.  inv is a power of two: the fma rounds exactly
  // as the add alone would.
  static __device__ __forceinline__ float bias_relu(In& in, Acc& acc, float inv, const float* v, int hh) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 y = __builtin_elementwise_fma(HD_GRP(acc[t], g), (f32x4)(inv), hd_lds4(v + 32 * t + 8 * g + 4 * hh));
        y = __builtin_elementwise_max(y, (f32x4)(0.f));
        m = hd_max4(m, y);
        in[4 * t + g] = y;
        if (g == 3) next_tile(in, acc, t);
      }
    return m;
  }

  // in = ReLU(LayerNorm(acc * inv + v[feature])) over the H features of every candidate (a candidate's features: the
  // 16 NT registers of lanes r, r + 32); returns the lane's largest result
  static __device__ __forceinline__ float bias_ln_relu(In& in, Acc& acc, float inv, const float* v, const float* gm, const float* bt,
                                                       float eps, int hh) {
    float P[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      f32x4 s4 = (f32x4)(0.f);
#pragma unroll
      for (int t = w * TQ; t < (w + 1) * TQ; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 y = __builtin_elementwise_fma(HD_GRP(acc[t], g), (f32x4)(inv), hd_lds4(v + 32 * t + 8 * g + 4 * hh));
          s4 += y;
          in[4 * t + g] = y;
          if (g == 3) next_tile(in, acc, t);
        }
      P[w] = hd_quarter(s4);
    }
    const float nmean = -(hd_quad(P) * (1.0f / (float)H));
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      f32x4 q4 = (f32x4)(0.f);
#pragma unroll
      for (int t = 4 * w * TQ; t < 4 * (w + 1) * TQ; ++t) {
        const f32x4 d = in[t] + (f32x4)(nmean);
        q4 = __builtin_elementwise_fma(d, d, q4);
        in[t] = d;
      }
      P[w] = hd_quarter(q4);
    }
    const float rstd = rsqrtf(hd_quad(P) * (1.0f / (float)H) + eps);
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fo = 32 * t + 8 * g + 4 * hh;
        f32x4 y = __builtin_elementwise_fma(in[4 * t + g] * (f32x4)(rstd), hd_lds4(gm + fo), hd_lds4(bt + fo));
        y = __builtin_elementwise_max(y, (f32x4)(0.f));
        m = hd_max4(m, y);
        in[4 * t + g] = y;
      }
    return m;
  }

  // a branch's share acc * inv of the last layer's input goes to this wave's park area (L2-resident) while the next
  // branch needs the registers: tile t, group g at pk[(4 t + g) * 64] (+ lane)
  static __device__ __forceinline__ void park_share(f32x4* pk_, Acc& acc, float inv, int lane) {
    __attribute__((address_space(1))) f32x4* pk = (__attribute__((address_space(1))) f32x4*)pk_;      // (global, not flat)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) pk[(4 * t + g) * 64 + lane] = HD_GRP(acc[t], g) * (f32x4)(inv);
      if (t + 1 < NT) asm volatile("" : "+a"(acc[t + 1]) : : "memory");                  // (tile by tile: see next_tile)
    }
  }
};

#define HD_PARK_BYTES(H) ((int64_t)HD_MAX_GRID * 4 * 2 * (H) * 32 * 4)
#define HD_MAX_GRID 256               /* one workgroup per CU (LDS and registers admit no second one): a persistent grid */
#ifdef OCN_X_HD_STAMPS               /* diagnostic build only (tools/headsbench.py): s_memtime at the phase boundaries of workgroup 0's first tile */
#define HD_STAMP(k) do { if (blockIdx.x == 0 && tile == 0 && threadIdx.x == 0) \
    reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.scratch) + HD_PARK_BYTES(H))[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HD_STAMP(k) do {} while (0)
#endif

template <int NT, bool LN>
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
  const i64 n_tiles = a.dump ? 1 : (a.B + HD_ROWS - 1) / HD_ROWS;
#ifdef OCN_X_HD_CLOCK                /* diagnostic build only: the clock the chip holds under this kernel (guide, DVFS give-back item 6) */
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), rt0 = __builtin_amdgcn_s_memrealtime();
#endif
  // class boundaries of the class-major order (both | cn1 only | cn2 only | none), or "every row runs every branch"
  i64 m3 = a.B, m32 = a.B, m321 = a.B;
  if (a.ranges) { m3 = a.ranges[2 * 1 + 1]; m32 = a.ranges[2 * 0 + 1]; m321 = a.ranges[2 * 3 + 1]; }
  // which branches a tile runs follows from the boundaries alone; branch 0 = xcn1lin, 1 = xcn2lin, 2 = xijlin
  auto tile_runs = [&](i64 tl, int br) -> bool {
    const i64 lo = tl * HD_ROWS, hi = lo + HD_ROWS < a.B ? lo + HD_ROWS : a.B;
    if (br == 0) return lo < m32;
    if (br == 1) return a.b_on_union ? lo < m321 : (lo < m3 || (lo > m32 ? lo : m32) < (hi < m321 ? hi : m321));
    return true;
  };
  auto row_has = [&](i64 slot, int br) -> bool {          // does this candidate's pooled input of branch br exist
    if (slot >= a.B) return false;
    if (br == 0) return slot < m32;
    if (br == 1) return a.b_on_union ? slot < m321 : (slot < m3 || (slot >= m32 && slot < m321));
    return true;
  };
  auto x_row = [&](i64 tl, int br) -> const float* {      // the lane's input row of branch br in tile tl (+ 4 hh floats)
    const i64 slot = tl * HD_ROWS + 32 * w + r;
    return a.x[br] + (slot < a.B ? slot : a.B - 1) * a.ldx + 4 * hh;
  };
  auto first_branch = [&](i64 tl) -> int { return tile_runs(tl, 0) ? 0 : (tile_runs(tl, 1) ? 1 : 2); };

  // uniform pointers as scalar pairs (the "s" operands of hd_dma)
  auto uni = [](const char* p) -> const char* {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
  };
  typename HD::In rA;                // the running layer's input (VGPRs)
  typename HD::Acc rB;               // its output (accumulator file)
  // where this wave parks the pooled branches' shares of the last layer's input
  // (a uniform base and a lane index: scalar-base addressing, no per-tile address registers)
  f32x4* park = reinterpret_cast<f32x4*>(const_cast<char*>(uni(reinterpret_cast<const char*>(
      reinterpret_cast<f32x4*>(a.scratch) + ((size_t)(blockIdx.x * 4 + w) * 2) * (NT * 4) * 64))));
  // the first tile's first input rows; every later first layer finds its rows prefetched by the layer before it
  {
    const float* xr = x_row(blockIdx.x, first_branch(blockIdx.x));
    hd_unroll<4 * NT>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      hd_xload<(32 * (i / 4) + 8 * (i % 4)) * 4>(rA[i], xr);
    });
    HD::xwait(rA);
  }
  // rows without a pooled input count as zero rows (the pooling never wrote them); returns the lane's largest |x|
  auto x_prepare = [&](bool has) -> float {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 4 * NT; ++t) {
      rA[t] = has ? rA[t] : (f32x4)(0.f);
      m = hd_max4(m, __builtin_elementwise_abs(rA[t]));
    }
    HD::pin_in(rA);
    return m;
  };

  // Tiles go to the workgroups in SNAKE order (round q: q G + b on even rounds, q G + G - 1 - b on odd ones): the rows are
  // class-major, so tile cost falls with the tile index (all three branches: 169 k cycles, xcn2lin + xijlin: 106 k, xijlin
  // only: 52 k) — the workgroup that drew a heavy tile first gets the lightest one next, instead of heavy + medium landing on
  // the first workgroups (collab shape: slowest workgroup 275 k -> 221 k cycles).
  auto tile_at = [&](i64 q) -> i64 {
    const i64 G = gridDim.x;
    const i64 t = q * G + ((q & 1) ? G - 1 - (i64)blockIdx.x : (i64)blockIdx.x);
    return t < n_tiles ? t : -1;
  };
#pragma unroll 1
  for (i64 round = 0;; ++round) {
    const i64 tile = tile_at(round);
    if (tile < 0) break;              // (only a last, partial round has tiles past the end: nothing follows it)
    const i64 slot = tile * HD_ROWS + 32 * w + r;
    const bool live = slot < a.B;
    const int wgA = tile_runs(tile, 0), wgB = tile_runs(tile, 1);
    const i64 ntile = tile_at(round + 1) >= 0 ? tile_at(round + 1) : tile;        // (past the end: this tile's rows once more)
    __syncthreads();                  // s_vec is written; nobody reads the previous tile's ring slots any more
    HD_STAMP(0);

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
    HD_STAMP(1);

    // ---- pooled branches a (xcn1lin) and b (xcn2lin): their shares M . act of the last layer's input -------------
#pragma unroll 1
    for (int br = 0; br < 2; ++br) {
      if (!(br == 0 ? wgA : wgB)) continue;
      const float* vb = s_vec + (br == 0 ? V_B0A : V_B0B) * H;                // b0, b3, gamma3, beta3 of this branch
      const char* p0 = uni(a.panel[3 * br]) + (size_t)w * 1024;
      const char* p1 = uni(a.panel[3 * br + 1]) + (size_t)w * 1024;
      const char* p2 = uni(a.panel[3 * br + 2]) + (size_t)w * 1024;
      const int nbr = (br == 0 && wgB) ? 1 : 2;                               // the branch whose rows the third layer prefetches
      const char* pn = uni(a.panel[3 * nbr]) + (size_t)w * 1024;
      float sc, inv;
      hd_row_scale(x_prepare(row_has(slot, br)), s_scal[1 + 3 * br], sc, inv);
      HD_STAMP(2 + 7 * br);
      HD::template layer<false, false>(rB, rA, sc, rg, p0, p1, lane16, nullptr);
      HD_STAMP(3 + 7 * br);
      float m = HD::bias_relu(rA, rB, inv, vb, hh);
      HD::pin_in(rA);
      hd_row_scale(m, s_scal[2 + 3 * br], sc, inv);
      HD_STAMP(4 + 7 * br);
      HD::template layer<false, false>(rB, rA, sc, rg, p1, p2, lane16, nullptr);
      HD_STAMP(5 + 7 * br);
      if constexpr (LN) m = HD::bias_ln_relu(rA, rB, inv, vb + H, vb + 2 * H, vb + 3 * H, a.eps, hh);
      else m = HD::bias_relu(rA, rB, inv, vb + H, hh);
      HD::pin_in(rA);
      hd_row_scale(m, s_scal[3 + 3 * br], sc, inv);
      HD_STAMP(6 + 7 * br);
      HD::template layer<false, true>(rB, rA, sc, rg, p2, pn, lane16, x_row(tile, nbr));
      HD_STAMP(7 + 7 * br);
      // (constants mode: B = 1, every lane holds row 0 — wave 0's share, as parked, IS the constant in the park layout)
      HD::park_share((a.dump && w == 0 ? reinterpret_cast<f32x4*>(a.dump) : park) + (size_t)br * (NT * 4) * 64, rB, inv, lane);
      HD_STAMP(8 + 7 * br);
    }
    // ---- xijlin, then out = ((share a + share b) + share c) + folded bias ---------------------------------------
    float inv;
    {
      const char* px = uni(a.panel[P_X0]) + (size_t)w * 1024;
      const char* pc = uni(a.panel[P_MC]) + (size_t)w * 1024;
      float sc;
      hd_row_scale(x_prepare(live), s_scal[1 + P_X0], sc, inv);
      HD_STAMP(16);
      HD::template layer<false, false>(rB, rA, sc, rg, px, pc, lane16, nullptr);
      HD_STAMP(17);
      float m;
      if constexpr (LN) m = HD::bias_ln_relu(rA, rB, inv, s_vec + V_B0X * H, s_vec + V_GX * H, s_vec + V_EX * H, a.eps, hh);
      else m = HD::bias_relu(rA, rB, inv, s_vec + V_B0X * H, hh);
      HD::pin_in(rA);
      hd_row_scale(m, s_scal[1 + P_MC], sc, inv);
      HD_STAMP(18);
      HD::template layer<true, true>(rB, rA, sc, rg, pc, pc, lane16, x_row(ntile, first_branch(ntile)));
      HD_STAMP(19);
    }
    // ---- lin: LayerNorm, ReLU, Linear(H, 1) on the accumulator file (the VGPR set already holds the next rows) -----
    float PS[4];                          // quarter sums of the last layer's input (hd_quad)
    f32x4 s4 = (f32x4)(0.f);
    {
      // ((share a + share b) + share c) + folded bias; a skipped branch's share is its constant
      typedef const __attribute__((address_space(1))) f32x4* gf4_t;          // (global, not flat: flat loads count out of order)
      gf4_t pk = (gf4_t)park;
      asm volatile("" : "+s"(pk));          // (opaque per tile: or the 64 load addresses are hoisted out of the tile loop and spilled)
      // (the constants come in the park layout too: both cases are global loads from a uniform base)
      gf4_t ck = (gf4_t)reinterpret_cast<const f32x4*>(a.cpark);
      asm volatile("" : "+s"(ck));
      gf4_t pa = wgA ? pk : ck;
      gf4_t pb = (wgB ? pk : ck) + (size_t)(NT * 4) * 64;
      f32x4 na[3][4], nb[3][4];            // the parked shares of tiles t + 1, t + 2 are requested before tile t is combined
      auto fetch = [&](int t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          na[t % 3][g] = pa[(4 * t + g) * 64 + lane];
          nb[t % 3][g] = pb[(4 * t + g) * 64 + lane];
        }
      };
      fetch(0);
      fetch(1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t + 2 < NT) fetch(t + 2);
        const f32x4 (&ca)[4] = na[t % 3];
        const f32x4 (&cb)[4] = nb[t % 3];
        f32x16 o;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = ((ca[g] + cb[g]) + HD_GRP(rB[t], g) * (f32x4)(inv)) + hd_lds4(s_vec + V_BF * H + 32 * t + 8 * g + 4 * hh);
          s4 += v;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[4 * g + j] = v[j];
        }
        rB[t] = o;
        if (t + 1 < NT) asm volatile("" : "+a"(rB[t]), "+a"(rB[t + 1]));
        else asm volatile("" : "+a"(rB[t]));
        if ((t + 1) % HD::TQ == 0) { PS[t / HD::TQ] = hd_quarter(s4); s4 = (f32x4)(0.f); }
      }
    }
    constexpr int TQ = HD::TQ;
    float PD4[4];                         // quarter sums of the dot product
    const float* dw = s_vec + V_DOTW * H;
    if constexpr (LN) {
      const float nmean = -(hd_quad(PS) * (1.0f / (float)H));
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        f32x4 q4 = (f32x4)(0.f);
#pragma unroll
        for (int t = w4 * TQ; t < (w4 + 1) * TQ; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 dd = HD_GRP(rB[t], g) + (f32x4)(nmean);
            q4 = __builtin_elementwise_fma(dd, dd, q4);
          }
        PS[w4] = hd_quarter(q4);
      }
      const float rstd = rsqrtf(hd_quad(PS) * (1.0f / (float)H) + a.eps);
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        f32x4 d4 = (f32x4)(0.f);
#pragma unroll
        for (int t = w4 * TQ; t < (w4 + 1) * TQ; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int fo = 32 * t + 8 * g + 4 * hh;
            f32x4 y = __builtin_elementwise_fma((HD_GRP(rB[t], g) + (f32x4)(nmean)) * (f32x4)(rstd), hd_lds4(s_vec + V_GL * H + fo),
                                                hd_lds4(s_vec + V_EL * H + fo));
            y = __builtin_elementwise_max(y, (f32x4)(0.f));
            d4 = __builtin_elementwise_fma(y, hd_lds4(dw + fo), d4);
          }
        PD4[w4] = hd_quarter(d4);
      }
    } else {
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        f32x4 d4 = (f32x4)(0.f);
#pragma unroll
        for (int t = w4 * TQ; t < (w4 + 1) * TQ; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            d4 = __builtin_elementwise_fma(__builtin_elementwise_max(HD_GRP(rB[t], g), (f32x4)(0.f)), hd_lds4(dw + 32 * t + 8 * g + 4 * hh), d4);
        PD4[w4] = hd_quarter(d4);
      }
    }
    const float d = hd_quad(PD4);
    if (live && hh == 0 && !a.dump) a.y[a.y_row_map ? a.y_row_map[slot] : slot] = d + s_scal[0];
    HD_STAMP(20);
  }
#ifdef OCN_X_HD_CLOCK
  if (threadIdx.x == 0) {            // into the scratch page of the diagnostic builds
    unsigned long long* o = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.scratch) + HD_PARK_BYTES(H)) + 64 + 2 * blockIdx.x;
    if (blockIdx.x < 192) { o[0] = __builtin_amdgcn_s_memtime() - clk0; o[1] = __builtin_amdgcn_s_memrealtime() - rt0; }
  }
#endif
}

// ---- small batches: the same head on a tile of 32 candidates per workgroup (latency instead of throughput) -------------
// heads_fused_kernel walks a candidate through eight layers on ONE wave: 8 x (12 288 matrix-pipe cycles + epilogue) = 70 us
// whatever the batch size — Cora's whole 1 152-candidate batch is nine such tiles on nine CUs.  Here the four waves of a
// workgroup share 32 candidates and split the OUTPUT features of every layer: wave w owns accumulator tiles w NT/4 ...,
// a quarter of the k-loop's MFMAs, and a quarter of every epilogue.  What the next layer needs of a finished one is its B
// operand: a lane's eight f16 pairs of k-step (t, ss) are the split of the SAME lane's accumulator registers 8 ss .. 8 ss + 7 of
// tile t, so the wave that owns tile t splits its results once and stores both fragments where all four waves read them —
// the operand buffer [k-step][hi, lo][lane], 2 NT KiB of LDS; no wave holds a whole activation row.  Row maxima (the scale)
// and the LayerNorm sums cross the waves through three small LDS arrays.  A weight fragment is used by exactly one wave of
// the workgroup: L2 -> registers, PD k-steps ahead (no LDS ring, no LDS-DMA).
// The arithmetic is heads_fused_kernel's, operation for operation — same panels, same k order, same split, same row scales,
// the sums over a row's features as four quarter sums combined (P0 + P1) + (P2 + P3) in both kernels (hd_quad) — so the two
// return the same BITS (tests/test_parity_gpu.py::test_heads_small_batch_kernel_is_bit_equal); a tile none of whose
// candidates has a branch's input takes the branch's constant from cpark, as there.
template <int NT>
struct HeadsN {
  using HD = Heads<NT>;
  static constexpr int H = 32 * NT, KS = HD::KS, TW = NT / 4, PD = 8, ROWS = 32;
  static constexpr int OB_VECS = KS * 2 * 64;                                   // the operand buffer, in 16-byte fragments per lane
  static constexpr int RED_FLOATS = 3 * 4 * 64;                                 // sums, squares / dot, maxima: [wave][lane]
  static constexpr size_t LDS_BYTES = (size_t)HD::LDS_W + (size_t)OB_VECS * 16 + (size_t)RED_FLOATS * 4;
  typedef h16x8 Frag[PD][TW][2];
  typedef f32x16 Acc[TW];
  typedef f32x4 Own[TW][4];            // this wave's part of a layer's output: [u][g][j] = feature 32 (w TW + u) + 8 g + 4 hh + j
  typedef const __attribute__((address_space(1))) h16x8* panel_t;
  static_assert(PD <= KS && KS % PD == 0, "prefetch ring");

  // The fragments (hi, lo) of k-step s for this wave's tiles: panel + vo + s NT 2 KiB, vo = (w TW 2) KiB + 16 lane.  Requested by
  // hand and waited for by count (wait<>): left to the compiler, the requests of a prefetch ring this deep sink down to their
  // uses — one L2 round trip per k-step.  Vector-memory operations complete in order, so "at most N outstanding" means the
  // N youngest; requests the compiler makes itself only make a count stricter than needed.
  template <int IMM>
  static __device__ __forceinline__ void gload(h16x8& f, unsigned vo, const char* panel) {
    static_assert(IMM >= 0 && IMM < 4096, "global offset");
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(f) : "v"(vo), "s"(panel), "i"(IMM));      // (read-only data: no memory clobber, LDS reads may pass)
  }
  template <int S>
  static __device__ __forceinline__ void fetch(Frag& f, unsigned vo, const char* panel) {
    const unsigned v = vo + (unsigned)(S * NT * 2048);
    hd_unroll<TW>([&](auto uc) {
      constexpr int u = decltype(uc)::value;
      gload<u * 2048>(f[S % PD][u][0], v, panel);
      gload<u * 2048 + 1024>(f[S % PD][u][1], v, panel);
    });
  }
  template <int N, int K>
  static __device__ __forceinline__ void wait(Frag& f) {
    if constexpr (TW == 2) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(f[K][0][0]), "+v"(f[K][0][1]), "+v"(f[K][1][0]), "+v"(f[K][1][1]) : "i"(N));
    else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(f[K][0][0]), "+v"(f[K][0][1]) : "i"(N));
  }

  // Every request has landed.  Wherever control flow joins (a skipped branch), the two paths may hold the ring in different
  // registers and the compiler reconciles them with moves — of registers it believes written: no request may be in flight there
  // (tools/check_heads_asm.py, rule 1, found exactly such a move).
  static __device__ __forceinline__ void drain(Frag& f) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < PD; ++k)
#pragma unroll
      for (int u = 0; u < TW; ++u) asm volatile("" : "+v"(f[k][u][0]), "+v"(f[k][u][1]));
  }

  // acc = (this wave's rows of Wp) . (the operand buffer); the fragments of the first PD k-steps are on their way, the layer
  // leaves those of the next panel so
  template <bool LAST>
  static __device__ __forceinline__ void layer(Acc& acc, const h16x8* ob /* + lane */, Frag& f, unsigned vo, const char* p_cur, const char* p_nxt) {
    h16x8 bq[2][2] = {{ob[0], ob[64]}, {}};                                              // the B operand, one k-step ahead
    hd_unroll<KS>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value, k = s % PD;
      constexpr int ahead = (LAST && KS - 1 - s < PD - 1) ? KS - 1 - s : PD - 1;       // k-steps requested after this one
      if constexpr (s + 1 < KS) { bq[(s + 1) & 1][0] = ob[(2 * s + 2) * 64]; bq[(s + 1) & 1][1] = ob[(2 * s + 3) * 64]; }
      const h16x8 bh = bq[s & 1][0], bl = bq[s & 1][1];
      wait<ahead * 2 * TW, k>(f);
#pragma unroll
      for (int u = 0; u < TW; ++u) {
        if constexpr (s == 0) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k][u][0], bl, (f32x16)(0.f), 0, 0, 0);     // wh . xl
        else acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k][u][0], bl, acc[u], 0, 0, 0);
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k][u][1], bh, acc[u], 0, 0, 0);                                  // wl . xh
        acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[k][u][0], bh, acc[u], 0, 0, 0);                                  // wh . xh
      }
      // the ring slot is free once its MFMAs have read it: the request is issued behind them in program order, and a load's
      // write-back cannot pass an earlier instruction's operand read
      if constexpr (s + PD < KS) { pin(acc); fetch<s + PD>(f, vo, p_cur); }
      else if constexpr (!LAST) { pin(acc); fetch<s + PD - KS + 0>(f, vo, p_nxt); }
    });
  }
  static __device__ __forceinline__ void pin(Acc& acc) {
#pragma unroll
    for (int u = 0; u < TW; ++u) asm volatile("" : "+a"(acc[u]));
  }

  // (R0 + R1) + (R2 + R3) of the four waves' values of this lane: `mine` goes to red[w][lane], everybody reads all four
  static __device__ __forceinline__ void cross_put(float* red, float mine, int w, int lane) { red[w * 64 + lane] = mine; }
  static __device__ __forceinline__ float cross_sum(const float* red, int lane) {
    return (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
  }
  static __device__ __forceinline__ float cross_max(const float* red, int lane) {
    return fmaxf(fmaxf(red[lane], red[64 + lane]), fmaxf(red[128 + lane], red[192 + lane]));
  }
  static __device__ __forceinline__ float quarter(const f32x4& p) { return hd_quarter(p); }      // this wave's P_w of hd_quad

  // y = acc * inv + v[feature] (ReLU) on this wave's tiles
  template <bool RELU>
  static __device__ __forceinline__ void bias(Own& y, Acc& acc, float inv, const float* v, int w, int hh) {
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        y[u][g] = __builtin_elementwise_fma(HD_GRP(acc[u], g), (f32x4)(inv), hd_lds4(v + 32 * (w * TW + u) + 8 * g + 4 * hh));
        if constexpr (RELU) y[u][g] = __builtin_elementwise_max(y[u][g], (f32x4)(0.f));
      }
  }
  // y = ReLU(LayerNorm(y)) over the whole row (two crossings); Heads::bias_ln_relu's arithmetic
  static __device__ __forceinline__ void ln_relu(Own& y, const float* gm, const float* bt, float eps, float* red, int w, int hh, int lane) {
    f32x4 s4 = (f32x4)(0.f);
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) s4 += y[u][g];
    cross_put(red, quarter(s4), w, lane);
    __syncthreads();
    const float nmean = -(cross_sum(red, lane) * (1.0f / (float)H));
    f32x4 q4 = (f32x4)(0.f);
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 d = y[u][g] + (f32x4)(nmean);
        q4 = __builtin_elementwise_fma(d, d, q4);
        y[u][g] = d;
      }
    cross_put(red + 256, quarter(q4), w, lane);
    __syncthreads();
    const float rstd = rsqrtf(cross_sum(red + 256, lane) * (1.0f / (float)H) + eps);
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fo = 32 * (w * TW + u) + 8 * g + 4 * hh;
        y[u][g] = __builtin_elementwise_max(__builtin_elementwise_fma(y[u][g] * (f32x4)(rstd), hd_lds4(gm + fo), hd_lds4(bt + fo)), (f32x4)(0.f));
      }
  }
  // hd_split's arithmetic as plain expressions (the sixteen splits of an operand step are independent chains the scheduler may
  // interleave; the asm form is one block each): H = f16x2(x sc), L = f16x2(x sc - f32(H)), every step correctly rounded
  static __device__ __forceinline__ void split(float x0, float x1, float sc, unsigned& Hh, unsigned& Ll) {
    typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
    const float t0 = x0 * sc, t1 = x1 * sc;
    const h16x2 h = {(_Float16)t0, (_Float16)t1};
    const h16x2 l = {(_Float16)(t0 - (float)h[0]), (_Float16)(t1 - (float)h[1])};
    Hh = __builtin_bit_cast(unsigned, h);
    Ll = __builtin_bit_cast(unsigned, l);
  }
  // the next layer's B operand from this layer's output: the row's scale (one crossing), this wave's values split and stored;
  // ABS: raw inputs (any sign) instead of ReLU outputs.  The closing barrier also ends the previous k-loop's reads of `ob`:
  // every wave is past it before the first store below (the crossing's barrier stands between).
  template <bool ABS>
  static __device__ __forceinline__ void operand(h16x8* ob /* + lane */, Own& y, float pinv, float& inv, float* red, int w, int lane) {
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) m = hd_max4(m, ABS ? __builtin_elementwise_abs(y[u][g]) : y[u][g]);
    cross_put(red + 512, m, w, lane);
    __syncthreads();
    float sc;
    hd_row_scale(cross_max(red + 512, lane), pinv, sc, inv);
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        unsigned xh[4], xl[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) split(y[u][2 * ss + (p >> 1)][2 * (p & 1)], y[u][2 * ss + (p >> 1)][2 * (p & 1) + 1], sc, xh[p], xl[p]);
        const int s = 2 * (w * TW + u) + ss;
        ob[(2 * s) * 64] = hd_frag(xh);
        ob[(2 * s + 1) * 64] = hd_frag(xl);
      }
    __syncthreads();
  }
};

#ifdef OCN_X_HN_STAMPS               /* diagnostic build only (tools/headslat.py -DOCN_X_HN_STAMPS): s_memtime at the phase boundaries of workgroup 0, wave 0 */
#define HN_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) \
    reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(a.scratch) + HD_PARK_BYTES(H))[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HN_STAMP(k) do {} while (0)
#endif

template <int NT, bool LN>
__global__ __launch_bounds__(OCN_BLOCK, 1) void heads_nsplit_kernel(const HeadsArgs a) {
  using HD = Heads<NT>;
  using HN = HeadsN<NT>;
  using panel_t = typename HN::panel_t;
  constexpr int H = HD::H, TW = HN::TW, PD = HN::PD, ROWS = HN::ROWS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // vectors | operand buffer | crossings
  float* s_vec = reinterpret_cast<float*>(smem);
  const float* s_scal = s_vec + HD_NVEC * H;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, hh = lane >> 5;
  h16x8* ob = reinterpret_cast<h16x8*>(smem + HD::LDS_W) + lane;
  float* red = reinterpret_cast<float*>(smem + HD::LDS_W + (size_t)HN::OB_VECS * 16);
  for (int q = threadIdx.x; q < HD::VEC_FLOATS; q += OCN_BLOCK) s_vec[q] = a.vec[q];
  i64 m3 = a.B, m32 = a.B, m321 = a.B;
  if (a.ranges) { m3 = a.ranges[2 * 1 + 1]; m32 = a.ranges[2 * 0 + 1]; m321 = a.ranges[2 * 3 + 1]; }
  const i64 tile = blockIdx.x;
  const i64 lo = tile * ROWS, hi = lo + ROWS < a.B ? lo + ROWS : a.B;
  const i64 slot = lo + r;
  const bool live = slot < a.B;
  const bool wgA = lo < m32;
  const bool wgB = a.b_on_union ? lo < m321 : (lo < m3 || (lo > m32 ? lo : m32) < (hi < m321 ? hi : m321));
  auto row_has = [&](int br) -> bool {
    if (!live) return false;
    if (br == 0) return slot < m32;
    if (br == 1) return a.b_on_union ? slot < m321 : (slot < m3 || (slot >= m32 && slot < m321));
    return true;
  };
  auto panel = [&](int P) -> const char* { return a.panel[P]; };
  const unsigned vo = (unsigned)(w * TW * 2048 + lane * 16);
  typename HN::Acc acc;
  typename HN::Frag fr;
  typename HN::Own xin[3];            // this wave's quarter of the three input rows of its lane's candidate
  const int br0 = wgA ? 0 : (wgB ? 1 : 2);
  {                                   // all inputs and the first fragments of the first panel are requested at once
#pragma unroll
    for (int br = 0; br < 3; ++br) {
      if (!(br == 0 ? wgA : (br == 1 ? wgB : true))) continue;
      const float* xr = a.x[br] + (live ? slot : a.B - 1) * a.ldx + 4 * hh + 32 * (w * TW);
#pragma unroll
      for (int u = 0; u < TW; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) xin[br][u][g] = *reinterpret_cast<const f32x4*>(xr + 32 * u + 8 * g);
    }
    const char* p0 = panel(br0 == 2 ? P_X0 : 3 * br0);
    hd_unroll<PD>([&](auto kc) { HN::template fetch<decltype(kc)::value>(fr, vo, p0); });
  }
  auto zero_unless = [&](typename HN::Own& y, bool has) {
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) y[u][g] = has ? y[u][g] : (f32x4)(0.f);
  };
  HN_STAMP(0);
  __syncthreads();                    // s_vec is written
  HN_STAMP(1);
  // (a + b): the pooled branches' shares of the last layer's input on this wave's tiles; -0 + x == x for every x
  f32x4 sab[TW][4];
#pragma unroll
  for (int u = 0; u < TW; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g) sab[u][g] = (f32x4)(-0.f);
  typedef const __attribute__((address_space(1))) f32x4* gf4_t;
  const gf4_t ck = (gf4_t)reinterpret_cast<const f32x4*>(a.cpark);
  typename HN::Own y;
  float inv;
  HN::drain(fr);                      // (the input rows are awaited here anyway)
#pragma unroll
  for (int br = 0; br < 2; ++br) {
    if (!(br == 0 ? wgA : wgB)) {     // nobody here has this branch's input: its constant (the park layout: [4 t + g][lane])
#pragma unroll
      for (int u = 0; u < TW; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) sab[u][g] += ck[(size_t)(br * 4 * NT + 4 * (w * TW + u) + g) * 64 + lane];
      continue;
    }
    const float* vb = s_vec + (br == 0 ? V_B0A : V_B0B) * H;
    const char *p0 = panel(3 * br), *p1 = panel(3 * br + 1), *p2 = panel(3 * br + 2);
    const int nbr = (br == 0 && wgB) ? 1 : 2;
    const char* pn = panel(nbr == 2 ? P_X0 : 3 * nbr);
    zero_unless(xin[br], row_has(br));
    HN::template operand<true>(ob, xin[br], s_scal[1 + 3 * br], inv, red, w, lane);
    HN_STAMP(2 + 8 * br);
    HN::template layer<false>(acc, ob, fr, vo, p0, p1);
    HN_STAMP(3 + 8 * br);
    HN::template bias<true>(y, acc, inv, vb, w, hh);
    HN::template operand<false>(ob, y, s_scal[2 + 3 * br], inv, red, w, lane);
    HN_STAMP(4 + 8 * br);
    HN::template layer<false>(acc, ob, fr, vo, p1, p2);
    HN_STAMP(5 + 8 * br);
    HN::template bias<!LN>(y, acc, inv, vb + H, w, hh);
    if constexpr (LN) HN::ln_relu(y, vb + 2 * H, vb + 3 * H, a.eps, red, w, hh, lane);
    HN_STAMP(6 + 8 * br);
    HN::template operand<false>(ob, y, s_scal[3 + 3 * br], inv, red, w, lane);
    HN_STAMP(7 + 8 * br);
    HN::template layer<false>(acc, ob, fr, vo, p2, pn);
    HN_STAMP(8 + 8 * br);
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) sab[u][g] += HD_GRP(acc[u], g) * (f32x4)(inv);
    HN::drain(fr);                    // (the paths join below)
    HN_STAMP(9 + 8 * br);
  }
  // ---- xijlin, then ((share a + share b) + share c) + folded bias on this wave's tiles --------------------------------
  const char *px = panel(P_X0), *pc = panel(P_MC);
  zero_unless(xin[2], live);
  HN::template operand<true>(ob, xin[2], s_scal[1 + P_X0], inv, red, w, lane);
  HN_STAMP(18);
  HN::template layer<false>(acc, ob, fr, vo, px, pc);
  HN_STAMP(19);
  HN::template bias<!LN>(y, acc, inv, s_vec + V_B0X * H, w, hh);
  if constexpr (LN) HN::ln_relu(y, s_vec + V_GX * H, s_vec + V_EX * H, a.eps, red, w, hh, lane);
  HN_STAMP(20);
  HN::template operand<false>(ob, y, s_scal[1 + P_MC], inv, red, w, lane);
  HN_STAMP(21);
  HN::template layer<true>(acc, ob, fr, vo, pc, pc);
  HN_STAMP(22);
#pragma unroll
  for (int u = 0; u < TW; ++u)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      y[u][g] = (sab[u][g] + HD_GRP(acc[u], g) * (f32x4)(inv)) + hd_lds4(s_vec + V_BF * H + 32 * (w * TW + u) + 8 * g + 4 * hh);
  // ---- lin: LayerNorm, ReLU, Linear(H, 1) -----------------------------------------------------------------------------
  const float* dw = s_vec + V_DOTW * H;
  f32x4 d4 = (f32x4)(0.f);
  if constexpr (LN) {
    f32x4 s4 = (f32x4)(0.f);
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) s4 += y[u][g];
    HN::cross_put(red, HN::quarter(s4), w, lane);
    __syncthreads();
    const float nmean = -(HN::cross_sum(red, lane) * (1.0f / (float)H));
    f32x4 q4 = (f32x4)(0.f);
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 dd = y[u][g] + (f32x4)(nmean);
        q4 = __builtin_elementwise_fma(dd, dd, q4);
      }
    HN::cross_put(red + 256, HN::quarter(q4), w, lane);
    __syncthreads();
    const float rstd = rsqrtf(HN::cross_sum(red + 256, lane) * (1.0f / (float)H) + a.eps);
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int fo = 32 * (w * TW + u) + 8 * g + 4 * hh;
        f32x4 z = __builtin_elementwise_fma((y[u][g] + (f32x4)(nmean)) * (f32x4)(rstd), hd_lds4(s_vec + V_GL * H + fo), hd_lds4(s_vec + V_EL * H + fo));
        z = __builtin_elementwise_max(z, (f32x4)(0.f));
        d4 = __builtin_elementwise_fma(z, hd_lds4(dw + fo), d4);
      }
  } else {
#pragma unroll
    for (int u = 0; u < TW; ++u)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        d4 = __builtin_elementwise_fma(__builtin_elementwise_max(y[u][g], (f32x4)(0.f)), hd_lds4(dw + 32 * (w * TW + u) + 8 * g + 4 * hh), d4);
  }
  HN::cross_put(red + 512, HN::quarter(d4), w, lane);
  __syncthreads();
  if (w == 0 && live && hh == 0) a.y[a.y_row_map ? a.y_row_map[slot] : slot] = HN::cross_sum(red + 512, lane) + s_scal[0];
  HN_STAMP(23);
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

template <int NT, bool LN>
static int heads_launch(const HeadsArgs& a, i64 tiles, hipStream_t st) {
  static bool raised_dev[64] = {};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return OCN_EINVAL;
  if (!raised_dev[devid]) {
    const hipError_t e = hipFuncSetAttribute((const void*)heads_fused_kernel<NT, LN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)Heads<NT>::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    raised_dev[devid] = true;
  }
  hipLaunchKernelGGL((heads_fused_kernel<NT, LN>), dim3((unsigned)(tiles < HD_MAX_GRID ? tiles : HD_MAX_GRID)), dim3(OCN_BLOCK),
                     Heads<NT>::LDS_BYTES, st, a);
  return launch_status();
}

template <int NT, bool LN>
static int heads_nsplit_launch(const HeadsArgs& a, hipStream_t st) {
  static bool raised_dev[64] = {};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid >= 64) return OCN_EINVAL;
  if (!raised_dev[devid]) {
    const hipError_t e = hipFuncSetAttribute((const void*)heads_nsplit_kernel<NT, LN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)HeadsN<NT>::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    raised_dev[devid] = true;
  }
  const i64 tiles = (a.B + HeadsN<NT>::ROWS - 1) / HeadsN<NT>::ROWS;
  hipLaunchKernelGGL((heads_nsplit_kernel<NT, LN>), dim3((unsigned)tiles), dim3(OCN_BLOCK), HeadsN<NT>::LDS_BYTES, st, a);
  return launch_status();
}

// batches up to this many candidates take heads_nsplit_kernel (two rounds of 32-candidate workgroups on the 256 CUs: 74 against
// 87 us with every branch on every row, 46 against 79 us with a Cora-like class mix; tools/headslat.py)
static int64_t g_heads_small_batch = 16384;

extern "C" {

int64_t ocn_heads_small_batch(int64_t max_rows) {
  const int64_t prev = g_heads_small_batch;
  if (max_rows >= 0) g_heads_small_batch = max_rows;
  return prev;
}

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

int64_t ocn_heads_const_bytes(int32_t H) { return (int64_t)2 * H * 32 * 4; }   /* two shares of one wave, park layout */

int64_t ocn_heads_scratch_bytes(int32_t H) { return HD_PARK_BYTES(H) + 4096; }   /* + a page of diagnostic stamps */

int ocn_heads_fused(const OcnHeadsArgs* h, void* stream) {
  if (!h || h->B < 0 || h->H <= 0) return OCN_EINVAL;
  if (h->B == 0) return 0;
  for (int i = 0; i < 3; ++i)
    if (!h->x[i] || !h->p_first[i] || !h->p_out[i]) return OCN_EINVAL;
  if (!h->p_mid[0] || !h->p_mid[1] || !h->vec || !h->scratch || (!h->y && !h->dump)) return OCN_EINVAL;
  if (h->dump ? h->B != 1 : !h->cpark) return OCN_EINVAL;
  const int64_t ldx = h->ldx ? h->ldx : h->H;
  if (ldx < h->H || (ldx & 3)) return OCN_EINVAL;
  HeadsArgs a;
  for (int i = 0; i < 3; ++i) a.x[i] = h->x[i];
  a.panel[P_A0] = (const char*)h->p_first[0]; a.panel[P_A3] = (const char*)h->p_mid[0]; a.panel[P_MA] = (const char*)h->p_out[0];
  a.panel[P_B0] = (const char*)h->p_first[1]; a.panel[P_B3] = (const char*)h->p_mid[1]; a.panel[P_MB] = (const char*)h->p_out[1];
  a.panel[P_X0] = (const char*)h->p_first[2]; a.panel[P_MC] = (const char*)h->p_out[2];
  a.ldx = ldx; a.B = h->B; a.vec = h->vec; a.ranges = (const i64*)h->ranges; a.y_row_map = (const i64*)h->y_row_map;
  a.y = h->y; a.dump = h->dump; a.cpark = h->cpark; a.scratch = h->scratch; a.eps = h->eps; a.ln = h->ln; a.b_on_union = h->b_on_union;
  const i64 tiles = h->dump ? 1 : (h->B + HD_ROWS - 1) / HD_ROWS;
  hipStream_t st = (hipStream_t)stream;
  // (the bound is quoted at H = 256; a head half as wide streams a quarter of the panel bytes per workgroup and stays ahead twice as far:
  // H = 128, 32 768 rows: 45.3 against 45.9 us with every branch, 30.6 against 36.6 with a class mix)
  if (!h->dump && h->B <= g_heads_small_batch * (256 / h->H)) switch (h->H) {
      case 128: return h->ln ? heads_nsplit_launch<4, true>(a, st) : heads_nsplit_launch<4, false>(a, st);
      case 256: return h->ln ? heads_nsplit_launch<8, true>(a, st) : heads_nsplit_launch<8, false>(a, st);
      default: return OCN_EINVAL;
    }
  switch (h->H) {
    case 128: return h->ln ? heads_launch<4, true>(a, tiles, st) : heads_launch<4, false>(a, tiles, st);
    case 256: return h->ln ? heads_launch<8, true>(a, tiles, st) : heads_launch<8, false>(a, tiles, st);
    default: return OCN_EINVAL;
  }
}

}  // extern "C"
