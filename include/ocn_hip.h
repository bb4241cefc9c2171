/*
 * ocn_hip.h — C ABI of libocn_hip.so, the MI355X (gfx950) implementation of OCN's
 * common-neighbour predictor hot path.
 *
 * The reference (qingpingmo/OCN) is pure Python: the boundary of this path is a Python
 * module API, and every entry below replaces the third-party native call
 * the reference makes at the cited file:line (paths relative to the reference root).
 * The Python mirror in ocn_amd/ binds these with ctypes.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller; nothing is allocated or freed
 *    inside the library and no entry synchronises the host (graph-capturable);
 *  - `stream` is a hipStream_t passed as void*; work is enqueued on it and the call returns;
 *  - CSR adjacency: rowptr int64 [n_rows+1], col int32 [nnz], columns ascending in a row;
 *  - candidate edges: int64 src[B], dst[B] (the reference's LongTensor tar_ei rows);
 *  - return value: 0 on success, a positive hipError_t from the launch, or a negative
 *    OCN_E* code for argument errors.  Never throws.
 */
#ifndef OCN_HIP_H
#define OCN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history — 2: workspaces zero on entry / left zero, ocn_check_edges, ocn_zero_regions, ocn_cn_colsum_exact, walk prep /
 * group entries, dense block route, ocn_heads_fused; 3: ocn_cn_flags takes bit rows of T1, ocn_bitrows_from_csr;
 * 4: ocn_batch_prep, ocn_order_by_node_finish; 5: slot records (`rec`) from ocn_cn_flags to ocn_cn_gather;
 * 6: ocn_heads_fused on f16 hi/lo panels (ocn_heads_split_weight replaces ocn_linear_split_weight_chained).
 * 7: + ocn_coo_to_csr, ocn_wgrad, ocn_cn_gather_backward_det, ocn_gather_schedule; ocn_cn_flags gains `gcost`,
 *    ocn_cn_gather gains `perm` and `rowsum`.
 * 8: a scan whose workspace was not zero ends in OCN_SCAN_POISON totals and a status bit instead of a GPU trap; `status` is
 *    int32[4] for every intersection entry, word 3 the sticky error word; ocn_cn_weights_cn7 takes the Chebyshev diagonals,
 *    ocn_gather_schedule a segment; + ocn_cn_gather3_backward, ocn_cn_gather_backward_det_lists, ocn_ln_drop_relu_*.
 * 9: + ocn_heads_small_batch (ocn_heads_fused picks its small-batch form by the batch size; same bits), ocn_spgemm_bit_rows
 *    (ocn_cn_flags accepts rowptrT2 == NULL beside bitmapT2). */
#define OCN_ABI_VERSION 9
#define OCN_EINVAL (-1)   /* null pointer / negative size / unsupported combination */
#define OCN_ECAP   (-2)   /* reported through the device status word: flags capacity too small */

/* Device status words of a candidate batch: int32[4].  [0] = error bits of THIS batch (cleared by the caller / by
 * ocn_batch_prep's resets with the batch's other scratch), [1], [2] = work-item tickets of the walk route, [3] = the same
 * error bits, STICKY: the library only ever ORs into it; the caller clears it when it has read it, so a scoring loop can
 * check a whole split with one read (ocn_amd.pipeline does) instead of one host sync per batch. */
#define OCN_ST_CAP  1     /* off[B] > flags_cap: the flag buffer is too small (nothing is written past the cap) */
#define OCN_ST_SCAN 2     /* the batch's offsets come from a scan that gave up (OCN_SCAN_POISON): its workspace was not zero */
#define OCN_SCAN_POISON (-1)   /* out[n] of a scan entry whose workspace was not zero on entry (see ocn_scan_workspace_bytes) */

/* bits of a CN flag byte */
#define OCN_F_CN1 1u      /* neighbour k of src is in adj1-target row of dst  (cn1 entry) */
#define OCN_F_CN2 2u      /* neighbour k of src is in adj2-target row of dst  (cn2 entry) */

int ocn_abi_version(void);

/* Scratch bytes the scan entries need for `n` items.  The workspace must be ZERO when first handed to the library
 * (the caller zeroes it once, when it allocates it); every entry leaves it zero again, so it can be reused from call
 * to call — inputs beyond one tile are scanned by a single launch whose tiles chain through this state.  A workspace
 * that is NOT zero breaks the chain (two launches in flight on one workspace, a buffer never cleared).  The launch then
 * neither hangs nor traps: after a bounded wait the tiles that cannot be chained write ZERO offsets for their items and
 * the total out[n] becomes OCN_SCAN_POISON (< 0).  Consumers test the total and do not trust the offsets: ocn_cn_flags /
 * ocn_cn_walk_flags raise OCN_ST_SCAN in their status words and treat the batch as empty (zero counts, empty slot
 * records), ocn_order_by_node_finish / ocn_class_order fall back to batch order (a correct order, only slower), and a
 * caller that reads a total on the host (nnz of a product, the flag capacity) sees a negative number.  Detection is
 * best effort: the contract (ZERO on entry, one launch at a time per workspace) stays the caller's. */
int64_t ocn_scan_workspace_bytes(int64_t n);

/* Zero up to 8 device arrays (4-byte aligned, byte counts multiples of 4) with one launch: the per-batch reset of
 * the histogram / counters / status words. */
int ocn_zero_regions(void* const* ptrs, const int64_t* bytes, int32_t n, void* stream);

/* off[e] = sum_{e'<e} deg_A(src[e']), off[B] = total: where each batch row starts in the
 * flag buffer.  Replaces the row bookkeeping of SparseTensor.__getitem__ (utils.py:256). */
int ocn_edge_offsets(const int64_t* rowptrA, const int64_t* src, int64_t B,
                     int64_t* off, void* workspace, void* stream);

/* bad[0] |= 1 when any src[e] is outside [0, n_src) or any dst[e] outside [0, n_dst) (bad[0] is NOT
 * cleared here).  The reference's row selection raises IndexError for such an id
 * (SparseTensor.__getitem__ -> index_select, utils.py:256-257); the kernels below would read out of
 * bounds instead, so the Python mirror runs this check first and raises the same exception. */
int ocn_check_edges(const int64_t* src, const int64_t* dst, int64_t B, int64_t n_src, int64_t n_dst,
                    int32_t* bad, void* stream);

/* A processing order for a candidate batch: order[] = the batch rows counting-sorted by the node id
 * in `node` (arbitrary order among equal ids).  Visiting rows with the same / nearby source node
 * together lets the rows they share be served from L2; it never changes a result.
 * workspace: ocn_order_workspace_bytes(n_nodes) bytes of device scratch, ZERO when first handed over and left zero
 * (per-node counters that each batch row clears again, and the scan state). */
int64_t ocn_order_workspace_bytes(int64_t n_nodes);
int ocn_order_by_node(const int64_t* node, int64_t B, int64_t n_nodes, int64_t* order, void* workspace,
                      void* stream);

/* One launch for everything in front of the intersection pass of a large batch: the flag offsets (ocn_edge_offsets), the
 * counting phase of ocn_order_by_node (order_workspace, or NULL: no processing order) and the batch's resets
 * (ocn_zero_regions' arguments) — the two latter ride on extra workgroups of the scan's launch.  Follow with
 * ocn_order_by_node_finish (scan of the counters + scatter) for the order itself.  Workspaces: ZERO on entry, left zero. */
int ocn_batch_prep(const int64_t* rowptrA, const int64_t* src, int64_t B, int64_t* off /* [B+1] */, void* scan_workspace,
                   int64_t n_nodes, void* order_workspace /* or NULL */, void* const* zero_ptrs, const int64_t* zero_bytes,
                   int32_t n_zero, void* stream);
int ocn_order_by_node_finish(const int64_t* node, int64_t B, int64_t n_nodes, int64_t* order, void* workspace,
                             void* stream);

/* Forward work-item offsets of the walk route: out[slot] = number of items of the earlier processing
 * slots, out[B] = number of items.  A batch row's items are groups of consecutive chunks of
 * ocn_walk_chunk() neighbours of its source; with nds (ocn_neighbor_degree_sum, or NULL) the group
 * size adapts so that an item sweeps a bounded number of elements — pass the SAME nds to
 * ocn_cn_walk_flags. */
int ocn_chunk_offsets(const int64_t* rowptrA, const int64_t* nds /* or NULL */, const int64_t* src,
                      const int64_t* order, int64_t B, int64_t* out, void* workspace, void* stream);

/* Class-major processing order for the MLP heads.  A candidate with no cn1 (cn2) entry has an all-zero
 * pooled xcn1 (xcn2), and a zero row through `xcn1lin` is the same constant vector for every such
 * candidate: sorted by class = (cnt1 > 0, cnt2 > 0) — 3 both, 2 cn1 only, 1 cn2 only, 0 none; stable, so
 * the order inside a class is that of `order_in` (or batch order) — the heads run on contiguous row
 * ranges and skip the rest (62 % / 47 % of a collab-shaped evaluation batch have no cn1 / cn2 entry).
 * order_out[slot] = batch row, inv_out[batch row] = slot; ranges[OCN_CLASS_RANGES][2] = {begin, end} of
 *   0: cn1 > 0   1: both   2: cn2 only   3: any   4: none   5: cn1 only   6: all rows
 * (device-resident: OcnLinearGroup.row_range points into it, no host sync).  prefix: int64[B+1] scratch;
 * workspace: ocn_scan_workspace_bytes(B). */
#define OCN_CLASS_RANGES 7
int ocn_class_order(const int32_t* cnt1, const int32_t* cnt2 /* or NULL */, const int64_t* order_in /* or NULL */,
                    int64_t B, int64_t* order_out, int64_t* inv_out, int64_t* ranges, int64_t* prefix,
                    void* workspace, void* stream);

/* Exclusive scan of int32 counts into int64 offsets (out[n] = total). */
int ocn_scan_i32(const int32_t* in, int64_t n, int64_t* out, void* workspace, void* stream);

/* The intersection: utils.adjoverlap -> spmoverlap_ (utils.py:162-183, 248-285), both calls
 * of a batch fused.  For edge e=(i,j) and the p-th neighbour k of i in A:
 *   flags[off[e]+p] = [k in T1 row j] | [k in T2 row j] << 1
 * cnt1[e] / cnt2[e] = |cn1_e| / |cn2_e| (the integer CN counts), and the per-column
 * histograms are accumulated (cn.sum(dim=0), model.py:2261,3114): hist[k] = {packed, walks}
 * with packed = n1 | n2 << 21 | n_union << 42 (one 64-bit atomic per CN entry) and walks = 0
 * here; must be zero on entry; B < 2^21.  T2 may be NULL (single adjoverlap call).
 * bitmapT2: optional dense bit rows of T2 (row j at bitmapT2 + j*bm_stride_words, bit k = column k;
 * written by ocn_spgemm_pattern_count): membership in the long A² row becomes one probe.
 * rec (or NULL): per processing SLOT a 32-byte record {batch row, src | dst << 32, start of N(src) | length << 40,
 * off | (cnt1 > 0) << 62 | (cnt2 > 0) << 63}: ocn_cn_gather given the same `rec` reads it instead of walking
 * order -> src / dst / off / counts -> rowptr (one dependent load in front of its first gather instead of three).
 * gcost (or NULL): int32[ceil(B / 4)]: per group of four consecutive processing slots the number of cn1 + cn2 entries
 * its candidates have — what the group will cost the pooling; see ocn_gather_schedule.
 * bitmapT1: the same for T1 (ocn_bitrows_from_csr; small dense graphs); rowptrT1 / colT1 may then be NULL.
 * status: device int32[4] (above): OCN_ST_CAP if off[B] > flags_cap (nothing is written past the cap), OCN_ST_SCAN if
 * off[B] is OCN_SCAN_POISON; both also ORed into the sticky word status[3].
 * order (here and below): optional permutation of 0..B-1 giving the order in which the batch rows
 * are PROCESSED (e.g. sorted by src so that rows sharing neighbourhoods meet in L2); every output
 * stays indexed by the batch row.  NULL = batch order. */
int ocn_cn_flags(const int64_t* rowptrA, const int32_t* colA,
                 const int64_t* rowptrT1, const int32_t* colT1,
                 const int64_t* rowptrT2, const int32_t* colT2,
                 const uint32_t* bitmapT1 /* or NULL */, int64_t bm1_stride_words,
                 const uint32_t* bitmapT2 /* or NULL */, int64_t bm_stride_words,
                 const int64_t* src, const int64_t* dst, const int64_t* order, int64_t B,
                 int64_t n_cols, const int64_t* off, uint8_t* flags, int64_t flags_cap,
                 uint64_t* hist /* [n_cols][2] */, int32_t* cnt1, int32_t* cnt2,
                 int32_t* status, uint64_t* rec /* [B][4] or NULL */, int32_t* gcost /* [ceil(B/4)] or NULL */, void* stream);

/* The pygho route get_cn1_cn2 (NeighborOverlap_large_ppa.py:147-173, NeighborOverlapCitation2.py:
 * 78-104) without forming Ej·A: cn1 = N(i) ∩ N(j) as above; cn2[e,k] = |N(k) ∩ N(j)| (number of
 * 2-walks j -> k) for k in N(i), kept where > 0.  wc[off[e]+p] receives the walk count, hist[k]
 * additionally accumulates walks = sum of the counts of column k; cnt2[e] = number of non-zero
 * entries of cn2 row e.  Work is cut into items of ocn_walk_chunk() neighbours of i (chunk_off from
 * ocn_chunk_offsets), so hub source nodes spread over many workgroups; cnt1 / cnt2 must be ZERO on
 * entry (a row's items add into them).  `status`: words [0] .. [2] ZERO on entry: [0] receives the
 * error bits as for ocn_cn_flags, [1] and [2] are the work-item ticket counters of the two sweeps, [3] the sticky word. */
int32_t ocn_walk_chunk(void);
int ocn_cn_walk_flags(const int64_t* rowptrA, const int32_t* colA, const int64_t* nds /* or NULL */,
                      const int64_t* src, const int64_t* dst, const int64_t* order, int64_t B,
                      const int64_t* chunk_off, const int64_t* rev_off /* NULL iff nds is */, const int64_t* off,
                      int64_t max_row_len /* longest row of A: sizes the LDS set; <= 0 = unknown */,
                      uint8_t* flags, int32_t* wc, int64_t flags_cap,
                      uint64_t* hist /* [N][2] */, int32_t* cnt1, int32_t* cnt2,
                      int32_t* status, void* stream);

/* Two-sided enumeration of the same walk counts: cn2[e,k] is the number of 2-walks j -> m -> k, which
 * can be swept from i's side (Σ_{k∈N(i)} deg k elements) or from j's side (Σ_{m∈N(j)} deg m).  With
 * nds[v] = Σ_{u∈N(v)} deg u (ocn_neighbor_degree_sum, once per adjacency) and rev_off = exclusive scan
 * of the reverse work items per batch row (ocn_walk_rev_offsets; 0 for rows swept forward),
 * ocn_cn_walk_flags sweeps each batch row from its cheaper endpoint; results are identical (integer
 * counts).  The pygho reference always expands from j (spspmm(Ej, 1, adj, 0)). */
int ocn_neighbor_degree_sum(const int64_t* rowptr, const int32_t* col, int64_t n_rows, int64_t* out,
                            void* stream);
int ocn_walk_rev_offsets(const int64_t* rowptrA, const int64_t* nds, const int64_t* src, const int64_t* dst,
                         const int64_t* order, int64_t B, int64_t* out /* [B+1] */, void* workspace,
                         void* stream);

/* Small batches of the walk route (B <= ocn_walk_prep_max_batch(): the ppa / citation2 drivers use 2048) — everything in
 * front of the walk kernels in ONE single-workgroup launch: order[] = the batch rows sorted by (source, batch row); off =
 * flag offsets (as ocn_edge_offsets); groups = runs of equal source cut into pieces of 64 (g_head[g] = first slot of
 * group g, g_head[n_groups] = B, meta[0] = n_groups); the members of a group whose target has at most 512 neighbours go
 * to the shared sweep ocn_cn_walk_group if their targets' rows sum to at most 4096 entries — if at least `min_share`
 * of them do (g_active[slot]; g_item_off = exclusive scan of the group's work items, one per meta[2] x 64 neighbours of the
 * source) —, every other candidate keeps its per-candidate work items in chunk_off / rev_off (as ocn_chunk_offsets /
 * ocn_walk_rev_offsets; 0 for the candidates of the shared sweep); cnt1, cnt2, status[0..2], scal[4] are cleared.  meta: int32[4] ([1] = the shared sweep's ticket).  rev_off NULL iff nds is.
 *
 * ocn_cn_walk_group: the rows N(k), k in N(i), that a candidate (i, j) sweeps for cn2[e,k] = |N(k) n N(j)| are the same
 * for every candidate with source i (the MRR layout scores 1000 negatives per source, NeighborOverlapCitation2.py:
 * 248-254): they are swept once per group, each element probed against ONE hash table (node -> 64-bit mask of the
 * group's targets it neighbours).  Same outputs as ocn_cn_walk_flags, for the candidates of the shared groups; run it
 * after ocn_cn_walk_flags of the same batch (which handles the others and zeroes wc). */
int32_t ocn_walk_prep_max_batch(void);
int ocn_walk_prep(const int64_t* rowptrA, const int64_t* nds /* or NULL */, const int64_t* src, const int64_t* dst,
                  int64_t B, int32_t min_share, int64_t* order, int64_t* off, int64_t* chunk_off,
                  int64_t* rev_off /* NULL iff nds is */, int32_t* g_head /* [B+1] */, int64_t* g_item_off /* [B+1] */,
                  int32_t* g_active /* [B]: slot goes to the shared sweep */, int32_t* meta, int32_t* cnt1,
                  int32_t* cnt2, int32_t* status, int32_t* scal, void* stream);
int ocn_cn_walk_group(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                      const int64_t* order, int64_t B, const int32_t* g_head, const int64_t* g_item_off,
                      const int32_t* g_active, int32_t* meta, const int64_t* off, uint8_t* flags, int32_t* wc,
                      int64_t flags_cap, uint64_t* hist, int32_t* cnt1, int32_t* cnt2, void* stream);

/* Per-column weights, written IN PLACE over hist (uint64[N][2] -> float[N][4]) as
 *   {w1, t, inv2, 0}: a cn1 entry pools with w1; a union entry whose cn2 value is c (1.0 for the
 *   pattern route, the walk count for the valued route) pools into xcn2 with
 *   (c*[in cn2] - t*[in cn1]) * inv2.
 * cn5 (model.py:2261-2272, 2352-2413): w1 = 1/S1 (0 if S1 < 2); scale = max w1 over cn1
 * entries; nip = innerprod/scale (scale>0); t = nip*w1; S2 = column sums of cn2 - nip*ncn1 over
 * the union pattern (0 -> 1); inv2 = 1/S2.  `innerprod` is a device float[1] (the module buffer);
 * `scalars` is the statistics word of ocn_cn5_column_stats — run that entry first whenever innerprod may be
 * non-zero; for innerprod == 0 (a fresh model) nip is 0 whatever the scale, and zeroed scalars suffice;
 * `valued` != 0 for walk-count cn2.
 * cn7 (model.py:3114-3126, 3186-3209): w1 = 1/S1, `sum_fill` where S1 < 2; cn2 raw ->
 * {w1, 0, 1, 0}.  diag1 / diag2 (or NULL = ones): the Chebyshev diagonals diag(T_k(linspace(-1, 1, N))) of
 * evaluate_polynomial (model.py:2958-3019) that the reference multiplies the normalised cn1 (:3141-3165) and the raw cn2
 * (:3186-3209) by — one fp32 product per entry: {w1 * diag1[c], 0, diag2[c], 0}.  The reference hard-wires k = 0. */
int ocn_cn_weights_cn5(uint64_t* hist, int64_t N, const float* innerprod, int32_t* scalars,
                       int32_t valued, const float* s2_exact /* or NULL: closed form from the counts */, void* stream);
int ocn_cn_weights_cn7(uint64_t* hist, int64_t N, float sum_fill, const float* diag1 /* [N] or NULL */,
                       const float* diag2 /* [N] or NULL */, void* stream);

/* scalars[0] (zero before the first call of a batch; idempotent afterwards) = the batch's scale statistic of
 * model.py:2370-2375: 0 = empty union, -1 = no column with S1 >= 2, else min{S1 >= 2} - INT_MAX - 1. */
int ocn_cn5_column_stats(const uint64_t* hist, int64_t N, int32_t* scalars, void* stream);

/* Order-exact column sums for a non-zero `innerprod` (every trained checkpoint).  The reference forms
 * S2[c] = sum_e v[e,c], v = cn2 - nip*ncn1 on the union pattern, with index_add_ over the coalesced COO
 * (model.py:2405-2406): one fp32 add per entry, in ascending batch-row order; where colsum(cn2) ~ nip the
 * result depends on that order.  This entry transposes the union pattern of the batch into
 * per-column lists of flag positions (count -> scan -> fill -> per-column sort: positions ascend with the batch
 * row) and adds each column's values sequentially in fp32: s2[c] is bit for bit the reference's sum (before its
 * `== 0 -> 1` fix-up); pass it to ocn_cn_weights_cn5 / _cn6 as s2_exact.  Must run BEFORE the weights entry
 * (hist still holds the counts).  cn6 (flagsB = the cn3 flags of the (A, A^3) pass, s3 != NULL): additionally
 * s3[c] = column sums of cn3 - nip*ncn1 - nip*ncn2' over the three-way union (model.py:2895-2913).
 * wc: walk counts of the valued route or NULL.  flags_cap < 2^32.  scalars as for ocn_cn5_column_stats.
 * s2_init (or NULL = zeros; cn5 only): the running sums of the EARLIER rows of an edge-sharded batch — rank r of
 * ocn_amd/dist.py continues the chains rank r-1 left off (the global batch's rows ascend with the rank), so
 * the last rank ends with the single-device sums; hist must then hold the all-reduced (global) counts, the
 * entry counts of this shard are taken from the flags.
 * workspace: ocn_cn_colsum_workspace_bytes(N, flags_cap) bytes of device scratch. */
int64_t ocn_cn_colsum_workspace_bytes(int64_t N, int64_t flags_cap);
int ocn_cn_colsum_exact(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, int64_t B,
                        const int64_t* off, const uint8_t* flagsA, const uint8_t* flagsB /* or NULL */,
                        const int32_t* wc /* or NULL */, int64_t flags_cap, const uint64_t* hist /* [N][2], counts */,
                        int64_t N, const float* innerprod, int32_t* scalars, const float* s2_init /* [N] or NULL */,
                        float* s2 /* [N] */, float* s3 /* [N], iff flagsB */, void* workspace, void* stream);

/* The pooling: spmm_add(ncn1, x), spmm_add(ncn2, x) and x[i]*x[j] (model.py:2426-2429,
 * 3213-3216) in one pass.  xcn1[e] = sum_{k in cn1_e} w1[k] h[k]; xcn2[e] = sum over the
 * union pattern as above; xij[e] = h[i] (.) h[j]; entries in ascending column order, fp32
 * multiply then add.  wc = per-neighbour cn2 values of the walk route or NULL (all 1.0).
 * h is [N][H] row-major fp32; outputs [B][H].  max_row_len = longest row of A (upper bound):
 * batch rows whose source row exceeds 1024 entries (hubs) are pooled by a whole workgroup — or, for narrow embeddings
 * in a large batch, a wave — of their own: the other lanes fetch, ONE wave adds, so those rows keep the strictly
 * sequential ascending-column sum as well.  out_row (or NULL): batch row e is written to row
 * out_row[e] of the three outputs (ocn_class_order's inv_out: class-major rows for the heads).
 * cnt1 / cnt2 (or NULL): the per-row CN counts of the intersection pass; a row with neither kind of
 * entry is not walked at all, and with out_row the xcn1 / xcn2 rows the class-major heads never read
 * (no cn1 entry; no entry at all) are not written.
 * rowsum (or NULL; pattern route with the cn7 weights only — the caller's promise that every cn2 entry has weight
 * exactly 1 and no cn1 correction): [N][H] rows (A h)[i] = sum_{k in N(i)} h[k] in ascending column order
 * (ocn_spmm_csr, mode sum).  A candidate whose WHOLE source row is cn2 (cnt2[e] = row length: the target's A^2 row
 * holds every neighbour of the source — ogbl-ddi, whose A^2 is full) has xcn2[e] = rowsum[src[e]], the same additions in
 * the same order, so the row is copied instead of summed again for every candidate of that source; only its cn1 entries
 * are gathered. */
int ocn_cn_gather(const int64_t* rowptrA, const int32_t* colA,
                  const int64_t* src, const int64_t* dst, const int64_t* order, int64_t B,
                  const int64_t* off, const uint8_t* flags, const int32_t* wc,
                  const float* weights /* [N][4] */, const float* h, int32_t H, int64_t max_row_len,
                  float* xcn1, float* xcn2, float* xij, const int64_t* out_row,
                  const int32_t* cnt1, const int32_t* cnt2, const uint64_t* rec /* or NULL */,
                  const int32_t* perm /* ocn_gather_schedule's, or NULL */, const float* rowsum /* or NULL */,
                  void* stream);
/* The pooling's visiting order at H = 256 (a workgroup = four candidates = one group of ocn_cn_flags' gcost): candidates
 * differ 100x in cost and the few with hundreds of rows, met late, end the kernel as stragglers (0.206 -> 0.17 ms at the
 * collab shape).  perm[] = inside each XCD's contiguous eighth of the groups, the groups stable-sorted by descending cost:
 * groups of one source keep one cost and stay neighbours (L2).  n_groups = B / 4, a multiple of 8, at most 8 * 65535.
 * segment > 0 (dividing an eighth): the sort runs inside consecutive segments of that many groups instead of over the whole
 * eighth — the sources in flight on an XCD then stay within a narrow range of the source order (what their rows' L2
 * residency depends on) while every segment still starts with its longest jobs; 0 = whole eighths. */
int ocn_gather_schedule(const int32_t* gcost, int64_t n_groups, int64_t segment, int32_t* perm, void* stream);

/* The 3-hop predictor cn6 (model.py:2535-2951), pattern route.  Two intersection passes over the same
 * candidate batch — (A, A, A²) into flagsA / histA and (A, A³) into flagsB / histB (bit 0 = cn3 entry,
 * n1 field = cn3 column count) — then:
 *   ocn_cn_weights_cn6: histA -> {inv1, t, inv2, 0} exactly as cn5 (:2546-2714), histB -> {1/S3,0,0,0}
 *     with S3 the column sums of cn3 - nip*ncn1 - nip*ncn2' over the union pattern (:2846-2931; in
 *     eval all three inner products are the stored buffer), nip_out[0] = nip;
 *   ocn_cn_gather3: xcn1, xcn2, xcn3 = spmm_add of the three normalised matrices (:2712-2713, :2933),
 *     xij = x_i * x_j.  H in {16, 32, 64, 128, 256, 512}. */
int ocn_cn_weights_cn6(uint64_t* histA, uint64_t* histB, int64_t N, const float* innerprod, int32_t* scalars,
                       float* nip_out, const float* s2_exact /* or NULL */, const float* s3_exact /* or NULL */,
                       void* stream);
int ocn_cn_gather3(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                   const int64_t* order, int64_t B, const int64_t* off, const uint8_t* flagsA,
                   const uint8_t* flagsB, const float* weightsA, const float* weightsB, const float* nip,
                   const float* h, int32_t H, float* xcn1, float* xcn2, float* xcn3, float* xij, void* stream);

/* Backward of the pooling with respect to h (training drop-in): for upstream
 * gradients g1, g2, g3 of xcn1, xcn2, xij ([B][H] each),
 *   dh[k] += w1[k] g1[e] + w2(e,k) g2[e]  over the CN entries (the transposed spmm_add),
 *   dh[i] += g3[e] (.) h[j],  dh[j] += g3[e] (.) h[i].
 * dh [N][H] must be initialised by the caller (zeros); fp32 atomic adds (summation order varies
 * between runs).  H in {16..512, power of two}. */
/* Backward of ocn_cn_gather3 with respect to h (cn6 under autograd, model.py:2535-2951): dh[k] += w1 g1[e] + w2 g2[e] + w3 g3[e]
 * over the union entries — the weights formed as the forward forms them —, dh[i] += g4[e] * h[j], dh[j] += g4[e] * h[i];
 * fp32 atomics into dh (zeroed by the caller). */
int ocn_cn_gather3_backward(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                            const int64_t* order, int64_t B, const int64_t* off, const uint8_t* flagsA,
                            const uint8_t* flagsB, const float* weightsA, const float* weightsB, const float* nip,
                            const float* h, int32_t H, const float* g1, const float* g2, const float* g3,
                            const float* g4, float* dh, void* stream);

int ocn_cn_gather_backward(const int64_t* rowptrA, const int32_t* colA,
                           const int64_t* src, const int64_t* dst, const int64_t* order, int64_t B,
                           const int64_t* off, const uint8_t* flags, const int32_t* wc,
                           const float* weights, const float* h, int32_t H,
                           const float* g1, const float* g2, const float* g3, float* dh, void* stream);

/* The same gradient without float atomics (the default of the Python layer): the batch's terms are transposed into
 * per-node lists (count -> scan -> fill -> per-list sort by flag position; the two endpoint terms of a candidate follow)
 * and ONE wave per node adds its list in that order — the same bits on every run.  dh [N][H] is read and written (the
 * caller initialises it); H a multiple of 4, <= 512; flags_cap + 2 B < 2^31. */
int64_t ocn_cn_gather_backward_det_workspace_bytes(int64_t N, int64_t B, int64_t flags_cap);
int ocn_cn_gather_backward_det(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                               int64_t B, const int64_t* off, const uint8_t* flags, const int32_t* wc, int64_t flags_cap,
                               const float* weights, const float* h, int64_t N, int32_t H, const float* g1,
                               const float* g2, const float* g3, float* dh, void* workspace, void* stream);
/* The first half of ocn_cn_gather_backward_det alone — the per-node key lists (count, scan, fill, per-list sort) — for
 * inspection: tests check the lists against a host-side reference BEFORE the accumulate pass forms addresses from them.
 * workspace as above; afterwards col_off = int64[N + 1] at its start, keys = int32[col_off[N]] at
 * ocn_cn_gather_backward_det_keys_offset(N) bytes: a key < flags_cap is the flag position of a CN entry of that node, a key
 * flags_cap + 2 e + s is candidate e's Hadamard term at its source (s = 0) or target (s = 1). */
int64_t ocn_cn_gather_backward_det_keys_offset(int64_t N);
int ocn_cn_gather_backward_det_lists(const int64_t* rowptrA, const int32_t* colA, const int64_t* src, const int64_t* dst,
                                     int64_t B, const int64_t* off, const uint8_t* flags, int64_t flags_cap, int64_t N,
                                     void* workspace, void* stream);

/* CSR SpMM of the encoders: torch_sparse spmm_add/mean/max (model.py:42-55), PyG GCNConv
 * propagate (model.py:58-68), pygho/torch COO @ dense (model.py:105-113).
 *   y[r] = post[r] * reduce_k( pre[r]^a * pre[k] * x[k] )  (+ self term)
 * pre/post may be NULL (= 1).  val: per-entry values of a valued adjacency (DropAdj's 1/(1-p)
 * rescale in training, model.py:198-229) or NULL (all 1).  mode: 0 sum, 1 mean, 2 max.
 * edge_scale: 0 -> entry weight is pre[k] applied to x[k] first (PureConv: n*x then A.);
 *             1 -> entry weight is fl(pre[r]*pre[k]) (GCNConv / PureConv2: normalised A).
 * self_mode: 0 none; 1 add the row's own term after the neighbours (PureConv gcn);
 *            2 insert it at its sorted column position (GCNConv fill_diag). */
int ocn_spmm_csr(const int64_t* rowptr, const int32_t* col, const float* val, int64_t n_rows,
                 const float* x, int32_t F, const float* pre, const float* post,
                 int32_t mode, int32_t edge_scale, int32_t self_mode,
                 float* y, void* stream);

/* out[r] = 1/sqrt(add + deg(r)) (0 where the argument is 0): rsqrt_(1+adj.sum(-1)) of
 * model.py:51,106 (add=1) and gcn_norm's deg^-1/2 (add=1 after fill_diag); deg = row length, or the
 * row sum of `val` for a valued adjacency. */
int ocn_deg_rsqrt(const int64_t* rowptr, const float* val, int64_t n_rows, float add, float* out,
                  void* stream);

/* Pattern of A*A (NeighborOverlap_large.py:68-74,112-119: spadj @ spadj, values dropped).
 * Two phases around a caller-side allocation: count -> ocn_scan_i32 -> fill.  The count phase can
 * also leave every output row as a dense bit row (288 GB of HBM make N*N/8 bytes affordable up to a
 * few hundred thousand nodes), which ocn_cn_flags then probes instead of searching the CSR row.
 * Needs n_cols <= ocn_spgemm_max_cols().  Output columns ascending per row. */
int64_t ocn_spgemm_max_cols(void);
int ocn_spgemm_pattern_count(const int64_t* rowptrA, const int32_t* colA, int64_t n_rows,
                             const int64_t* rowptrB, const int32_t* colB, int64_t n_colsB,
                             int32_t* row_count, uint32_t* bitmap /* or NULL: [n_rows][bm_stride_words] */,
                             int64_t bm_stride_words, void* stream);
int ocn_spgemm_pattern_fill(const int64_t* rowptrA, const int32_t* colA, int64_t n_rows,
                            const int64_t* rowptrB, const int32_t* colB, int64_t n_colsB,
                            const int64_t* rowptrC, int32_t* colC, void* stream);
/* Bit rows of A*B for the rows a caller is about to probe (a training step's per-batch A², NeighborOverlap_large.py:68-74, is
 * read at the candidates' target rows only): request i names row rows[i] (int64, duplicates and out-of-range ids allowed — the
 * latter are ignored); `done` (int32 [n_rows], zero before the first call, kept by the caller between calls) marks the rows that
 * exist, a row is built by the request that turns its word from 0 to 1.  Rows never requested stay unwritten.  ocn_cn_flags
 * takes such a T2 with rowptrT2 == NULL (n_cols > 8192 only: small graphs read the row lengths too). */
/* Up to this many columns ocn_cn_flags keeps a workgroup's column histogram in LDS and reads T2's row lengths beside its bit rows
 * (so T2 must come with its row pointers there). */
int32_t ocn_cn_flags_small_graph_cols(void);
int ocn_spgemm_bit_rows(const int64_t* rowptrA, const int32_t* colA, int64_t n_rows, const int64_t* rowptrB, const int32_t* colB,
                        int64_t n_colsB, const int64_t* rows, int64_t n_req, int32_t* done, uint32_t* bitmap,
                        int64_t bm_stride_words, void* stream);

/* The block route of utils.block_matrix_multiply (utils.py:287-323; the drivers' --adj2byblock, ogbl-ddi): A as a dense
 * 0/1 int8 matrix (ocn_dense_from_csr: dense[r][c] and its transpose, row stride ld bytes, ld a multiple of 64 >= n, both
 * ZERO on entry), each (row block, column block) of the reference's tile loop multiplied on the integer matrix cores
 * (v_mfma_i32_32x32x32_i8; the entries are walk counts, exact in int32) and the non-zero pattern OR-ed into bit rows
 * (ocn_dense_block_mm_bits: rows [r0, r1) x columns [c0, c1), block starts multiples of 32, K = inner dimension).
 * fold != 0 writes the block at its block-LOCAL position — the reference adds SparseTensor.from_dense(block), whose
 * indices are block-local, without the block's offset (utils.py:318-321): every block then lands on the
 * top-left corner; fold == 0 places it where it belongs (the intended A^2).  ocn_bitrows_count / _fill turn bit rows
 * into CSR (columns ascending). */
int ocn_dense_from_csr(const int64_t* rowptr, const int32_t* col, int64_t n, int64_t ld, int8_t* dense, int8_t* denseT,
                       void* stream);
int ocn_dense_block_mm_bits(const int8_t* A, const int8_t* Bt, int64_t ld, int64_t K, int32_t r0, int32_t r1, int32_t c0,
                            int32_t c1, int32_t fold, uint32_t* bits, int64_t bm_stride_words, void* stream);
/* Dense bit rows of a CSR pattern ([n_rows][bm_stride_words] words, ZERO on entry): what ocn_cn_flags probes as
 * bitmapT1 on small dense graphs (ogbl-ddi: 4267 nodes, 500+ neighbours each), one load per membership test instead of a
 * binary search of the target row. */
int ocn_bitrows_from_csr(const int64_t* rowptr, const int32_t* col, int64_t n_rows, uint32_t* bits, int64_t bm_stride_words,
                         void* stream);
int ocn_bitrows_count(const uint32_t* bits, int64_t bm_stride_words, int64_t n_rows, int64_t n_cols, int32_t* row_count,
                      void* stream);
int ocn_bitrows_fill(const uint32_t* bits, int64_t bm_stride_words, int64_t n_rows, int64_t n_cols, const int64_t* rowptr,
                     int32_t* col, void* stream);

/* Glue for head layouts the fused Linear kernel below does not cover (widths outside 32..256, training
 * mode; model.py:2203-2235, 2429-2437): y = LayerNorm(x) (eps, affine gamma/beta) followed by ReLU when `relu` != 0,
 * one pass over [rows][H] (replaces nn.LayerNorm + Dropout(eval) + nn.ReLU); and the branch mix
 * out = c[0]*x1 + c[1]*x2 + c[2]*x3 with the three coefficients read from device memory
 * (alpha.sigmoid().cumprod() and beta, model.py:2435-2436).  H in {16..512, power of two}. */
int ocn_rows_ln_relu(const float* x, const float* gamma, const float* beta, float eps, int32_t relu,
                     int64_t rows, int32_t H, float* y, void* stream);
int ocn_combine3(const float* coef, const float* x1, const float* x2, const float* x3, int64_t n,
                 float* out, void* stream);

/* The nn.Linear layers of the heads (model.py:2192-2235) on the matrix cores:
 *   Y[M][N] = epilogue(X[M][K] . W^T + bias),  X/W/Y fp32, W = nn.Linear.weight [N][K].
 * Each fp32 operand is split into three bf16 terms and the six leading cross products are
 * accumulated in fp32 on v_mfma_f32_32x32x16_bf16 (error below fp32's own rounding).
 * ocn_linear_split_weight writes the pre-split, fragment-ordered panel Wp
 * (ocn_linear_panel_bytes(N, K) bytes; redo it whenever W changes).  N in {32,64,128,256}, K % 16 == 0.
 * Epilogue, in this order: + bias (or NULL); LayerNorm over the N columns with gamma/beta/eps (or
 * both NULL); ReLU if relu != 0; then either store Y[M][N], or — when dotw != NULL — the trailing
 * Linear(N -> 1) of `lin`: Y[M] = <row, dotw> + dotb[0]. */
/* dst[row][0 .. n_cols) = vec[0 .. n_cols) for the rows of a device-side range (clamped to max_rows): the
 * constant activations of the candidates whose pooled input is all zero (ocn_class_order). */
int ocn_fill_rows(float* dst, int64_t ld, int32_t n_cols, const float* vec, const int64_t* row_range,
                  int64_t max_rows, void* stream);

typedef struct OcnLinearGroup {
  const float* X; int64_t ldX;       /* input [M][K], row stride in floats (0 = K) */
  int64_t M;
  const void* Wp;                    /* panel of this group's weight */
  const float* bias;                 /* or NULL */
  const float* gamma; const float* beta; float eps;   /* LayerNorm, or both NULL */
  int32_t relu;
  const float* scale;                /* device float[1] multiplied in after ReLU, or NULL */
  const float* addend; int64_t ldAdd;/* [M][N] added last (row stride, 0 = N), or NULL */
  const float* dotw; const float* dotb;   /* trailing Linear(N -> 1): Y is [M]; or NULL */
  float* Y; int64_t ldY;             /* output rows, row stride in floats (0 = N) */
  /* Rows [row_range[0], row_range[1]) of the M-row buffers only (device int64[2], clamped to [0, M]), or
   * NULL for all M rows: the bounds stay on the device, e.g. the class boundaries of a batch sorted by
   * "has common neighbours" (ocn_class_order) — no host sync to launch on a sub-range. */
  const int64_t* row_range;
  const int64_t* y_row_map;          /* dot epilogue only: Y[y_row_map[row]] instead of Y[row], or NULL */
  int32_t add_bcast;                 /* != 0: addend is ONE row [N] added to every row */
} OcnLinearGroup;

int64_t ocn_linear_panel_bytes(int32_t N, int32_t K);
int ocn_linear_split_weight(const float* W, int32_t N, int32_t K, void* Wp, void* stream);
int ocn_linear_bf16x6(const float* X, int64_t M, int32_t K, const void* Wp, int32_t N,
                      const float* bias, const float* gamma, const float* beta, float eps,
                      int32_t relu, const float* dotw, const float* dotb, float* Y, void* stream);
/* Up to 5 independent Linear layers of the same (K, N) in ONE launch (e.g. the first layers of
 * xcn1lin, xcn2lin and xijlin): more workgroups than CU slots, so one group's LayerNorm epilogue
 * overlaps another's MFMA loop, and strided outputs let two branches write the halves of a
 * [M][2N] buffer that a K = 2N Linear then consumes (the branch mix of model.py:2436 folded into
 * its weights).  `groups` is a HOST array. */
int ocn_linear_grouped(const OcnLinearGroup* groups, int32_t n_groups, int32_t K, int32_t N, void* stream);

/* The MLP heads of cn5 / cn7 (model.py:2203-2235, 2429-2437 == 3216-3223) as one launch per candidate batch:
 *   a = ReLU(LN(W3a ReLU(W0a xcn1 + b0a) + b3a)),  b likewise from xcn2,  c = ReLU(LN(W0x (x_i*x_j) + b0x)),
 *   y = Linear(H,1)(ReLU(LN(Ma a + Mb b + Mc c + bf)))
 * where the host has folded the reference's products without a non-linearity between them — the third layers
 * of xcn1lin / xcn2lin, the second layer of xijlin, the mix alpha0*xcn1 + alpha1*xcn2 + beta*xij (model.py:2436)
 * and lin[0] — into Ma, Mb, Mc, bf.  Every activation stays in registers; only x[3] is read and y written.
 * An f32 product is three f16 MFMAs on hi/lo splits of both operands (v_mfma_f32_32x32x16_f16, fp32 accumulate):
 * ocn_heads_split_weight writes the panel of scale * W (scale = a power of two that puts max |W| into
 * [2^13, 2^14); ocn_heads_panel_bytes(N, K) bytes) in the k order in which the previous layer's accumulator
 * registers arrive; activation rows are scaled per row inside the kernel.
 * p_first: panels of xcn1lin.0, xcn2lin.0, xijlin.0; p_mid: of xcn1lin.3, xcn2lin.3; p_out: of Ma, Mb, Mc.
 * vec: ocn_heads_nvec() vectors of H floats — b0a b3a g3a e3a  b0b b3b g3b e3b  b0x gx ex  bf gl el  dotw  constA constB
 * — followed by ocn_heads_nscal() scalars: the dot bias, then 1 / scale of the eight panels in the order
 * xcn1lin.0 xcn1lin.3 Ma xcn2lin.0 xcn2lin.3 Mb xijlin.0 Mc, then zeros (constA / constB are unused since ABI 6).
 * What a skipped branch contributes — Ma a, Mb b of an all-zero pooled row — comes from this entry in constants mode
 * (dump != NULL, B == 1, x = one zero row: fills ocn_heads_const_bytes(H) bytes, no score) and is handed back as
 * `cpark` in scoring mode.  ranges (or NULL) = ocn_class_order's table: the candidates then come class-major and a
 * workgroup without any cn1 (cn2) row adds the constant instead of running the branch; b_on_union: cn5 (xcn2 lives on
 * cn1 u cn2) vs cn7 (cn2 only).  H in {128, 256}; in_channels == H (narrower heads: ocn_linear_grouped). */
typedef struct OcnHeadsArgs {
  const float* x[3];
  int64_t ldx, B;
  int32_t H;
  const void* p_first[3];
  const void* p_mid[2];
  const void* p_out[3];
  const float* vec;
  const int64_t* ranges;
  const int64_t* y_row_map;
  float* y;
  float* dump;               /* constants mode: ocn_heads_const_bytes(H) bytes out, B == 1 */
  const float* cpark;        /* scoring mode: the buffer a constants-mode call filled */
  float* scratch;            /* ocn_heads_scratch_bytes(H) bytes: where a wave parks a finished branch's share of the output */
  float eps;
  int32_t ln, b_on_union;
} OcnHeadsArgs;
int32_t ocn_heads_nvec(void);
int32_t ocn_heads_nscal(void);
int64_t ocn_heads_scratch_bytes(int32_t H);
int64_t ocn_heads_const_bytes(int32_t H);
int64_t ocn_heads_panel_bytes(int32_t N, int32_t K);
int ocn_heads_split_weight(const float* W, int32_t N, int32_t K, float scale, void* Wp, void* stream);
int ocn_heads_fused(const OcnHeadsArgs* args, void* stream);
/* Batches of up to `max_rows` candidates are scored by the small-batch form of the same head — 32 candidates per workgroup,
 * the four waves splitting every layer's output features — which returns the same bits as the throughput form at a fifth of
 * its latency (Cora's 1 152-candidate batch: nine 128-row tiles = 70 us whatever the batch size).  Sets the bound (process-wide;
 * default 16384 = two rounds of workgroups on 256 CUs; quoted at H = 256 — at H = 128 twice as many rows take the small form) and
 * returns the previous one; a negative argument only queries. */
int64_t ocn_heads_small_batch(int64_t max_rows);

/* Training-side pieces (NeighborOverlap_large.py:56-63, 76-90).
 *
 * ocn_coo_to_csr: the per-batch masked adjacency — SparseTensor.from_edge_index(tei, sparse_sizes) (a sort by (row, col)
 * in torch_sparse; dedupe = 0: duplicates are kept, as there) and .to_symmetric() (symmetrize = 1, dedupe = 1: the
 * transposed entries join and the pattern is coalesced), pattern only.  Count -> scan -> fill -> per-row sort (wave rank
 * sort / workgroup bitonic network) -> unique count -> scan -> compact, all on `stream`.  rowptr [n_rows + 1]; col_out must
 * hold nnz * (symmetrize ? 2 : 1) entries (the upper bound); result[0] = entries written (= rowptr[n_rows]), result[1] = 1
 * if an index was out of range (such entries are dropped) — device memory, the caller reads it when it needs to. */
int64_t ocn_coo_to_csr_workspace_bytes(int64_t nnz, int64_t n_rows, int32_t symmetrize, int32_t dedupe);
int ocn_coo_to_csr(const int64_t* row, const int64_t* col, int64_t nnz, int64_t n_rows, int64_t n_cols,
                   int32_t symmetrize, int32_t dedupe, int64_t* rowptr, int32_t* col_out, void* workspace,
                   int64_t* result, void* stream);

/* ocn_wgrad: weight / bias gradient of a Linear layer, dW[N][K] = dY^T X, db[N] = column sums of dY (db may be NULL),
 * dY [B][N] row stride ldY, X [B][K] row stride ldX, fp32.  bf16x6 on the matrix cores, split over the batch; the
 * slices' partial results are added in slice order (no float atomics: the same bits on every run).  Any N, K >= 1. */
int64_t ocn_wgrad_workspace_bytes(int64_t B, int32_t N, int32_t K);
int ocn_wgrad(const float* dY, int64_t ldY, const float* X, int64_t ldX, int64_t B, int32_t N, int32_t K,
              float* dW, float* db, void* workspace, void* stream);

/* The heads' LayerNorm -> Dropout -> ReLU tails under autograd (the nn.Sequential layouts of model.py:2203-2235 in train(),
 * NeighborOverlap_large.py:76-90), one forward and two backward launches:
 *   y = relu?( dropout_p( gamma == NULL ? x : LN(x; gamma, beta, eps) ) ),   x, y [rows][H] fp32, H in {16 .. 512}
 * stats (with gamma): float[rows][2] = {mean, 1 / sqrt(var + eps)} for the backward.  Dropout keeps an element iff a
 * counter-based hash of (seed, element index) passes p — no mask is stored; the backward recomputes it from the same seed
 * (ocn_dropout_keep_mask exposes the decisions for tests).  This is the library's random stream, not torch's.
 * backward: dx; with gamma also dgamma / dbeta — every one of a FIXED number of lane groups walks a contiguous block of rows
 * and keeps its own partial sums, a second launch adds them in block order: the same bits on every run.
 * workspace: ocn_ln_drop_relu_workspace_bytes(H) (with gamma). */
int64_t ocn_ln_drop_relu_workspace_bytes(int32_t H);
int ocn_ln_drop_relu_forward(const float* x, const float* gamma /* or NULL */, const float* beta /* or NULL */, float eps, float p,
                             uint64_t seed, int32_t relu, int64_t rows, int32_t H, float* y, float* stats /* [rows][2], with gamma */,
                             void* stream);
int ocn_ln_drop_relu_backward(const float* g, const float* x, const float* y, const float* stats, const float* gamma /* or NULL */,
                              float p, uint64_t seed, int32_t relu, int64_t rows, int32_t H, float* dx,
                              float* dgamma, float* dbeta, void* workspace, void* stream);
int ocn_dropout_keep_mask(uint64_t seed, float p, int64_t n, uint8_t* out, void* stream);

/* Backward of the branch mix z = coef[0] x1 + coef[1] x2 + coef[2] x3 (ocn_combine3; model.py:2436, 3222): d_k = coef[k] g and
 * dcoef[k] = <g, x_k> — per-workgroup partial sums over contiguous chunks, added in workgroup order (deterministic).  n a
 * multiple of 4; workspace: ocn_mix3_workspace_bytes(). */
int64_t ocn_mix3_workspace_bytes(void);
int ocn_mix3_backward(const float* coef, const float* g, const float* x1, const float* x2, const float* x3, int64_t n,
                      float* d1, float* d2, float* d3, float* dcoef, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OCN_HIP_H */
