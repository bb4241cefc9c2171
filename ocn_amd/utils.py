"""Drop-in for the hot-path names of the reference's ``utils.py``.

``adjoverlap`` (utils.py:248-285), ``PermIterator`` (utils.py:8-36), ``sparse_tensor_multiply`` /
``block_matrix_multiply`` (utils.py:287-329) keep their signatures; ``get_cn1_cn2`` is the pygho
route of the ppa / citation2 drivers (NeighborOverlap_large_ppa.py:147-173).  ``adjoverlap`` and
``get_cn1_cn2`` return lazy :class:`CNBatch` handles instead of materialised [B, N] SparseTensors;
the predictors in ``ocn_amd.model`` recognise them and run the fused HIP path, ``.materialize()``
yields the explicit matrix for inspection and parity tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .sparse import SparseTensor


class PermIterator:
    """Batch index iterator (utils.py:8-36): ``randperm`` + drop-last when training, ``arange`` +
    ragged tail kept in eval."""

    def __init__(self, device, size, bs, training=True) -> None:
        self.bs = bs
        self.training = training
        self.idx = torch.randperm(size, device=device) if training else torch.arange(size, device=device)

    def __len__(self):
        return (self.idx.shape[0] + (self.bs - 1) * (not self.training)) // self.bs

    def __iter__(self):
        self.ptr = 0
        return self

    def __next__(self):
        if self.ptr + self.bs * self.training > self.idx.shape[0]:
            raise StopIteration
        ret = self.idx[self.ptr:self.ptr + self.bs]
        self.ptr += self.bs
        return ret


class CNState:
    """Device state of one candidate batch after the intersection kernel: where each batch row
    starts (``off``), one flag byte per neighbour of the source node (bit 0: cn1 entry, bit 1: cn2
    entry), the walk counts of the valued route, the packed per-column histograms
    {n1, n2, n_union, walks}, and the integer CN counts."""

    def __init__(self, adj: SparseTensor, t1: Optional[SparseTensor], t2: Optional[SparseTensor],
                 tarei: Tensor, walk: bool = False, ws: Optional[dict] = None):
        if tarei.dim() != 2 or tarei.shape[0] != 2:
            raise ValueError("tarei must be [2, B]")
        for t in (t1, t2):
            if t is not None and t.sparse_sizes() != adj.sparse_sizes():
                raise ValueError("adjoverlap: adjacency sizes differ")    # utils.py:164 assert
        if walk and adj.size(0) != adj.size(1):
            raise ValueError("get_cn1_cn2 needs a square adjacency")
        self.adj, self.walk, self.t2 = adj, walk, t2
        self.ws = ws                           # a predictor's scratch cache: this state is then transient
        self.src = tarei[0].to(torch.int64).contiguous()
        self.dst = tarei[1].to(torch.int64).contiguous()
        self.B = self.src.numel()
        self.N = adj.size(1)
        ops.check_edges(self.src, self.dst, adj.size(0), adj.size(0) if walk else t1.size(0))
        # per-slot records the intersection pass leaves for the pooling (pattern route)
        self.rec = None if (walk or self.B == 0) else ops.buf(ws, "rec", (self.B, 4), torch.int64, self.src.device)
        # ... and what each group of four slots will cost the pooling, for its longest-first schedule (+ room for the schedule)
        self.sched = (ops.buf(ws, "sched", 2 * ((self.B + 3) // 4), torch.int32, self.src.device)
                      if (self.rec is not None and ops.heavy_first and self.B >= ops.sort_edges_min_batch) else None)
        # a product whose rows are built on demand (a training step's A²): the rows this batch probes, and no row pointers — on
        # one stream only (a loop with several phase-A streams completes the product before it forks them: pipeline.score_edges)
        t2_rows = (t2 is not None and not walk and t2.rows_on_demand() and not getattr(ops, "_overlap_active", False))
        (self.order, self.off, self.flags, self.wc, self.hist, self.cnt1, self.cnt2, self.status, self.scal) = ops.cn_flags(
            adj._rowptr, adj._col, None if walk else (t1._rowptr, t1._col),
            None if (walk or t2 is None) else ((None, None) if t2_rows else
                                               (t2._rowptr, t2._col if (t2.col_materialized() or t2._bitmap is None) else None)),
            self.src, self.dst, self.N,
            adj.max_rowcount(), walk=walk,
            t2_bitmap=None if (walk or t2 is None) else (t2.product_bit_rows(self.dst) if t2_rows else t2.product_bit_rows()), wsd=ws,
            nds=adj.neighbor_degree_sum() if (walk and ops.walk_two_sided) else None,
            t1_bitmap=None if walk else t1.bit_rows(), rec=self.rec, sched=self.sched)
        self._hist_live = True

    @classmethod
    def from_materialized(cls, adj: SparseTensor, cn1: SparseTensor, cn2: SparseTensor, tarei: Tensor,
                          ws: Optional[dict] = None) -> "CNState":
        """Batch state from two explicit [B, N] matrices, as the reference's own ``adjoverlap`` / ``get_cn1_cn2`` return
        them (every row e a subset of N_adj(tarei[0][e]); cn1 values 1.0; cn2 values 1.0 or walk counts).  A format
        conversion in torch ops (searchsorted into the source rows), not a hot path: the handles of
        ``ocn_amd.utils.adjoverlap`` skip it."""
        st = cls.__new__(cls)
        st.adj, st.ws = adj, ws
        st.src = tarei[0].to(torch.int64).contiguous()
        st.dst = tarei[1].to(torch.int64).contiguous()
        st.B, st.N = st.src.numel(), adj.size(1)
        dev = st.src.device
        if tuple(cn1.sizes()) != (st.B, st.N) or tuple(cn2.sizes()) != (st.B, st.N):
            raise ValueError("cn1 / cn2 must be [B, N] with B = tar_ei.shape[1], N = adj.size(1)")
        ops.check_edges(st.src, st.dst, adj.size(0), adj.size(0))
        st.order = None
        st.off = ops.edge_offsets(adj._rowptr, st.src)
        total = int(st.off[-1])
        st.flags = torch.zeros(max(total, 1), dtype=torch.uint8, device=dev)
        rows_sel = adj[st.src]                                    # row e = N(src[e]), columns ascending
        r_s, c_s, _ = rows_sel.coo()
        key_s = r_s * st.N + c_s                                  # ascending: position p of the flag buffer
        v2 = cn2.storage.value()
        st.walk = bool(v2 is not None and v2.numel() and bool((v2 != 1).any()))
        st.wc = torch.zeros(max(total, 1), dtype=torch.int32, device=dev) if st.walk else None
        for bit, m in ((1, cn1), (2, cn2)):
            r, c, v = m.coo()
            key = r * st.N + c
            pos = torch.searchsorted(key_s, key).clamp_(max=max(key_s.numel() - 1, 0))
            if key.numel() and not bool((key_s[pos] == key).all()):
                raise NotImplementedError("cn1 / cn2 entries outside the source rows of `adj` cannot be expressed as CN flags")
            if bit == 2 and v is not None:                        # explicit zeros (pygho Hadamard) are no entries
                keep = v != 0
                pos, v = pos[keep], v[keep]
            st.flags[pos] |= bit
            if bit == 2 and st.walk:
                st.wc[pos] = v.to(torch.int32)
        f = st.flags[:total].to(torch.int64)
        e_of = r_s
        st.cnt1 = torch.zeros(st.B, dtype=torch.int64, device=dev).index_add_(0, e_of, f & 1).to(torch.int32)
        st.cnt2 = torch.zeros(st.B, dtype=torch.int64, device=dev).index_add_(0, e_of, (f >> 1) & 1).to(torch.int32)
        z = torch.zeros(st.N, dtype=torch.int64, device=dev)
        n1 = z.clone().index_add_(0, c_s, f & 1)
        n2 = z.clone().index_add_(0, c_s, (f >> 1) & 1)
        nu = z.clone().index_add_(0, c_s, (f != 0).to(torch.int64))
        walks = z.clone().index_add_(0, c_s, st.wc[:total].to(torch.int64)) if st.walk else z
        st.hist = torch.stack([n1 | (n2 << ops.HIST_FIELD_BITS) | (nu << (2 * ops.HIST_FIELD_BITS)), walks], dim=1).contiguous()
        st.status = torch.zeros(4, dtype=torch.int32, device=dev)
        st.scal = torch.zeros(4, dtype=torch.int32, device=dev)
        st._hist_live = True
        return st

    def check_status(self) -> None:
        """Raise for the error bits the intersection pass left for this batch (one host sync; ocn_hip.h: OCN_ST_*)."""
        bits = int(self.status[0].item())
        if bits:
            self.status[3:].zero_()
            raise RuntimeError(ops.status_message(bits))

    def hist_counts(self) -> Tensor:
        """int64 [N, 4] = {n1, n2, n_union, walk-count sum} per column."""
        assert self._hist_live, "histogram already consumed"
        return ops.hist_counts(self.hist)

    def weights_cn5(self, innerprod: Tensor) -> Tensor:
        assert self._hist_live, "histogram already consumed"
        self._hist_live = False
        s2, scal = None, self.scal                   # zeroed with the batch's other scratch (ops.cn_flags)
        if ops.colsum_wanted(innerprod):
            # innerprod != 0 (a trained checkpoint): S2 summed entry by entry in the reference's order
            run = lambda init: ops.cn_colsum_exact(self.adj._rowptr, self.adj._col, self.src, self.off, self.flags, None,
                                                   self.wc, self.hist, innerprod, scal, self.ws, s2_init=init)[0]
            if getattr(self, "sharded", False):
                from .dist import ring_colsum          # an edge shard continues the chains of the ranks before it
                s2 = ring_colsum(run, self.hist.shape[0], self.hist.device, self.shard_group)
            else:
                s2 = run(None)
        return ops.cn_weights_cn5(self.hist, innerprod, valued=self.walk, wsd=self.ws, s2_exact=s2, scal=scal)

    def weights_cn7(self, sum_fill: float, diag1: Optional[Tensor] = None, diag2: Optional[Tensor] = None) -> Tensor:
        assert self._hist_live, "histogram already consumed"
        self._hist_live = False
        return ops.cn_weights_cn7(self.hist, sum_fill, diag1, diag2)

    def gather(self, weights: Tensor, h: Tensor, order: Optional[Tensor] = None, out_row: Optional[Tensor] = None,
               rowsum: Optional[Tensor] = None):
        """(xcn1, xcn2, x_i * x_j); ``order`` overrides the processing order, ``out_row`` sends batch row e
        to output row out_row[e] (class-major rows for the heads, ops.class_order); ``rowsum`` = A·h for the cn7 weights
        (ocn_hip.h, ocn_cn_gather: candidates whose whole source row is cn2 copy their xcn2 row)."""
        return ops.cn_gather(self.adj._rowptr, self.adj._col, self.src, self.dst, self.off,
                             self.flags, self.wc, weights, h, order=self.order if order is None else order,
                             max_row_len=self.adj.max_rowcount(), wsd=self.ws, out_row=out_row,
                             cnt1=self.cnt1, cnt2=self.cnt2, rec=getattr(self, "rec", None) if order is None else None,
                             sched=getattr(self, "sched", None) if order is None else None,
                             rowsum=None if self.walk else rowsum, sched_ready=getattr(self, "_sched_ready", False))

    def prepare_schedule(self, H: int) -> None:
        """Phase A of a scoring loop: the pooling's visiting order needs the intersection pass only."""
        if H == 256 and getattr(self, "sched", None) is not None and getattr(self, "rec", None) is not None:
            self._sched_ready = ops.gather_schedule(self.sched, self.B)

    def gather_backward(self, weights: Tensor, h: Tensor, g1: Tensor, g2: Tensor, g3: Tensor) -> Tensor:
        return ops.cn_gather_backward(self.adj._rowptr, self.adj._col, self.src, self.dst, self.off, self.flags,
                                      self.wc, weights, h, g1.contiguous(), g2.contiguous(), g3.contiguous(),
                                      order=self.order)

    def cn5_batch_innerprod(self) -> Tensor:
        """Σ (cn2 ⊙ ncn1) of this batch (model.py:2241-2244), from the integer column counts:
        every entry in both sets contributes 1/S1 of its column.  Pattern route only."""
        hc = self.hist_counts()
        if self.walk:
            # valued cn2: Σ over entries in both sets of walk count / S1[col] (host-side format code,
            # training only — the eval path never needs it)
            n1 = hc[:, 0]
            inv1 = torch.where(n1 >= 2, 1.0 / n1.clamp(min=1).to(torch.float32), torch.zeros((), device=n1.device))
            both = self.materialize(1)
            r, c, _ = both.coo()
            cn2 = self.materialize(2)
            key2 = cn2.coo()[0] * self.N + cn2.coo()[1]
            key1 = r * self.N + c
            idx = torch.searchsorted(key2, key1).clamp(max=max(key2.numel() - 1, 0))
            hit = (key2[idx] == key1) if key2.numel() else torch.zeros_like(key1, dtype=torch.bool)
            return (cn2.storage.value()[idx[hit]] * inv1[c[hit]]).sum()
        n1 = hc[:, 0]
        nb = n1 + hc[:, 1] - hc[:, 2]
        inv1 = torch.where(n1 >= 2, 1.0 / n1.clamp(min=1).to(torch.float32), torch.zeros((), device=n1.device))
        return (nb.to(torch.float32) * inv1).sum()

    def materialize(self, bit: int) -> SparseTensor:
        """[B, N] matrix of the entries whose flag has ``bit`` set; values 1.0, or the walk counts
        for bit 2 of the valued route (host-side format conversion for tests; not on the product
        path)."""
        row_sel = self.adj[self.src]
        r, c, _ = row_sel.coo()
        pos = torch.arange(r.numel(), device=r.device) - row_sel._rowptr[:-1][r] + self.off[:-1][r]
        keep = (self.flags[pos] & bit) != 0
        val = torch.ones(int(keep.sum()), device=r.device)
        if self.walk and bit == 2:
            val = self.wc[pos][keep].to(torch.float32)
        return SparseTensor(row=r[keep], col=c[keep], value=val, sparse_sizes=(self.B, self.N),
                            is_sorted=True, trust_data=True)


class CNBatch:
    """Lazy ``adjoverlap(adj1, adj2, tarei)``: rows N_adj1(tarei[0][e]) ∩ N_adj2(tarei[1][e]); or one
    half of ``get_cn1_cn2(adj, tedge)`` (``mode`` "walk1" / "walk2")."""

    def __init__(self, adj1: SparseTensor, adj2: Optional[SparseTensor], tarei: Tensor, mode: str = "pattern"):
        self.adj1, self.adj2, self.tarei, self.mode = adj1, adj2, tarei, mode
        self._state: Optional[CNState] = None

    def sizes(self):
        return [self.tarei.shape[1], self.adj1.size(1)]

    def size(self, dim: int) -> int:
        return self.sizes()[dim]

    def device(self):
        return self.adj1.device()

    def _single(self) -> CNState:
        if self._state is None:
            if self.mode == "pattern":
                self._state = CNState(self.adj1, self.adj2, None, self.tarei)
            else:
                self._state = CNState(self.adj1, None, None, self.tarei, walk=True)
        return self._state

    def counts(self) -> Tensor:
        """Integer CN count per candidate edge (per-row count of non-zero entries)."""
        st = self._single()
        return st.cnt2 if self.mode == "walk2" else st.cnt1

    def materialize(self) -> SparseTensor:
        return self._single().materialize(2 if self.mode == "walk2" else 1)


class CNState3:
    """The 3-hop predictor's batch state: two intersection passes over the same candidates — (A, A, A²) and
    (A, A³) — sharing the row offsets (both walk the source rows of A)."""

    def __init__(self, adj: SparseTensor, adj2: SparseTensor, adj3: SparseTensor, tarei: Tensor):
        self.a = CNState(adj, adj, adj2, tarei)
        self.b = CNState(adj, adj3, None, tarei)
        self.adj, self.B, self.N = adj, self.a.B, self.a.N

    @property
    def cnt3(self) -> Tensor:
        return self.b.cnt1

    def weights(self, innerprod: Tensor, sharded: bool = False):
        """``sharded``: the histograms were summed over edge shards — the closed form over the global counts is
        used (the order-exact sums need every rank's entries in order; cn5 chains them through ocn_amd.dist.ring_colsum,
        the two-stage cn6 does not)."""
        assert self.a._hist_live and self.b._hist_live, "histograms already consumed"
        self.a._hist_live = self.b._hist_live = False
        exact = None if sharded else (self.adj._rowptr, self.adj._col, self.a.src, self.a.off, self.a.flags, self.b.flags)
        return ops.cn_weights_cn6(self.a.hist, self.b.hist, innerprod, exact=exact, scal=self.a.scal)

    def gather(self, wa: Tensor, wb: Tensor, nip: Tensor, h: Tensor):
        return ops.cn_gather3(self.adj._rowptr, self.adj._col, self.a.src, self.a.dst, self.a.off, self.a.flags,
                              self.b.flags, wa, wb, nip, h, order=self.a.order)

    def gather_backward(self, wa: Tensor, wb: Tensor, nip: Tensor, h: Tensor, g1, g2, g3, g4) -> Tensor:
        return ops.cn_gather3_backward(self.adj._rowptr, self.adj._col, self.a.src, self.a.dst, self.a.off, self.a.flags,
                                       self.b.flags, wa, wb, nip, h, g1.contiguous(), g2.contiguous(), g3.contiguous(),
                                       g4.contiguous(), order=self.a.order)


def _same_edges(a: Tensor, b: Tensor) -> bool:
    """The drivers hand the SAME tensor to both adjoverlap calls and to the predictor
    (NeighborOverlap_large.py:121-159): identity (or the same view of one storage) needs no device
    comparison, anything else costs a torch.equal (one host sync)."""
    if a is b or (a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() == b.stride()
                  and a.dtype == b.dtype and a.device == b.device):
        return True
    return torch.equal(a, b)


def fuse3(cn1: "CNBatch", cn2: "CNBatch", cn3: "CNBatch", tar_ei: Tensor) -> CNState3:
    """(cn1, cn2, cn3) = adjoverlap(adj, adj | adj2 | adj3, e) of one candidate batch."""
    for c in (cn1, cn2, cn3):
        if not isinstance(c, CNBatch) or c.mode != "pattern":
            raise NotImplementedError("cn6 takes the handles returned by ocn_amd.utils.adjoverlap")
    if not (cn1.adj1 is cn2.adj1 is cn3.adj1) or cn1.adj2 is not cn1.adj1:
        raise NotImplementedError("cn1/cn2/cn3 must come from adjoverlap(adj, adj|adj2|adj3, e) of one adjacency")
    if ops.validate_indices and not (_same_edges(cn1.tarei, cn2.tarei) and _same_edges(cn1.tarei, cn3.tarei)
                                     and _same_edges(cn1.tarei, tar_ei)):
        raise NotImplementedError("cn1, cn2, cn3 and tar_ei must be built from the same candidate edges")
    return CNState3(cn1.adj1, cn2.adj2, cn3.adj2, cn1.tarei)


def fuse(cn1, cn2, tar_ei: Tensor, ws: Optional[dict] = None, adj: Optional[SparseTensor] = None) -> CNState:
    """One intersection pass for the (cn1, cn2) pair every driver builds from the same candidate
    edges (NeighborOverlap_large.py:76-82,121-159; NeighborOverlap_large_ppa.py:98-133).  Explicit [B, N]
    SparseTensors (what the reference's own adjoverlap returns) are accepted too, given the adjacency the
    predictor was called with: they are converted to the flag form (CNState.from_materialized)."""
    if isinstance(cn1, CNBatch) != isinstance(cn2, CNBatch):
        cn1 = cn1.materialize() if isinstance(cn1, CNBatch) else cn1
        cn2 = cn2.materialize() if isinstance(cn2, CNBatch) else cn2
    if isinstance(cn1, SparseTensor) and isinstance(cn2, SparseTensor):
        if not isinstance(adj, SparseTensor):
            raise TypeError("explicit cn1 / cn2 matrices need the adjacency (the predictor's `adj` argument) to be rebuilt "
                            "as CN flags")
        return CNState.from_materialized(adj, cn1, cn2, tar_ei, ws=None)
    if not isinstance(cn1, CNBatch) or not isinstance(cn2, CNBatch):
        raise TypeError("ocn_amd predictors take the CNBatch handles returned by ocn_amd.utils.adjoverlap "
                        "/ get_cn1_cn2, or explicit [B, N] SparseTensors")
    if cn1.adj1 is not cn2.adj1:
        raise NotImplementedError("cn1 and cn2 must select their source rows from the same adjacency")
    if cn1.tarei.shape != cn2.tarei.shape or cn1.tarei.shape != tar_ei.shape:
        raise ValueError("cn1, cn2 and tar_ei describe different numbers of candidate edges")
    if ops.validate_indices and not (_same_edges(cn1.tarei, cn2.tarei) and _same_edges(cn1.tarei, tar_ei)):
        raise NotImplementedError("cn1, cn2 and tar_ei must be built from the same candidate edges")
    if cn1.mode == "walk1" and cn2.mode == "walk2":
        return CNState(cn1.adj1, None, None, cn1.tarei, walk=True, ws=ws)
    if cn1.mode != "pattern" or cn2.mode != "pattern":
        raise NotImplementedError("mixing adjoverlap and get_cn1_cn2 handles in one predictor call")
    return CNState(cn1.adj1, cn1.adj2, cn2.adj2, cn1.tarei, ws=ws)


def adjoverlap(adj1: SparseTensor, adj2: SparseTensor, tarei: Tensor, filled1: bool = False,
               calresadj: bool = False, cnsampledeg: int = -1, ressampledeg: int = -1) -> CNBatch:
    """utils.py:248-285.  ``calresadj`` / sampling belong to the cn2-cn4 predictors, which no
    reference driver can call."""
    if calresadj or cnsampledeg > 0 or ressampledeg > 0:
        raise NotImplementedError("calresadj / neighbour sampling are outside the cn5/cn7 path")
    return CNBatch(adj1, adj2, tarei)


def get_cn1_cn2(adj: SparseTensor, tedge: Tensor) -> Tuple[CNBatch, CNBatch]:
    """NeighborOverlap_large_ppa.py:147-173 / NeighborOverlapCitation2.py:78-104:
    cn1 = Ei ⊙ Ej, cn2 = Ei ⊙ (Ej · A) with the number of 2-walks as values — no global A²."""
    return CNBatch(adj, None, tedge, "walk1"), CNBatch(adj, None, tedge, "walk2")


def block_matrix_multiply(spadj: SparseTensor, block_size: int, fold_quirk: bool = False) -> SparseTensor:
    """utils.py:287-323: A·A for the dense ddi graph by the reference's tile loop — (row block, column block) products of
    ``block_size`` on the integer matrix cores over A as a dense 0/1 matrix (ops.dense_block_adj2).  Values are not
    formed: ``adjoverlap`` discards them (utils.py:150-151).

    ``fold_quirk=False`` (default, what every parity claim uses): the offset-correct pattern of A².
    ``fold_quirk=True``: the reference as written — each block's ``SparseTensor.from_dense`` carries block-local
    indices and is added without the block's offset (utils.py:318-321), so all blocks fold onto the
    top-left ``block_size`` corner (the CPU restatement used by the tests has the same switch)."""
    n = spadj.size(0)
    if spadj.size(1) != n:
        raise ValueError("block_matrix_multiply needs a square adjacency")
    if n <= ops.dense_adj2_max_nodes and block_size > 0 and block_size % 32 == 0 and spadj._col.is_cuda:
        rowptr, col, bits = ops.dense_block_adj2(spadj._rowptr, spadj._col, n, int(block_size), fold=fold_quirk)
        out = SparseTensor(rowptr=rowptr, col=col, sparse_sizes=(n, n))
        if not fold_quirk:
            out._bitmap = bits                   # dense bit rows of A²: one probe per membership test in the intersection
            out._published("bitmap")
        return out
    if fold_quirk:
        raise NotImplementedError("fold_quirk needs the dense block route (n <= ops.dense_adj2_max_nodes, block_size % 32 == 0)")
    return SparseTensor.from_torch_sparse_coo_tensor(
        spadj.to_torch_sparse_coo_tensor() @ spadj.to_torch_sparse_coo_tensor(), False)


def sparse_tensor_multiply(spadj: SparseTensor, block_size: int = 1024) -> SparseTensor:
    """utils.py:326-329.  Which reading of utils.py:318-321 the unchanged ddi command gets is a process-wide
    switch: ``ops.adj2_fold_quirk`` (environment ``OCN_ADJ2_FOLD_QUIRK=1``) selects the reference as written — every
    block folded onto the top-left corner; the default is the intended offset-correct A²."""
    return block_matrix_multiply(spadj, block_size, fold_quirk=bool(ops.adj2_fold_quirk))
