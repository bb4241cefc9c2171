"""CPU oracle for the OCN common-neighbour predictor hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ocn_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may use it, and there only as the checker / the timed CPU port.

PARITY UNPINNED.  The reference (qingpingmo/OCN @ 2025-10-24) ships no tests,
no golden vectors and cannot be imported in the build container (its
arithmetic lives in un-vendored third-party packages: torch-sparse 0.6.18,
torch-scatter 2.1.2, torch-geometric 2.6.1, pygho (unpinned)).
This file therefore restates the reference's algorithm from source reading,
op for op, with the torch-native calls the reference itself makes
(``searchsorted``, ``unique(dim=1)``, ``sparse_coo_tensor().coalesce()``,
``index_add_``) and with int64 indices / fp32 values as the reference uses.
It is pinned only by (a) the hand-derived known answers in
``tests/golden/appendix_c.json`` and (b) the independent set-based model in
``oracle/naive_model.py``.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor


# ----------------------------------------------------------------------------
# Sparse containers (stand-ins for torch_sparse.SparseTensor, int64 like it)
# ----------------------------------------------------------------------------
@dataclass
class SpM:
    """COO/CSR pattern (optionally valued) sorted by (row, col).

    Mirrors the pieces of torch_sparse.SparseTensor the hot path touches:
    storage.row()/col()/value(), sizes(), csr().
    """
    row: Tensor            # int64 [nnz], non-decreasing
    col: Tensor            # int64 [nnz]
    val: Optional[Tensor]  # fp32 [nnz] or None (pattern only)
    n_rows: int
    n_cols: int

    @property
    def nnz(self) -> int:
        return int(self.row.numel())

    def rowptr(self) -> Tensor:
        """CSR row pointer, cached on first use as torch_sparse's SparseStorage caches its ``_rowptr``
        (the reference's ``adj[idx]`` at utils.py:256-257 therefore pays the bincount once per adjacency,
        not once per batch)."""
        out = self.__dict__.get("_rowptr_cache")
        if out is None:
            cnt = torch.bincount(self.row, minlength=self.n_rows)
            out = torch.zeros(self.n_rows + 1, dtype=torch.long)
            torch.cumsum(cnt, 0, out=out[1:])
            self.__dict__["_rowptr_cache"] = out
        return out

    def rowcount(self) -> Tensor:
        return torch.bincount(self.row, minlength=self.n_rows)

    def to_dense(self) -> Tensor:
        d = torch.zeros(self.n_rows, self.n_cols)
        v = self.val if self.val is not None else torch.ones(self.nnz)
        d.index_put_((self.row, self.col), v, accumulate=True)
        return d


def from_edge_index(ei: Tensor, n: int) -> SpM:
    """SparseTensor.from_edge_index(ei, sparse_sizes=(n, n)): sort by (row, col),
    duplicates kept, no values (NeighborOverlap_large.py:59-60, ogbdataset.py:44)."""
    key = ei[0] * n + ei[1]
    perm = torch.argsort(key, stable=True)
    return SpM(ei[0][perm].contiguous(), ei[1][perm].contiguous(), None, n, n)


def coalesce(m: SpM) -> SpM:
    key = torch.unique(m.row * m.n_cols + m.col)  # sorted
    return SpM(torch.div(key, m.n_cols, rounding_mode="floor"), key % m.n_cols, None,
               m.n_rows, m.n_cols)


def to_symmetric(m: SpM) -> SpM:
    """SparseTensor.to_symmetric() on a value-less matrix: pattern of A ∪ Aᵀ,
    coalesced (NeighborOverlap_large.py:63, ogbdataset.py:45)."""
    r = torch.cat([m.row, m.col])
    c = torch.cat([m.col, m.row])
    return coalesce(SpM(r, c, None, m.n_rows, m.n_cols))


def row_select(m: SpM, idx: Tensor) -> SpM:
    """``adj[idx]`` — SparseTensor.__getitem__(LongTensor) = index_select on dim 0
    (utils.py:256-257).  Row e of the result is row idx[e] of ``m``; order and
    duplicates of ``idx`` are preserved."""
    rowptr = m.rowptr()
    start = rowptr[idx]
    deg = rowptr[idx + 1] - start
    out_ptr = torch.zeros(idx.numel() + 1, dtype=torch.long)
    torch.cumsum(deg, 0, out=out_ptr[1:])
    total = int(out_ptr[-1])
    row = torch.repeat_interleave(torch.arange(idx.numel()), deg)
    pos = torch.arange(total) - out_ptr[:-1][row] + start[row]
    val = m.val[pos] if m.val is not None else None
    return SpM(row, m.col[pos], val, int(idx.numel()), m.n_cols)


# ----------------------------------------------------------------------------
# utils.py: intersection
# ----------------------------------------------------------------------------
def spm2elem(m: SpM) -> Tensor:
    """utils.py:154-160 — int64 key (row << 32) + col."""
    return torch.bitwise_left_shift(m.row, 32).add_(m.col)


def elem2spm(elem: Tensor, n_rows: int, n_cols: int) -> SpM:
    """utils.py:146-151 — split the key again; every value becomes 1.0 fp32."""
    col = torch.bitwise_and(elem, 0xFFFFFFFF)
    row = torch.bitwise_right_shift(elem, 32)
    return SpM(row, col, torch.ones(elem.numel(), dtype=torch.float32), n_rows, n_cols)


def spmoverlap_(a1: SpM, a2: SpM) -> SpM:
    """utils.py:162-183 — sorted-key intersection; the longer key vector is the
    haystack, ``searchsorted(hay[:-1], needles)`` + equality test."""
    assert (a1.n_rows, a1.n_cols) == (a2.n_rows, a2.n_cols)
    e1, e2 = spm2elem(a1), spm2elem(a2)
    if e2.shape[0] > e1.shape[0]:
        e1, e2 = e2, e1
    idx = torch.searchsorted(e1[:-1], e2)
    mask = e1[idx] == e2
    return elem2spm(e2[mask], a1.n_rows, a1.n_cols)


def adjoverlap(adj1: SpM, adj2: SpM, tarei: Tensor) -> SpM:
    """utils.py:248-285, non-calresadj branch, cnsampledeg <= 0."""
    return spmoverlap_(row_select(adj1, tarei[0]), row_select(adj2, tarei[1]))


# ----------------------------------------------------------------------------
# A² builders
# ----------------------------------------------------------------------------
def adj2_sparse(adj: SpM) -> SpM:
    """NeighborOverlap_large.py:68-74,112-119 — ``spadj @ spadj`` on the fp32-ones
    COO tensor, then ``from_torch_sparse_coo_tensor(.., has_value=False)``:
    pattern of the coalesced product."""
    sp = torch.sparse_coo_tensor(torch.stack([adj.row, adj.col]),
                                 torch.ones(adj.nnz), (adj.n_rows, adj.n_cols))
    prod = torch.sparse.mm(sp, sp).coalesce()
    r, c = prod.indices()
    return SpM(r.contiguous(), c.contiguous(), None, adj.n_rows, adj.n_cols)


def adj2_by_block(adj: SpM, block_size: int = 1024, fold_quirk: bool = False) -> SpM:
    """utils.py:287-329 — dense row-block × col-block products, each turned
    back into a sparse matrix and accumulated.

    ``fold_quirk=False`` (default, used by every parity claim): tiles are placed
    at their (i, j) offset, i.e. the intended A² with walk-count values.
    ``fold_quirk=True``: reproduces SURVEY.md Q7 — ``SparseTensor.from_dense``
    yields block-local indices that the reference adds without an offset, so all
    tiles fold onto the top-left corner ([3P-memory]: depends on torch_sparse's
    SparseTensor + SparseTensor).  Both readings are pinned by hand-derived path-graph
    answers (tests/test_oracle.py) and asserted against the product's block route."""
    n = adj.n_rows
    dense = adj.to_dense()
    rows, cols, vals = [], [], []
    for i in range(0, n, block_size):
        ie = min(i + block_size, n)
        for j in range(0, n, block_size):
            je = min(j + block_size, n)
            blk = dense[i:ie, :] @ dense[:, j:je]
            r, c = blk.nonzero(as_tuple=True)
            rows.append(r if fold_quirk else r + i)
            cols.append(c if fold_quirk else c + j)
            vals.append(blk[r, c])
    r, c, v = torch.cat(rows), torch.cat(cols), torch.cat(vals)
    key, inv = torch.unique(r * n + c, return_inverse=True)
    val = torch.zeros(key.numel()).index_add_(0, inv, v)
    return SpM(torch.div(key, n, rounding_mode="floor"), key % n, val, n, n)


def get_cn1_cn2(adj: SpM, tedge: Tensor) -> Tuple[SpM, SpM]:
    """NeighborOverlap_large_ppa.py:147-173 (pygho route, [3P-memory] for pygho):
    cn1 = Ei ⊙ Ej; cn2 = Ei ⊙ (Ej · A) whose value at (e, k) is the number of
    2-walks j_e → k = |N(k) ∩ N(j_e)|.  Entries of Ei with no match are dropped
    here (explicit zeros are immaterial to every downstream sum/product)."""
    n = adj.n_cols
    Ei, Ej = row_select(adj, tedge[0]), row_select(adj, tedge[1])
    cn1 = spmoverlap_(Ei, Ej)
    spEj = torch.sparse_coo_tensor(torch.stack([Ej.row, Ej.col]), torch.ones(Ej.nnz),
                                   (Ej.n_rows, n))
    spA = torch.sparse_coo_tensor(torch.stack([adj.row, adj.col]), torch.ones(adj.nnz), (n, n))
    Ej2 = torch.sparse.mm(spEj, spA).coalesce()
    r2, c2 = Ej2.indices()
    k2 = torch.bitwise_left_shift(r2, 32) + c2
    ki = spm2elem(Ei)
    idx = torch.searchsorted(k2, ki).clamp_(max=max(k2.numel() - 1, 0))
    hit = (k2[idx] == ki) if k2.numel() else torch.zeros_like(ki, dtype=torch.bool)
    cn2 = SpM(Ei.row[hit], Ei.col[hit], Ej2.values()[idx[hit]].to(torch.float32),
              Ei.n_rows, n)
    return cn1, cn2


# ----------------------------------------------------------------------------
# small sparse helpers used by the predictors
# ----------------------------------------------------------------------------
def col_sum(m: SpM) -> Tensor:
    """``cn.sum(dim=0)`` (model.py:2261,3114,3168) — fp32 scatter-add by column."""
    return torch.zeros(m.n_cols, dtype=torch.float32).index_add_(0, m.col, m.val)


def spmm_add(m: SpM, x: Tensor, chunk: int = 1 << 18) -> Tensor:
    """torch_sparse ``spmm_add`` (model.py:2426-2427,3213-3214): out[r] = Σ val·x[col],
    entries of a row taken in ascending column order, fp32 multiply then add."""
    out = torch.zeros(m.n_rows, x.shape[1], dtype=torch.float32)
    for s in range(0, m.nnz, chunk):
        e = min(s + chunk, m.nnz)
        out.index_add_(0, m.row[s:e], m.val[s:e, None] * x[m.col[s:e]])
    return out


# ----------------------------------------------------------------------------
# predictor MLP heads (model.py:2192-2235 ≡ 3044-3087), functional over a state_dict
# ----------------------------------------------------------------------------
def _seq(sd: Dict[str, Tensor], prefix: str, x: Tensor, layout) -> Tensor:
    """Run an nn.Sequential restated as a list of (kind, index); dropouts are
    identities in eval mode and are omitted."""
    for kind, i in layout:
        if kind == "lin":
            x = F.linear(x, sd[f"{prefix}.{i}.weight"], sd[f"{prefix}.{i}.bias"])
        elif kind == "ln":
            w = sd[f"{prefix}.{i}.weight"]
            x = F.layer_norm(x, (w.numel(),), w, sd[f"{prefix}.{i}.bias"], 1e-5)
        elif kind == "relu":
            x = torch.relu(x)
    return x


def _xcn_layout(ln: bool):
    # Lin(0) Drop(1) ReLU(2) Lin(3) LN?(4) Drop(5) ReLU(6) Lin(7)     model.py:2203-2214
    return ([("lin", 0), ("relu", 2), ("lin", 3)] + ([("ln", 4)] if ln else [])
            + [("relu", 6), ("lin", 7)])


def _xij_layout(ln: bool, tailact: bool):
    # Lin(0) LN?(1) Drop(2) ReLU(3) (Lin(4) | Identity)               model.py:2223-2226
    return ([("lin", 0)] + ([("ln", 1)] if ln else []) + [("relu", 3)]
            + ([] if tailact else [("lin", 4)]))


def _lin_layout(ln: bool, twolayerlin: bool):
    # Lin(0) LN?(1) Drop(2) ReLU(3) [Lin(4) LN?(5) Drop(6) ReLU(7)] Lin(8)   model.py:2227-2235
    lay = [("lin", 0)] + ([("ln", 1)] if ln else []) + [("relu", 3)]
    if twolayerlin:
        lay += [("lin", 4)] + ([("ln", 5)] if ln else []) + [("relu", 7)]
    return lay + [("lin", 8)]


def _heads(sd, x, xcn1, xcn2, tar_ei, ln, tailact, twolayerlin) -> Tensor:
    """model.py:2429-2437 ≡ 3216-3223."""
    xij = _seq(sd, "xijlin", x[tar_ei[0]] * x[tar_ei[1]], _xij_layout(ln, tailact))
    xcn1 = _seq(sd, "xcn1lin", xcn1, _xcn_layout(ln))
    xcn2 = _seq(sd, "xcn2lin", xcn2, _xcn_layout(ln))
    alpha = torch.sigmoid(sd["alpha"]).cumprod(-1)
    z = alpha[0] * xcn1 + alpha[1] * xcn2 + sd["beta"] * xij
    return _seq(sd, "lin", z, _lin_layout(ln, twolayerlin))


# ----------------------------------------------------------------------------
# cn5 = CNLinkPredictorOringin.multidomainforward, eval mode (model.py:2252-2440)
# ----------------------------------------------------------------------------
def cn5_pool(x: Tensor, cn1: SpM, cn2: SpM, innerprod: Tensor):
    """Steps 1-6 of SURVEY Appendix A.3; returns (xcn1, xcn2, aux)."""
    S1 = col_sum(cn1)                                   # :2261
    S1[S1 == 0] = 1                                     # :2263
    inv1 = 1 / S1                                       # :2264
    inv1[~(S1 != 1)] = 0                                # :2265-2266  (Q2)
    ncn1 = SpM(cn1.row, cn1.col, inv1[cn1.col] * cn1.val, cn1.n_rows, cn1.n_cols)  # :2272
    B, N = cn1.n_rows, cn1.n_cols

    sp_ncn1 = torch.sparse_coo_tensor(torch.stack([ncn1.row, ncn1.col]), ncn1.val, (B, N)).coalesce()
    sp_cn2 = torch.sparse_coo_tensor(torch.stack([cn2.row, cn2.col]), cn2.val, (B, N)).coalesce()
    i2, i1 = sp_cn2.indices(), sp_ncn1.indices()        # :2336-2345
    uniq, inv = torch.unique(torch.cat((i2, i1), dim=1), dim=1, return_inverse=True)  # :2352-2356
    a2 = torch.zeros(uniq.size(1))
    a2[inv[:i2.size(1)]] = sp_cn2.values()              # :2358-2359
    a1 = torch.zeros(uniq.size(1))
    a1[inv[i2.size(1):i2.size(1) + i1.size(1)]] = sp_ncn1.values()   # :2361-2362
    scale = a1.abs().max().item() if a1.numel() > 0 else 1.0         # :2370-2375
    nip = innerprod / scale if scale > 0 else innerprod              # :2376 (eval: stored buffer)
    newv = a2 - nip * a1                                # :2380-2384
    sp2 = torch.sparse_coo_tensor(uniq, newv, (B, N)).coalesce()     # :2386-2398
    idx, vals = sp2.indices(), sp2.values()
    S2 = torch.zeros(N).index_add_(0, idx[1], vals)     # :2405-2406
    S2[S2 == 0] = 1                                     # :2409
    inv2 = 1 / S2                                       # :2410
    ncn2 = SpM(idx[0], idx[1], vals * inv2[idx[1]], B, N)            # :2413-2423
    xcn1 = spmm_add(ncn1, x)                            # :2426
    xcn2 = spmm_add(ncn2, x)                            # :2427
    aux = dict(S1=S1, inv1=inv1, scale=scale, nip=nip, S2=S2, ncn1=ncn1, ncn2=ncn2)
    return xcn1, xcn2, aux


def cn5_batch_innerprod(cn1: SpM, cn2: SpM) -> Tensor:
    """innerprod1 in training mode (model.py:2241-2244): Σ of the Hadamard product of cn2 and the
    column-normalised cn1 (step 1 of A.3), in the entry order of cn2."""
    S1 = col_sum(cn1)
    S1[S1 == 0] = 1
    inv1 = 1 / S1
    inv1[~(S1 != 1)] = 0
    k1, k2 = spm2elem(cn1), spm2elem(cn2)
    idx = torch.searchsorted(k1, k2).clamp_(max=max(k1.numel() - 1, 0))
    hit = (k1[idx] == k2) if k1.numel() else torch.zeros_like(k2, dtype=torch.bool)
    return (cn2.val[hit] * (inv1[cn1.col] * cn1.val)[idx[hit]]).sum()


def cn5_forward(sd: Dict[str, Tensor], x: Tensor, cn1: SpM, cn2: SpM, tar_ei: Tensor,
                ln: bool = False, tailact: bool = False, twolayerlin: bool = False) -> Tensor:
    xcn1, xcn2, _ = cn5_pool(x, cn1, cn2, sd["innerprod"])
    return _heads(sd, x, xcn1, xcn2, tar_ei, ln, tailact, twolayerlin)


# ----------------------------------------------------------------------------
# cn6 = CNLinkPredictor3hopCNs.multidomainforward, eval mode (model.py:2535-2951)
# ----------------------------------------------------------------------------
def adj3_sparse(adj: SpM, adj2: SpM) -> SpM:
    """Pattern of A·A·A = A² · A (values discarded), the 3-hop analogue of NeighborOverlap_large.py:
    68-74.  No reference driver builds it (cn6 is registered in predictor_dict, model.py:3725, but no
    driver passes a cn3); cn3 = adjoverlap(adj, adj3, e) is the natural continuation of the cn1 / cn2
    pair and what ``forward(x, adj, cn1, cn2, cn3, tar_ei, args)`` (model.py:2950) expects."""
    sp2 = torch.sparse_coo_tensor(torch.stack([adj2.row, adj2.col]), torch.ones(adj2.nnz), (adj.n_rows, adj.n_cols))
    sp1 = torch.sparse_coo_tensor(torch.stack([adj.row, adj.col]), torch.ones(adj.nnz), (adj.n_rows, adj.n_cols))
    prod = torch.sparse.mm(sp2, sp1).coalesce()
    idx = prod.indices()
    return SpM(idx[0].contiguous(), idx[1].contiguous(), None, adj.n_rows, adj.n_cols)


def cn6_pool(x: Tensor, cn1: SpM, cn2: SpM, cn3: SpM, innerprod: Tensor):
    """model.py:2546-2940 in eval mode (``innerprod1`` returns the stored buffer three times):
    stage 1 = cn5's normalise / orthogonalise / normalise of (cn1, cn2); stage 2 orthogonalises cn3
    against BOTH normalised matrices over the union pattern and column-normalises it."""
    xcn1, xcn2, aux = cn5_pool(x, cn1, cn2, innerprod)            # :2546-2714 == cn5 :2261-2427
    ncn1, ncn2 = aux["ncn1"], aux["ncn2"]
    B, N = cn1.n_rows, cn1.n_cols
    sp_ncn1 = torch.sparse_coo_tensor(torch.stack([ncn1.row, ncn1.col]), ncn1.val, (B, N)).coalesce()   # :2734-2748
    sp_ncn2 = torch.sparse_coo_tensor(torch.stack([ncn2.row, ncn2.col]), ncn2.val, (B, N)).coalesce()
    sp_cn3 = torch.sparse_coo_tensor(torch.stack([cn3.row, cn3.col]), cn3.val, (B, N)).coalesce()
    i3, i1, i2 = sp_cn3.indices(), sp_ncn1.indices(), sp_ncn2.indices()                                  # :2846-2852
    uniq, inv = torch.unique(torch.cat((i3, i1, i2), dim=1), dim=1, return_inverse=True)                 # :2860-2866
    a3 = torch.zeros(uniq.size(1))
    a3[inv[:i3.size(1)]] = sp_cn3.values()                                                               # :2868-2869
    a1 = torch.zeros(uniq.size(1))
    a1[inv[i3.size(1):i3.size(1) + i1.size(1)]] = sp_ncn1.values()                                       # :2871-2875
    a2 = torch.zeros(uniq.size(1))
    a2[inv[i3.size(1) + i1.size(1):]] = sp_ncn2.values()                                                 # :2877-2881
    scale = a1.abs().max().item() if a1.numel() > 0 else 1.0                                             # :2883-2886
    nip1 = innerprod / scale if scale > 0 else innerprod                                                 # :2888-2893
    nip2 = innerprod / scale if scale > 0 else innerprod
    newv = a3 - nip1 * a1 - nip2 * a2                                                                    # :2895-2899
    sp3 = torch.sparse_coo_tensor(uniq, newv, (B, N)).coalesce()                                         # :2901-2907
    idx, vals = sp3.indices(), sp3.values()
    S3 = torch.zeros(N).index_add_(0, idx[1], vals)                                                      # :2912-2913
    S3[S3 == 0] = 1                                                                                      # :2917
    inv3 = 1 / S3
    ncn3 = SpM(idx[0], idx[1], vals * inv3[idx[1]], B, N)                                                # :2921-2931
    xcn3 = spmm_add(ncn3, x)                                                                             # :2933
    aux = dict(aux, S3=S3, ncn3=ncn3, scale2=scale)
    return xcn1, xcn2, xcn3, aux


def cn6_forward(sd: Dict[str, Tensor], x: Tensor, cn1: SpM, cn2: SpM, cn3: SpM, tar_ei: Tensor,
                ln: bool = False, tailact: bool = False, twolayerlin: bool = False) -> Tensor:
    """model.py:2934-2947: four branches, alpha has three live entries."""
    xcn1, xcn2, xcn3, _ = cn6_pool(x, cn1, cn2, cn3, sd["innerprod"])
    xij = _seq(sd, "xijlin", x[tar_ei[0]] * x[tar_ei[1]], _xij_layout(ln, tailact))
    xcn1 = _seq(sd, "xcn1lin", xcn1, _xcn_layout(ln))
    xcn2 = _seq(sd, "xcn2lin", xcn2, _xcn_layout(ln))
    xcn3 = _seq(sd, "xcn3lin", xcn3, _xcn_layout(ln))
    alpha = torch.sigmoid(sd["alpha"]).cumprod(-1)
    z = alpha[0] * xcn1 + alpha[1] * xcn2 + alpha[2] * xcn3 + sd["beta"] * xij
    return _seq(sd, "lin", z, _lin_layout(ln, twolayerlin))


# ----------------------------------------------------------------------------
# cn7 = CNLinkPredictorbaselearn.multidomainforward, eval (model.py:3102-3226)
# ----------------------------------------------------------------------------
# model.py:2958-2990: the Chebyshev polynomials T0 .. T10, written as the reference writes them (fp32: the order of the
# terms and the ``**`` powers are part of the value)
POLYNOMIALS = [
    lambda x: torch.ones_like(x),
    lambda x: x,
    lambda x: 2 * x**2 - 1,
    lambda x: 4 * x**3 - 3 * x,
    lambda x: 8 * x**4 - 8 * x**2 + 1,
    lambda x: 16 * x**5 - 20 * x**3 + 5 * x,
    lambda x: 32 * x**6 - 48 * x**4 + 18 * x**2 - 1,
    lambda x: 64 * x**7 - 112 * x**5 + 56 * x**3 - 7 * x,
    lambda x: 128 * x**8 - 256 * x**6 + 160 * x**4 - 32 * x**2 + 1,
    lambda x: 256 * x**9 - 576 * x**7 + 432 * x**5 - 120 * x**3 + 9 * x,
    lambda x: 512 * x**10 - 1280 * x**8 + 1120 * x**6 - 400 * x**4 + 50 * x**2 - 1,
]


def evaluate_polynomial(n: int, poly_index: int) -> Tensor:
    """model.py:2995-3019: the diagonal of ``diag(T_k(linspace(-1, 1, n)))`` (the reference wraps it in an n x n
    SparseTensor; only the diagonal carries data)."""
    if poly_index < 0 or poly_index >= len(POLYNOMIALS):
        raise ValueError(f"Invalid poly_index. Must be between 0 and {len(POLYNOMIALS)-1}.")
    return POLYNOMIALS[poly_index](torch.linspace(-1, 1, n))


def cn7_pool(x: Tensor, cn1: SpM, cn2: SpM, sum_fill: float, polyfirst: int = 0, polysecond: int = 0):
    """``polyfirst`` / ``polysecond``: the index the reference passes to evaluate_polynomial at :3141 / :3186 — the literal 0
    there (T0 = identity; the drivers' --polyfirst / --polysecond flags are never read)."""
    S1 = col_sum(cn1)                                   # :3114
    S1[S1 == 0] = 1
    inv1 = 1 / S1
    inv1[~(S1 != 1)] = sum_fill                         # :3120 (Q2 with args.sum)
    # × diag(T_k(linspace(-1,1,N))) (model.py:3141-3165): spspmm with a diagonal = one product per entry
    d1 = evaluate_polynomial(cn1.n_cols, polyfirst)
    ncn1 = SpM(cn1.row, cn1.col, (inv1[cn1.col] * cn1.val) * d1[cn1.col], cn1.n_rows, cn1.n_cols)
    # normalized_cn2 is computed and discarded (:3168-3180, Q5); the RAW cn2 × the second diagonal is used (:3186-3209).
    d2 = evaluate_polynomial(cn2.n_cols, polysecond)
    rcn2 = SpM(cn2.row, cn2.col, cn2.val * d2[cn2.col], cn2.n_rows, cn2.n_cols)
    xcn1 = spmm_add(ncn1, x)                            # :3213
    xcn2 = spmm_add(rcn2, x)                            # :3214
    return xcn1, xcn2, dict(S1=S1, inv1=inv1, ncn1=ncn1)


def cn7_forward(sd, x, cn1, cn2, tar_ei, sum_fill: float, ln=False, tailact=False,
                twolayerlin=False, polyfirst: int = 0, polysecond: int = 0) -> Tensor:
    xcn1, xcn2, _ = cn7_pool(x, cn1, cn2, sum_fill, polyfirst, polysecond)
    return _heads(sd, x, xcn1, xcn2, tar_ei, ln, tailact, twolayerlin)


# ----------------------------------------------------------------------------
# Encoders (model.py:32-55, 58-82, 85-142, 232-511)
# ----------------------------------------------------------------------------
def spmm_pattern(adj: SpM, x: Tensor, w: Optional[Tensor] = None) -> Tensor:
    v = w if w is not None else torch.ones(adj.nnz)
    return spmm_add(SpM(adj.row, adj.col, v, adj.n_rows, adj.n_cols), x)


def pureconv_gcn(adj: SpM, x: Tensor) -> Tensor:
    """model.py:50-55: n = rsqrt(1+deg); y = n ⊙ (A(n ⊙ x) + n ⊙ x)."""
    norm = torch.rsqrt(1 + adj.rowcount().to(torch.float32)).reshape(-1, 1)
    x = norm * x
    x = spmm_pattern(adj, x) + x
    return norm * x


def pureconv(adj: SpM, x: Tensor, aggr: str) -> Tensor:
    if aggr == "gcn":
        return pureconv_gcn(adj, x)
    if aggr == "sum":                                   # model.py:48-49
        return spmm_pattern(adj, x)
    if aggr == "mean":                                  # model.py:44-45
        deg = adj.rowcount().to(torch.float32).clamp_(min=1).reshape(-1, 1)
        return spmm_pattern(adj, x) / deg
    if aggr == "max":                                   # model.py:46-47
        out = torch.full((adj.n_rows, x.shape[1]), float("-inf"))
        out = out.index_reduce_(0, adj.row, x[adj.col], "amax", include_self=True)
        out[adj.rowcount() == 0] = 0
        return out
    raise ValueError(aggr)


def gcnconv(adj: SpM, x: Tensor, weight: Tensor, bias: Tensor, normalize: bool) -> Tensor:
    """PyG GCNConv as wired by convdict (model.py:58-68) [3P-memory]:
    gin key: A·(xWᵀ) + b;  gcn key: D̃^-½(A+I)D̃^-½·(xWᵀ) + b."""
    x = F.linear(x, weight)
    if not normalize:
        return spmm_pattern(adj, x) + bias
    n = adj.n_rows
    keep = adj.row != adj.col
    r = torch.cat([adj.row[keep], torch.arange(n)])
    c = torch.cat([adj.col[keep], torch.arange(n)])
    perm = torch.argsort(r * n + c)
    a = SpM(r[perm], c[perm], None, n, n)               # fill_diag(adj, 1.0)
    deg = a.rowcount().to(torch.float32)
    dinv = deg.pow(-0.5)
    dinv[dinv == float("inf")] = 0
    w = (torch.ones(a.nnz) * dinv[a.row]) * dinv[a.col]
    return spmm_pattern(a, x, w) + bias


def pureconv23_gcn(adj: SpM, x: Tensor) -> Tensor:
    """model.py:105-113 / 136-140: n = rsqrt(1+deg); e = n[row]·n[col]; y = (A ⊙ e) x."""
    val = adj.val if adj.val is not None else torch.ones(adj.nnz)
    deg = torch.zeros(adj.n_rows).index_add_(0, adj.row, val)
    norm = torch.rsqrt(1 + deg)
    enorm = norm[adj.row] * norm[adj.col]
    return spmm_pattern(adj, x, val * enorm)


def gcn_forward(sd: Dict[str, Tensor], x: Tensor, adj: SpM, *, num_layers: int, conv_fn: str,
                ln: bool = False, res: bool = False, jk: bool = False, max_x: int = -1,
                variant: int = 1) -> Tensor:
    """GCN / GCN2 / GCN3 .forward in eval mode (model.py:308-323, 402-417, 496-511)."""
    pure = "pure" in conv_fn
    # xemb (model.py:253-262)
    if max_x >= 0:
        x = sd["xemb.0.weight"][x]
    elif "xemb.1.weight" in sd:
        x = F.linear(x, sd["xemb.1.weight"], sd["xemb.1.bias"])
    if num_layers == 0 or conv_fn == "none":
        return x
    jkx = []
    for i in range(num_layers):
        if variant == 1:
            if pure:
                y = pureconv(adj, x, conv_fn[4:])
            else:
                norm = {"gcn": True, "gcn_cached": True, "gin": False}[conv_fn]
                y = gcnconv(adj, x, sd[f"convs.{i}.lin.weight"], sd[f"convs.{i}.bias"], norm)
        else:
            aggr = conv_fn[4:] if pure else {"gcn": "gcn", "gcn_cached": "gcn", "gin": "sum"}[conv_fn]
            if aggr == "gcn":
                y = pureconv23_gcn(adj, x)
            else:
                y = spmm_pattern(adj, x, adj.val)
            if not pure:                                # Linear(no bias) + ReLU after aggregation
                y = torch.relu(F.linear(y, sd[f"convs.{i}.lin.0.weight"]))
        # lins[i] (model.py:281-305): non-pure, not last -> LN? Drop ReLU; else identity in eval
        if not pure and (i == 0 or i < num_layers - 1):
            if ln:
                w = sd[f"lins.{i}.0.weight"]
                y = F.layer_norm(y, (w.numel(),), w, sd[f"lins.{i}.0.bias"], 1e-5)
            y = torch.relu(y)
        x = y + x if (res and y.shape[-1] == x.shape[-1]) else y
        if jk:
            jkx.append(x)
    if jk:
        x = torch.sum(torch.stack(jkx, 0) * sd["jkparams"].reshape(-1, 1, 1), dim=0)
    return x


def perm_batches(size: int, bs: int):
    """utils.PermIterator(training=False) (utils.py:8-36): contiguous slices, ragged tail kept."""
    return [torch.arange(s, min(s + bs, size)) for s in range(0, size, bs)]
