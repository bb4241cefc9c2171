"""`from pygho.backend.Spspmm import spsphadamard, spspmm` (NeighborOverlap_large_ppa.py:24,148-154; model.py:17): the two
calls of the drivers' get_cn1_cn2 on deferred expressions (shims/pygho/__init__.py).  spsphadamard keeps Ei's pattern;
the CN count is the number of NON-ZERO values — which is what the CNBatch handle counts."""
from pygho import LazyCN, RowSelect, TwoHop, SparseTensor


def spspmm(A, dim1: int, B, dim2: int, aggr: str = "sum", *unused, **kw):
    """spspmm(Ej, 1, adj, 0): (adj[dst]) @ adj, values = number of 2-walks — deferred."""
    if isinstance(A, RowSelect) and isinstance(B, SparseTensor) and (dim1, dim2) == (1, 0) and aggr == "sum" and A.adj is B:
        return TwoHop(A)
    raise NotImplementedError(f"pygho stand-in: spspmm({type(A).__name__}, {dim1}, {type(B).__name__}, {dim2}, aggr={aggr!r}) — only "
                              "spspmm(adj.index_select([0], ids), 1, adj, 0) of the drivers' get_cn1_cn2 is emulated")


def spsphadamard(A, B, *unused, **kw):
    """spsphadamard(Ei, Ej) -> cn1;  spsphadamard(Ei, Ej @ adj) -> cn2 — deferred."""
    if isinstance(A, RowSelect) and isinstance(B, RowSelect) and A.adj is B.adj:
        return LazyCN(A.adj, A.idx, B.idx, "walk1")
    if isinstance(A, RowSelect) and isinstance(B, TwoHop) and A.adj is B.sel.adj:
        return LazyCN(A.adj, A.idx, B.sel.idx, "walk2")
    raise NotImplementedError(f"pygho stand-in: spsphadamard({type(A).__name__}, {type(B).__name__}) — only Ei (.) Ej and "
                              "Ei (.) (Ej @ adj) over one adjacency are emulated (model.py:2243 calls it on explicit matrices in "
                              "training: ocn_amd's predictors compute that inner product from the CN flags instead)")
