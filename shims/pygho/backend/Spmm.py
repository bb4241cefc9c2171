"""`from pygho.backend.Spmm import spmm` (NeighborOverlap_large_ppa.py:25): imported by the drivers, called nowhere on the
cn5 / cn7 path (the encoders' SpMM is ocn_amd.ops.spmm_csr)."""
from pygho import SparseTensor


def spmm(A, dim1: int, X, aggr: str = "sum"):
    """A @ X for the 2-D adjacency (dim1 = 1)."""
    from ocn_amd import ops
    if not isinstance(A, SparseTensor) or dim1 != 1 or aggr not in ("sum", "mean", "max"):
        raise NotImplementedError("pygho stand-in: spmm(adj, 1, X, aggr) on the 2-D adjacency only")
    return ops.spmm_csr(A._rowptr, A._col, X.contiguous(), val=A._value, mode=aggr)
