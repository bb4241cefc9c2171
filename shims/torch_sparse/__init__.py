"""`from torch_sparse import SparseTensor` (NeighborOverlap_large.py:6) -> ocn_amd.sparse.SparseTensor: the
torch_sparse surface the drivers and utils.py touch, over int64 rowptr + int32 col resident in HBM."""
from _shimguard import shadowing as _shadowing

_shadowing("torch_sparse")
from ocn_amd.sparse import SparseTensor  # noqa: F401


def spmm_add(src, other):
    """torch_sparse.spmm_add (model.py:2426): rows of `src` times dense `other`, summed."""
    from ocn_amd import ops
    return ops.spmm_csr(src._rowptr, src._col, other.contiguous(), val=src._value, mode="sum")
