"""`from pygho import SparseTensor as pSparseTensor` (NeighborOverlap_large_ppa.py:23,83-90; NeighborOverlapCitation2.py:22,151).

pygho is an un-vendored, unpinned dependency of the reference.  The drivers use exactly one pattern of it: wrap
the [N, N] adjacency, and in their local `get_cn1_cn2` (…_ppa.py:147-173, …Citation2.py:78-104)

    Ei = adj.index_select([0], tedge[0].unsqueeze(0));  Ej = adj.index_select([0], tedge[1].unsqueeze(0))
    cn1 = spsphadamard(Ei, Ej);  Ej2 = spspmm(Ej, 1, adj, 0);  cn2 = spsphadamard(Ei, Ej2)
    cn = cn.to_torch_sparse_coo();  row, col = cn.indices();  val = cn.values()
    torch_sparse.SparseTensor(row=row, col=col, value=val, sparse_sizes=(B, N))

This package is a LAZY ALGEBRA for that pattern: nothing is computed; every step returns a deferred expression, and the
last line (ocn_amd.sparse.SparseTensor, which shims/torch_sparse re-exports) collapses it to the CNBatch handle
`ocn_amd.utils.get_cn1_cn2` returns — which the predictors fuse into ONE intersection pass.  Anything outside the pattern
raises NotImplementedError naming the expression; a deferred index / value vector used as a real tensor materialises the
explicit matrix (values = number of 2-walks, zeros dropped: SURVEY Appendix A.2)."""
from typing import Optional, Sequence

import torch

from ocn_amd.sparse import SparseTensor as _OcnSparseTensor
from ocn_amd.utils import CNBatch


class SparseTensor(_OcnSparseTensor):
    """pygho.SparseTensor(indices [2, nnz], values [nnz], shape, is_coalesced): the 2-D adjacency only.  It IS an
    ocn_amd SparseTensor (CSR in HBM), so the encoders (GCN2 / GCN3) and the predictors take it as they take adj_t."""

    def __init__(self, indices: torch.Tensor, values: Optional[torch.Tensor] = None, shape: Optional[Sequence[int]] = None,
                 is_coalesced: bool = False):
        if indices.dim() != 2 or indices.shape[0] != 2:
            raise NotImplementedError("pygho stand-in: only 2-D sparse tensors (the drivers' adjacency) are emulated")
        if shape is not None and len(shape) != 2:
            raise NotImplementedError("pygho stand-in: dense trailing dimensions are not emulated")
        super().__init__(row=indices[0], col=indices[1], value=values, sparse_sizes=None if shape is None else tuple(shape),
                         is_sorted=bool(is_coalesced))

    @property
    def shape(self):
        return tuple(self.sizes())

    @property
    def indices(self) -> torch.Tensor:
        return torch.stack([self.storage.row(), self.storage.col()])

    @property
    def values(self) -> Optional[torch.Tensor]:
        return self.storage.value()

    def index_select(self, dims, idx: torch.Tensor) -> "RowSelect":
        """pygho: `adj.index_select([0], ids[None, :])` -> the rows `ids` of adj as a [B, N] sparse tensor (deferred)."""
        d = list(dims) if isinstance(dims, (list, tuple)) else [dims]
        if d != [0] or idx.dim() != 2 or idx.shape[0] != 1:
            raise NotImplementedError(f"pygho stand-in: index_select({dims}, idx{tuple(idx.shape)}) — only rows: ([0], ids[None, :])")
        return RowSelect(self, idx)

    def to_torch_sparse_coo(self) -> torch.Tensor:
        return self.to_torch_sparse_coo_tensor()


def _ids(idx: torch.Tensor) -> torch.Tensor:
    return idx.reshape(-1)


def _edges(src_idx: torch.Tensor, dst_idx: torch.Tensor) -> torch.Tensor:
    """The [2, B] candidate tensor the two row selections were cut from (no copy when they are rows 0 and 1 of one
    contiguous tensor, as in `tedge[0].unsqueeze(0)`, `tedge[1].unsqueeze(0)`), else a fresh stack."""
    b = getattr(src_idx, "_base", None)
    if (b is not None and b is getattr(dst_idx, "_base", None) and b.dim() == 2 and b.shape[0] == 2 and b.is_contiguous()
            and src_idx.numel() == b.shape[1] and dst_idx.numel() == b.shape[1]
            and src_idx.data_ptr() == b.data_ptr() and dst_idx.data_ptr() == b[1].data_ptr()):
        return b
    return torch.stack([_ids(src_idx), _ids(dst_idx)])


class _Deferred:
    kind = "?"

    def _no(self, what: str):
        raise NotImplementedError(f"pygho stand-in: {what} of a deferred {self.kind} expression is outside the drivers' "
                                  "get_cn1_cn2 pattern (NeighborOverlap_large_ppa.py:147-173)")

    def to_torch_sparse_coo(self):
        self._no("to_torch_sparse_coo()")


class RowSelect(_Deferred):
    """adj[ids] : [B, N]"""
    kind = "row selection"

    def __init__(self, adj: SparseTensor, idx: torch.Tensor):
        self.adj, self.idx = adj, idx

    @property
    def shape(self):
        return (self.idx.numel(), self.adj.size(1))

    def to_torch_sparse_coo(self) -> torch.Tensor:
        return self.adj[_ids(self.idx)].to_torch_sparse_coo_tensor()


class TwoHop(_Deferred):
    """adj[ids] @ adj : [B, N], values = number of 2-walks"""
    kind = "row selection times adjacency"

    def __init__(self, sel: RowSelect):
        self.sel = sel

    @property
    def shape(self):
        return self.sel.shape


class LazyCN(_Deferred):
    """Ei (.) Ej  (walk1)  or  Ei (.) (Ej @ adj)  (walk2): what ocn_amd.utils.get_cn1_cn2 returns as a CNBatch."""
    kind = "common-neighbour matrix"

    def __init__(self, adj: SparseTensor, src_idx: torch.Tensor, dst_idx: torch.Tensor, mode: str):
        self.adj, self.src_idx, self.dst_idx, self.mode = adj, src_idx, dst_idx, mode
        self._batch: Optional[CNBatch] = None

    @property
    def shape(self):
        return (self.src_idx.numel(), self.adj.size(1))

    def batch(self) -> CNBatch:
        if self._batch is None:
            self._batch = CNBatch(self.adj, None, _edges(self.src_idx, self.dst_idx), self.mode)
        return self._batch

    def to_torch_sparse_coo(self) -> "LazyCoo":
        return LazyCoo(self)


class LazyCoo:
    """cn.to_torch_sparse_coo(): `.shape`, `.indices()` (unpacks into row, col), `.values()` — still deferred."""

    def __init__(self, cn: LazyCN):
        self.cn = cn

    @property
    def shape(self):
        return self.cn.shape

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def indices(self):
        return LazyVec(self.cn, "row"), LazyVec(self.cn, "col")

    def values(self) -> "LazyVec":
        return LazyVec(self.cn, "val")

    def coalesce(self) -> "LazyCoo":
        return self


class LazyVec:
    """One of the row / col / value vectors of a deferred CN matrix.  ocn_amd.sparse.SparseTensor(row=…, col=…, value=…)
    recognises it (`_ocn_lazy_cn`) and returns the CNBatch handle; touched as a tensor, it materialises."""

    def __init__(self, cn: LazyCN, which: str):
        self._ocn_lazy_cn, self._which = cn, which

    def tensor(self) -> torch.Tensor:
        m = self._ocn_lazy_cn.batch().materialize()
        row, col, val = m.coo()
        return {"row": row, "col": col, "val": val}[self._which]

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.tensor(), name)

    def __len__(self):
        return self.tensor().shape[0]
