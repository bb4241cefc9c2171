"""Adjacency container with the torch_sparse.SparseTensor surface the OCN drivers touch.

The reference keeps adjacencies as ``torch_sparse.SparseTensor`` (int64 row/col) and calls on
it: ``from_edge_index``, ``to_device``, ``to_symmetric``, ``to_torch_sparse_coo_tensor``,
``from_torch_sparse_coo_tensor``, ``sizes``, ``device`` (NeighborOverlap_large.py:51-74,103-119),
``__getitem__``, ``storage.row/col/rowcount``, ``csr``, ``coo``, ``sparse_sizes``,
``fill_value_``, ``coalesce``, ``to_dense``, ``from_dense`` (utils.py:42-44,150-158,256-257,
302-321), ``sum(dim)``, ``mul``, ``size`` (model.py:2261-2276).  This class offers those names
over the layout the HIP kernels want: int64 rowptr + int32 col (columns ascending in a row)
resident in HBM.  Format conversions (sort, unique) are torch ops; arithmetic is in the HIP
library.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import ops


class _Storage:
    def __init__(self, owner: "SparseTensor"):
        self._o = owner

    def row(self) -> Tensor:
        return self._o._row64()

    def col(self) -> Tensor:
        return self._o._col.to(torch.int64)

    def rowptr(self) -> Tensor:
        return self._o._rowptr

    def rowcount(self) -> Tensor:
        return self._o._rowptr[1:] - self._o._rowptr[:-1]

    def value(self) -> Optional[Tensor]:
        return self._o._value

    def has_value(self) -> bool:
        return self._o._value is not None


class SparseTensor:
    def __new__(cls, row=None, rowptr=None, col=None, value=None, sparse_sizes=None, *a, **k):
        # `torch_sparse.SparseTensor(row=row1, col=col1, value=value1, sparse_sizes=(B, N))` at the end of the pygho drivers'
        # get_cn1_cn2 (NeighborOverlap_large_ppa.py:170-171) on the deferred vectors of shims/pygho: the handle itself
        cn = getattr(row, "_ocn_lazy_cn", None)
        if cn is not None:
            if getattr(col, "_ocn_lazy_cn", None) is not cn or (value is not None and getattr(value, "_ocn_lazy_cn", None) is not cn):
                raise NotImplementedError("row / col / value of different deferred common-neighbour matrices")
            if sparse_sizes is not None and tuple(int(v) for v in sparse_sizes) != tuple(cn.shape):
                raise ValueError(f"sparse_sizes {tuple(sparse_sizes)} != {tuple(cn.shape)} of the deferred matrix")
            return cn.batch()
        return super().__new__(cls)

    def __init__(self, row: Optional[Tensor] = None, rowptr: Optional[Tensor] = None,
                 col: Optional[Tensor] = None, value: Optional[Tensor] = None,
                 sparse_sizes: Optional[Tuple[int, int]] = None, is_sorted: bool = False,
                 trust_data: bool = False):
        assert col is not None and (row is not None or rowptr is not None)
        if sparse_sizes is None:
            n_rows = (int(row.max()) + 1 if row.numel() else 0) if rowptr is None else rowptr.numel() - 1
            n_cols = int(col.max()) + 1 if col.numel() else 0
            sparse_sizes = (n_rows, n_cols)
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        if rowptr is None and value is None and not is_sorted and col.is_cuda:
            # the per-batch adjacency of the training loop (NeighborOverlap_large.py:56-59): counting pass + per-row sorts
            # on the device (ocn_coo_to_csr) instead of a comparison sort over the edge list
            rowptr, col = ops.coo_to_csr(row.to(torch.int64), col.to(torch.int64), self._sizes[0], self._sizes[1],
                                         check_range=not trust_data)
        if rowptr is None:
            row = row.to(torch.int64)
            col64 = col.to(torch.int64)
            if not is_sorted:
                perm = torch.argsort(row * max(self._sizes[1], 1) + col64, stable=True)
                row, col64 = row[perm], col64[perm]
                if value is not None:
                    value = value[perm]
            if not trust_data and row.numel():
                if int(row.min()) < 0 or int(row.max()) >= self._sizes[0] or int(col64.min()) < 0 \
                        or int(col64.max()) >= self._sizes[1]:
                    raise IndexError("SparseTensor: index out of range for sparse_sizes")
            cnt = torch.bincount(row, minlength=self._sizes[0])
            rowptr = torch.zeros(self._sizes[0] + 1, dtype=torch.int64, device=col.device)
            torch.cumsum(cnt, 0, out=rowptr[1:])
            col = col64
        self._rowptr = rowptr.to(torch.int64).contiguous()
        self._col_thunk = None
        self._col = col.to(torch.int32).contiguous()
        self._value = None if value is None else value.contiguous()
        self._row_cache: Optional[Tensor] = None
        self._maxdeg: Optional[int] = None
        self._bitmap: Optional[Tensor] = None
        self._nds: Optional[Tensor] = None
        self._ready: dict = {}                 # cache name -> event recorded behind its asynchronous build

    @property
    def storage(self) -> _Storage:
        """``adj.storage.row() / .col() / .rowcount()`` (utils.py:42-44, 150-158).  A fresh view per access: an object that
        held its storage while the storage held it back would be a reference cycle, freed only when Python's cycle collector
        gets round to it — and a training loop's per-batch A² (7 GB of bit rows at the collab shape) must go when its step ends."""
        return _Storage(self)

    # The column ids of a PRODUCT (A @ A) may be deferred: its row pointers and dense bit rows come out of the counting pass, and
    # on a large graph the intersection kernel probes the bit rows only (ocn_hip.h: ocn_cn_flags `bitmapT2`) — the fill pass
    # (2.75 ms and a host sync for the output size per call at the collab shape: an eighth of a training step, which rebuilds
    # A² of the masked graph per batch, NeighborOverlap_large.py:68-74) runs when somebody actually asks for the ids.
    @property
    def _col(self) -> Tensor:
        if self._col_v is None and self._col_thunk is None and getattr(self, "_lazy", None) is not None:
            self._complete()
        if self._col_v is None and self._col_thunk is not None:
            self._col_v, self._col_thunk = self._col_thunk(), None
        return self._col_v

    # One step further for a product formed while autograd records (a training step's per-batch A²): not even the counting pass
    # runs before somebody needs the row pointers — the intersection pass of a large graph asks for the bit rows of its candidates'
    # TARGET rows only (``product_bit_rows(rows)``: ocn_hip.h, ocn_spgemm_bit_rows), two fifths of the rows at the collab shape, and
    # every dense row costs a 29 KiB write.  Anything that treats the product as a matrix (row pointers, ids, nnz, all bit rows)
    # completes it with the ordinary counting pass.  Single-stream use only (the loops that fork streams warm() first = complete).
    @property
    def _rowptr(self) -> Tensor:
        if self._rowptr_v is None and getattr(self, "_lazy", None) is not None:
            self._complete()
        return self._rowptr_v

    @_rowptr.setter
    def _rowptr(self, v) -> None:
        self._rowptr_v = v

    def _complete(self) -> None:
        a_rp, a_col, b_rp, b_col = self._lazy
        self._lazy = self._done = self._bitmap = None               # (the partial bit rows go before the full ones come)
        rowptr, col, bitmap = ops.spgemm_pattern(a_rp, a_col, b_rp, b_col, self._sizes[1], defer_fill=True)
        self._rowptr_v = rowptr
        if callable(col):
            self._col_v, self._col_thunk = None, col
        else:
            self._col_v, self._col_thunk = col, None
        self._bitmap = bitmap
        if bitmap is not None:
            self._published("bitmap")

    @classmethod
    def _lazy_product(cls, a: "SparseTensor", b: "SparseTensor") -> "SparseTensor":
        out = cls.__new__(cls)
        out._sizes = (int(a._sizes[0]), int(b._sizes[1]))
        out._rowptr_v = None
        out._col_v = out._col_thunk = None
        out._value = None
        out._row_cache = out._maxdeg = out._nds = None
        out._ready = {}
        out._lazy = (a._rowptr, a._col, b._rowptr, b._col)
        dev = a._rowptr.device
        out._bitmap = torch.empty(out._sizes[0], (out._sizes[1] + 31) // 32, dtype=torch.int32, device=dev)      # rows on demand
        out._done = torch.zeros(out._sizes[0], dtype=torch.int32, device=dev)
        return out

    @_col.setter
    def _col(self, v) -> None:
        self._col_v = v

    def col_materialized(self) -> bool:
        return self._col_v is not None

    @classmethod
    def _deferred_product(cls, rowptr: Tensor, col_thunk, bitmap: Tensor, sparse_sizes) -> "SparseTensor":
        out = cls.__new__(cls)
        out._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        out._rowptr = rowptr
        out._col_v, out._col_thunk = None, col_thunk
        out._value = None
        out._row_cache = out._maxdeg = out._nds = None
        out._bitmap = bitmap
        out._ready = {}
        return out

    # ---- constructors ------------------------------------------------------------------
    @classmethod
    def from_edge_index(cls, edge_index: Tensor, edge_attr: Optional[Tensor] = None,
                        sparse_sizes: Optional[Tuple[int, int]] = None, is_sorted: bool = False,
                        trust_data: bool = False) -> "SparseTensor":
        return cls(row=edge_index[0], col=edge_index[1], value=edge_attr, sparse_sizes=sparse_sizes,
                   is_sorted=is_sorted, trust_data=trust_data)

    @classmethod
    def from_csr(cls, rowptr: Tensor, col: Tensor, sparse_sizes, value: Optional[Tensor] = None):
        return cls(rowptr=rowptr, col=col, value=value, sparse_sizes=sparse_sizes)

    @classmethod
    def from_dense(cls, mat: Tensor) -> "SparseTensor":
        r, c = mat.nonzero(as_tuple=True)
        return cls(row=r, col=c, value=mat[r, c], sparse_sizes=tuple(mat.shape), is_sorted=True,
                   trust_data=True)

    @classmethod
    def from_torch_sparse_coo_tensor(cls, mat, has_value: bool = True) -> "SparseTensor":
        if isinstance(mat, CooView):
            sp = mat.sp
            if has_value and sp._value is None:
                if mat.is_product:
                    raise NotImplementedError(
                        "walk-count values of A@A are not formed; the reference drops them "
                        "(from_torch_sparse_coo_tensor(spadj @ spadj, False), "
                        "NeighborOverlap_large.py:74,119)")
                return sp.fill_value(1.0)
            return sp if has_value or sp._value is None else sp.set_value(None)
        mat = mat.coalesce()
        r, c = mat.indices()
        return cls(row=r, col=c, value=mat.values() if has_value else None,
                   sparse_sizes=tuple(mat.shape), is_sorted=True, trust_data=True)

    # ---- basic accessors ---------------------------------------------------------------
    def sizes(self) -> List[int]:
        return list(self._sizes)

    def sparse_sizes(self) -> Tuple[int, int]:
        return self._sizes

    def size(self, dim: int) -> int:
        return self._sizes[dim]

    def nnz(self) -> int:
        return int(self._col.numel())

    def device(self):
        return self._bitmap.device if getattr(self, "_lazy", None) is not None else self._rowptr.device

    def has_value(self) -> bool:
        return self._value is not None

    def csr(self):
        return self._rowptr, self._col.to(torch.int64), self._value

    def coo(self):
        return self._row64(), self._col.to(torch.int64), self._value

    def _row64(self) -> Tensor:
        if self._row_cache is None:
            deg = self._rowptr[1:] - self._rowptr[:-1]
            self._row_cache = torch.repeat_interleave(
                torch.arange(self._sizes[0], device=self._col.device), deg, output_size=self.nnz())
        return self._row_cache

    def max_rowcount(self) -> int:
        if self._maxdeg is None:
            self._maxdeg = int((self._rowptr[1:] - self._rowptr[:-1]).max()) if self._sizes[0] else 0
        return self._maxdeg

    # The caches below are built ASYNCHRONOUSLY on whatever stream is current at their first use.  A scoring loop runs its
    # batches on several streams (pipeline.overlapped_steps): the stream that finds the cache filled must not read it
    # before the stream that is still filling it is done, so each cache carries the event recorded behind its build and
    # every later reader's stream waits for it — until the event has completed, after which it is dropped.
    def _published(self, name: str) -> None:
        if self._rowptr.is_cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self._rowptr.device))
            self._ready[name] = ev

    def _await(self, name: str) -> None:
        ev = self._ready.get(name)
        if ev is None:
            return
        if torch.cuda.is_current_stream_capturing():       # (no event query while capturing; the warm-up passes ran before it)
            return
        if ev.query():
            del self._ready[name]
        else:
            torch.cuda.current_stream(self._rowptr.device).wait_event(ev)

    def bit_rows(self) -> Optional[Tensor]:
        """The pattern as dense bit rows (cached, built once by ocn_bitrows_from_csr) when they fit
        ``ops.a1_bitmap_max_bytes``, else None: what the intersection kernel probes instead of searching a row
        (a product of ``matmul`` arrives with its bit rows already)."""
        if getattr(self, "_lazy", None) is not None:        # (a product with rows on demand, asked for ALL its rows)
            self._complete()
        if self._bitmap is None and self._rowptr.is_cuda:
            n, m = self._sizes
            if 0 < n * ((m + 31) // 32) * 4 <= ops.a1_bitmap_max_bytes:
                self._bitmap = ops.bitrows_from_csr(self._rowptr, self._col, m)
                self._published("bitmap")
        else:
            self._await("bitmap")
        return self._bitmap

    def neighbor_degree_sum(self) -> Tensor:
        """Σ_{u∈N(v)} deg(u) per node, cached: lets the walk-count route sweep each candidate edge from
        its cheaper endpoint (ocn_hip.h: ocn_neighbor_degree_sum).  Square adjacencies only."""
        if self._nds is None:
            self._nds = ops.neighbor_degree_sum(self._rowptr, self._col)
            self._published("nds")
        else:
            self._await("nds")
        return self._nds

    def product_bit_rows(self, rows: Optional[Tensor] = None) -> Optional[Tensor]:
        """The bit rows this matrix ARRIVED with (``A @ A``, the block route), never built on demand: A² of a large graph
        only has them when they fit ``ops.a2_bitmap_max_bytes`` at construction.  ``rows`` (int64 ids): the caller will probe
        these rows only — a product whose rows are built on demand (``_lazy_product``) builds the missing ones of them now, on
        the current stream."""
        if getattr(self, "_lazy", None) is not None:
            if rows is None:
                self._complete()
            else:
                a_rp, a_col, b_rp, b_col = self._lazy
                ops.spgemm_bit_rows(a_rp, a_col, b_rp, b_col, self._sizes[1], rows, self._done, self._bitmap)
                return self._bitmap
        if self._bitmap is not None:
            self._await("bitmap")
        return self._bitmap

    def rows_on_demand(self) -> bool:
        """A product none of whose matrix-level data exists yet (see ``_lazy_product``)."""
        return getattr(self, "_lazy", None) is not None

    def warm(self, walk: bool = False) -> None:
        """Build the lazy caches a candidate batch reads (bit rows, longest row, the walk route's degree sums) on the
        CURRENT stream: what a scoring loop calls before it forks its side streams."""
        self.max_rowcount()
        if walk:
            self.neighbor_degree_sum()
        else:
            self.bit_rows()

    # ---- device movement ---------------------------------------------------------------
    def to_device(self, device, non_blocking: bool = False) -> "SparseTensor":
        out = SparseTensor(rowptr=self._rowptr.to(device, non_blocking=non_blocking),
                           col=self._col.to(device, non_blocking=non_blocking),
                           value=None if self._value is None else self._value.to(device, non_blocking=non_blocking),
                           sparse_sizes=self._sizes)
        out._maxdeg = self._maxdeg
        return out

    def to(self, device, non_blocking: bool = False):
        return self.to_device(device, non_blocking)

    def cuda(self):
        return self.to_device("cuda")

    def cpu(self):
        return self.to_device("cpu")

    # ---- structure ops -----------------------------------------------------------------
    def set_value(self, value: Optional[Tensor]) -> "SparseTensor":
        out = SparseTensor(rowptr=self._rowptr, col=self._col, value=value, sparse_sizes=self._sizes)
        out._row_cache, out._maxdeg = self._row_cache, self._maxdeg
        return out

    def fill_value(self, v: float, dtype=torch.float32) -> "SparseTensor":
        return self.set_value(torch.full((self.nnz(),), v, dtype=dtype, device=self._col.device))

    def fill_value_(self, v: float, dtype=torch.float32) -> "SparseTensor":
        self._value = torch.full((self.nnz(),), v, dtype=dtype, device=self._col.device)
        return self

    def coalesce(self, reduce: str = "sum") -> "SparseTensor":
        n = max(self._sizes[1], 1)
        key = self._row64() * n + self._col.to(torch.int64)
        if self._value is None:
            ukey = torch.unique_consecutive(key)
            val = None
        else:
            ukey, inv = torch.unique_consecutive(key, return_inverse=True)
            val = torch.zeros(ukey.numel(), dtype=self._value.dtype, device=key.device).index_add_(0, inv, self._value)
        return SparseTensor(row=torch.div(ukey, n, rounding_mode="floor"), col=ukey % n, value=val,
                            sparse_sizes=self._sizes, is_sorted=True, trust_data=True)

    def to_symmetric(self, reduce: str = "sum") -> "SparseTensor":
        """Pattern of A ∪ Aᵀ, coalesced (values, if any, summed)."""
        r, c = self._row64(), self._col.to(torch.int64)
        if self._value is None and self._col.is_cuda and self._sizes[0] == self._sizes[1]:
            # NeighborOverlap_large.py:63: transposed entries join, duplicates leave — one pass of ocn_coo_to_csr
            rowptr, col = ops.coo_to_csr(r, c, self._sizes[0], self._sizes[1], symmetrize=True, dedupe=True, check_range=False)
            return SparseTensor(rowptr=rowptr, col=col, sparse_sizes=self._sizes)
        v = None if self._value is None else torch.cat([self._value, self._value])
        return SparseTensor(row=torch.cat([r, c]), col=torch.cat([c, r]), value=v,
                            sparse_sizes=self._sizes, trust_data=True).coalesce(reduce)

    def t(self) -> "SparseTensor":
        """Transpose (cached): CSR of Aᵀ with the values carried along — the operator of the SpMM
        backward when the adjacency is not symmetric (DropAdj masks directed entries)."""
        if getattr(self, "_t_cache", None) is None:
            r, c = self._row64(), self._col.to(torch.int64)
            self._t_cache = SparseTensor(row=c, col=r, value=self._value, sparse_sizes=(self._sizes[1], self._sizes[0]),
                                         trust_data=True)
        return self._t_cache

    def __getitem__(self, idx: Tensor) -> "SparseTensor":
        """Row select (utils.py:256-257): row e of the result is row idx[e]."""
        idx = idx.to(torch.int64)
        start = self._rowptr[idx]
        deg = self._rowptr[idx + 1] - start
        optr = torch.zeros(idx.numel() + 1, dtype=torch.int64, device=idx.device)
        torch.cumsum(deg, 0, out=optr[1:])
        total = int(optr[-1])
        row = torch.repeat_interleave(torch.arange(idx.numel(), device=idx.device), deg, output_size=total)
        pos = torch.arange(total, device=idx.device) - optr[:-1][row] + start[row]
        return SparseTensor(rowptr=optr, col=self._col[pos],
                            value=None if self._value is None else self._value[pos],
                            sparse_sizes=(idx.numel(), self._sizes[1]))

    def to_dense(self) -> Tensor:
        d = torch.zeros(self._sizes, device=self._col.device)
        v = self._value if self._value is not None else torch.ones(self.nnz(), device=self._col.device)
        d.index_put_((self._row64(), self._col.to(torch.int64)), v.to(d.dtype), accumulate=True)
        return d

    def sum(self, dim: int) -> Tensor:
        v = self._value if self._value is not None else torch.ones(self.nnz(), device=self._col.device)
        idx = self._col.to(torch.int64) if dim == 0 else self._row64()
        return torch.zeros(self._sizes[1 - dim] if dim in (0, 1) else 0, dtype=v.dtype,
                           device=v.device).index_add_(0, idx, v)

    def mul(self, other: Tensor) -> "SparseTensor":
        """torch_sparse ``SparseTensor.mul(dense)`` with a broadcastable [1, N] or [M, 1] dense operand
        (model.py:2272: ``cn1.mul(inv_col_sum.view(1, -1))``): scales the stored values (1.0 where none)."""
        v = self._value if self._value is not None else torch.ones(self.nnz(), device=self._col.device)
        other = other if torch.is_tensor(other) else torch.as_tensor(other, device=v.device)
        if other.dim() == 2 and other.shape[0] == 1 and other.shape[1] == self._sizes[1]:
            scale = other[0][self._col.to(torch.int64)]
        elif other.dim() == 2 and other.shape[1] == 1 and other.shape[0] == self._sizes[0]:
            scale = other[:, 0][self._row64()]
        elif other.numel() == 1:
            scale = other.reshape(())
        else:
            raise NotImplementedError("SparseTensor.mul: dense operand must broadcast as [1, N], [M, 1] or a scalar")
        out = SparseTensor(rowptr=self._rowptr, col=self._col, value=v * scale.to(v.dtype), sparse_sizes=self._sizes)
        return out

    def __mul__(self, other):
        return self.mul(other)

    def add(self, other: "SparseTensor") -> "SparseTensor":
        """torch_sparse ``SparseTensor + SparseTensor`` (utils.py:318-321): entries concatenated and coalesced by
        sum; value-less operands count as ones."""
        if not isinstance(other, SparseTensor):
            raise NotImplementedError("SparseTensor + dense")
        if other._sizes != self._sizes:
            raise ValueError("SparseTensor.add: sizes differ")
        dev = self._col.device
        va = self._value if self._value is not None else torch.ones(self.nnz(), device=dev)
        vb = other._value if other._value is not None else torch.ones(other.nnz(), device=dev)
        key = torch.cat([self._row64() * self._sizes[1] + self._col.to(torch.int64),
                         other._row64() * self._sizes[1] + other._col.to(torch.int64)])
        uniq, inv = torch.unique(key, return_inverse=True)
        val = torch.zeros(uniq.numel(), dtype=va.dtype, device=dev).index_add_(0, inv, torch.cat([va, vb.to(va.dtype)]))
        return SparseTensor(row=torch.div(uniq, self._sizes[1], rounding_mode="floor"), col=uniq % self._sizes[1], value=val,
                            sparse_sizes=self._sizes, is_sorted=True, trust_data=True)

    def __add__(self, other):
        return self.add(other)

    def to_torch_sparse_coo_tensor(self) -> "CooView":
        return CooView(self)

    def __repr__(self) -> str:
        return f"SparseTensor(sizes={self._sizes}, nnz={self.nnz()}, device={self.device()})"


class CooView:
    """What ``adj.to_torch_sparse_coo_tensor()`` hands back.  The only thing the drivers do with it
    is ``spadj @ spadj`` followed by ``SparseTensor.from_torch_sparse_coo_tensor(.., False)``
    (NeighborOverlap_large.py:68-74,112-119); ``@`` runs the HIP A·B pattern kernels."""

    def __init__(self, sp: SparseTensor, is_product: bool = False):
        self.sp = sp
        self.is_product = is_product

    @property
    def shape(self):
        return tuple(self.sp._sizes)

    def __matmul__(self, other: "CooView") -> "CooView":
        a, b = self.sp, other.sp
        if a._sizes[1] != b._sizes[0]:
            raise ValueError("shape mismatch in sparse @ sparse")
        n, m = a._sizes[0], b._sizes[1]
        if (ops.lazy_product_rows and torch.is_grad_enabled() and a._rowptr.is_cuda and m > ops.small_graph_cols()
                and 0 < n * ((m + 31) // 32) * 4 <= ops.a2_bitmap_max_bytes and m <= ops.spgemm_max_cols()):
            return CooView(SparseTensor._lazy_product(a, b), is_product=True)
        rowptr, col, bitmap = ops.spgemm_pattern(a._rowptr, a._col, b._rowptr, b._col, b._sizes[1], defer_fill=True)
        if callable(col):                      # (bit rows exist: the column ids wait until somebody reads them)
            out = SparseTensor._deferred_product(rowptr, col, bitmap, (a._sizes[0], b._sizes[1]))
        else:
            out = SparseTensor(rowptr=rowptr, col=col, sparse_sizes=(a._sizes[0], b._sizes[1]))
            out._bitmap = bitmap               # dense bit rows of the product, probed by the intersection kernel
        if bitmap is not None:
            out._published("bitmap")
        return CooView(out, is_product=True)

    def to_torch(self) -> Tensor:
        sp = self.sp
        v = sp._value if sp._value is not None else torch.ones(sp.nnz(), device=sp.device())
        return torch.sparse_coo_tensor(torch.stack([sp._row64(), sp._col.to(torch.int64)]), v, sp._sizes)
