"""Drop-in for the hot-path classes of the reference's ``model.py``.

Exports ``predictor_dict`` (``cn5`` = CNLinkPredictorOringin, model.py:2171-2443; ``cn7`` =
CNLinkPredictorbaselearn, model.py:3021-3229), ``convdict`` / ``convdict2`` / ``convdict3``,
``GCN`` / ``GCN2`` / ``GCN3`` (model.py:232-511), ``PureConv*``, ``DropAdj``, ``DropEdge`` with the
reference's constructor signatures and ``state_dict`` key layout, so checkpoints and the unchanged
drivers work.  The sparse arithmetic runs in libocn_hip.so (no torch_sparse / pygho / PyG); in eval the
dense ``Linear`` / ``LayerNorm`` / ``ReLU`` heads run on the bf16x6 MFMA kernel of the same library
(one fused launch ``ocn_heads_fused`` from H = 128 up, else ``_seq_eval`` / ``_heads_grouped``); under autograd the
heads' Linear layers run through ``_LinearFn`` (forward and input gradient on the same kernel).  The ``nn`` modules hold
the parameters.
"""
from __future__ import annotations

from typing import Final, Iterable, Optional

import torch
import torch.nn as nn
from torch import Tensor

import os

from . import ops
from .sparse import SparseTensor
from .utils import fuse


# ------------------------------------------------------------------------------------------
# edge dropout (model.py:198-229) — identity in eval; train-time regulariser
# ------------------------------------------------------------------------------------------
class DropEdge(nn.Module):
    def __init__(self, dp: float = 0.0) -> None:
        super().__init__()
        self.dp = dp

    def forward(self, edge_index: Tensor):
        if self.dp == 0:
            return edge_index
        mask = torch.rand_like(edge_index[0], dtype=torch.float) > self.dp
        return edge_index[:, mask]


class DropAdj(nn.Module):
    doscale: Final[bool]

    def __init__(self, dp: float = 0.0, doscale=True) -> None:
        super().__init__()
        self.dp = dp
        self.register_buffer("ratio", torch.tensor(1 / (1 - dp)))
        self.doscale = doscale

    def forward(self, adj: SparseTensor) -> SparseTensor:
        if self.dp < 1e-6 or not self.training:
            return adj
        row, col, val = adj.coo()
        mask = torch.rand_like(col, dtype=torch.float) > self.dp
        if self.doscale:
            val = (val[mask] * self.ratio) if val is not None else torch.full(
                (int(mask.sum()),), 1 / (1 - self.dp), device=col.device)
        else:
            val = None if val is None else val[mask]
        return SparseTensor(row=row[mask], col=col[mask], value=val, sparse_sizes=adj.sparse_sizes(),
                            is_sorted=True, trust_data=True)


# ------------------------------------------------------------------------------------------
# autograd glue (training drop-in): the HIP kernels are linear in x / h, their transposes are HIP
# kernels too
# ------------------------------------------------------------------------------------------
class _SpmmFn(torch.autograd.Function):
    """y = M x for one of the encoder operators.  The drivers' adjacencies are symmetric
    (``to_symmetric``, NeighborOverlap_large.py:63, ogbdataset.py:45), and so are A, P(A+I)P and
    D^-½(A+I)D^-½: the backward is the same kernel applied to the gradient.  A valued adjacency
    (DropAdj masks directed entries, so it is no longer symmetric) uses its transpose.  ``mean``
    (D⁻¹A) has the transpose A D⁻¹; ``max`` is not differentiated here."""

    @staticmethod
    def forward(ctx, x, adj, kw):
        ctx.adj, ctx.kw = adj, kw
        return ops.spmm_csr(adj._rowptr, adj._col, x, val=adj._value, **kw)

    @staticmethod
    def backward(ctx, g):
        adj, kw = ctx.adj, dict(ctx.kw)
        g = g.contiguous()
        mode = kw.get("mode", "sum")
        if mode == "max":
            raise NotImplementedError("backward of max aggregation")
        if adj._value is not None:
            adj = adj.t()
            if mode == "mean":
                raise NotImplementedError("backward of mean aggregation over a valued adjacency")
        if mode == "mean":
            deg = (adj._rowptr[1:] - adj._rowptr[:-1]).clamp(min=1).to(torch.float32)
            return ops.spmm_csr(adj._rowptr, adj._col, g, pre=1.0 / deg, mode="sum"), None, None
        return ops.spmm_csr(adj._rowptr, adj._col, g, val=adj._value, **kw), None, None


def _spmm(adj: SparseTensor, x: Tensor, **kw) -> Tensor:
    if torch.is_grad_enabled() and x.requires_grad:
        return _SpmmFn.apply(x, adj, kw)
    return ops.spmm_csr(adj._rowptr, adj._col, x, val=adj._value, **kw)


class _PoolFn(torch.autograd.Function):
    """(xcn1, xcn2, x_i ⊙ x_j) = pooling(h): linear in h except for the Hadamard term; the column
    weights depend on the graph and the batch only."""

    @staticmethod
    def forward(ctx, h, st, w):
        ctx.st = st
        ctx.save_for_backward(h, w)
        return st.gather(w, h)

    @staticmethod
    def backward(ctx, g1, g2, g3):
        h, w = ctx.saved_tensors
        return ctx.st.gather_backward(w, h, g1, g2, g3), None, None


class _PoolFn3(torch.autograd.Function):
    """cn6's (xcn1, xcn2, xcn3, x_i ⊙ x_j) = pooling(h): as ``_PoolFn`` with the third pool (model.py:2933)."""

    @staticmethod
    def forward(ctx, h, st, wa, wb, nip):
        ctx.st = st
        ctx.save_for_backward(h, wa, wb, nip)
        return st.gather(wa, wb, nip, h)

    @staticmethod
    def backward(ctx, g1, g2, g3, g4):
        h, wa, wb, nip = ctx.saved_tensors
        return ctx.st.gather_backward(wa, wb, nip, h, g1, g2, g3, g4), None, None, None, None


# Chebyshev bases of cn7 (model.py:2958-3019).  T_k as the reference writes them — the same terms in the same order, each
# power by ``**`` — because the diagonal is fp32 data of the forward: (coefficient, power) pairs added left to right.
_CHEB_TERMS = (
    ((1, 0),),
    ((1, 1),),
    ((2, 2), (-1, 0)),
    ((4, 3), (-3, 1)),
    ((8, 4), (-8, 2), (1, 0)),
    ((16, 5), (-20, 3), (5, 1)),
    ((32, 6), (-48, 4), (18, 2), (-1, 0)),
    ((64, 7), (-112, 5), (56, 3), (-7, 1)),
    ((128, 8), (-256, 6), (160, 4), (-32, 2), (1, 0)),
    ((256, 9), (-576, 7), (432, 5), (-120, 3), (9, 1)),
    ((512, 10), (-1280, 8), (1120, 6), (-400, 4), (50, 2), (-1, 0)),
)
_cheb_cache: dict = {}


def chebyshev_diag(n: int, k: int, device) -> Optional[Tensor]:
    """diag(T_k(linspace(-1, 1, n))) as float32 [n] on ``device`` — what ``evaluate_polynomial(n, k)`` (model.py:2995-3019)
    puts on the diagonal.  Evaluated with torch on the CPU exactly as the reference evaluates it (torch.linspace, ``x ** p``,
    the terms left to right) and uploaded once per (n, k, device); None for k = 0 (T0 = 1: no multiply at all)."""
    if k < 0 or k >= len(_CHEB_TERMS):
        raise ValueError(f"Invalid poly_index. Must be between 0 and {len(_CHEB_TERMS) - 1}.")      # model.py:2997-2998
    if k == 0:
        return None
    key = (int(n), int(k), str(device))
    d = _cheb_cache.get(key)
    if d is None:
        x = torch.linspace(-1, 1, int(n))                                  # model.py:3001 (CPU, as the reference)
        acc = None
        for coef, p in _CHEB_TERMS[k]:
            a = abs(coef)
            term = a if p == 0 else ((x if a == 1 else a * x) if p == 1 else a * x ** p)
            acc = (term if coef > 0 else -term) if acc is None else (acc + term if coef > 0 else acc - term)
        if len(_cheb_cache) > 32:
            _cheb_cache.clear()
        d = _cheb_cache[key] = acc.to(torch.float32).to(device).contiguous()
    return d


# ------------------------------------------------------------------------------------------
# message passing layers
# ------------------------------------------------------------------------------------------
def _values(adj: SparseTensor):
    """fp32 entry values of a valued adjacency (DropAdj's rescale in training) or None."""
    v = adj._value
    return None if v is None else v.to(torch.float32)


def _lin_eval(lin: nn.Linear, x: Tensor, relu: bool = False) -> Tensor:
    """An encoder's dense ``X Wᵀ (+ b)`` (GCNConv.lin, PureConv2.lin, xemb: model.py:58-68, 98-113, 253-262): under
    no_grad on the library's own MFMA Linear kernel (widths 32..256, K a multiple of 16), else the torch module."""
    if (not torch.is_grad_enabled() and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and ops.linear_ok(x.contiguous(), lin.weight)):
        return ops.linear(x.contiguous(), lin.weight, lin.bias, None, relu)
    y = lin(x)
    return torch.relu_(y) if relu else y


class PureConv(nn.Module):
    """model.py:32-55 — parameter-free aggregation; ``gcn`` = n ⊙ (A(n ⊙ x) + n ⊙ x), n = (1+deg)^-½."""
    aggr: Final[str]

    def __init__(self, indim, outdim, aggr="gcn") -> None:
        super().__init__()
        self.aggr = aggr
        if indim == outdim:
            self.lin = nn.Identity()
        else:
            raise NotImplementedError

    def forward(self, x, adj_t: SparseTensor):
        x = self.lin(x).contiguous()
        if self.aggr in ("mean", "max", "sum"):
            return _spmm(adj_t, x, mode=self.aggr)
        if self.aggr == "gcn":
            norm = ops.deg_rsqrt(adj_t._rowptr, 1.0, val=_values(adj_t))
            return _spmm(adj_t, x, pre=norm, post=norm, mode="sum", edge_scale=False, self_mode=1)
        raise ValueError(self.aggr)


class GCNConv(nn.Module):
    """The slice of torch_geometric.nn.GCNConv that ``convdict`` uses (model.py:58-72): a bias-free
    ``lin``, propagation over ``adj_t`` with ``aggr``, then ``+ bias``.  ``normalize`` applies
    D̃^-½(A+I)D̃^-½.  Parameter names (``lin.weight``, ``bias``) follow PyG 2.6.1."""

    def __init__(self, in_channels, out_channels, aggr="add", normalize=True, add_self_loops=True,
                 cached=False, bias=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.aggr = "sum" if aggr == "add" else aggr
        self.normalize, self.add_self_loops = normalize, add_self_loops
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        nn.init.xavier_uniform_(self.lin.weight)            # PyG: glorot weight, zero bias
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def forward(self, x, adj_t: SparseTensor):
        x = _lin_eval(self.lin, x).contiguous()
        if self.normalize:
            dinv = ops.deg_rsqrt(adj_t._rowptr, 1.0, val=_values(adj_t))   # degree of A + I (A has no self loops)
            out = _spmm(adj_t, x, pre=dinv, mode="sum", edge_scale=True,
                        self_mode=2 if self.add_self_loops else 0)
        else:
            out = _spmm(adj_t, x, mode=self.aggr)
        return out if self.bias is None else out + self.bias


convdict = {
    "gcn": GCNConv,
    "gcn_cached": lambda indim, outdim: GCNConv(indim, outdim, cached=True),
    "sage": lambda indim, outdim: GCNConv(indim, outdim, aggr="mean", normalize=False, add_self_loops=False),
    "gin": lambda indim, outdim: GCNConv(indim, outdim, aggr="sum", normalize=False, add_self_loops=False),
    "max": lambda indim, outdim: GCNConv(indim, outdim, aggr="max", normalize=False, add_self_loops=False),
    "puremax": lambda indim, outdim: PureConv(indim, outdim, aggr="max"),
    "puresum": lambda indim, outdim: PureConv(indim, outdim, aggr="sum"),
    "puremean": lambda indim, outdim: PureConv(indim, outdim, aggr="mean"),
    "puregcn": lambda indim, outdim: PureConv(indim, outdim, aggr="gcn"),
    "none": None,
}


class PureConv2(nn.Module):
    """model.py:85-113 (and PureConv3, :115-142): aggregate first — ``gcn`` = (A ⊙ n nᵀ) x with no
    self term — then optional ``Linear(no bias) + ReLU``."""
    aggr: Final[str]

    def __init__(self, indim, outdim, aggr="gcn", use_lin=False) -> None:
        super().__init__()
        self.aggr = aggr
        if not use_lin:
            if indim == outdim:
                self.lin = nn.Identity()
            else:
                raise NotImplementedError
        else:
            self.lin = nn.Sequential(nn.Linear(indim, outdim, bias=False), nn.ReLU(inplace=True))

    def forward(self, x, adj_t: SparseTensor):
        x = x.contiguous()
        if self.aggr in ("mean", "max", "sum"):
            x = _spmm(adj_t, x, mode=self.aggr)
        elif self.aggr == "gcn":
            norm = ops.deg_rsqrt(adj_t._rowptr, 1.0, val=_values(adj_t))
            x = _spmm(adj_t, x, pre=norm, mode="sum", edge_scale=True)
        if isinstance(self.lin, nn.Sequential) and not self.training:
            return _lin_eval(self.lin[0], x, relu=True)                  # Linear(no bias) + ReLU in one launch
        return self.lin(x)


class PureConv3(PureConv2):
    pass


def _convdict23(cls):
    return {
        "gcn": lambda indim, outdim: cls(indim, outdim, aggr="gcn", use_lin=True),
        "gcn_cached": lambda indim, outdim: cls(indim, outdim, aggr="gcn", use_lin=True),
        "sage": lambda indim, outdim: cls(indim, outdim, aggr="mean", use_lin=True),
        "gin": lambda indim, outdim: cls(indim, outdim, aggr="sum", use_lin=True),
        "max": lambda indim, outdim: cls(indim, outdim, aggr="max", use_lin=True),
        "puremax": lambda indim, outdim: cls(indim, outdim, aggr="max"),
        "puresum": lambda indim, outdim: cls(indim, outdim, aggr="sum"),
        "puremean": lambda indim, outdim: cls(indim, outdim, aggr="mean"),
        "puregcn": lambda indim, outdim: cls(indim, outdim, aggr="gcn"),
        "none": None,
    }


convdict2 = _convdict23(PureConv2)
convdict3 = _convdict23(PureConv3)


# ------------------------------------------------------------------------------------------
# encoders
# ------------------------------------------------------------------------------------------
class _Encoder(nn.Module):
    """Shared body of GCN / GCN2 / GCN3 (model.py:232-511): input embedding, L conv layers with
    their LayerNorm/Dropout/ReLU tails, residual, JumpingKnowledge with raw weights."""
    _convs = convdict
    _use_adjdrop = True

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, ln=False,
                 res=False, max_x=-1, conv_fn="gcn", jk=False, edrop=0.0, xdropout=0.0,
                 taildropout=0.0, noinputlin=False):
        super().__init__()
        self.adjdrop = DropAdj(edrop)
        if max_x >= 0:
            emb = nn.Embedding(max_x + 1, hidden_channels)
            nn.init.orthogonal_(emb.weight)
            self.xemb = nn.Sequential(emb, nn.Dropout(dropout))
            in_channels = hidden_channels
        else:
            self.xemb = nn.Sequential(nn.Dropout(xdropout))
            if not noinputlin and ("pure" in conv_fn or num_layers == 0):
                self.xemb.append(nn.Linear(in_channels, hidden_channels))
                self.xemb.append(nn.Dropout(dropout, inplace=True) if dropout > 1e-6 else nn.Identity())
        self.res = res
        self.jk = jk
        if jk:
            self.register_parameter("jkparams", nn.Parameter(torch.randn((num_layers,))))
        if num_layers == 0 or conv_fn == "none":
            self.jk = False
            return
        make = self._convs[conv_fn]
        norm = (lambda dim: nn.LayerNorm(dim)) if ln else (lambda dim: nn.Identity())
        if num_layers == 1:
            hidden_channels = out_channels
        self.convs = nn.ModuleList()
        self.lins = nn.ModuleList()
        if "pure" in conv_fn:
            self.convs.append(make(hidden_channels, hidden_channels))
            for _ in range(num_layers - 1):
                self.lins.append(nn.Identity())
                self.convs.append(make(hidden_channels, hidden_channels))
            self.lins.append(nn.Dropout(taildropout, True))
        else:
            self.convs.append(make(in_channels, hidden_channels))
            self.lins.append(nn.Sequential(norm(hidden_channels), nn.Dropout(dropout, True), nn.ReLU(True)))
            for i in range(num_layers - 1):
                last = i == num_layers - 2
                self.convs.append(make(hidden_channels, hidden_channels if last else out_channels))
                if not last:
                    self.lins.append(nn.Sequential(norm(out_channels), nn.Dropout(dropout, True), nn.ReLU(True)))
                else:
                    self.lins.append(nn.Identity())

    def forward(self, x, adj_t):
        if (not self.training and not torch.is_grad_enabled() and torch.is_tensor(x) and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 2):
            x = _seq_eval(self.xemb, x)                                  # eval: Dropouts vanish, the input Linear on the MFMA kernel
        else:
            x = self.xemb(x)
        jkx = []
        for i, conv in enumerate(self.convs):
            a = self.adjdrop(adj_t) if self._use_adjdrop else adj_t
            x1 = conv(x, a)
            if (isinstance(self.lins[i], nn.Sequential) and not self.training and not torch.is_grad_enabled()
                    and x1.is_cuda and x1.dim() == 2 and x1.dtype == torch.float32 and x1.is_contiguous()):
                # eval: LayerNorm + ReLU of a layer in one HIP pass (torch's LayerNorm kernel runs at 0.26 TB/s on
                # the 32-wide rows of the citation2 encoder: 2.9 ms per layer against 0.2 ms)
                x1 = _seq_eval(self.lins[i], x1)
            elif (isinstance(self.lins[i], nn.Sequential) and torch.is_grad_enabled() and x1.is_cuda and x1.dim() == 2
                  and x1.dtype == torch.float32 and ops.train_tails):
                x1 = _seq_train(self.lins[i], x1.contiguous())        # autograd on: the layer's LayerNorm -> Dropout -> ReLU as one launch each way
            else:
                x1 = self.lins[i](x1)
            x = x1 + x if (self.res and x1.shape[-1] == x.shape[-1]) else x1
            if self.jk:
                jkx.append(x)
        if self.jk:
            x = torch.sum(torch.stack(jkx, dim=0) * self.jkparams.reshape(-1, 1, 1), dim=0)
        return x


class GCN(_Encoder):
    """model.py:232-323."""


class GCN2(_Encoder):
    """model.py:326-417 — convdict2, no adjacency dropout inside the layer loop."""
    _convs = convdict2
    _use_adjdrop = False


class GCN3(_Encoder):
    """model.py:420-511."""
    _convs = convdict3
    _use_adjdrop = False


# ------------------------------------------------------------------------------------------
# predictors
# ------------------------------------------------------------------------------------------
def _stages(seq: nn.Sequential, H: int):
    """Parse an eval-mode head into fused stages [(Linear, LayerNorm | None, relu)], or None when a
    module does not fit the ``Linear [-> LayerNorm] [-> ReLU]`` pattern at width H."""
    mods = [m for m in seq if not isinstance(m, (nn.Dropout, nn.Identity))]
    out, i = [], 0
    while i < len(mods):
        m = mods[i]
        if not (isinstance(m, nn.Linear) and m.in_features == H and m.out_features == H and m.bias is not None):
            return None
        ln, relu, j = None, False, i + 1
        if j < len(mods) and isinstance(mods[j], nn.LayerNorm):
            if not (mods[j].elementwise_affine and tuple(mods[j].normalized_shape) == (H,)):
                return None
            ln, j = mods[j], j + 1
        if j < len(mods) and isinstance(mods[j], nn.ReLU):
            relu, j = True, j + 1
        out.append((m, ln, relu))
        i = j
    return out


def _grp(x, st, y, **extra):
    lin, ln, relu = st
    g = dict(x=x, weight=lin.weight, bias=lin.bias, relu=relu, y=y,
             ln=None if ln is None else (ln.weight, ln.bias, ln.eps))
    g.update(extra)
    return g


def _seq_eval(seq: nn.Sequential, x: Tensor, y_row_map: Optional[Tensor] = None) -> Tensor:
    """Eval-mode walk of one of the predictor's ``nn.Sequential`` heads.  Dropout/Identity vanish;
    ``Linear [-> LayerNorm] [-> ReLU] [-> Linear(H, 1)]`` runs as one bf16x6 MFMA kernel with the
    tail fused into its epilogue; anything that does not fit falls back to the module itself (the
    remaining ``LayerNorm -> ReLU`` pairs still as one HIP pass)."""
    mods = [m for m in seq if not isinstance(m, (nn.Dropout, nn.Identity))]
    fresh, i = False, 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear) and ops.linear_ok(x, m.weight):
            j, ln, relu, dot = i + 1, None, False, None
            if (j < len(mods) and isinstance(mods[j], nn.LayerNorm) and mods[j].elementwise_affine
                    and tuple(mods[j].normalized_shape) == (m.out_features,)):
                ln = (mods[j].weight, mods[j].bias, mods[j].eps)
                j += 1
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                relu = True
                j += 1
            if (j == len(mods) - 1 and isinstance(mods[j], nn.Linear) and mods[j].out_features == 1
                    and mods[j].in_features == m.out_features):
                dot = (mods[j].weight, mods[j].bias)
                j += 1
            x = ops.linear(x, m.weight, m.bias, ln, relu, dot, y_row_map=y_row_map if dot is not None else None)
            i = j
        elif (isinstance(m, nn.LayerNorm) and m.elementwise_affine and x.dim() == 2
                and x.shape[-1] in ops.LN_WIDTHS and x.is_contiguous()):
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            x = ops.rows_ln_relu(x, m.weight, m.bias, m.eps, relu, inplace=fresh)
            i += 2 if relu else 1
        elif isinstance(m, nn.ReLU):
            x = torch.relu_(x) if fresh else torch.relu(x)
            i += 1
        else:
            x = m(x)
            i += 1
        fresh = True
    return x


class _LinearFn(torch.autograd.Function):
    """y = x @ W^T + b with autograd on (training): the forward and the input gradient run on the library's bf16x6
    MFMA Linear kernel; the weight gradient dY^T @ X and the bias gradient (reductions over the whole batch) on
    ``ocn_wgrad`` (split over the batch, partial sums added in a fixed order).  The nn.Linear module still owns the
    parameters (state_dict keys unchanged)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return ops.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            gx = ops.linear_t(gy, weight) if ops.linear_ok_t(gy, weight) else gy @ weight
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            gw, gb = ops.wgrad(gy, x, with_bias=want_b)
        elif want_b:
            gb = gy.sum(0)
        return gx, gw, gb


class _TailFn(torch.autograd.Function):
    """y = ReLU?(Dropout_p(LayerNorm?(x))) — the tail between two Linear layers of the heads (model.py:2203-2235) — as one
    launch each way (ocn_ln_drop_relu_forward / _backward).  The dropout decisions come from a counter-based hash of a seed
    drawn from torch's CPU generator (``torch.manual_seed`` makes a run repeatable; no host sync) and are recomputed in the
    backward; they are this library's random stream, not torch's Philox."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, p, relu):
        seed = int(torch.randint(0, 1 << 62, (1,)).item()) if p > 0.0 else 0
        y, stats = ops.ln_drop_relu_forward(x, gamma, beta, eps, p, seed, relu)
        ctx.save_for_backward(x, y, stats, gamma)
        ctx.cfg = (p, seed, relu)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, stats, gamma = ctx.saved_tensors
        p, seed, relu = ctx.cfg
        dx, dg, db = ops.ln_drop_relu_backward(g, x, y, stats, gamma, p, seed, relu)
        return dx, dg, db, None, None, None


class _MixFn(torch.autograd.Function):
    """z = coef[0] x1 + coef[1] x2 + coef[2] x3 — the branch mix of model.py:2436 / 3222 with coef = [σ(α0), σ(α0)σ(α1), β] —
    as one launch forward (ocn_combine3) and two backward (ocn_mix3_backward: the three input gradients in one pass, the three
    dot products <g, x_k> deterministic); the coefficients' own graph (sigmoid, cumprod) stays torch's."""

    @staticmethod
    def forward(ctx, coef, x1, x2, x3):
        coef = coef.contiguous()
        ctx.save_for_backward(coef, x1, x2, x3)
        return ops.combine3(coef, x1, x2, x3)

    @staticmethod
    def backward(ctx, g):
        coef, x1, x2, x3 = ctx.saved_tensors
        d1, d2, d3, dc = ops.mix3_backward(coef, g, x1, x2, x3)
        return dc, d1, d2, d3


def _mix(alpha: Tensor, beta: Tensor, x1: Tensor, x2: Tensor, x3: Tensor) -> Tensor:
    """alpha[0] * x1 + alpha[1] * x2 + beta * x3 (alpha = cumprod of sigmoids, already formed)."""
    if (ops.train_tails and x1.is_cuda and x1.dtype == torch.float32 and x1.shape == x2.shape == x3.shape and x1.numel() % 4 == 0
            and x1.is_contiguous() and x2.is_contiguous() and x3.is_contiguous()):
        return _MixFn.apply(torch.cat([alpha[:2], beta.reshape(1)]), x1, x2, x3)
    return alpha[0] * x1 + alpha[1] * x2 + beta * x3


def _seq_train(seq: nn.Sequential, x: Tensor) -> Tensor:
    """Autograd-mode walk of one of the predictor's heads: every ``nn.Linear`` the MFMA kernel takes runs through
    ``_LinearFn``; a ``[LayerNorm] [Dropout] [ReLU]`` run between them through ``_TailFn`` (one launch each way); anything
    else stays the torch module."""
    mods = [m for m in seq if not isinstance(m, nn.Identity)]
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Linear) and x.dim() == 2 and x.is_contiguous() and ops.linear_ok(x, m.weight):
            x = _LinearFn.apply(x, m.weight, m.bias)
            i += 1
            continue
        if (ops.train_tails and isinstance(m, (nn.LayerNorm, nn.Dropout, nn.ReLU)) and x.is_cuda and x.dim() == 2
                and x.dtype == torch.float32 and x.shape[1] in ops.LN_WIDTHS):
            j, ln, p, relu = i, None, 0.0, False
            if isinstance(mods[j], nn.LayerNorm) and mods[j].elementwise_affine and tuple(mods[j].normalized_shape) == (x.shape[1],):
                ln, j = mods[j], j + 1
            if j < len(mods) and isinstance(mods[j], nn.Dropout):
                p, j = (float(mods[j].p) if mods[j].training else 0.0), j + 1
            if j < len(mods) and isinstance(mods[j], nn.ReLU):
                relu, j = True, j + 1
            if j > i and p < 1.0:
                x = _TailFn.apply(x.contiguous(), None if ln is None else ln.weight, None if ln is None else ln.bias,
                                  0.0 if ln is None else ln.eps, p, relu)
                i = j
                continue
        x = m(x)
        i += 1
    return x


class _CNPredictorBase(nn.Module):
    """Parameters and MLP heads shared by cn5 and cn7 (model.py:2173-2239 ≡ 3023-3089).  The
    ``nn.Sequential`` layouts are part of the checkpoint contract (state_dict keys)."""
    cndeg: Final[int]
    _xcn2_on_union = True          # pooled xcn2 lives on cn1 ∪ cn2 (cn5); cn7 overrides: raw cn2 only
    _weights_need_args = False     # cn7: its column weights read ``args.sum``

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, edrop=0.0,
                 ln=False, cndeg=-1, use_xlin=False, tailact=False, twolayerlin=False, beta=1.0):
        super().__init__()
        self.register_parameter("beta", nn.Parameter(beta * torch.ones((1))))
        self.dropadj = DropAdj(edrop)
        H, p = hidden_channels, dropout
        norm = (lambda: nn.LayerNorm(H)) if ln else (lambda: nn.Identity())
        drop = lambda: nn.Dropout(p, inplace=True)
        relu = lambda: nn.ReLU(inplace=True)

        def pooled_head(tail_identity=False):
            return nn.Sequential(nn.Linear(in_channels, H), drop(), relu(), nn.Linear(H, H), norm(),
                                 drop(), relu(), nn.Identity() if tail_identity else nn.Linear(H, H))

        self.xlin = nn.Sequential(nn.Linear(H, H), drop(), relu(), nn.Linear(H, H), norm(), drop(),
                                  relu()) if use_xlin else (lambda x: 0)
        self.xcnlin = pooled_head(tail_identity=tailact)     # allocated, unused by cn5/cn7 forward
        self.xcn1lin = pooled_head()
        self.xcn2lin = pooled_head()
        self.xcn4lin = pooled_head()                         # allocated, unused
        self.xijlin = nn.Sequential(nn.Linear(in_channels, H), norm(), drop(), relu(),
                                    nn.Identity() if tailact else nn.Linear(H, H))
        two = twolayerlin
        self.lin = nn.Sequential(nn.Linear(H, H), norm(), drop(), relu(),
                                 nn.Linear(H, H) if two else nn.Identity(),
                                 norm() if two else nn.Identity(),
                                 drop() if two else nn.Identity(),
                                 relu() if two else nn.Identity(),
                                 nn.Linear(H, out_channels))
        self.cndeg = cndeg
        self.register_parameter("alpha", nn.Parameter(torch.ones((3))))
        self.register_buffer("innerprod", torch.tensor([0.0]))
        self.n = 0
        self._shard_group, self._sharded = None, False
        self._ws = {}                          # scratch reused from batch to batch on the no-grad path

    def set_edge_sharding(self, group=None, enabled: bool = True) -> None:
        """The candidate batch handed to ``forward`` is one rank's slice of a global batch: sum the
        column histograms over ``group`` (RCCL all-reduce) before normalising (ocn_amd.dist)."""
        self._shard_group, self._sharded = group, enabled

    def _exchange(self, st, x=None):
        """Edge-sharded batches: sum the column histograms over the ranks.  The collective is started, the class
        ordering of the batch rows (which reads the per-row counts only) is enqueued beside it, then the current stream
        waits for the sum."""
        if self._sharded:
            from .dist import allreduce_hist_finish, allreduce_hist_start
            handle = allreduce_hist_start(st.hist, self._shard_group, valued=st.walk, slice_edges=st.B)
            st.sharded, st.shard_group = True, self._shard_group
            if x is not None:
                self._class_order(st, x)
            allreduce_hist_finish(handle)
            ops._mark("allreduce_hist")
        return st

    # ---- two-phase scoring (eval, no grad) for loops that keep two batches in flight ---------------------------
    # begin(t + 1) is enqueued before finish(t): in the edge-sharded mode the histogram all-reduce of batch t then
    # runs beside the intersection pass of batch t + 1 instead of stalling the stream (ocn_amd/dist.py; bench.py).
    def begin(self, x, adj, cn1, cn2, tar_ei, slot: int = 0, args=None):
        """Phase A: the intersection pass of one batch (scratch set ``slot`` mod ``ops.overlap_depth`` — a batch in phase A
        must not share buffers with the ones still in flight) and, sharded, the START of the histogram sum.  Returns a token."""
        if self.training or torch.is_grad_enabled():
            raise RuntimeError("begin / finish is the no-grad scoring path: call .eval() under torch.no_grad()")
        n_sets = max(2, int(ops.overlap_depth), int(ops.overlap_depth_small))
        if len(getattr(self, "_ws_slots", ())) != n_sets:
            self._ws_slots = [dict() for _ in range(n_sets)]
        st = fuse(cn1, cn2, tar_ei, self._ws_slots[slot % n_sets], adj=adj)
        handle = None
        if self._sharded:
            from .dist import allreduce_hist_start
            st.sharded, st.shard_group = True, self._shard_group
            if ops.shard_reduce_in_finish if ops.shard_reduce_in_finish is not None else ops._overlap_active:
                handle = "late"                    # (started by finish(), on ITS stream: see there)
            else:
                # the whole interleaved buffer, no copy-out / copy-back of the packed word: the collective is hidden behind the
                # next batch's intersection pass, its extra bytes are free, the two copies were not
                handle = allreduce_hist_start(st.hist, self._shard_group, valued=True, slice_edges=st.B)
        # Unsharded, the batch's histogram is complete when the intersection pass ends: the column weights — and, for a
        # trained cn5 / cn6 (innerprod != 0), the order-exact column sums in front of them, 0.2 ms of latency-bound chains at
        # the collab shape — belong to phase A, which the scoring loops run beside the previous batch's pooling and heads.
        # (cn7's weights need ``args.sum``: pass ``args``.)
        w = None
        if handle is None and not self._sharded:
            if args is not None or not self._weights_need_args:
                w = self._weights(st, args)
            if ops.phase_a_extras:
                self._class_order(st, x)           # reads the per-row counts only: off the critical phase too
                st.prepare_schedule(x.shape[1])    # ... as does the pooling's visiting order (group costs of the intersection pass)
                if ops.phase_a_pool and w is not None:
                    # ... and the pooling itself: it needs the weights and the embeddings only.  Phase B is then the heads alone
                    # (the caller's stream carried pooling + heads, 0.32 ms of kernels back to back, beside side streams with 0.2 ms
                    # each: the pipeline's stages are better matched this way)
                    st.pooled = self._pool(st, w, x)
        return st, handle, w

    def finish(self, x, token, args=None):
        """Phase B: class order (beside the collective), wait for the histogram sum, weights (unless phase A had them),
        pooling, heads."""
        st, handle, w_early = token
        ops._mark("begin")                     # (stage timers: this phase may run on another stream than phase A did)
        if isinstance(handle, str):            # "late": the collective starts here, the class ordering runs beside it, and the
            from .dist import allreduce_hist_start      # next batch's phase A (another stream) fills the rest of the wait
            handle = allreduce_hist_start(st.hist, self._shard_group, valued=True, slice_edges=st.B)
        if not getattr(st, "_cls_decided", False):
            self._class_order(st, x)
        if handle is not None:
            from .dist import allreduce_hist_finish
            allreduce_hist_finish(handle)
            ops._mark("allreduce_hist")
        pooled = getattr(st, "pooled", None)
        if pooled is None:
            w = w_early if w_early is not None else self._weights(st, args)
            pooled = self._pool(st, w, x)
        xcn1, xcn2, xij = pooled
        return self._heads(x, xcn1, xcn2, xij, getattr(st, "cls", None))

    def check_errors(self) -> None:
        """Read the STICKY status words of this predictor's scratch sets (one host sync) and raise for what the batches since
        the last check left there: a flag buffer that was too small, offsets from a scan whose workspace was not zero
        (ocn_hip.h: OCN_ST_CAP / OCN_ST_SCAN).  The scoring loops call it once per split; a driver that calls ``forward`` per
        batch can call it whenever it reads scores anyway."""
        words = [t for d in ([self._ws] + list(getattr(self, "_ws_slots", ()))) for k, t in d.items() if k[0] == "status"]
        if not words:
            return
        bits = 0
        for v in torch.stack([t[3] for t in words]).tolist():
            bits |= int(v)
        if bits:
            for t in words:
                t[3:].zero_()
            raise RuntimeError(ops.status_message(bits))

    def _class_order(self, st, x) -> None:
        """Class-major rows (candidates without cn1 / cn2 entries end up in contiguous ranges the heads skip) where
        that pays; sets ``st.cls`` (None = batch order)."""
        st.cls = None
        st._cls_decided = True
        if (ops.skip_zero_rows and not self.training and not torch.is_grad_enabled() and st.B >= ops.skip_zero_min_batch
                and st.cnt2 is not None and (self._fused_plan(x.shape[1]) is not None or self._heads_plan(x.shape[1]) is not None)
                and self._skip_worthwhile()):
            st.cls = ops.class_order(st.cnt1, st.cnt2, st.order, st.ws)
            self._skip_probe(st.cls[2], st.B)

    def _scratch(self, x):
        """Reuse scratch across batches only where nothing outlives the call: no autograd graph (it
        would hold the flag / weight buffers for the backward)."""
        return None if (torch.is_grad_enabled() and (x.requires_grad or self.training)) else self._ws

    def _pool(self, st, w, x):
        x = x.contiguous()
        if torch.is_grad_enabled() and x.requires_grad:
            return _PoolFn.apply(x, st, w)
        if not getattr(st, "_cls_decided", False):
            self._class_order(st, x)
        rs = self._rowsum(st, x)
        if st.cls is not None:
            # (the pooling keeps its own source-sorted, XCD-balanced processing order and only WRITES to the
            # class-major rows: processed class-major, the XCDs holding the heavy classes ran 50 % longer)
            return st.gather(w, x, out_row=st.cls[1], rowsum=rs)
        return st.gather(w, x, rowsum=rs)

    def _rowsum(self, st, x):
        """None here: cn5's cn2 weights depend on the batch and on the entry's cn1 flag.  cn7 overrides."""
        return None

    # Whether skipping pays is a property of the data (a dense graph such as ddi has common neighbours for
    # every candidate: the re-ordering then only costs).  The class boundaries of a batch are copied to pinned
    # host memory WITHOUT waiting; whenever a copy has landed by the time a later batch arrives, its skippable
    # share decides: below ``ops.skip_zero_min_share`` the next ``ops.skip_zero_backoff`` batches run plain.
    def _skip_worthwhile(self) -> bool:
        pr = getattr(self, "_skip_state", None)
        if pr is None:
            return True
        if torch.cuda.is_current_stream_capturing():                           # no event query while capturing:
            return pr["off"] == 0                                                # replay what was last decided
        if pr["pending"] is not None and pr["pending"].query():
            r = pr["host"]
            B = max(int(pr["B"]), 1)
            cn1, any_, none = int(r[ops.R_CN1][1]), int(r[ops.R_ANY][1]), B - int(r[ops.R_ANY][1])
            pr["share"] = ((B - cn1) + 2 * none) / (3.0 * B)          # rows skipped in branch a, branch b (cn5) and the mix
            pr["pending"] = None
            if pr["share"] < ops.skip_zero_min_share:
                pr["off"] = ops.skip_zero_backoff
        if pr["off"] > 0:
            pr["off"] -= 1
            return False
        return True

    def _skip_probe(self, ranges: Tensor, B: int) -> None:
        pr = getattr(self, "_skip_state", None)
        if pr is None:
            pr = self._skip_state = dict(host=torch.empty((ops.CLASS_RANGES, 2), dtype=torch.int64).pin_memory(),
                                         pending=None, B=0, off=0, share=1.0)
        pr["calls"] = pr.get("calls", 0) + 1
        if torch.cuda.is_current_stream_capturing():                     # (a graph replays whatever was decided at capture)
            return
        if pr["pending"] is None and pr["calls"] % 8 == 1:               # a 112-byte copy every 8th batch
            pr["host"].copy_(ranges, non_blocking=True)
            pr["B"] = B
            pr["pending"] = torch.cuda.current_stream(ranges.device).record_event()

    def _heads_plan(self, H: int):
        """Parsed stages of the four heads when the zero-row-skipping evaluation applies (3-layer pooled
        heads ending in a plain Linear, 1- or 2-layer xijlin, `lin` ending in the fused Linear(H, 1) dot),
        else None.  Cached per module structure."""
        key = (H, ops.fast_linear)
        if getattr(self, "_plan_key", None) != key:
            plan = None
            if H in ops.LINEAR_WIDTHS and ops.fast_linear:
                sa, sb, sx = _stages(self.xcn1lin, H), _stages(self.xcn2lin, H), _stages(self.xijlin, H)
                mods = [m for m in self.lin if not isinstance(m, (nn.Dropout, nn.Identity))]
                tail_ok = (len(mods) >= 2 and isinstance(mods[-1], nn.Linear) and mods[-1].out_features == 1
                           and mods[-1].in_features == H and _stages(nn.Sequential(*mods[:-1]), H) is not None)
                if (sa is not None and sb is not None and sx is not None and len(sa) == 3 and len(sb) == 3
                        and len(sx) in (1, 2) and sa[2][1] is None and not sa[2][2] and sb[2][1] is None
                        and not sb[2][2] and tail_ok):
                    plan = (sa, sb, sx)
            self._plan, self._plan_key = plan, key
        return self._plan

    def innerprod1(self, st):
        """model.py:2241-2250: in training the running mean of Σ(cn2 ⊙ ncn1) is updated and used;
        in eval the stored buffer is used as is."""
        if self.training:
            with torch.no_grad():
                ip = st.cn5_batch_innerprod()
                self.n += 1
                beta = self.n ** -1
                self.innerprod *= (1 - beta)
                self.innerprod += beta * ip
        return self.innerprod

    def _drop_caches(self) -> None:
        self._coef_key = self._mixw_key = self._zc_key = self._plan_key = self._fplan_key = self._fpack_key = None
        self._zc_params = self._fp_params = None
        for p in self.parameters():
            ops._panels.pop(id(p), None)

    def _apply(self, fn, *a, **kw):
        """.to() / .cuda() / .float(): parameters are re-created, so every cached derivative goes."""
        out = super()._apply(fn, *a, **kw)
        self._drop_caches()
        return out

    def train(self, mode: bool = True):
        """Mode switches (the drivers call .train() / .eval() around every pass) also drop the cached
        eval-path constants, so edits that bypass the version counters (``param.data``) cannot go stale
        across a training pass."""
        self._drop_caches()
        return super().train(mode)

    def _mix_coef(self) -> Tensor:
        """[σ(α0), σ(α0)σ(α1), β] on the device, recomputed only when alpha / beta change
        (their ``_version`` / storage): three tiny launches less per candidate batch."""
        key = (self.alpha.data_ptr(), self.alpha._version, self.beta.data_ptr(), self.beta._version)
        if getattr(self, "_coef_key", None) != key:
            alpha = torch.sigmoid(self.alpha.detach()).cumprod(-1)
            self._coef = torch.cat([alpha[:2], self.beta.detach()]).contiguous()
            self._coef_key = key
        return self._coef

    def _mix_weight(self, la: nn.Linear, lb: nn.Linear):
        """[σ(α0)·W3a | σ(α0)σ(α1)·W3b] and the matching bias: the branch mix of model.py:2436 folded
        into one K = 2H Linear over the concatenated second-layer outputs.  Cached like the panels."""
        key = (self._mix_coef().data_ptr(), self._coef_key, la.weight.data_ptr(), la.weight._version,
               lb.weight.data_ptr(), lb.weight._version, la.bias._version, lb.bias._version)
        if getattr(self, "_mixw_key", None) != key:
            c = self._mix_coef()
            with torch.no_grad():
                self._mixw = torch.cat([c[0] * la.weight, c[1] * lb.weight], dim=1).contiguous()
                self._mixb = (c[0] * la.bias + c[1] * lb.bias).contiguous()
            self._mixw_key = key
        return self._mixw, self._mixb

    def _zero_consts(self, sa, sb, H: int, dev):
        """What a candidate with an all-zero pooled xcn1 (xcn2) contributes: the activations after the second
        layer of xcn1lin (xcn2lin) on a zero row, and the mix layer of the two — computed by the SAME
        kernels on a one-row zero input, so a skipped row gets bit for bit what it would have computed.
        Cached on the parameter versions."""
        plist = getattr(self, "_zc_params", None)
        if plist is None:             # (walking the module tree every batch cost 60 us of host time)
            plist = self._zc_params = [p for seq in (self.xcn1lin, self.xcn2lin) for p in seq.parameters()]
        self._mix_coef()
        key = (tuple(p._version for p in plist), self._coef_key)
        if getattr(self, "_zc_key", None) != key:
            z1 = torch.zeros(1, H, device=dev)
            t = torch.empty(2, 1, H, device=dev)
            c2 = torch.empty(1, 2 * H, device=dev)
            ops.linear_grouped([_grp(z1, sa[0], t[0]), _grp(z1, sb[0], t[1])], H, H)
            ops.linear_grouped([_grp(t[0], sa[1], c2[:, :H]), _grp(t[1], sb[1], c2[:, H:])], H, H)
            w3, b3 = self._mix_weight(sa[2][0], sb[2][0])
            zc = torch.empty(1, H, device=dev)
            ops.linear_grouped([dict(x=c2, weight=w3, bias=b3, relu=False, y=zc)], 2 * H, H)
            self._zc = (c2[:, :H].contiguous(), c2[:, H:].contiguous(), zc)
            self._zc_key = key
        return self._zc

    def _heads_skipping(self, xcn1, xcn2, xij, cls):
        """The heads on class-major rows (ops.class_order): layer 1-2 of xcn1lin only where cnt1 > 0, of
        xcn2lin only where cnt2 > 0, the mix layer only where either is, the constants elsewhere; `lin`
        scatters the scores back to batch order."""
        B, H = xij.shape
        order2, _, r = cls
        sa, sb, sx = self._heads_plan(H)
        a2c, b2c, zc = self._zero_consts(sa, sb, H, xij.device)
        coef = self._mix_coef()
        dev = xij.device
        two = len(sx) == 2
        y0 = ops.buf(self._ws, "y0", (B, H), torch.float32, dev)
        z = ops.buf(self._ws, "z", (B, H), torch.float32, dev)
        t1 = ops.buf(self._ws, "t1", (3 if two else 2, B, H), torch.float32, dev)
        cat = ops.buf(self._ws, "cat", (B, 2 * H), torch.float32, dev)
        R = ops

        def xij_last(x, st):         # beta * xijlin(..): into y0 where the mix layer runs, + zc straight into z elsewhere
            return [_grp(x, st, y0, scale=coef[2:3], row_range=r[R.R_ANY]),
                    _grp(x, st, z, scale=coef[2:3], row_range=r[R.R_NONE], addend=zc, add_bcast=True)]

        # rows whose pooled xcn2 can be non-zero: cn5 orthogonalises cn2 against cn1 over the UNION pattern
        # (a cn1-only entry carries -t*inv2 when innerprod != 0), cn7 pools the raw cn2
        b_rows = [R.R_ANY] if self._xcn2_on_union else [R.R_BOTH, R.R_CN2_ONLY]
        g1 = [_grp(xcn1, sa[0], t1[0], row_range=r[R.R_CN1])] + [_grp(xcn2, sb[0], t1[1], row_range=r[q]) for q in b_rows]
        g1 += [_grp(xij, sx[0], t1[2])] if two else xij_last(xij, sx[0])
        ops.linear_grouped(g1, H, H)
        g2 = [_grp(t1[0], sa[1], cat[:, :H], row_range=r[R.R_CN1])] + \
             [_grp(t1[1], sb[1], cat[:, H:], row_range=r[q]) for q in b_rows]
        if two:
            g2 += xij_last(t1[2], sx[1])
        ops.linear_grouped(g2, H, H)
        if not self._xcn2_on_union:
            ops.fill_rows(cat[:, H:], b2c, r[R.R_CN1_ONLY])
        ops.fill_rows(cat[:, :H], a2c, r[R.R_CN2_ONLY])
        w3, b3 = self._mix_weight(sa[2][0], sb[2][0])
        ops.linear_grouped([dict(x=cat, weight=w3, bias=b3, relu=False, y=z, addend=y0, row_range=r[R.R_ANY])], 2 * H, H)
        return _seq_eval(self.lin, z, y_row_map=order2)

    def _heads_grouped(self, xcn1, xcn2, xij):
        """The three branches layer by layer instead of branch by branch: launch 1 = first layers of
        xcn1lin / xcn2lin / xijlin, launch 2 = their second layers (the two pooled branches write the
        halves of one [B, 2H] buffer), launch 3 = one K = 2H Linear that is both third layers and the
        branch mix, with the xij branch as its addend.  Returns None when the heads do not have that
        shape (then the per-branch walk runs)."""
        B, H = xij.shape
        if H not in ops.LINEAR_WIDTHS or not ops.fast_linear or B == 0:
            return None
        rows = B * H * 4
        if not (xcn1.is_contiguous() and xcn2.data_ptr() == xcn1.data_ptr() + rows
                and xij.data_ptr() == xcn2.data_ptr() + rows):
            return None
        sa, sb, sx = _stages(self.xcn1lin, H), _stages(self.xcn2lin, H), _stages(self.xijlin, H)
        if sa is None or sb is None or sx is None or len(sa) != 3 or len(sb) != 3 or len(sx) not in (1, 2):
            return None
        if sa[2][1] is not None or sa[2][2] or sb[2][1] is not None or sb[2][2]:
            return None
        coef = self._mix_coef()
        dev = xij.device
        y0 = ops.buf(self._ws, "y0", (B, H), torch.float32, dev)           # β · xijlin(x_i ⊙ x_j)
        t1 = ops.buf(self._ws, "t1", (3 if len(sx) == 2 else 2, B, H), torch.float32, dev)
        g1 = [_grp(xcn1, sa[0], t1[0]), _grp(xcn2, sb[0], t1[1]),
              _grp(xij, sx[0], t1[2]) if len(sx) == 2 else _grp(xij, sx[0], y0, scale=coef[2:3])]
        ops.linear_grouped(g1, H, H)
        cat = ops.buf(self._ws, "cat", (B, 2 * H), torch.float32, dev)
        g2 = [_grp(t1[0], sa[1], cat[:, :H]), _grp(t1[1], sb[1], cat[:, H:])]
        if len(sx) == 2:
            g2.append(_grp(t1[2], sx[1], y0, scale=coef[2:3]))
        ops.linear_grouped(g2, H, H)
        w3, b3 = self._mix_weight(sa[2][0], sb[2][0])
        z = ops.buf(self._ws, "z", (B, H), torch.float32, dev)
        ops.linear_grouped([dict(x=cat, weight=w3, bias=b3, relu=False, y=z, addend=y0)], 2 * H, H)
        return z

    # ---- the whole head as one launch (ocn_heads_fused) -------------------------------------------------------------
    def _fused_plan(self, H: int):
        """The parsed head when it has the layout ocn_heads_fused evaluates (the drivers': 3-layer pooled heads, 1- or
        2-layer xijlin, single-layer lin with the Linear(H, 1) tail, in_channels == hidden), else None."""
        key = (H, ops.fused_heads)
        if getattr(self, "_fplan_key", None) != key:
            plan = None
            if ops.fused_heads and H in ops.HEADS_WIDTHS:
                sa, sb, sx = _stages(self.xcn1lin, H), _stages(self.xcn2lin, H), _stages(self.xijlin, H)
                mods = [m for m in self.lin if not isinstance(m, (nn.Dropout, nn.Identity))]
                sl = _stages(nn.Sequential(*mods[:-1]), H) if len(mods) >= 2 else None
                tail = mods[-1] if mods else None

                def pooled_ok(st):
                    return (st is not None and len(st) == 3 and st[0][1] is None and st[0][2] and st[1][2]
                            and st[2][1] is None and not st[2][2])
                if (pooled_ok(sa) and pooled_ok(sb) and sx is not None and len(sx) in (1, 2) and sx[0][2]
                        and (len(sx) == 1 or (sx[1][1] is None and not sx[1][2]))
                        and sl is not None and len(sl) == 1 and sl[0][2]
                        and isinstance(tail, nn.Linear) and tail.out_features == 1 and tail.in_features == H):
                    lns = [sa[1][1], sb[1][1], sx[0][1], sl[0][1]]
                    if all(l is None for l in lns) or (all(l is not None for l in lns) and len({l.eps for l in lns}) == 1):
                        plan = (sa, sb, sx, sl[0], tail)
            self._fplan, self._fplan_key = plan, key
        return self._fplan

    def _fused_pack(self, H: int, dev):
        """Panels, folded output matrices, epilogue vectors and skipped-branch constants of ocn_heads_fused, cached on
        the parameter versions.  Folding (fp64, rounded once): the products of the reference without a non-linearity
        between them — third layers of xcn1lin / xcn2lin, second layer of xijlin, the mix of model.py:2436, lin[0]."""
        sa, sb, sx, sl, tail = self._fused_plan(H)
        plist = getattr(self, "_fp_params", None)
        if plist is None:
            plist = self._fp_params = [p for seq in (self.xcn1lin, self.xcn2lin, self.xijlin, self.lin) for p in seq.parameters()]
        self._mix_coef()
        key = (tuple(p._version for p in plist), tuple(p.data_ptr() for p in plist), self._coef_key, str(dev))
        if getattr(self, "_fpack_key", None) == key:
            return self._fpack
        with torch.no_grad():
            c = self._mix_coef().double()                       # [s(a0), s(a0) s(a1), beta]
            W0l, b0l = sl[0].weight.double(), sl[0].bias.double()
            Ma = c[0] * (W0l @ sa[2][0].weight.double())
            Mb = c[1] * (W0l @ sb[2][0].weight.double())
            bsum = c[0] * sa[2][0].bias.double() + c[1] * sb[2][0].bias.double()
            if len(sx) == 2:
                Mc = c[2] * (W0l @ sx[1][0].weight.double())
                bsum = bsum + c[2] * sx[1][0].bias.double()
            else:
                Mc = c[2] * W0l
            bf = W0l @ bsum + b0l
            ln = sa[1][1] is not None
            zeros, ones = torch.zeros(H, device=dev), torch.ones(H, device=dev)

            def gb(l):
                return (l.weight.float(), l.bias.float()) if l is not None else (ones, zeros)
            vecs = [sa[0][0].bias, sa[1][0].bias, *gb(sa[1][1]), sb[0][0].bias, sb[1][0].bias, *gb(sb[1][1]),
                    sx[0][0].bias, *gb(sx[0][1]), bf.float(), *gb(sl[1]), tail.weight.reshape(-1), zeros, zeros]
            assert len(vecs) == int(ops._lib.lib().ocn_heads_nvec())
            # f16 hi/lo panels in the order of the kernel's stream: xcn1lin.0 .3 Ma | xcn2lin.0 .3 Mb | xijlin.0 Mc
            pans = [ops.heads_panel(m.float().contiguous()) for m in
                    (sa[0][0].weight, sa[1][0].weight, Ma, sb[0][0].weight, sb[1][0].weight, Mb, sx[0][0].weight, Mc)]
            nscal = int(ops._lib.lib().ocn_heads_nscal())
            scal = torch.zeros(nscal, device=dev)
            scal[0] = tail.bias.detach().float().reshape(())
            scal[1:9] = torch.tensor([p[1] for p in pans], device=dev)
            vec = torch.cat([v.detach().float().reshape(-1) for v in vecs] + [scal]).contiguous()
            pack = dict(first=[pans[0][0], pans[3][0], pans[6][0]], mid=[pans[1][0], pans[4][0]],
                        out=[pans[2][0], pans[5][0], pans[7][0]],
                        vec=vec, ln=ln, eps=(sa[1][1].eps if ln else 1e-5), flops_per_row=2.0 * H * H * 8)
            # the constants a skipped branch contributes: the branch's share of the output on an all-zero pooled row,
            # computed by the kernel itself (constants mode) so that skipping changes no bit
            z = torch.zeros(1, H, device=dev)
            cpark = torch.empty(int(ops._lib.lib().ocn_heads_const_bytes(H)) // 4, device=dev)
            scratch = ops.buf(self._ws, "heads_scratch", int(ops._lib.lib().ocn_heads_scratch_bytes(H)) // 4, torch.float32, dev)
            ops.heads_fused(z, z, z, pack, None, None, self._xcn2_on_union, scratch, dump=cpark)
            pack["cpark"] = cpark
        self._fpack, self._fpack_key = pack, key
        return pack

    def _heads_fused(self, xcn1, xcn2, xij, cls):
        B, H = xij.shape
        pack = self._fused_pack(H, xij.device)
        scratch = ops.buf(self._ws, "heads_scratch", int(ops._lib.lib().ocn_heads_scratch_bytes(H)) // 4, torch.float32, xij.device)
        ranges, rowmap = (None, None) if cls is None else (cls[2], cls[0])
        return ops.heads_fused(xcn1, xcn2, xij, pack, ranges, rowmap, self._xcn2_on_union, scratch)

    def _heads(self, x, xcn1, xcn2, xij, cls=None):
        if (not self.training and not torch.is_grad_enabled() and xij.is_cuda and xij.dim() == 2 and xij.shape[0] > 0
                and xij.is_contiguous() and xcn1.is_contiguous() and xcn2.is_contiguous()
                and xcn1.shape == xij.shape and xij.shape[1] >= ops.fused_heads_min_width
                and self._fused_plan(xij.shape[1]) is not None):
            return self._heads_fused(xcn1, xcn2, xij, cls)
        if cls is not None:
            return self._heads_skipping(xcn1, xcn2, xij, cls)
        if self.training or torch.is_grad_enabled() or not xij.is_cuda or xij.shape[-1] % 4:
            alpha = torch.sigmoid(self.alpha).cumprod(-1)
            run = _seq_train if (xij.is_cuda and ops.train_linear) else (lambda seq, t: seq(t))
            xij = run(self.xijlin, xij)
            xcn1 = run(self.xcn1lin, xcn1)
            xcn2 = run(self.xcn2lin, xcn2)
            return run(self.lin, _mix(alpha, self.beta, xcn1, xcn2, xij))
        # eval under no_grad (the drivers' test()): same modules, same parameters, on the bf16x6 MFMA
        # Linear kernel with fused LayerNorm/ReLU epilogues.  With autograd on, the torch modules
        # above run instead so that the graph is recorded.
        z = self._heads_grouped(xcn1, xcn2, xij)
        if z is not None:
            return _seq_eval(self.lin, z)
        xij = _seq_eval(self.xijlin, xij)
        xcn1 = _seq_eval(self.xcn1lin, xcn1)
        xcn2 = _seq_eval(self.xcn2lin, xcn2)
        z = ops.combine3(self._mix_coef(), xcn1, xcn2, xij)
        return _seq_eval(self.lin, z)


class CNLinkPredictorOringin(_CNPredictorBase):
    """cn5 (model.py:2171-2443): column-normalised cn1, cn2 orthogonalised against it with the
    running inner product, column-normalised again; pooled embeddings -> MLP heads -> logit."""

    def multidomainforward(self, x, adj, cn1, cn2, tar_ei, filled1: bool = False,
                           cndropprobs: Iterable[float] = []):
        st = self._exchange(fuse(cn1, cn2, tar_ei, self._scratch(x), adj=adj), x)
        w = self._weights(st, None)
        xcn1, xcn2, xij = self._pool(st, w, x)
        return self._heads(x, xcn1, xcn2, xij, getattr(st, "cls", None))

    def _weights(self, st, args):
        return st.weights_cn5(self.innerprod1(st))

    def forward(self, x, adj, cn1, cn2, tar_ei, filled1: bool = False):
        return self.multidomainforward(x, adj, cn1, cn2, tar_ei, filled1, [])


class CNLinkPredictorbaselearn(_CNPredictorBase):
    """cn7 (model.py:3021-3229): cn1 column-normalised with ``args.sum`` for columns hit fewer than
    twice, Chebyshev diagonal hard-wired to T0 = identity, raw cn2; same heads."""
    _xcn2_on_union = False
    _weights_need_args = True
    # Chebyshev basis index of the cn1 / cn2 branch (model.py:3141, 3186: ``evaluate_polynomial(num_cols, 0)`` — the reference
    # hard-wires 0 = T0 = identity, its --polyfirst / --polysecond flags are parsed and never read).  Plain
    # attributes, 0 by default: set them on the instance to evaluate the T1 .. T10 variants the reference file defines.
    polyfirst: int = 0
    polysecond: int = 0

    def multidomainforward(self, x, adj, cn1, cn2, tar_ei, args, filled1: bool = False,
                           cndropprobs: Iterable[float] = []):
        st = self._exchange(fuse(cn1, cn2, tar_ei, self._scratch(x), adj=adj), x)
        w = self._weights(st, args)
        xcn1, xcn2, xij = self._pool(st, w, x)
        return self._heads(x, xcn1, xcn2, xij, getattr(st, "cls", None))

    def _weights(self, st, args):
        dev = st.hist.device
        return st.weights_cn7(float(args.sum), chebyshev_diag(st.N, int(self.polyfirst), dev),
                              chebyshev_diag(st.N, int(self.polysecond), dev))

    def _rowsum(self, st, x):
        """A·h, once per (embeddings, adjacency): cn7's cn2 weights are exactly 1 (raw cn2, model.py:3186-3209), so a
        candidate whose target's A² row holds EVERY neighbour of its source — the rule on a dense graph, ogbl-ddi's A² is
        full — has xcn2 = (A·h)[source], the same additions in the same order; the pooling copies that row instead of
        summing ~500 embedding rows again for each of the source's candidates (ocn_hip.h, ocn_cn_gather `rowsum`).  Only
        where it can pay: pattern route, A² at least half full."""
        adj, t2 = st.adj, getattr(st, "t2", None)
        if not ops.share_full_rows or st.walk or t2 is None or not x.is_cuda or x.dim() != 2 or int(self.polysecond) != 0:
            return None                        # (a non-trivial basis weights the cn2 entries: the row sum is not their pool)
        n = adj.size(0)
        if t2.nnz() * 2 < n * n or adj.size(1) != x.shape[0] or n != x.shape[0]:
            return None
        # Keyed on the live tensor OBJECTS (weak references) and x's version counter, never on addresses: the drivers' test()
        # computes a fresh h = model(x, adj) per evaluation, and the caching allocator hands the new h the freed one's
        # address (a data_ptr key then matched and the pooling copied the previous epoch's row sums).
        import weakref
        hit = getattr(self, "_rowsum_cache", None)
        if (hit is None or hit[0]() is not x or hit[1] != x._version or hit[2]() is not adj
                or hit[3] != (x.data_ptr(), tuple(x.shape))):
            if torch.cuda.is_current_stream_capturing() and hit is not None:
                raise RuntimeError("embeddings changed since the last eager call: run one eager batch before capturing")
            hit = self._rowsum_cache = (weakref.ref(x), x._version, weakref.ref(adj), (x.data_ptr(), tuple(x.shape)),
                                        ops.spmm_csr(adj._rowptr, adj._col, x))
        return hit[4]

    def forward(self, x, adj, cn1, cn2, tar_ei, filled1: bool = False):
        # the drivers pass the argparse Namespace in this slot (NeighborOverlap_large.py:122)
        return self.multidomainforward(x, adj, cn1, cn2, tar_ei, filled1, [])


class CNLinkPredictor3hopCNs(_CNPredictorBase):
    """cn6 (model.py:2445-2951): the 3-hop predictor.  cn1 / cn2 as cn5; cn3 = N(i) ∩ N³(j) is
    orthogonalised against BOTH normalised matrices and column-normalised; a fourth head ``xcn3lin`` and
    ``alpha[2]`` join the mix.  ``forward(x, adj, cn1, cn2, cn3, tar_ei, args)`` as in the reference
    (:2950); no reference driver builds a cn3 — ``adjoverlap(adj, adj3, e)`` with adj3 the pattern of
    A·A·A is its natural source.  Forward only (eval / no-grad): the reference's training path updates the
    one ``innerprod`` buffer three times per call with three different quantities (:2527-2533) and is
    reached by no driver."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, dropout, edrop=0.0,
                 ln=False, cndeg=-1, use_xlin=False, tailact=False, twolayerlin=False, beta=1.0):
        super().__init__(in_channels, hidden_channels, out_channels, num_layers, dropout, edrop, ln, cndeg,
                         use_xlin, tailact, twolayerlin, beta)
        H, p = hidden_channels, dropout
        norm = nn.LayerNorm(H) if ln else nn.Identity()
        del self.xcn4lin                                        # cn6 has xcn3lin in its place (:2496-2501)
        self.xcn3lin = nn.Sequential(nn.Linear(in_channels, H), nn.Dropout(p, inplace=True), nn.ReLU(inplace=True),
                                     nn.Linear(H, H), norm, nn.Dropout(p, inplace=True), nn.ReLU(inplace=True),
                                     nn.Linear(H, H))

    def multidomainforward(self, x, adj, cn1, cn2, cn3, tar_ei, args=None, cndropprobs: Iterable[float] = []):
        from .utils import fuse3
        if self.training:
            raise NotImplementedError("cn6 in training mode updates the one `innerprod` buffer three times per call with three "
                                      "different quantities (model.py:2527-2533, 2636, 2817, 2842) and is reached by no driver: "
                                      "call .eval(); gradients with respect to x and the head parameters flow in eval mode")
        st = fuse3(cn1, cn2, cn3, tar_ei)
        if self._sharded:
            from .dist import allreduce_hist
            allreduce_hist(st.a.hist, self._shard_group, valued=False)
            allreduce_hist(st.b.hist, self._shard_group, valued=False)
        wa, wb, nip = st.weights(self.innerprod, sharded=self._sharded)
        if torch.is_grad_enabled():
            # autograd on (row f3): the pooling through _PoolFn3 (its transpose is ocn_cn_gather3_backward), the
            # heads through the modules / the MFMA Linear under autograd, as cn5 / cn7 do
            x = x.contiguous()
            xcn1, xcn2, xcn3, xij = (_PoolFn3.apply(x, st, wa, wb, nip) if x.requires_grad else st.gather(wa, wb, nip, x))
            alpha = torch.sigmoid(self.alpha).cumprod(-1)
            run = _seq_train if (xij.is_cuda and ops.train_linear) else (lambda seq, t: seq(t))
            z = (alpha[0] * run(self.xcn1lin, xcn1) + alpha[1] * run(self.xcn2lin, xcn2)
                 + alpha[2] * run(self.xcn3lin, xcn3) + self.beta * run(self.xijlin, xij))
            return run(self.lin, z)
        xcn1, xcn2, xcn3, xij = st.gather(wa, wb, nip, x.contiguous())
        with torch.no_grad():
            alpha = torch.sigmoid(self.alpha).cumprod(-1)
            z = (alpha[0] * _seq_eval(self.xcn1lin, xcn1) + alpha[1] * _seq_eval(self.xcn2lin, xcn2)
                 + alpha[2] * _seq_eval(self.xcn3lin, xcn3) + self.beta * _seq_eval(self.xijlin, xij))
            return _seq_eval(self.lin, z)

    def forward(self, x, adj, cn1, cn2, cn3, tar_ei, args=None):
        return self.multidomainforward(x, adj, cn1, cn2, cn3, tar_ei, args)


predictor_dict = {
    "cn5": CNLinkPredictorOringin,
    "cn6": CNLinkPredictor3hopCNs,
    "cn7": CNLinkPredictorbaselearn,
}
