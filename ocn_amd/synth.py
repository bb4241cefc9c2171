"""Seeded synthetic graphs of the reference's dataset shapes.

There is no network on the build or GPU boxes, so ``ogbdataset.loaddataset``
(/root/reference/ogbdataset.py:29-71) cannot fetch anything; these generators give
undirected simple graphs (no self loops, no multi-edges) with a truncated power-law
degree sequence matched to (N, nnz, d_max) of the named dataset.  ``clique_frac`` mixes
in small cliques ("papers"), which is what makes a co-authorship graph such as
ogbl-collab rich in triangles and therefore in 1-hop common neighbours.
"""
from __future__ import annotations

from typing import Tuple

import torch
from torch import Tensor

# public OGB / Planetoid statistics; nnz = stored directed entries
SHAPES = {
    "cora":      dict(n=2_708,     nnz=7_400,      max_deg=120,   feat=1433, clique_frac=0.3),
    "collab":    dict(n=235_868,   nnz=2_360_000,  max_deg=671,   feat=128,  clique_frac=0.6),
    "ppa":       dict(n=576_289,   nnz=42_500_000, max_deg=3_241, feat=58,   clique_frac=0.3),
    "citation2": dict(n=2_927_963, nnz=60_000_000, max_deg=10_000, feat=128, clique_frac=0.2),
    "ddi":       dict(n=4_267,     nnz=2_140_000,  max_deg=2_234, feat=0,    clique_frac=0.0),
}


def _powerlaw_weights(n: int, avg_deg: float, max_deg: int, gamma: float, g: torch.Generator) -> Tensor:
    u = torch.rand(n, generator=g, dtype=torch.float64)
    lo, hi = 1.0, float(max_deg)
    a = 1.0 - gamma
    w = (lo ** a + u * (hi ** a - lo ** a)) ** (1.0 / a)       # inverse-CDF of a truncated Pareto
    return (w * (avg_deg / w.mean())).clamp_(max=float(max_deg))


def chung_lu_graph(n: int, avg_deg: float, max_deg: int, seed: int = 0, gamma: float = 2.3,
                   clique_frac: float = 0.0, window: int = 512) -> Tensor:
    """Return an undirected edge list [2, m] (each edge once, src < dst).

    Endpoints are drawn in proportion to power-law node weights (Chung-Lu).  A ``clique_frac``
    share of the edges comes from cliques of 3-6 nodes whose members are drawn, again by weight,
    from a window of +-``window`` ids around a weight-drawn lead: hubs then sit in many
    overlapping cliques, which gives both the heavy degree tail and the triangles."""
    g = torch.Generator().manual_seed(seed)
    w = _powerlaw_weights(n, avg_deg, max_deg, gamma, g)
    cdf = torch.cumsum(w, 0)
    cdf /= cdf[-1].clone()
    cdf0 = torch.cat([torch.zeros(1, dtype=cdf.dtype), cdf])       # cdf0[i] = mass of ids < i
    m_target = int(n * avg_deg / 2)

    def draw(k: int) -> Tensor:
        return torch.searchsorted(cdf, torch.rand(k, generator=g, dtype=torch.float64)).clamp_(max=n - 1)

    def cliques(m_edges: int) -> Tensor:
        kmax = 6
        sizes = torch.randint(3, kmax + 1, (max(m_edges // 7, 1),), generator=g)
        lead = draw(sizes.numel())
        lo = (lead - window).clamp_(min=0)
        hi = (lead + window + 1).clamp_(max=n)
        u = torch.rand(sizes.numel(), kmax, generator=g, dtype=torch.float64)
        mass = cdf0[lo][:, None] + u * (cdf0[hi] - cdf0[lo])[:, None]
        mem = torch.searchsorted(cdf, mass).clamp_(max=n - 1)
        mem[:, 0] = lead
        valid = torch.arange(kmax)[None, :] < sizes[:, None]
        iu, ju = torch.triu_indices(kmax, kmax, offset=1)
        ok = valid[:, iu] & valid[:, ju]
        return torch.stack([mem[:, iu][ok], mem[:, ju][ok]])

    def dedup(parts) -> Tensor:
        ei = torch.cat(parts, dim=1)
        lo, hi = torch.minimum(ei[0], ei[1]), torch.maximum(ei[0], ei[1])
        keep = lo != hi
        return torch.unique(lo[keep] * n + hi[keep])

    m_cl = int(m_target * clique_frac)
    key_cl = torch.zeros(0, dtype=torch.long)
    for _ in range(8):                       # duplicates inside windows are common: top up
        if key_cl.numel() >= m_cl:
            break
        need = m_cl - key_cl.numel()
        new = dedup([cliques(int(need * 1.15) + 16)])
        key_cl = torch.unique(torch.cat([key_cl, new]))
    key = key_cl
    for _ in range(8):
        if key.numel() >= m_target:
            break
        need = int((m_target - key.numel()) * 1.05) + 8
        key = torch.unique(torch.cat([key, dedup([torch.stack([draw(need), draw(need)])])]))
    if key.numel() > m_target:
        key = key[torch.randperm(key.numel(), generator=g)[:m_target]].sort().values
    return torch.stack([torch.div(key, n, rounding_mode="floor"), key % n])


def dataset_like(name: str, seed: int = 0, scale: float = 1.0) -> Tuple[Tensor, int, dict]:
    """Undirected edge list + N of a graph shaped like ``name``; ``scale`` < 1 shrinks N and
    nnz together (tests)."""
    s = SHAPES[name]
    n = max(int(s["n"] * scale), 16)
    nnz = s["nnz"] * scale
    ei = chung_lu_graph(n, avg_deg=nnz / n, max_deg=min(s["max_deg"], n - 1), seed=seed,
                        clique_frac=s["clique_frac"])
    return ei, n, s


def sample_edges(row: Tensor, col: Tensor, n: int, B: int, seed: int = 1, pos_frac: float = 0.5) -> Tensor:
    """Candidate batch [2, B] int64: ``pos_frac`` sampled from stored entries (row, col) of the
    adjacency (positives: high overlap), the rest uniform random pairs (negatives) — SURVEY §8d."""
    g = torch.Generator().manual_seed(seed)
    n_pos = int(B * pos_frac) if row.numel() else 0
    pick = torch.randint(0, max(row.numel(), 1), (n_pos,), generator=g)
    pos = torch.stack([row[pick], col[pick]]) if n_pos else torch.zeros(2, 0, dtype=torch.long)
    neg = torch.randint(0, n, (2, B - n_pos), generator=g)
    e = torch.cat([pos.to(torch.long), neg], dim=1)
    return e[:, torch.randperm(B, generator=g)].contiguous()


def loaddataset_like(name: str, use_valedges_as_input: bool = False, seed: int = 0, scale: float = 1.0,
                     val_ratio: float = 0.05, test_ratio: float = 0.10, n_neg: int = 0, feat: int = 0):
    """(data, split_edge) with the fields the reference drivers read from ``ogbdataset.loaddataset``
    (/root/reference/ogbdataset.py:29-71), on a seeded synthetic graph of the named shape:

    ``data.x`` (features, or node ids with ``data.max_x`` for ppa / ddi as in :47-52), ``data.adj_t``
    (symmetric adjacency of the TRAINING edges), ``data.full_adj_t`` (+ validation edges when
    ``use_valedges_as_input``, :61-68), ``data.num_nodes``; ``split_edge[train|valid|test]['edge']``
    [m, 2] (each undirected edge once) and ``['edge_neg']`` [n_neg, 2] uniform random non-self pairs.
    citation2's per-positive negatives (``target_node_neg``) are given as ``['edge_neg']`` of shape
    [m, n_neg_per, 2] when ``name == 'citation2'``."""
    from types import SimpleNamespace
    from .sparse import SparseTensor
    ei, n, shape = dataset_like(name, seed=seed, scale=scale)
    g = torch.Generator().manual_seed(seed + 17)
    m = ei.shape[1]
    perm = torch.randperm(m, generator=g)
    n_val, n_test = int(m * val_ratio), int(m * test_ratio)
    idx = {"valid": perm[:n_val], "test": perm[n_val:n_val + n_test], "train": perm[n_val + n_test:]}
    split_edge = {k: {"edge": ei[:, v].t().contiguous()} for k, v in idx.items()}
    n_neg = n_neg or max(n_val, 1)

    def negatives(shape_):
        a = torch.randint(0, n, shape_, generator=g)
        b = torch.randint(0, n - 1, shape_, generator=g)
        return torch.stack([a, b + (b >= a).long()], dim=-1)          # no self pairs

    if name == "citation2":
        for k in ("valid", "test"):
            src = split_edge[k]["edge"][:, 0]
            per = max(n_neg // max(src.numel(), 1), 1) if n_neg else 1000
            tgt = torch.randint(0, n, (src.numel(), per), generator=g)
            split_edge[k]["edge_neg"] = torch.stack([src[:, None].expand_as(tgt), tgt], dim=-1)
    else:
        for k in ("valid", "test"):
            split_edge[k]["edge_neg"] = negatives((n_neg,))
    data = SimpleNamespace(num_nodes=n, max_x=-1, edge_weight=None)
    train_ei = split_edge["train"]["edge"].t()
    data.edge_index = torch.cat([train_ei, train_ei.flip(0)], dim=1)
    data.adj_t = SparseTensor.from_edge_index(data.edge_index, sparse_sizes=(n, n)).to_symmetric().coalesce()
    gx = torch.Generator().manual_seed(seed + 29)
    if name == "ppa":
        data.x = torch.randint(0, shape["feat"], (n,), generator=gx)
        data.max_x = int(data.x.max())
    elif name == "ddi":
        data.x = torch.arange(n)
        data.max_x = n
    else:
        data.x = torch.randn(n, feat or shape["feat"], generator=gx)
    if use_valedges_as_input:
        full = torch.cat([data.edge_index, split_edge["valid"]["edge"].t()], dim=1)
        data.full_adj_t = SparseTensor.from_edge_index(full, sparse_sizes=(n, n)).coalesce().to_symmetric()
    else:
        data.full_adj_t = data.adj_t
    return data, split_edge
