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
