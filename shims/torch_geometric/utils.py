"""`from torch_geometric.utils import negative_sampling` (NeighborOverlap_large.py:12,51)."""
import torch


def negative_sampling(edge_index, num_nodes=None, num_neg_samples=None, **kwargs):
    """Uniform random node pairs that are not edges of `edge_index` (PyG's sparse method: sample, reject hits)."""
    n = int(num_nodes if num_nodes is not None else int(edge_index.max()) + 1)
    m = int(num_neg_samples if num_neg_samples is not None else edge_index.shape[1])
    key = torch.unique(edge_index[0].long() * n + edge_index[1].long())
    out = []
    need = m
    for _ in range(64):                                    # rejection sampling, topped up until the count is met
        if need <= 0:
            break
        cand = torch.randint(0, n * n, (int(need * 1.2) + 16,), device=edge_index.device)
        idx = torch.searchsorted(key, cand).clamp_(max=max(key.numel() - 1, 0))
        ok = (key[idx] != cand) if key.numel() else torch.ones_like(cand, dtype=torch.bool)
        cand = cand[ok][:need]
        out.append(cand)
        need -= cand.numel()
    if need > 0:
        raise RuntimeError(f"negative_sampling: the graph is too dense to draw {m} non-edges ({m - need} found)")
    c = torch.cat(out) if out else torch.zeros(0, dtype=torch.long, device=edge_index.device)
    return torch.stack([torch.div(c, n, rounding_mode="floor"), c % n])


def to_undirected(edge_index, *a, **k):
    return torch.unique(torch.cat([edge_index, edge_index.flip(0)], dim=1), dim=1)
