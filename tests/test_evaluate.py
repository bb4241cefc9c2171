"""Hits@K / MRR (ocn_amd.evaluate) against brute-force restatements of the published OGB definitions,
and the synthetic loaddataset_like surface the drivers read (ogbdataset.py:29-71).  CPU only."""
import numpy as np
import pytest
import torch

from ocn_amd.evaluate import Evaluator
from ocn_amd.synth import loaddataset_like


def _hits_brute(pos, neg, k):
    if len(neg) < k:
        return 1.0
    kth = np.sort(neg)[::-1][k - 1]
    return float(np.sum(pos > kth)) / len(pos)


@pytest.mark.parametrize("seed", range(4))
def test_hits_at_k_matches_the_definition(seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.randn(500, generator=g).round(decimals=1)          # rounded: ties with the threshold happen
    neg = torch.randn(3000, generator=g).round(decimals=1)
    ev = Evaluator("ogbl-collab")
    assert ev.K == 50 and ev.eval_metric == "hits@50"
    for k in (1, 20, 50, 100, 2999, 3000, 3001):
        ev.K = k
        got = ev.eval({"y_pred_pos": pos, "y_pred_neg": neg})[f"hits@{k}"]
        assert got == pytest.approx(_hits_brute(pos.numpy(), neg.numpy(), k), abs=1e-12)
    assert Evaluator("ogbl-ppa").K == 100 and Evaluator("ogbl-ddi").K == 20


def test_mrr_matches_the_definition():
    g = torch.Generator().manual_seed(1)
    pos = torch.randn(200, generator=g).round(decimals=1)
    neg = torch.randn(200, 50, generator=g).round(decimals=1)
    out = Evaluator("ogbl-citation2").eval({"y_pred_pos": pos, "y_pred_neg": neg})
    p, n = pos.numpy(), neg.numpy()
    rank = np.array([0.5 * ((n[i] > p[i]).sum() + (n[i] >= p[i]).sum()) + 1 for i in range(len(p))])
    assert np.allclose(out["mrr_list"].numpy(), 1.0 / rank)
    for k in (1, 3, 10):
        assert np.array_equal(out[f"hits@{k}_list"].numpy(), (rank <= k).astype(np.float32))


def test_evaluator_errors_like_the_original():
    ev = Evaluator("ogbl-collab")
    with pytest.raises(RuntimeError):
        ev.eval({"y_pred_pos": torch.zeros(3)})
    with pytest.raises(RuntimeError):
        Evaluator("ogbl-citation2").eval({"y_pred_pos": torch.zeros(3), "y_pred_neg": torch.zeros(4, 5)})


@pytest.mark.parametrize("name,valedges", [("cora", False), ("cora", True), ("ddi", False)])
def test_loaddataset_like_surface(name, valedges):
    data, split = loaddataset_like(name, valedges, seed=3, scale=0.3 if name == "ddi" else 1.0)
    n = data.num_nodes
    tr, va, te = (split[k]["edge"] for k in ("train", "valid", "test"))
    assert tr.shape[1] == 2 and va.shape[1] == 2 and te.shape[1] == 2
    # the three splits partition the undirected edges; the adjacency holds the training edges only
    allp = torch.cat([tr, va, te])
    key = allp.min(1).values * n + allp.max(1).values
    assert key.unique().numel() == key.numel()
    r, c, _ = data.adj_t.coo()
    akey = set((r * n + c).tolist())
    assert all(int(a * n + b) in akey and int(b * n + a) in akey for a, b in tr[:200].tolist())
    assert not any(int(a * n + b) in akey for a, b in te[:200].tolist())
    assert data.adj_t.nnz() == 2 * tr.shape[0]
    if valedges:
        assert data.full_adj_t.nnz() == 2 * (tr.shape[0] + va.shape[0])
    else:
        assert data.full_adj_t is data.adj_t
    neg = split["valid"]["edge_neg"]
    assert neg.shape[1] == 2 and (neg[:, 0] != neg[:, 1]).all() and int(neg.max()) < n
    if name == "ddi":
        assert data.max_x == n and torch.equal(data.x, torch.arange(n))
    else:
        assert data.max_x == -1 and data.x.shape == (n, 1433)
