"""Hits@K and MRR as the reference's drivers obtain them from ``ogb.linkproppred.Evaluator``
(NeighborOverlap_large.py:11,161-179,264-266; NeighborOverlapCitation2.py uses the ``mrr`` form).

``ogb`` is a third-party dependency of the reference and is not vendored there (``environment.yml``);
this is its published metric definition, computed on whatever device the scores live on so that the
evaluation loop needs no per-batch host copy:

* hits@K  = share of positive scores strictly above the K-th largest negative score
            (1.0 when there are fewer than K negatives);
* mrr     = per positive, rank = 1 + (#negatives > pos + #negatives >= pos) / 2 among its own row of
            negatives; returns ``mrr_list`` and ``hits@{1,3,10}_list``.
"""
from __future__ import annotations

from typing import Dict

import torch
from torch import Tensor

_METRIC = {"ogbl-collab": ("hits", 50), "ogbl-ppa": ("hits", 100), "ogbl-ddi": ("hits", 20),
           "ogbl-citation2": ("mrr", None)}


class Evaluator:
    """Drop-in for ``ogb.linkproppred.Evaluator(name=...)``: attribute ``K`` (the drivers overwrite it,
    ``evaluator.K = K``), attribute ``eval_metric``, method ``eval(input_dict)``."""

    def __init__(self, name: str):
        kind, k = _METRIC.get(name, ("hits", 100))         # the Planetoid runs construct 'ogbl-ppa' (driver :264)
        self.name = name
        self.eval_metric = "mrr" if kind == "mrr" else f"hits@{k}"
        self.K = k

    def eval(self, input_dict: Dict[str, Tensor]) -> Dict[str, object]:
        for key in ("y_pred_pos", "y_pred_neg"):
            if key not in input_dict:
                raise RuntimeError(f"Missing key of {key}")
        pos, neg = input_dict["y_pred_pos"], input_dict["y_pred_neg"]
        if not (torch.is_tensor(pos) and torch.is_tensor(neg)):
            raise ValueError("y_pred_pos and y_pred_neg must be torch tensors")
        if self.eval_metric == "mrr":
            return self._mrr(pos, neg)
        return {f"hits@{self.K}": self._hits(pos, neg, int(self.K))}

    @staticmethod
    def _hits(pos: Tensor, neg: Tensor, k: int) -> float:
        if pos.dim() != 1 or neg.dim() != 1:
            raise RuntimeError("hits@K expects 1-d score tensors")
        if neg.numel() < k:
            return 1.0
        kth = torch.topk(neg, k).values[-1]
        return float((pos > kth).sum().item()) / max(pos.numel(), 1)

    @staticmethod
    def _mrr(pos: Tensor, neg: Tensor) -> Dict[str, Tensor]:
        if pos.dim() != 1 or neg.dim() != 2 or neg.shape[0] != pos.shape[0]:
            raise RuntimeError("mrr expects y_pred_pos [n] and y_pred_neg [n, n_neg]")
        p = pos.reshape(-1, 1)
        optimistic = (neg > p).sum(dim=1)
        pessimistic = (neg >= p).sum(dim=1)
        rank = 0.5 * (optimistic + pessimistic).to(torch.float32) + 1.0
        return {"hits@1_list": (rank <= 1).to(torch.float32), "hits@3_list": (rank <= 3).to(torch.float32),
                "hits@10_list": (rank <= 10).to(torch.float32), "mrr_list": 1.0 / rank}
