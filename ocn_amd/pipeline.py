"""Batched scoring with the memory-bound CN stage and the MFMA-bound MLP heads on two HIP streams.

``score_edges`` computes exactly what the reference's ``test()`` computes with

    torch.cat([predictor(h, adj, adjoverlap(adj, adj, e[perm].t()), adjoverlap(adj, adj2, e[perm].t()),
                         e[perm].t(), args).squeeze().cpu() for perm in PermIterator(dev, n, bs, False)])

(NeighborOverlap_large.py:121-159) — same batch composition, so the same batch-coupled
normalisation — but keeps the scores on the device and lets batch t+1's intersection / pooling
kernels (HBM- and L2-bound, no LDS, < 64 VGPRs) run beside batch t's Linear layers (fp32 MFMA
GEMMs) instead of behind them.  One D2H copy at the end replaces one blocking ``.cpu()`` per batch.
"""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import Tensor

from . import ops
from .utils import CNState, PermIterator


class TwoStreamScorer:
    """Reusable pair of side streams for one predictor on one device."""

    def __init__(self, predictor, device=None):
        self.pred = predictor
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.dev = dev
        self.cn_stream = torch.cuda.Stream(device=dev)
        self.mlp_stream = torch.cuda.Stream(device=dev)

    def begin(self) -> None:
        """Make both side streams wait for whatever produced the inputs on the current stream."""
        cur = torch.cuda.current_stream(self.dev)
        self.cn_stream.wait_stream(cur)
        self.mlp_stream.wait_stream(cur)

    def submit(self, h: Tensor, adj, adj2, edges: Tensor, args=None) -> Tensor:
        """Enqueue one candidate batch; returns the [B,1] score tensor (valid after ``end()`` or a
        wait on ``mlp_stream``)."""
        pred = self.pred
        is_cn7 = hasattr(args, "sum") and type(pred).__name__ == "CNLinkPredictorbaselearn"
        # the batch slice was produced on the caller's stream just now
        cur = torch.cuda.current_stream(self.dev)
        self.cn_stream.wait_event(cur.record_event())
        edges.record_stream(self.cn_stream)
        with torch.cuda.stream(self.cn_stream):
            st = pred._exchange(CNState(adj, adj, adj2, edges))
            w = st.weights_cn7(float(args.sum)) if is_cn7 else st.weights_cn5(pred.innerprod)
            xcn1, xcn2, xij = st.gather(w, h)
            ready = self.cn_stream.record_event()
        for t in (xcn1, xcn2, xij):
            t.record_stream(self.mlp_stream)
        with torch.cuda.stream(self.mlp_stream):
            self.mlp_stream.wait_event(ready)
            ops._mark("begin")
            out = pred._heads(h, xcn1, xcn2, xij)
            ops._mark("mlp")
        return out

    def end(self, outs: List[Tensor]) -> None:
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_stream(self.mlp_stream)
        cur.wait_stream(self.cn_stream)
        for o in outs:
            o.record_stream(cur)


@torch.no_grad()
def score_edges(predictor, h: Tensor, adj, adj2, edges: Tensor, batch_size: int, args=None,
                scorer: Optional[TwoStreamScorer] = None, run_ahead: int = 6) -> Tensor:
    """Scores for ``edges`` [n, 2] (the layout of ``split_edge[...]['edge']``), batched like
    ``PermIterator(.., training=False)``; returns a [n] fp32 tensor on the device.  The host stays at
    most ``run_ahead`` batches ahead of the GPU: an unbounded backlog makes the HIP runtime block the
    host until the queue has drained completely, and it bounds the live scratch."""
    if predictor.training:
        raise RuntimeError("score_edges is the eval path; call predictor.eval() first")
    scorer = scorer or TwoStreamScorer(predictor, h.device)
    h = h.contiguous()
    scorer.begin()
    outs, done = [], []
    for perm in PermIterator(edges.device, edges.shape[0], batch_size, training=False):
        if len(done) >= max(run_ahead, 1):
            done.pop(0).synchronize()
        outs.append(scorer.submit(h, adj, adj2, edges[perm].t().contiguous(), args))
        done.append(scorer.mlp_stream.record_event())
    scorer.end(outs)
    return torch.cat(outs, dim=0).squeeze(-1) if outs else h.new_zeros(0)


@torch.no_grad()
def score_mrr_split(predictor, h: Tensor, adj, source: Tensor, target: Tensor, target_neg: Tensor,
                    batch_size: int, args=None, evaluator=None):
    """``test_split`` of the citation2 driver (NeighborOverlapCitation2.py:227-254): positives
    (source, target) and, per positive, ``target_neg.shape[1]`` negatives sharing its source, each
    scored through ``get_cn1_cn2`` in ``PermIterator(.., training=False)`` batches — same batch
    composition as the reference, scores kept on the device.  Returns (pos_pred [n], neg_pred [n, n_neg])
    or, with an ``evaluator``, the mean of its ``mrr_list``."""
    from .utils import get_cn1_cn2
    if predictor.training:
        raise RuntimeError("score_mrr_split is the eval path; call predictor.eval() first")

    def run(src_all: Tensor, dst_all: Tensor) -> Tensor:
        outs = []
        for perm in PermIterator(src_all.device, src_all.shape[0], batch_size, training=False):
            e = torch.stack((src_all[perm], dst_all[perm]))
            cn1, cn2 = get_cn1_cn2(adj, e)
            outs.append(predictor(h, adj, cn1, cn2, e, args).reshape(-1))
        return torch.cat(outs, dim=0) if outs else h.new_zeros(0)

    pos_pred = run(source, target)
    n_neg = target_neg.shape[1]
    neg_pred = run(source.view(-1, 1).repeat(1, n_neg).view(-1), target_neg.reshape(-1)).view(-1, n_neg)
    if evaluator is None:
        return pos_pred, neg_pred
    return evaluator.eval({"y_pred_pos": pos_pred, "y_pred_neg": neg_pred})["mrr_list"].mean().item()


class GraphedScorer:
    """One candidate batch of a FIXED size as a captured HIP graph: ~40 kernel launches per batch become one
    graph launch.  For the small-batch configurations (Cora / Citeseer / Pubmed drivers: 1152-edge batches on
    a 2.7 k-node graph) the host spends 0.20 ms of Python per batch — replayed, 0.02 ms: one copy of the edge
    ids and one launch.  (Measured on MI355X: the GPU side of such a batch, ~40 dependent dispatches of a
    few microseconds each, stays at 0.23 ms either way — the graph frees the host, it does not shorten the
    chain.)

        scorer = GraphedScorer(predictor, h, adj, adj2, batch_size=1152, args=args)
        scores = scorer(edges_2xB)            # [B, 1]; valid until the next call

    The graph bakes in the addresses of ``h``, the adjacency arrays and the predictor's parameters: build a
    new scorer when any of them is replaced (in-place updates are seen).  Candidate ids are NOT bounds-
    checked on replay (that check is a host sync); pass ``check=True`` to pay for it."""

    def __init__(self, predictor, h: Tensor, adj, adj2, batch_size: int, args=None, route: str = "pattern"):
        if predictor.training:
            raise RuntimeError("GraphedScorer is the eval path; call predictor.eval() first")
        if int(batch_size) >= ops.sort_edges_min_batch:
            # validated for the small-batch regime only (where the host is the cost); a 65 536-edge capture
            # ended in a GPU memory fault on MI355X and buys nothing: such a batch is GPU-bound
            raise ValueError(f"GraphedScorer is for small batches (< {ops.sort_edges_min_batch} candidates)")
        self.pred, self.h, self.adj, self.adj2, self.args, self.route = predictor, h.contiguous(), adj, adj2, args, route
        self.B = int(batch_size)
        dev = h.device
        self.edges = torch.zeros(2, self.B, dtype=torch.int64, device=dev)
        self.graph = torch.cuda.CUDAGraph()
        keep = ops.validate_indices
        ops.validate_indices = False                      # a bounds check is a host sync: not capturable
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side), torch.no_grad():  # warm-up: scratch buffers, weight panels, attributes
                for _ in range(2):
                    self._step()
            torch.cuda.current_stream(dev).wait_stream(side)
            with torch.cuda.graph(self.graph), torch.no_grad():
                self.out = self._step()
        finally:
            ops.validate_indices = keep

    def _step(self) -> Tensor:
        from .utils import adjoverlap, get_cn1_cn2
        if self.route == "walk":
            cn1, cn2 = get_cn1_cn2(self.adj, self.edges)
        else:
            cn1, cn2 = adjoverlap(self.adj, self.adj, self.edges), adjoverlap(self.adj, self.adj2, self.edges)
        return self.pred(self.h, self.adj, cn1, cn2, self.edges, self.args)

    def __call__(self, edges: Tensor, check: bool = False) -> Tensor:
        if tuple(edges.shape) != (2, self.B):
            raise ValueError(f"GraphedScorer was captured for [2, {self.B}] candidate edges, got {tuple(edges.shape)}")
        if check:
            ops.check_edges(edges[0], edges[1], self.adj.size(0), self.adj.size(0))
        self.edges.copy_(edges, non_blocking=True)
        self.graph.replay()
        return self.out
