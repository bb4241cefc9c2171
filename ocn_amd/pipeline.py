"""Batched scoring loops of the drivers' ``test()`` functions with the scores kept on the device.

``score_edges`` computes exactly what the reference's ``test()`` computes with

    torch.cat([predictor(h, adj, adjoverlap(adj, adj, e[perm].t()), adjoverlap(adj, adj2, e[perm].t()),
                         e[perm].t(), args).squeeze().cpu() for perm in PermIterator(dev, n, bs, False)])

(NeighborOverlap_large.py:121-159) — same batch composition, so the same batch-coupled normalisation —
with one D2H copy at the end instead of one blocking ``.cpu()`` per batch.  (An earlier version ran the
CN stage of batch t+1 on a second stream beside the MLP heads of batch t: +4 % only, the MFMA kernels
leave no registers for a co-resident wave — DESIGN.md §4; the single-stream loop below, which gets the
predictor's scratch reuse and zero-row skipping, is faster.)
"""
from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import ops
from .utils import PermIterator


def _graphed(predictor, h, adj, make_handles, batch_size, args, n_batches):
    """A ``GraphedPhases`` for this loop where replaying pays: ``ops.graph_loops``, a GPU, an unsharded predictor and enough
    batches to amortise two eager uses and one capture per scratch set."""
    if not (ops.graph_loops and h.is_cuda and not getattr(predictor, "_sharded", False)):
        return None
    g = GraphedPhases(predictor, h, adj, make_handles, batch_size, args)
    return g if n_batches >= ops.graph_loops_min_batches_per_set * g.n_sets else None


def _dealt(perms, group):
    """(indices of the batches this rank scores, world size): whole batches dealt round robin over ``group`` (dist.deal_batches),
    or every batch when there is no group / one rank / fewer batches than ranks."""
    import torch.distributed as dist
    from .dist import deal_batches
    if group is None or not (dist.is_available() and dist.is_initialized()):
        return list(range(len(perms))), 1
    world = dist.get_world_size(group if group is not True else None)
    if world == 1 or len(perms) < world:
        return list(range(len(perms))), 1
    return deal_batches(len(perms), world, dist.get_rank(group if group is not True else None)), world


@torch.no_grad()
def score_edges(predictor, h: Tensor, adj, adj2, edges: Tensor, batch_size: int, args=None,
                run_ahead: int = 6, group=None) -> Tensor:
    """Scores for ``edges`` [n, 2] (the layout of ``split_edge[...]['edge']``), batched like
    ``PermIterator(.., training=False)``; returns a [n] fp32 tensor on the device.  The host stays at
    most ``run_ahead`` batches ahead of the GPU: an unbounded backlog makes the HIP runtime block the
    host until the queue has drained completely.

    ``group`` (a process group, or True for the default one): the ``PermIterator`` batches are DEALT round robin to the
    ranks — every batch stays whole on one GPU, so its batch-coupled normalisation is the single-device one and no
    histogram travels; ONE all-gather of the scores closes the split (``dist.gather_dealt``).  Every rank returns all n
    scores, bit-equal to the single-process call."""
    from .utils import adjoverlap
    if predictor.training:
        raise RuntimeError("score_edges is the eval path; call predictor.eval() first")
    h = h.contiguous()
    outs, done = [], []
    if edges.shape[0] == 0:
        return h.new_zeros(0)
    # the adjacency's lazy caches (bit rows, longest row) are built HERE, on the caller's stream, before the loop forks its
    # side streams: two phase-A streams must never race on a half-built cache (the caches also carry their own events)
    adj.warm(walk=False)
    if adj2 is not None and adj2.rows_on_demand():          # (a product formed under autograd: its rows are built on ONE stream)
        adj2.product_bit_rows()
    # the ids of the whole split are bounds-checked once; the batches then run without a host sync each
    with ops.prevalidated(edges[:, 0], edges[:, 1], adj.size(0), adj.size(0)):
        perms = list(PermIterator(edges.device, edges.shape[0], batch_size, training=False))
        mine, world = _dealt(perms, group)
        graphed = _graphed(predictor, h, adj, lambda e: (adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e)), batch_size, args, len(mine))

        def begin(it):
            e = edges[perms[mine[it]]].t()
            if graphed is not None:
                return graphed.begin(it, e)
            e = e.contiguous()
            return predictor.begin(h, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e, slot=it, args=args)

        def flow(it):
            if len(done) >= max(run_ahead, 1):
                done.pop(0).synchronize()

        fin = graphed.finish if graphed is not None else (lambda tok: predictor.finish(h, tok, args))
        for out in overlapped_steps(begin, fin, len(mine), before_step=flow, batch=batch_size, device=h.device):
            outs.append(out.reshape(-1).clone() if graphed is not None else out.reshape(-1))     # (a replay's scores live in the graph's pool)
            done.append(torch.cuda.current_stream(h.device).record_event())
    if world > 1:
        from .dist import gather_dealt
        scores = gather_dealt(outs, [int(p.numel()) for p in perms], None if group is True else group)
    else:
        scores = torch.cat(outs, dim=0)
    predictor.check_errors()       # the batches' sticky status words, read once per split (flag capacity, scan state)
    return scores


@torch.no_grad()
def score_mrr_split(predictor, h: Tensor, adj, source: Tensor, target: Tensor, target_neg: Tensor,
                    batch_size: int, args=None, evaluator=None, group=None):
    """``test_split`` of the citation2 driver (NeighborOverlapCitation2.py:227-254): positives
    (source, target) and, per positive, ``target_neg.shape[1]`` negatives sharing its source, each
    scored through ``get_cn1_cn2`` in ``PermIterator(.., training=False)`` batches — same batch
    composition as the reference, scores kept on the device.  Returns (pos_pred [n], neg_pred [n, n_neg])
    or, with an ``evaluator``, the mean of its ``mrr_list``.  ``group``: whole batches dealt over the ranks, as in
    ``score_edges`` — the partition that suits the drivers' 2 048-candidate batches (a dense histogram all-reduce per such
    batch costs more link time than the batch costs compute)."""
    from .utils import get_cn1_cn2
    if predictor.training:
        raise RuntimeError("score_mrr_split is the eval path; call predictor.eval() first")

    def run(src_all: Tensor, dst_all: Tensor) -> Tensor:
        outs = []
        if src_all.numel() == 0:
            return h.new_zeros(0)
        adj.warm(walk=True)            # (degree sums of the two-sided sweep: built on the caller's stream, before the side streams fork)
        with ops.prevalidated(src_all, dst_all, adj.size(0), adj.size(0)):
            perms = list(PermIterator(src_all.device, src_all.shape[0], batch_size, training=False))
            mine, world = _dealt(perms, group)
            graphed = _graphed(predictor, h, adj, lambda e: get_cn1_cn2(adj, e), batch_size, args, len(mine))

            def begin(it):
                e = torch.stack((src_all[perms[mine[it]]], dst_all[perms[mine[it]]]))
                if graphed is not None:
                    return graphed.begin(it, e)
                cn1, cn2 = get_cn1_cn2(adj, e)
                return predictor.begin(h, adj, cn1, cn2, e, slot=it, args=args)

            fin = graphed.finish if graphed is not None else (lambda tok: predictor.finish(h, tok, args))
            for out in overlapped_steps(begin, fin, len(mine), batch=batch_size, device=h.device):
                outs.append(out.reshape(-1).clone() if graphed is not None else out.reshape(-1))
        if world > 1:
            from .dist import gather_dealt
            scores = gather_dealt(outs, [int(p.numel()) for p in perms], None if group is True else group)
        else:
            scores = torch.cat(outs, dim=0)
        predictor.check_errors()
        return scores

    pos_pred = run(source, target)
    n_neg = target_neg.shape[1]
    neg_pred = run(source.view(-1, 1).repeat(1, n_neg).view(-1), target_neg.reshape(-1)).view(-1, n_neg)
    if evaluator is None:
        return pos_pred, neg_pred
    return evaluator.eval({"y_pred_pos": pos_pred, "y_pred_neg": neg_pred})["mrr_list"].mean().item()


class GraphedPhases:
    """Phase A and phase B of a scoring loop's batches as captured HIP graphs, one pair per scratch set.

    A scoring loop enqueues ~25 launches per batch through ctypes — 0.2 – 0.3 ms of host time, which is the whole step at
    the drivers' 2 048-candidate batches and on the dense ddi shape once the pipeline keeps four to eight batches in flight
    (host enqueue 0.27 – 0.32 ms against a 0.33 ms step: the GPU waits for Python).  The batches of one size that use one scratch
    set run the same launches on the same buffers — only the candidate ids differ — so after two eager uses of a set (buffers,
    panels, cached decisions) its phase A (``predictor.begin``: prep, intersection, weights, class order, schedule, pooling)
    and its phase B (``predictor.finish``: the heads) are captured once and replayed: per batch one copy of the ids into the
    set's static buffer and two graph launches.  The events that order the phases across streams stay outside the graphs.

    ``make_handles(e)`` -> (cn1, cn2) handles for the static id buffer ``e`` [2, B].  A batch of another size (the ragged
    tail of a split) runs eagerly.  The score tensor a replay returns is the graph's own: valid until the set's next batch,
    so the loops copy it out.  Unsharded only (a collective is not captured)."""

    WARM = 2

    def __init__(self, predictor, h: Tensor, adj, make_handles, batch_size: int, args=None):
        self.pred, self.h, self.adj, self.handles, self.B, self.args = predictor, h, adj, make_handles, int(batch_size), args
        self.n_sets = max(2, int(ops.overlap_depth), int(ops.overlap_depth_small))
        self.slots: dict = {}

    def begin(self, it: int, e: Tensor):
        """``e``: the batch's candidate ids, [2, b] (any strides).  Returns a token for ``finish``."""
        b = int(e.shape[1])
        s = it % self.n_sets
        if b != self.B or getattr(self.pred, "_sharded", False):
            e = e.contiguous()
            return ("eager", self.pred.begin(self.h, self.adj, *self.handles(e), e, slot=s, args=self.args), None)
        sl = self.slots.get(s)
        if sl is None:
            sl = self.slots[s] = dict(calls=0, edges=torch.zeros(2, self.B, dtype=torch.int64, device=e.device), gA=None, gB=None,
                                      tok=None, out=None)
        sl["calls"] += 1
        sl["edges"].copy_(e, non_blocking=True)
        timed = ops.stage_timer is not None and getattr(ops.stage_timer, "active", False)      # (a step whose stages are being timed runs launch by launch)
        if sl["calls"] <= self.WARM or timed:
            return ("eager", self.pred.begin(self.h, self.adj, *self.handles(sl["edges"]), sl["edges"], slot=s, args=self.args), sl)
        if sl["gA"] is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sl["tok"] = self.pred.begin(self.h, self.adj, *self.handles(sl["edges"]), sl["edges"], slot=s, args=self.args)
            sl["gA"] = g
        sl["gA"].replay()
        return ("graph", sl["tok"], sl)

    def finish(self, token) -> Tensor:
        kind, tok, sl = token
        if kind == "eager" or (ops.stage_timer is not None and getattr(ops.stage_timer, "active", False)):
            return self.pred.finish(self.h, tok, self.args)
        if sl["gB"] is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sl["out"] = self.pred.finish(self.h, tok, self.args)
            sl["gB"] = g
        sl["gB"].replay()
        return sl["out"]


_side_streams: dict = {}


def _side_stream(index: int, q: int = 0):
    s = _side_streams.get((index, q))
    if s is None:
        s = _side_streams[(index, q)] = torch.cuda.Stream(device=index)
    return s


def overlapped_steps(begin, finish, n_steps: int, before_step=None, after_step=None, overlap=None, batch: Optional[int] = None,
                     device=None):
    """Generator over ``finish(begin(it))`` for it = 0 .. n_steps - 1 with TWO batches in flight: ``begin(it + 1)`` (the
    predictor's phase A: the intersection pass, scratch set ``it & 1``) is enqueued before ``finish(it)`` (phase B:
    weights, pooling, heads) — and, on a GPU, on a SECOND HIP stream, so that the latency-bound intersection kernels of
    the next batch run beside the pooling and the heads of the current one instead of in front of them (measured, one
    MI355X: collab shape 0.507 -> 0.474 ms per batch, ddi 0.629 -> 0.488, ppa 0.564 -> 0.439; same scores).  Events order
    the two streams: phase B of batch t waits for phase A of batch t; phase A of batch t + 2 waits for phase B of batch t
    (they share a scratch set).  ``overlap`` = None: ``ops.overlap_streams`` where CUDA/HIP is up and the batch has at least
    ``ops.overlap_min_batch`` candidates (``batch``: per-rank batch size, if known), else one stream.  ``device``: the device
    of the tensors the two phases work on (default: the current device) — the streams and events are created there."""
    if overlap is None:
        overlap = (bool(ops.overlap_streams) and torch.cuda.is_available() and torch.cuda.is_initialized()
                   and (batch is None or batch >= ops.overlap_min_batch))
    if not overlap:
        ahead = None
        for it in range(n_steps):
            if before_step is not None:
                before_step(it)
            tok = ahead if ahead is not None else begin(it)
            ahead = begin(it + 1) if it + 1 < n_steps else None
            yield finish(tok)
            if after_step is not None:
                after_step(it)
        return
    index = torch.cuda.current_device() if device is None else (torch.device(device).index if torch.device(device).index is not None
                                                                 else torch.cuda.current_device())
    # (tensors on another device than the current one, ADVICE r3: the ops launch on THEIR device's current stream —
    # ops._on_device — so the loop's main stream, side streams and events are that device's, whatever the current device is)
    main = torch.cuda.current_stream(index)
    # batches in flight: depth - 1 in phase A, one in phase B
    depth = ops.loop_depth(batch)
    sides = [_side_stream(index, q) for q in range(depth - 1)]
    begun = [torch.cuda.Event() for _ in range(depth)]
    done = [torch.cuda.Event() for _ in range(depth)]
    for sd in sides:
        sd.wait_stream(main)

    def phase_a(it):
        sd = sides[it % (depth - 1)]
        with torch.cuda.stream(sd):
            if it >= depth:
                sd.wait_event(done[it % depth])        # phase B of batch it - depth read the scratch set this batch overwrites
            was_active, ops._overlap_active = ops._overlap_active, True      # (predictor.begin: the sharded collective then starts in
            try:                                                            # phase B; set around begin() only — the consumer's own
                tok = begin(it)                                             # calls between two yields must not see it)
            finally:
                ops._overlap_active = was_active
            begun[it % depth].record(sd)
        return tok

    ahead = []                                             # tokens of the batches already in phase A, oldest first
    try:
        for it in range(n_steps):
            if before_step is not None:
                before_step(it)
            while len(ahead) < depth and it + len(ahead) < n_steps:
                ahead.append(phase_a(it + len(ahead)))
            tok = ahead.pop(0)
            main.wait_event(begun[it % depth])
            out = finish(tok)
            done[it % depth].record(main)
            yield out
            if after_step is not None:
                after_step(it)
    finally:
        for sd in sides:
            main.wait_stream(sd)


def pipelined_shard_loop(begin, finish, n_steps: int, batch_total: int, group=None, gather_at_end: bool = True,
                         before_step=None, after_step=None, overlap=None):
    """The edge-sharded scoring loop (bench.py's timed region at N > 1; one rank's view).  ``begin(it)`` = the predictor's
    phase A of batch ``it`` (intersection pass + START of the histogram all-reduce) and returns a token, ``finish(token)``
    = phase B (wait, weights, pooling, heads) and returns this rank's scores ``[b, C]``.  Two batches are in flight:
    ``begin(it + 1)`` is enqueued before ``finish(it)``, so the all-reduce of batch ``it`` runs beside the intersection work
    of batch ``it + 1``.  Scores: ``gather_at_end`` keeps every rank's slices and closes the loop with ONE all-gather (what a
    scoring loop that consumes its scores afterwards wants; needs equal slices), else each batch is gathered asynchronously
    and waited for one step later.  Returns (scores of all steps ``[n_steps, batch_total, C]`` in batch order, pattern
    string).  ``before_step(it)`` / ``after_step(it)``: the caller's flow control and timers."""
    from .dist import gather_scores
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    if gather_at_end and batch_total % world != 0:
        gather_at_end = False                                  # ragged slices: per-batch gather pads and trims
    kept, outs, pending = [], [], None
    streams, deep = "", 2
    if overlap is None:
        overlap = (bool(ops.overlap_streams) and torch.cuda.is_available() and torch.cuda.is_initialized()
                   and batch_total // max(world, 1) >= ops.overlap_min_batch)
    if overlap:
        deep = ops.loop_depth(batch_total // max(world, 1))
        streams = ("; phase A (intersection pass) of batch t + 1 on a second HIP stream beside phase B of batch t" if deep == 2 else
                   f"; phase A (intersection pass) of batches t + 1 .. t + {deep - 1} on {deep - 1} more HIP streams beside phase B of batch t")
    it = -1
    for loc in overlapped_steps(begin, finish, n_steps, before_step, None, overlap, batch=batch_total // max(world, 1)):
        it += 1
        if gather_at_end:
            kept.append(loc)
        else:
            if pending is not None:
                pending[1].wait() if pending[1] is not None else None
                outs.append(pending[0])
            pending = gather_scores(loc, batch_total, group, async_op=True)
        if after_step is not None:
            after_step(it)
    if pending is not None:
        pending[1].wait() if pending[1] is not None else None
        outs.append(pending[0])
    if gather_at_end:
        if not kept:
            return None, "one all-gather at the end" + streams
        per = kept[0].shape[0]
        allsc = gather_scores(torch.cat(kept, 0), len(kept) * batch_total, group)       # rank-major: [world][n_steps][per]
        w = allsc.shape[0] // (len(kept) * per)
        scores = allsc.view(w, len(kept), per, -1).permute(1, 0, 2, 3).reshape(len(kept), w * per, -1)
        return scores, f"{deep} batches in flight (begin/finish), local scores kept, ONE all-gather closes the loop" + streams
    return (torch.stack(outs, 0) if outs else None), f"{deep} batches in flight (begin/finish), one async all-gather per batch" + streams


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
