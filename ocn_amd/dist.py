"""Edge-batch sharding over the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference is single-device; this is the new data-parallel layer north_star asks
for.  The adjacency, A², the embeddings and the weights are replicated; a candidate batch is cut
into contiguous slices, one per rank.  Because the predictors normalise per column over the WHOLE
batch (``cn.sum(dim=0)``, model.py:2261,3114), ranks exchange exactly one thing before pooling: the
per-column histograms {n1, n2, n_union, walks} (packed int64, sum all-reduce — integer, so exact and
order-independent; cn5's ``scale`` and — for a fresh model, innerprod == 0 — S2 are functions of those counts
and need no extra collective; with a trained innerprod S2 depends on the reference's entry order and is chained
through the ranks, ``ring_colsum``).  Scores come back with one all-gather.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


force_collectives = False        # tests: run the collectives even with one rank (a one-GPU box can then exercise the RCCL path)


def _solo(group) -> bool:
    return (not (dist.is_available() and dist.is_initialized())) or (dist.get_world_size(group) == 1 and not force_collectives)


def shard_bounds(total: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced slices (first ``total % world`` ranks get one extra edge)."""
    base, rem = divmod(total, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


def _host_staged(t: Tensor, group) -> bool:
    """gloo rehearsals (several ranks on one GPU, CPU tests) move device tensors through the host;
    the production backend is RCCL ("nccl"), which works on device memory directly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def allreduce_hist(hist: Tensor, group=None, valued: bool = True) -> Tensor:
    """In-place sum of the packed per-column histograms (int64 [N, 2]: integer fields, so the sum is
    exact and order-independent) over the edge shards.  ``valued=False`` (pattern route: the second
    word, the walk-count sums, is all zero) moves only the packed word: half the bytes on the wire."""
    if not _solo(group):
        buf = hist if valued else hist[:, 0].contiguous()
        if _host_staged(buf, group):
            tmp = buf.cpu()
            dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
            buf.copy_(tmp)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        if not valued:
            hist[:, 0].copy_(buf)
    return hist


# Sparse exchange (VERDICT r3 #7b).  A 2 048-candidate slice of a citation2-shaped batch touches ~1e5 of 2.9 M columns, yet the
# dense all-reduce moves the whole [N, 2] int64 buffer: 47 MB per batch and rank.  Here only the touched columns travel, as
# (column, packed counts, walk-count sum) triples: one all-gather of the per-rank counts (read on the host: the only sync),
# one all-gather of the triples padded to the longest list, and every rank adds the other ranks' triples into its own
# histogram — the same integer sums as the all-reduce, in any order.  Whether it pays is decided from the gathered counts, by
# every rank alike: if the longest list reaches N / sparse_exchange_min_ratio columns the dense all-reduce runs instead.
sparse_exchange = "auto"         # "auto" | True | False: when edge-sharded batches exchange touched columns only
sparse_exchange_min_cols = 1 << 19      # auto: only histograms of at least this many columns (collab's 236 k stay dense: 3.8 MB) ...
sparse_exchange_slice_ratio = 512       # ... and slices of fewer than N / this candidates (citation2: 2 048 of 2.9 M -> yes; ppa: 2 048 of 576 k -> no)
sparse_exchange_min_ratio = 8           # fall back to the dense all-reduce when a rank's touched columns reach N / this


def sparse_exchange_wanted(n_cols: int, slice_edges: int) -> bool:
    if sparse_exchange == "auto":
        return n_cols >= sparse_exchange_min_cols and slice_edges * sparse_exchange_slice_ratio < n_cols
    return bool(sparse_exchange)


def allreduce_hist_sparse(hist: Tensor, group=None) -> str:
    """In-place sum of the packed histograms over the edge shards, moving the touched columns only.  Returns "sparse" or
    "dense" (the fallback every rank takes together when a list is too long to pay).  One host sync (the counts)."""
    if _solo(group):
        return "sparse"
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    N = hist.shape[0]
    host = _host_staged(hist, group) or not hist.is_cuda
    idx = torch.nonzero((hist[:, 0] != 0) | (hist[:, 1] != 0)).flatten()          # (sizes the list: a host sync)
    k = int(idx.numel())
    cnt = torch.tensor([k], dtype=torch.int64, device="cpu" if host else hist.device)
    counts = [torch.empty_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    kmax = max(counts)
    if kmax * sparse_exchange_min_ratio >= N:                                     # every rank sees the same counts: one decision
        allreduce_hist(hist, group, valued=True)
        return "dense"
    if kmax == 0:
        return "sparse"
    mine = torch.zeros((kmax, 3), dtype=torch.int64, device=hist.device)
    mine[:k, 0] = idx
    mine[:k, 1:] = hist[idx]
    if host:
        parts = [torch.empty((kmax, 3), dtype=torch.int64) for _ in range(world)]
        dist.all_gather(parts, mine.cpu(), group=group)
        allp = torch.stack(parts, 0).to(hist.device)
    else:
        allp = torch.empty((world, kmax, 3), dtype=torch.int64, device=hist.device)
        dist.all_gather_into_tensor(allp.view(world * kmax, 3), mine, group=group)
    for r in range(world):
        if r != rank and counts[r]:
            hist.index_add_(0, allp[r, :counts[r], 0], allp[r, :counts[r], 1:])
    return "sparse"


def allreduce_hist_start(hist: Tensor, group=None, valued: bool = True, slice_edges: Optional[int] = None):
    """``allreduce_hist`` split in two so that work which does not read the histogram (the class ordering of the batch
    rows) runs between them: on RCCL the collective is enqueued asynchronously (its own stream, ordered after the
    kernels already on the current one) and a handle is returned for ``allreduce_hist_finish``; gloo rehearsals and the
    one-rank case finish here and return None.  ``slice_edges``: this rank's candidates in the batch — lets the exchange
    take the sparse form (``allreduce_hist_sparse``) where that pays (``sparse_exchange_wanted``)."""
    if _solo(group):
        return None
    if slice_edges is not None and sparse_exchange_wanted(hist.shape[0], slice_edges):
        allreduce_hist_sparse(hist, group)          # (touched columns only; finishes here: its counts are read on the host)
        return None
    buf = hist if valued else hist[:, 0].contiguous()
    if _host_staged(buf, group) or not buf.is_cuda:
        allreduce_hist(hist, group, valued)
        return None
    work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return work, buf, hist, valued


def allreduce_hist_finish(handle) -> None:
    """Make the current stream wait for the histogram sum started by ``allreduce_hist_start``."""
    if handle is None:
        return
    work, buf, hist, valued = handle
    work.wait()                                  # a stream dependency, not a host wait
    if not valued:
        hist[:, 0].copy_(buf)


def ring_colsum(run, n_cols: int, device, group=None) -> Tensor:
    """Order-exact S2 of an edge-sharded cn5 batch (innerprod != 0).  The reference adds a column's entries in
    ascending batch-row order (model.py:2405-2406) and the shards are contiguous row ranges, so the sum is a chain
    through the ranks: rank 0 starts from zeros, every rank continues from its predecessor's running sums
    (``run(init)`` = ocn_cn_colsum_exact with ``s2_init``), the last rank holds the single-device result and
    broadcasts it.  One [N] fp32 vector per hop — serial in the world size by construction, which is the price of
    the reference's summation order; a fresh model (innerprod == 0) never takes this path."""
    if _solo(group):
        return run(None)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    ranks = [dist.get_global_rank(group, r) if group is not None else r for r in range(world)]
    host = dist.get_backend(group) == "gloo"
    stage = torch.empty(n_cols, dtype=torch.float32, device="cpu" if host else device)
    if rank == 0:
        init = torch.zeros(n_cols, dtype=torch.float32, device=device)
    else:
        dist.recv(stage, src=ranks[rank - 1], group=group)
        init = stage.to(device)
    s2 = run(init)
    if rank < world - 1:
        dist.send(s2.cpu() if host else s2, dst=ranks[rank + 1], group=group)
    out = s2.cpu() if host else s2
    dist.broadcast(out, src=ranks[world - 1], group=group)
    if host:
        s2.copy_(out)
    return s2


def check_global_batch(total: int) -> None:
    """The column counts travel in 21-bit fields (ocn_hip.h): they must hold the GLOBAL batch."""
    from .ops import MAX_BATCH
    if total > MAX_BATCH:
        raise ValueError(f"global candidate batch of {total} edges exceeds the histogram field width ({MAX_BATCH})")


def gather_scores(local: Tensor, total: int, group=None, async_op: bool = False):
    """All-gather the per-edge scores of every shard back into batch order: [total, C].
    ``async_op`` (RCCL, equal slices): returns ``(out, work)`` with the collective still in flight — a scoring loop
    that only consumes the scores later (``bench.py``, ``pipeline.score_edges``) calls ``work.wait()`` then, and the next
    batch's kernels do not queue behind this batch's all-gather."""
    if _solo(group):
        return (local, None) if async_op else local
    if async_op:
        world = dist.get_world_size(group)
        if total % world == 0 and local.is_contiguous() and not _host_staged(local, group) and local.is_cuda:
            out = local.new_empty((total,) + tuple(local.shape[1:]))
            return out, dist.all_gather_into_tensor(out, local, group=group, async_op=True)
        return gather_scores(local, total, group), None
    world = dist.get_world_size(group)
    bounds = shard_bounds(total, world)
    width = max(e - s for s, e in bounds)
    if local.shape[0] == width and local.is_contiguous():
        pad = local                              # equal slices (total % world == 0): nothing to pad
    else:
        pad = local.new_zeros((width,) + tuple(local.shape[1:]))
        pad[: local.shape[0]] = local
    if _host_staged(local, group):
        parts = [torch.empty_like(pad, device="cpu") for _ in range(world)]
        dist.all_gather(parts, pad.cpu(), group=group)
        out = torch.cat(parts, dim=0).to(local.device)
    else:
        out = local.new_empty((world * width,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, pad, group=group)
    if all(e - s == width for s, e in bounds):
        return out
    return torch.cat([out[r * width: r * width + (e - s)] for r, (s, e) in enumerate(bounds)], dim=0)


# ---------------------------------------------------------------------------------------------
# whole-batch dealing: the partition for the drivers' loops over independent candidate batches
# ---------------------------------------------------------------------------------------------
def deal_batches(n_batches: int, world: int, rank: int) -> List[int]:
    """The batches of a scoring loop that rank ``rank`` owns: round robin.  The reference's test loops
    (NeighborOverlap_large.py:121-159, NeighborOverlap_large_ppa.py:98-133, NeighborOverlapCitation2.py:227-254) score
    independent ``PermIterator`` batches; a batch's column normalisation couples only ITS candidates, so a batch
    kept whole on one GPU needs no exchange at all and its scores equal the single-device ones bit for bit."""
    return list(range(rank, n_batches, world))


def gather_dealt(local: List[Tensor], sizes: List[int], group=None) -> Tensor:
    """Scores of a dealt scoring loop back in split order.  ``local``: this rank's per-batch score vectors (its
    ``deal_batches`` share, in that order, each ``[sizes[b]]`` or ``[sizes[b], C]``); ``sizes``: the length of EVERY batch of the
    loop.  ONE all-gather of the ranks' concatenated scores, padded to the longest share.  Returns ``[sum(sizes), ...]``."""
    if _solo(group):
        return torch.cat(local, 0) if local else torch.zeros(0)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    shares = [deal_batches(len(sizes), world, r) for r in range(world)]
    lens = [sum(sizes[b] for b in sh) for sh in shares]
    width = max(lens)
    ref = local[0] if local else None
    if ref is None:                                     # (a rank without a batch still takes part in the collective)
        raise ValueError("gather_dealt: every rank needs at least one batch (fewer batches than ranks: score them undealt)")
    tail = tuple(ref.shape[1:])
    mine = torch.cat(local, 0)
    if mine.shape[0] != lens[rank]:
        raise ValueError("gather_dealt: local scores do not match this rank's share of `sizes`")
    pad = mine if mine.shape[0] == width else torch.cat([mine, mine.new_zeros((width - mine.shape[0],) + tail)], 0)
    pad = pad.contiguous()
    if _host_staged(pad, group) or not pad.is_cuda:
        parts = [torch.empty(pad.shape, dtype=pad.dtype) for _ in range(world)]
        dist.all_gather(parts, pad.cpu(), group=group)
        allsc = torch.stack(parts, 0).to(ref.device)
    else:
        allsc = pad.new_empty((world, width) + tail)
        dist.all_gather_into_tensor(allsc.view((world * width,) + tail), pad, group=group)
    out, cursor = [None] * len(sizes), [0] * world
    for b in range(len(sizes)):                         # batch b is rank b % world's next piece
        r = b % world
        out[b] = allsc[r, cursor[r]: cursor[r] + sizes[b]]
        cursor[r] += sizes[b]
    return torch.cat(out, 0)


def sharded_predict(predictor, h: Tensor, adj, adj2, edges: Tensor, args=None, group=None) -> Tensor:
    """Score the global candidate batch ``edges`` [2, B] with the batch cut over the ranks of
    ``group``; every rank returns the full [B, 1] score vector, equal to the single-device result
    on the same batch.  In eval under no_grad it takes the predictor's two-phase path (``begin`` / ``finish``): the
    histogram sum runs on the whole interleaved buffer — no copy-out / copy-back of the packed word — as in the
    pipelined loop (``pipeline.pipelined_shard_loop``)."""
    from .utils import adjoverlap
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    check_global_batch(edges.shape[1])
    s, e = shard_bounds(edges.shape[1], world)[rank]
    mine = edges[:, s:e].contiguous()
    predictor.set_edge_sharding(group if world > 1 else None, enabled=world > 1 or force_collectives)
    try:
        c1, c2 = adjoverlap(adj, adj, mine), adjoverlap(adj, adj2, mine)
        if not predictor.training and not torch.is_grad_enabled():
            local = predictor.finish(h, predictor.begin(h, adj, c1, c2, mine), args)
        else:
            local = predictor(h, adj, c1, c2, mine, args)
    finally:
        predictor.set_edge_sharding(None, enabled=False)
    return gather_scores(local, edges.shape[1], group)


# ---------------------------------------------------------------------------------------------
# dry run: what an N-rank run holds per GPU and moves per batch (no GPU, no process group needed)
# ---------------------------------------------------------------------------------------------
XGMI_LINK_GBPS = 153.0           # per link and direction; 7 links per GPU, full mesh (MI355X_MICROARCH.md / task notes)
SHAPES = {
    # name: nodes, stored entries of A, entries of A² (None: walk route, no A²), H, candidates per rank, valued histogram
    "collab": dict(n=235_868, nnz=2_360_000, nnz2=127_996_321, H=256, batch=65_536, walk=False, bitrows=True, mean_deg=38),
    "citation2": dict(n=2_927_963, nnz=60_000_000, nnz2=None, H=32, batch=2_048, walk=True, bitrows=False, mean_deg=90),
}


def shard_plan(shape: str = "collab", worlds=(1, 2, 4, 8)) -> List[dict]:
    """Per world size: bytes resident per GPU (everything but the candidate slice is replicated), bytes per collective
    and the time they take on xGMI if RCCL reaches the link rate — a floor, not a measurement (no N > 1 run exists yet).
    All-reduce of S bytes moves 2 (w-1)/w S per rank; spread over the min(w-1, 7) links of the mesh (direct
    reduce-scatter + all-gather) or over ONE link (ring).  Weak scaling: ``batch`` candidates per rank."""
    c = SHAPES[shape]
    n, H, B = c["n"], c["H"], c["batch"]
    csr = (n + 1) * 8 + c["nnz"] * 4
    a2 = 0 if c["nnz2"] is None else (n + 1) * 8 + c["nnz2"] * 4
    bits = n * ((n + 31) // 32) * 4 if c["bitrows"] else 0
    hbytes = n * H * 4
    hist = n * 16                                           # interleaved {packed counts, walk-count sums}
    scratch = 2 * (B * c["mean_deg"] * (5 if c["walk"] else 1) + B * 32 + hist + 3 * B * H * 4) + (256 * 4 * 2 * H * 32 * 4 if H >= 128 else 0)
    rows = []
    for w in worlds:
        ar = 0.0 if w == 1 else 2.0 * (w - 1) / w * hist
        ag = 0.0 if w == 1 else (w - 1) * B * 4             # every rank receives the other ranks' slices
        links = max(min(w - 1, 7), 1)
        rows.append(dict(shape=shape, world=w, global_batch=w * B,
                         resident_GB=(csr + a2 + bits + hbytes + scratch) / 1e9,
                         graph_MB=csr / 1e6, adj2_MB=a2 / 1e6, bitrows_GB=bits / 1e9, h_MB=hbytes / 1e6, scratch_MB=scratch / 1e6,
                         hist_allreduce_MB_per_rank=ar / 1e6, score_allgather_KB_per_rank=ag / 1e3,
                         allreduce_us_mesh=ar / (links * XGMI_LINK_GBPS * 1e3), allreduce_us_ring=ar / (XGMI_LINK_GBPS * 1e3),
                         allgather_us_mesh=ag / (links * XGMI_LINK_GBPS * 1e3)))
    return rows


def partition_plan(shape: str = "citation2", worlds=(1, 2, 4, 8), touched_cols: Optional[int] = None) -> List[dict]:
    """What each partition of the scoring loop moves per candidate batch and rank (bytes; a dry run like ``shard_plan``):
    ``intra_dense`` — the batch cut over the ranks, dense [N, 2] int64 histogram all-reduce (``sharded_predict``,
    ``pipelined_shard_loop``); ``intra_sparse`` — the same cut, but only the touched columns' (column, counts, walks) triples
    all-gathered (24 B per triple: ``allreduce_hist_sparse``); ``dealt`` — whole batches dealt round robin
    (``deal_batches``): nothing per batch, one all-gather of the scores per split.  ``touched_cols``: distinct columns with a
    union entry per rank slice (default: batch x mean degree, an upper bound)."""
    c = SHAPES[shape]
    n, B = c["n"], c["batch"]
    touched = touched_cols if touched_cols is not None else min(n, B * c["mean_deg"])
    rows = []
    for w in worlds:
        dense = 0.0 if w == 1 else 2.0 * (w - 1) / w * n * 16
        sparse = 0.0 if w == 1 else (w - 1) * touched * 24
        rows.append(dict(shape=shape, world=w, intra_dense_MB=dense / 1e6, intra_sparse_MB=sparse / 1e6, dealt_MB=0.0,
                         dealt_final_allgather_KB_per_batch=0.0 if w == 1 else (w - 1) * B * 4 / 1e3,
                         sparse_pays=bool(w > 1 and touched < n / 8)))
    return rows


def shard_plan_markdown(shapes=("collab", "citation2"), worlds=(1, 2, 4, 8)) -> str:
    out = ["| shape | ranks | global batch | resident / GPU | of which A² + bit rows | h | histogram all-reduce / rank | floor, mesh / ring | "
           "score all-gather / rank | floor |", "|---|---|---|---|---|---|---|---|---|---|"]
    for sh in shapes:
        for r in shard_plan(sh, worlds):
            out.append(f"| {sh} | {r['world']} | {r['global_batch']} | {r['resident_GB']:.2f} GB | {r['adj2_MB'] / 1e3 + r['bitrows_GB']:.2f} GB | "
                       f"{r['h_MB']:.0f} MB | {r['hist_allreduce_MB_per_rank']:.2f} MB | {r['allreduce_us_mesh']:.1f} / {r['allreduce_us_ring']:.1f} µs | "
                       f"{r['score_allgather_KB_per_rank']:.0f} KB | {r['allgather_us_mesh']:.2f} µs |")
    return "\n".join(out)


if __name__ == "__main__":
    print(shard_plan_markdown())
    print()
    print("| shape | ranks | intra-batch, dense all-reduce / rank | intra-batch, sparse triples / rank | whole batches dealt: per batch | + scores at the end, per batch |")
    print("|---|---|---|---|---|---|")
    for sh, tc in (("collab", 180_000), ("citation2", 100_000)):
        for r in partition_plan(sh, touched_cols=tc):
            print(f"| {sh} | {r['world']} | {r['intra_dense_MB']:.2f} MB | {r['intra_sparse_MB']:.2f} MB | {r['dealt_MB']:.0f} | {r['dealt_final_allgather_KB_per_batch']:.0f} KB |")
