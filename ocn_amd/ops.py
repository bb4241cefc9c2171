"""Host-side wrappers of the C ABI: argument validation + raw-pointer calls.

torch is used for device memory and the current stream only.  Every function here requires
CUDA(HIP) tensors and raises if the library is missing — there is no CPU path.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import check, ptr, stream_ptr

FLAGS_NOSYNC_LIMIT = 1 << 30     # bytes of flag buffer we are willing to over-allocate to avoid a host sync
validate_indices = True          # bounds-check candidate edges on the host side (one sync per batch)
stage_timer = None               # optional object with .mark(name): bench.py records HIP events between stages
LN_WIDTHS = (16, 32, 64, 128, 256, 512)   # row widths of the lane-group kernels (LayerNorm, pooling backward)
a1_bitmap_max_bytes = int(os.environ.get("OCN_A1_BITMAP_MAX_BYTES", 64 << 20))   # keep A itself as dense bit rows too when they fit this (ogbl-ddi: 2.3 MB): cn1 membership = one probe
a2_bitmap_max_bytes = 16 << 30   # keep A·B also as dense bit rows when n_rows * n_cols / 8 fits this budget
skip_zero_rows = True            # heads: skip the layers whose pooled input row is all zero (class-major rows)
skip_zero_min_share = 0.15       # ... and when fewer than this share of the head rows could be skipped (probed asynchronously)
skip_zero_backoff = 32           # batches evaluated plainly before the share is probed again
skip_zero_min_batch = 4096       # below this the extra small launches cost more than the skipped rows save
walk_two_sided = True            # walk route: sweep each candidate from its cheaper endpoint (needs nds of the adjacency)
walk_share_min = 2               # walk route, B <= 4096: candidates sharing a source are swept together from this group size on (0 = never)
sort_edges_min_batch = 4096      # batches at least this large are processed in src order (L2 reuse of shared rows)
heavy_first = os.environ.get("OCN_HEAVY_FIRST", "1") != "0"   # ... and the pooling (H = 256) visits its slot groups longest first (ocn_cn_flags' gcost -> ocn_gather_schedule)
sched_segment = int(os.environ.get("OCN_SCHED_SEG", 0))        # ... inside segments of this many groups of an XCD's eighth (0 = the whole eighth)
overlap_depth = int(os.environ.get("OCN_OVERLAP_DEPTH", 4))   # scratch sets of a predictor = the most batches a scoring loop keeps in flight (round 4: four, with the pooling in phase A)
overlap_depth_small = int(os.environ.get("OCN_OVERLAP_DEPTH_SMALL", 8))   # ... and for batches of at most overlap_small_batch candidates (the drivers' 2 048: latency chains, not bandwidth — citation2 shape 5.58 -> 5.83 M edges/s, ppa unchanged)
overlap_small_batch = 4096
overlap_deep_max_batch = int(os.environ.get("OCN_OVERLAP_DEEP_MAX_BATCH", 1 << 30))   # ... which it does for batches up to this size: every size by
                                 # default.  Two intersection passes beside one pooling + heads pay where phase A is the longer one: the drivers'
                                 # 2 048-candidate walk-route batches (citation2 shape 3.96 -> 4.81 M edges/s) and ANY trained cn5 model, whose
                                 # order-exact column sums sit in phase A (collab shape 104 -> 126 M); a fresh model's large batches lose < 1 %
# Sharded scoring loops: where a batch's histogram all-reduce starts.  None = by the loop: in phase B (on the caller's stream, the
# class ordering and the NEXT batch's phase A on the other stream fill the wait) when the loop overlaps two streams — started from
# the side stream the collective cost 25 % of the step in the one-rank rehearsal (105 M against 140 M edges/s) — and in phase A (a
# whole step ahead, hidden behind the next intersection pass in stream order) on one stream.  OCN_REDUCE_IN_FINISH=0|1 forces it.
shard_reduce_in_finish = {"0": False, "1": True}.get(os.environ.get("OCN_REDUCE_IN_FINISH", ""), None)
_overlap_active = False          # set by pipeline.overlapped_steps while it runs phase A on side streams
phase_a_extras = os.environ.get("OCN_PHASE_A_EXTRAS", "1") != "0"   # unsharded loops: class ordering + pooling schedule in phase A (else phase B)
# Unsharded loops: the POOLING runs in phase A too, so that phase B — the caller's stream — is the heads alone.  Round 3's split
# (side streams: prep + intersection + weights, 0.2 ms of kernels each; caller's stream: pooling + heads, 0.32 ms back to back) left
# the caller's stream the critical one; with the pooling moved and three side streams (overlap_depth 4) one box gives collab
# 150.6 -> 158.4 M edges/s, ddi 70.3 -> 99.5 M, ppa 5.08 -> 6.05 M, citation2 4.82 -> 5.58 M, a trained cn5 model unchanged; same bits.
phase_a_pool = os.environ.get("OCN_PHASE_A_POOL", "1") != "0"
# Scoring loops: phase A / phase B of a scratch set as captured HIP graphs (pipeline.GraphedPhases).  OFF by default: measured on one
# box, the host's enqueue time per step drops 0.26 -> 0.07 ms (ddi / ppa / citation2 shapes; collab 0.20 -> 0.08) and the
# throughput does not move (ddi 97.0 vs 97.8 M edges/s, ppa 5.5 - 6.0 vs 6.0, citation2 5.8 vs 5.6 - 5.8, collab 155.5 both): the
# steps are bound on the device, as round 3 found for the one-stream capture.  For a caller whose host has other work to do.
graph_loops = os.environ.get("OCN_GRAPH_LOOPS", "0") == "1"
graph_loops_min_batches_per_set = 6                                # ... from this many batches per scratch set on (two eager uses + one capture each)
overlap_min_batch = 2048         # ... from this many candidates per batch (Cora-sized batches: the two event hand-offs cost more than the overlap gives)
overlap_streams = os.environ.get("OCN_ONE_STREAM", "0") != "1"   # scoring loops: phase A of batch t + 1 on a second stream beside phase B of batch t (pipeline.overlapped_steps)
share_full_rows = True           # cn7 on a dense graph: candidates whose whole source row is cn2 copy (A h)[source] (ocn_cn_gather `rowsum`)
deterministic_backward = os.environ.get("OCN_ATOMIC_BACKWARD", "0") != "1"   # pooling backward node by node in a fixed order (ocn_cn_gather_backward_det); else fp32 atomics


def loop_depth(batch) -> int:
    """Batches a scoring loop keeps in flight (= streams it uses) for batches of this size (None: unknown -> two)."""
    if batch is None or batch > overlap_deep_max_batch:
        return 2
    return max(2, int(overlap_depth_small if batch <= overlap_small_batch else overlap_depth))


def _on_device(fn):
    """Launch on the device of the operands.  ``stream_ptr()`` is the current stream of the CURRENT
    device; tensors living on another device (a predictor on cuda:1 in a process whose current device is
    cuda:0) would otherwise be handed to a kernel enqueued on the wrong GPU.  Tensors on different
    devices are an error, as they are for torch's own operators."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        idx = -1
        for a in args:
            if isinstance(a, Tensor) and a.is_cuda:
                if idx < 0:
                    idx = a.device.index
                elif a.device.index != idx:
                    raise RuntimeError(f"{fn.__name__}: operands on different devices (cuda:{idx}, cuda:{a.device.index})")
        if idx >= 0 and idx != torch._C._cuda_getDevice():
            with torch.cuda.device(idx):
                return fn(*args, **kw)
        return fn(*args, **kw)
    return wrapper


def _mark(name: str, flops: float = 0.0) -> None:
    if stage_timer is not None:
        stage_timer.mark(name, flops)


def _req(t: Tensor, dtype, name: str, ndim: Optional[int] = None) -> Tensor:
    if not isinstance(t, Tensor) or not t.is_cuda:
        raise _lib.OcnHipError(f"{name}: expected a CUDA/HIP tensor — ocn_amd has no CPU path")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if ndim is not None and t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim}-d, got shape {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


ST_CAP, ST_SCAN = 1, 2             # include/ocn_hip.h: OCN_ST_CAP, OCN_ST_SCAN (bits of status[0] and of the sticky status[3])


def _total(t: Tensor) -> int:
    """The grand total of a scan read on the host.  Negative = OCN_SCAN_POISON: the scan's workspace was not zero on entry
    (ocn_hip.h, ocn_scan_workspace_bytes) — a Python error here, where round 3 ended in a GPU trap."""
    v = int(t.item())
    if v < 0:
        raise _lib.OcnHipError("scan workspace was not zero on entry (two launches in flight on one workspace?)")
    return v


def status_message(bits: int) -> str:
    return "; ".join(m for b, m in ((ST_CAP, "CN flag buffer capacity exceeded"),
                                    (ST_SCAN, "a scan workspace was not zero on entry: the batch's offsets are void")) if bits & b)


def _ws(n: int, device) -> Tensor:
    return torch.zeros(int(_lib.lib().ocn_scan_workspace_bytes(n)) // 8 + 1, dtype=torch.int64, device=device)


def buf(ws, name: str, shape, dtype, device, zero: bool = False, zero_init: bool = False) -> Tensor:
    """Scratch tensor.  ``ws`` = None: a fresh allocation (states the caller keeps: tests, materialize).
    ``ws`` = a dict owned by a predictor: the buffer is cached by (name, shape, dtype) and reused by the
    next batch — stream order makes that safe, and a dozen allocator round trips per batch disappear
    from the host's critical path.  ``zero``: cleared on every call; ``zero_init``: cleared when it is
    allocated only (workspaces the library hands back zero: ocn_hip.h, ocn_scan_workspace_bytes)."""
    shape = tuple(int(v) for v in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    if ws is None:
        return (torch.zeros if (zero or zero_init) else torch.empty)(shape, dtype=dtype, device=device)
    key = (name, shape, dtype, str(device))
    t = ws.get(key)
    if t is None:
        t = ws[key] = (torch.zeros if zero_init else torch.empty)(shape, dtype=dtype, device=device)
    if zero:
        t.zero_()
    return t


def zero_regions(tensors) -> None:
    """One launch that zeroes every tensor of the list (contiguous, 4-byte element multiples)."""
    ts = [t for t in tensors if t is not None and t.numel()]
    if not ts:
        return
    n = len(ts)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
    nbytes = (ctypes.c_int64 * n)(*[t.numel() * t.element_size() for t in ts])
    idx = ts[0].device.index
    if idx != torch._C._cuda_getDevice():
        with torch.cuda.device(idx):
            return zero_regions(ts)
    check(_lib.lib().ocn_zero_regions(ptrs, nbytes, n, stream_ptr()), "ocn_zero_regions")


@_on_device
def edge_offsets(rowptr: Tensor, src: Tensor, wsd=None) -> Tensor:
    _req(rowptr, torch.int64, "rowptr", 1)
    _req(src, torch.int64, "src", 1)
    B = src.numel()
    off = buf(wsd, "off", B + 1, torch.int64, src.device)
    ws = buf(wsd, "scan_ws", int(_lib.lib().ocn_scan_workspace_bytes(B)) // 8 + 1, torch.int64, src.device, zero_init=True)
    check(_lib.lib().ocn_edge_offsets(ptr(rowptr), ptr(src), B, ptr(off), ptr(ws), stream_ptr()),
          "ocn_edge_offsets")
    return off


@_on_device
def scan_i32(cnt: Tensor) -> Tensor:
    _req(cnt, torch.int32, "cnt", 1)
    n = cnt.numel()
    out = torch.empty(n + 1, dtype=torch.int64, device=cnt.device)
    ws = _ws(n, cnt.device)
    check(_lib.lib().ocn_scan_i32(ptr(cnt), n, ptr(out), ptr(ws), stream_ptr()), "ocn_scan_i32")
    return out


@_on_device
def check_edges(src: Tensor, dst: Tensor, n_src: int, n_dst: int) -> None:
    """Reference behaviour for an out-of-range node id is an IndexError from index_select; a raw
    kernel would fault instead, so the ids are checked here (costs one host sync)."""
    if not validate_indices or src.numel() == 0:
        return
    if (int(n_src), int(n_dst)) in getattr(_checked, "scopes", ()):      # inside ``prevalidated`` for an adjacency of this size
        return
    _req(src, torch.int64, "src", 1); _req(dst, torch.int64, "dst", 1)
    bad = torch.zeros(1, dtype=torch.int32, device=src.device)
    check(_lib.lib().ocn_check_edges(ptr(src), ptr(dst), src.numel(), int(n_src), int(n_dst), ptr(bad), stream_ptr()),
          "ocn_check_edges")
    if int(bad.item()):
        raise IndexError("candidate edge endpoint out of range for the adjacency")


import threading  # noqa: E402
_checked = threading.local()      # .scopes: (n_src, n_dst) of the splits validated by the ``prevalidated`` scopes of THIS thread


class prevalidated:
    """``with ops.prevalidated(src_all, dst_all, n_src, n_dst):`` — bounds-check a whole split of candidate
    edges once (one host sync), then run the batches inside without the per-batch check.  This is what
    ``pipeline.score_edges`` / ``score_mrr_split`` do: the check is hoisted out of the batch loop, the
    exception for a bad id is the same IndexError, raised before the first batch.  The waiver is scoped: it
    holds for the calling thread only and only for adjacencies of the validated size — another thread, or a call on a
    graph of another size inside the scope, is checked as usual (``validate_indices`` itself is not touched)."""

    def __init__(self, src: Tensor, dst: Tensor, n_src: int, n_dst: int):
        check_edges(src.contiguous(), dst.contiguous(), n_src, n_dst)
        self._key = (int(n_src), int(n_dst))

    def __enter__(self):
        if not hasattr(_checked, "scopes"):
            _checked.scopes = []
        _checked.scopes.append(self._key)
        return self

    def __exit__(self, *exc):
        _checked.scopes.remove(self._key)
        return False


HIST_FIELD_BITS = 21             # hist word 0 packs n1 | n2 << 21 | n_union << 42
MAX_BATCH = (1 << HIST_FIELD_BITS) - 1


def hist_counts(hist: Tensor) -> Tensor:
    """Decode the packed histogram [N,2] int64 into int64 [N,4] = {n1, n2, n_union, walks}."""
    m = (1 << HIST_FIELD_BITS) - 1
    p = hist[:, 0]
    return torch.stack([p & m, (p >> HIST_FIELD_BITS) & m, (p >> (2 * HIST_FIELD_BITS)) & m, hist[:, 1]], dim=1)


@_on_device
def cn_flags(rowptrA: Tensor, colA: Tensor, t1: Optional[Tuple[Tensor, Tensor]],
             t2: Optional[Tuple[Tensor, Tensor]], src: Tensor, dst: Tensor, n_cols: int, max_deg_a: int,
             walk: bool = False, t2_bitmap: Optional[Tensor] = None, wsd=None, nds: Optional[Tensor] = None,
             t1_bitmap: Optional[Tensor] = None, rec: Optional[Tensor] = None, sched: Optional[Tensor] = None):
    """Intersection pass.  ``walk=False``: flags of N(src) against the rows of dst in t1 (and t2).
    ``walk=True``: the pygho route on A itself (t1/t2 ignored): cn1 flags + walk counts; with ``nds``
    (``neighbor_degree_sum`` of A) every batch row is swept from its cheaper endpoint.
    ``rec`` (int64 [B, 4], pattern route): receives the per-slot records ``cn_gather`` reads instead of walking
    order -> src / dst / off / counts -> rowptr.
    Returns (order|None, off, flags, wc|None, hist[N,2] int64 packed, cnt1, cnt2|None, status, scal) — ``scal``
    is the zeroed int32[4] statistics scratch the weights stage of the same batch uses."""
    dev = src.device
    B = src.numel()
    _req(rowptrA, torch.int64, "rowptrA", 1); _req(colA, torch.int32, "colA", 1)
    if not walk:
        _req(t1[0], torch.int64, "rowptrT1", 1); _req(t1[1], torch.int32, "colT1", 1)
        if t2 is not None:
            if t2[0] is not None or t2_bitmap is None or n_cols <= small_graph_cols():      # (a product with rows on demand: bit rows only)
                _req(t2[0], torch.int64, "rowptrT2", 1)
            if t2[1] is not None or t2_bitmap is None:           # (with bit rows the kernel never reads T2's column ids: they may be deferred)
                _req(t2[1], torch.int32, "colT2", 1)
    _req(src, torch.int64, "src", 1); _req(dst, torch.int64, "dst", 1)
    if dst.numel() != B:
        raise ValueError("src/dst length mismatch")
    if B > MAX_BATCH:
        raise ValueError(f"candidate batch of {B} edges exceeds the histogram field width ({MAX_BATCH})")
    _mark("begin")
    if walk and 0 < B <= int(_lib.lib().ocn_walk_prep_max_batch()) and walk_share_min > 0:
        return _cn_flags_walk_small(rowptrA, colA, src, dst, n_cols, max_deg_a, wsd, nds)
    # processing order: candidates with the same / nearby source node share most of the rows they
    # gather, so visiting them together turns HBM row fetches into L2 hits (outputs stay in batch order)
    want_order = B >= sort_edges_min_batch
    n_src = rowptrA.numel() - 1
    order = buf(wsd, "order", B, torch.int64, dev) if want_order else None
    ows = (buf(wsd, "order_ws", int(_lib.lib().ocn_order_workspace_bytes(n_src)) // 8 + 1, torch.int64, dev, zero_init=True)
           if want_order else None)
    off = buf(wsd, "off", B + 1, torch.int64, dev)
    sws = buf(wsd, "scan_ws", int(_lib.lib().ocn_scan_workspace_bytes(B)) // 8 + 1, torch.int64, dev, zero_init=True)
    hist = buf(wsd, "hist", (n_cols, 2), torch.int64, dev)
    cnt1 = buf(wsd, "cnt1", B, torch.int32, dev)
    cnt2 = buf(wsd, "cnt2", B, torch.int32, dev) if (walk or t2 is not None) else None
    status = buf(wsd, "status", 4, torch.int32, dev, zero_init=True)     # [0] error bits of this batch; [1], [2] walk-route work tickets; [3] sticky error bits
    scal = buf(wsd, "scal", 4, torch.int32, dev)         # the column statistics word of the weights stage
    # the flag offsets, the counting phase of the order and the batch's resets: ONE launch (ocn_batch_prep)
    zs = [t for t in [hist, status[:3], scal] + ([cnt1, cnt2] if walk else []) if t is not None and t.numel()]     # (status[3]: sticky)
    if sched is not None:             # group costs for the pooling's schedule (ocn_hip.h: ocn_cn_flags `gcost`): first half of `sched`
        if rec is None or walk or _req(sched, torch.int32, "sched", 1).numel() < 2 * ((B + 3) // 4):
            raise ValueError("sched: int32[2 * ceil(B / 4)] beside the slot records of the pattern route")
    zp = (ctypes.c_void_p * len(zs))(*[t.data_ptr() for t in zs])
    zb = (ctypes.c_int64 * len(zs))(*[t.numel() * t.element_size() for t in zs])
    check(_lib.lib().ocn_batch_prep(ptr(rowptrA), ptr(src), B, ptr(off), ptr(sws), n_src, ptr(ows), zp, zb, len(zs),
                                    stream_ptr()), "ocn_batch_prep")
    if want_order:
        check(_lib.lib().ocn_order_by_node_finish(ptr(src), B, n_src, ptr(order), ptr(ows), stream_ptr()),
              "ocn_order_by_node_finish")
    bound = B * max(int(max_deg_a), 0)
    cap = bound if bound <= FLAGS_NOSYNC_LIMIT else _total(off[-1])
    flags = buf(wsd, "flags", max(cap, 1), torch.uint8, dev)
    wc = buf(wsd, "wc", max(cap, 1), torch.int32, dev) if walk else None
    chunk_off = rev_off = None
    if walk:
        if nds is not None and walk_two_sided:
            _req(nds, torch.int64, "nds", 1)
            if nds.numel() != rowptrA.numel() - 1:
                raise ValueError("nds does not match the adjacency")
        else:
            nds = None
        chunk_off = buf(wsd, "chunk_off", B + 1, torch.int64, dev)
        cws = buf(wsd, "scan_ws", int(_lib.lib().ocn_scan_workspace_bytes(B)) // 8 + 1, torch.int64, dev, zero_init=True)
        check(_lib.lib().ocn_chunk_offsets(ptr(rowptrA), ptr(nds), ptr(src), ptr(order), B, ptr(chunk_off), ptr(cws),
                                           stream_ptr()), "ocn_chunk_offsets")
        if nds is not None:
            rev_off = buf(wsd, "rev_off", B + 1, torch.int64, dev)
            check(_lib.lib().ocn_walk_rev_offsets(ptr(rowptrA), ptr(nds), ptr(src), ptr(dst), ptr(order), B,
                                                  ptr(rev_off), ptr(cws), stream_ptr()), "ocn_walk_rev_offsets")
    _mark("cn_prep")
    if walk:
        check(_lib.lib().ocn_cn_walk_flags(ptr(rowptrA), ptr(colA), ptr(nds), ptr(src), ptr(dst), ptr(order), B,
                                           ptr(chunk_off), ptr(rev_off), ptr(off), int(max_deg_a), ptr(flags),
                                           ptr(wc), cap, ptr(hist), ptr(cnt1), ptr(cnt2), ptr(status),
                                           stream_ptr()), "ocn_cn_walk_flags")
    else:
        if t2_bitmap is not None:
            _req(t2_bitmap, torch.int32, "t2_bitmap", 2)
            if t2 is None or (t2[0] is not None and t2_bitmap.shape[0] != t2[0].numel() - 1) or t2_bitmap.shape[1] * 32 < n_cols:
                raise ValueError("t2_bitmap does not match the T2 adjacency")
        if t1_bitmap is not None:
            _req(t1_bitmap, torch.int32, "t1_bitmap", 2)
            if t1_bitmap.shape[0] != t1[0].numel() - 1 or t1_bitmap.shape[1] * 32 < n_cols:
                raise ValueError("t1_bitmap does not match the T1 adjacency")
        check(_lib.lib().ocn_cn_flags(ptr(rowptrA), ptr(colA), ptr(t1[0]), ptr(t1[1]),
                                      ptr(t2[0] if t2 else None), ptr(t2[1] if t2 else None),
                                      ptr(t1_bitmap), t1_bitmap.shape[1] if t1_bitmap is not None else 0,
                                      ptr(t2_bitmap), t2_bitmap.shape[1] if t2_bitmap is not None else 0,
                                      ptr(src), ptr(dst), ptr(order), B, n_cols, ptr(off), ptr(flags), cap, ptr(hist),
                                      ptr(cnt1), ptr(cnt2), ptr(status), ptr(rec), ptr(sched), stream_ptr()), "ocn_cn_flags")
    _mark("cn_flags")
    return order, off, flags, wc, hist, cnt1, cnt2, status, scal


def _cn_flags_walk_small(rowptrA, colA, src, dst, n_cols, max_deg_a, wsd, nds):
    """Walk route, small batch (the drivers' 2048): one prep launch (ocn_walk_prep), the per-candidate sweeps for
    candidates with a source of their own, the shared sweep for candidates that share one (ocn_cn_walk_group)."""
    dev, B = src.device, src.numel()
    l = _lib.lib()
    if nds is not None and walk_two_sided:
        _req(nds, torch.int64, "nds", 1)
        if nds.numel() != rowptrA.numel() - 1:
            raise ValueError("nds does not match the adjacency")
    else:
        nds = None
    order = buf(wsd, "order", B, torch.int64, dev)
    off = buf(wsd, "off", B + 1, torch.int64, dev)
    chunk_off = buf(wsd, "chunk_off", B + 1, torch.int64, dev)
    rev_off = buf(wsd, "rev_off", B + 1, torch.int64, dev) if nds is not None else None
    g_head = buf(wsd, "g_head", B + 1, torch.int32, dev)
    g_item_off = buf(wsd, "g_item_off", B + 1, torch.int64, dev)
    g_active = buf(wsd, "g_active", B, torch.int32, dev)
    meta = buf(wsd, "walk_meta", 4, torch.int32, dev)
    hist = buf(wsd, "hist", (n_cols, 2), torch.int64, dev)
    cnt1 = buf(wsd, "cnt1", B, torch.int32, dev)
    cnt2 = buf(wsd, "cnt2", B, torch.int32, dev)
    status = buf(wsd, "status", 4, torch.int32, dev, zero_init=True)     # ([3]: the sticky error word, never cleared by the library)
    scal = buf(wsd, "scal", 4, torch.int32, dev)
    check(l.ocn_walk_prep(ptr(rowptrA), ptr(nds), ptr(src), ptr(dst), B, int(walk_share_min), ptr(order), ptr(off),
                          ptr(chunk_off), ptr(rev_off), ptr(g_head), ptr(g_item_off), ptr(g_active), ptr(meta), ptr(cnt1),
                          ptr(cnt2), ptr(status), ptr(scal), stream_ptr()), "ocn_walk_prep")
    zero_regions([hist])
    bound = B * max(int(max_deg_a), 0)
    cap = bound if bound <= FLAGS_NOSYNC_LIMIT else _total(off[-1])
    flags = buf(wsd, "flags", max(cap, 1), torch.uint8, dev)
    wc = buf(wsd, "wc", max(cap, 1), torch.int32, dev)
    _mark("cn_prep")
    check(l.ocn_cn_walk_flags(ptr(rowptrA), ptr(colA), ptr(nds), ptr(src), ptr(dst), ptr(order), B, ptr(chunk_off),
                              ptr(rev_off), ptr(off), int(max_deg_a), ptr(flags), ptr(wc), cap, ptr(hist), ptr(cnt1),
                              ptr(cnt2), ptr(status), stream_ptr()), "ocn_cn_walk_flags")
    check(l.ocn_cn_walk_group(ptr(rowptrA), ptr(colA), ptr(src), ptr(dst), ptr(order), B, ptr(g_head), ptr(g_item_off),
                              ptr(g_active), ptr(meta), ptr(off), ptr(flags), ptr(wc), cap, ptr(hist), ptr(cnt1), ptr(cnt2),
                              stream_ptr()), "ocn_cn_walk_group")
    _mark("cn_flags")
    return order, off, flags, wc, hist, cnt1, cnt2, status, scal


@_on_device
def neighbor_degree_sum(rowptr: Tensor, col: Tensor) -> Tensor:
    """nds[v] = Σ_{u∈N(v)} deg(u) (int64): the elements a sweep of v's neighbour rows touches."""
    _req(rowptr, torch.int64, "rowptr", 1); _req(col, torch.int32, "col", 1)
    n = rowptr.numel() - 1
    out = torch.empty(n, dtype=torch.int64, device=rowptr.device)
    check(_lib.lib().ocn_neighbor_degree_sum(ptr(rowptr), ptr(col), n, ptr(out), stream_ptr()),
          "ocn_neighbor_degree_sum")
    return out


exact_colsum = True              # cn5 / cn6 with innerprod != 0: column sums in the reference's entry order (ocn_cn_colsum_exact)


def _ip_nonzero(innerprod: Tensor) -> bool:
    """Whether the ``innerprod`` buffer is non-zero, read back once per value: cached per live tensor object
    (weak reference) and version counter, so an eval loop pays one host sync in total, not one per batch — and a
    freed tensor's address being reused by another one cannot hit."""
    import weakref
    hit = _ip_cache.get("k")
    if hit is None or hit[0]() is not innerprod or hit[1] != innerprod.data_ptr() or hit[2] != innerprod._version:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("innerprod changed since the last eager call: run one eager batch before capturing")
        _ip_cache["k"] = (weakref.ref(innerprod), innerprod.data_ptr(), innerprod._version)
        _ip_cache["v"] = bool(innerprod.detach().reshape(-1)[0].item() != 0.0)
    return _ip_cache["v"]


_ip_cache: dict = {}


def colsum_wanted(innerprod: Tensor) -> bool:
    """cn5 / cn6: is the second column normalisation to be summed in the reference's entry order?  Only a
    non-zero ``innerprod`` makes the order matter (for 0 the closed form over the integer counts is exact)."""
    return exact_colsum and _ip_nonzero(innerprod)


@_on_device
def cn_colsum_exact(rowptrA, colA, src, off, flagsA, flagsB, wc, hist: Tensor, innerprod: Tensor, scal: Tensor, wsd=None,
                    s2_init: Optional[Tensor] = None):
    """Order-exact S2 (and S3 with ``flagsB``) of the batch; ocn_hip.h: ocn_cn_colsum_exact.  ``scal``: the
    batch's zeroed int32[4] statistics scratch (shared with the weights call).  Returns (s2, s3|None)."""
    _req(hist, torch.int64, "hist", 2)
    N, dev = hist.shape[0], hist.device
    cap = flagsA.numel()
    if cap >= (1 << 32) - 1:
        raise NotImplementedError("order-exact column sums address flag positions with 32 bits")
    ip = _req(innerprod.detach().reshape(1).to(torch.float32), torch.float32, "innerprod", 1)
    if s2_init is not None and _req(s2_init, torch.float32, "s2_init", 1).numel() != N:
        raise ValueError("s2_init must have one entry per column")
    s2 = buf(wsd, "s2_exact", N, torch.float32, dev)
    s3 = buf(wsd, "s3_exact", N, torch.float32, dev) if flagsB is not None else None
    ws = buf(wsd, "colsum_ws", int(_lib.lib().ocn_cn_colsum_workspace_bytes(N, cap)) // 8 + 1, torch.int64, dev)
    check(_lib.lib().ocn_cn_colsum_exact(ptr(rowptrA), ptr(colA), ptr(src), src.numel(), ptr(off), ptr(flagsA), ptr(flagsB),
                                         ptr(wc), cap, ptr(hist), N, ptr(ip), ptr(scal), ptr(s2_init), ptr(s2), ptr(s3),
                                         ptr(ws), stream_ptr()), "ocn_cn_colsum_exact")
    _mark("cn_colsum")
    return s2, s3


@_on_device
def cn_weights_cn5(hist: Tensor, innerprod: Tensor, valued: bool = False, wsd=None, s2_exact: Optional[Tensor] = None,
                   scal: Optional[Tensor] = None) -> Tensor:
    """In place: packed int64 [N,2] histogram -> float32 [N,4] weights {w1, t, inv2, 0} (same storage).
    ``s2_exact``: the column sums of ``cn_colsum_exact`` (then ``scal`` is the statistics scratch that call
    used); None: the closed form over the integer counts (exact for innerprod == 0)."""
    _req(hist, torch.int64, "hist", 2)
    ip = _req(innerprod.detach().reshape(1).to(torch.float32), torch.float32, "innerprod", 1)
    if scal is None:
        scal = buf(wsd, "scal", 4, torch.int32, hist.device, zero=True)
    if _ip_nonzero(innerprod) and s2_exact is None:     # the batch's scale statistic: only a non-zero innerprod reads it
        check(_lib.lib().ocn_cn5_column_stats(ptr(hist), hist.shape[0], ptr(scal), stream_ptr()), "ocn_cn5_column_stats")
    check(_lib.lib().ocn_cn_weights_cn5(ptr(hist), hist.shape[0], ptr(ip), ptr(scal), int(valued), ptr(s2_exact),
                                        stream_ptr()), "ocn_cn_weights_cn5")
    _mark("cn_weights")
    return hist.view(torch.float32)


@_on_device
def cn_weights_cn7(hist: Tensor, sum_fill: float, diag1: Optional[Tensor] = None, diag2: Optional[Tensor] = None) -> Tensor:
    """``diag1`` / ``diag2``: the Chebyshev diagonals of the cn1 / cn2 branch (float32 [N]; None = T0 = ones)."""
    _req(hist, torch.int64, "hist", 2)
    for d, nm in ((diag1, "diag1"), (diag2, "diag2")):
        if d is not None and _req(d, torch.float32, nm, 1).numel() != hist.shape[0]:
            raise ValueError(f"{nm}: one entry per column")
    check(_lib.lib().ocn_cn_weights_cn7(ptr(hist), hist.shape[0], float(sum_fill), ptr(diag1), ptr(diag2), stream_ptr()),
          "ocn_cn_weights_cn7")
    _mark("cn_weights")
    return hist.view(torch.float32)


@_on_device
def cn_gather(rowptrA, colA, src, dst, off, flags, wc: Optional[Tensor], weights: Tensor, h: Tensor,
              order: Optional[Tensor] = None, max_row_len: int = 0, wsd=None, out_row: Optional[Tensor] = None,
              cnt1: Optional[Tensor] = None, cnt2: Optional[Tensor] = None, rec: Optional[Tensor] = None,
              sched: Optional[Tensor] = None, rowsum: Optional[Tensor] = None, sched_ready: bool = False):
    _req(weights, torch.float32, "weights", 2)
    _req(h, torch.float32, "h", 2)
    if weights.shape[0] != h.shape[0] or weights.shape[1] != 4:
        raise ValueError("weights must be [N,4] with N = h.shape[0]")
    B, H = src.numel(), h.shape[1]
    out = buf(wsd, "pooled", (3, B, H), torch.float32, h.device)
    if rowsum is not None and (wc is not None or cnt2 is None or _req(rowsum, torch.float32, "rowsum", 2).shape != h.shape):
        raise ValueError("rowsum: [N, H] row sums of h, pattern route with per-row counts only")
    perm = None
    if sched is not None and rec is not None and H == 256 and B % 32 == 0 and B // 32 <= 65535:
        n_groups = B // 4             # the intersection pass left the groups' costs in sched[:n_groups]; their visiting order follows
        perm = sched[n_groups:]
        if not sched_ready:           # (a scoring loop's phase A has run gather_schedule already)
            check(_lib.lib().ocn_gather_schedule(ptr(sched), n_groups, int(sched_segment), ptr(perm), stream_ptr()), "ocn_gather_schedule")
    _mark("cn_pre")                   # (stage timers: what sits between the intersection pass and the pooling on this stream)
    check(_lib.lib().ocn_cn_gather(ptr(rowptrA), ptr(colA), ptr(src), ptr(dst), ptr(order), B, ptr(off), ptr(flags),
                                   ptr(wc), ptr(weights), ptr(h), H, int(max_row_len), ptr(out[0]), ptr(out[1]),
                                   ptr(out[2]), ptr(out_row), ptr(cnt1), ptr(cnt2), ptr(rec), ptr(perm), ptr(rowsum),
                                   stream_ptr()), "ocn_cn_gather")
    _mark("cn_gather")
    return out[0], out[1], out[2]


CLASS_RANGES = 7                 # include/ocn_hip.h: OCN_CLASS_RANGES
R_CN1, R_BOTH, R_CN2_ONLY, R_ANY, R_NONE, R_CN1_ONLY, R_ALL = range(CLASS_RANGES)


@_on_device
def gather_schedule(sched: Tensor, B: int) -> bool:
    """The pooling's longest-first visiting order from the group costs the intersection pass left in ``sched[:B // 4]``
    (ocn_gather_schedule) into ``sched[B // 4:]``; False where the schedule does not apply (cn_gather's own conditions)."""
    if sched is None or B % 32 != 0 or B // 32 > 65535 or B == 0:
        return False
    n_groups = B // 4
    check(_lib.lib().ocn_gather_schedule(ptr(sched), n_groups, int(sched_segment), ptr(sched[n_groups:]), stream_ptr()), "ocn_gather_schedule")
    _mark("cn_sched")
    return True


@_on_device
def class_order(cnt1: Tensor, cnt2: Optional[Tensor], order: Optional[Tensor], wsd=None):
    """Class-major processing order for the heads (ocn_hip.h: ocn_class_order).
    Returns (order2 [B] slot -> batch row, inv [B] batch row -> slot, ranges int64 [7, 2] on the device)."""
    _req(cnt1, torch.int32, "cnt1", 1)
    B = cnt1.numel()
    dev = cnt1.device
    order2 = buf(wsd, "cls_order", B, torch.int64, dev)
    inv = buf(wsd, "cls_inv", B, torch.int64, dev)
    ranges = buf(wsd, "cls_ranges", (CLASS_RANGES, 2), torch.int64, dev)
    prefix = buf(wsd, "cls_prefix", B + 1, torch.int64, dev)
    ws = buf(wsd, "scan_ws", int(_lib.lib().ocn_scan_workspace_bytes(B)) // 8 + 1, torch.int64, dev, zero_init=True)
    check(_lib.lib().ocn_class_order(ptr(cnt1), ptr(cnt2), ptr(order), B, ptr(order2), ptr(inv), ptr(ranges),
                                     ptr(prefix), ptr(ws), stream_ptr()), "ocn_class_order")
    _mark("cn_class")
    return order2, inv, ranges


@_on_device
def cn_weights_cn6(histA: Tensor, histB: Tensor, innerprod: Tensor, exact=None, scal: Optional[Tensor] = None):
    """In place: histA -> float32 [N,4] {inv1, t, inv2, 0}, histB -> {1/S3, 0, 0, 0}; returns them and the
    device scalar nip (ocn_hip.h: ocn_cn_weights_cn6).  ``exact`` = (rowptrA, colA, src, off, flagsA, flagsB):
    order-exact S2 / S3 for a non-zero ``innerprod``, as for cn5."""
    _req(histA, torch.int64, "histA", 2); _req(histB, torch.int64, "histB", 2)
    if histA.shape != histB.shape:
        raise ValueError("histA / histB shape mismatch")
    ip = _req(innerprod.detach().reshape(1).to(torch.float32), torch.float32, "innerprod", 1)
    if scal is None:
        scal = torch.zeros(4, dtype=torch.int32, device=histA.device)
    nip = torch.empty(1, dtype=torch.float32, device=histA.device)
    s2 = s3 = None
    if exact is not None and colsum_wanted(innerprod):
        rowptrA, colA, src, off, flagsA, flagsB = exact
        s2, s3 = cn_colsum_exact(rowptrA, colA, src, off, flagsA, flagsB, None, histA, innerprod, scal)
    elif _ip_nonzero(innerprod):
        check(_lib.lib().ocn_cn5_column_stats(ptr(histA), histA.shape[0], ptr(scal), stream_ptr()), "ocn_cn5_column_stats")
    check(_lib.lib().ocn_cn_weights_cn6(ptr(histA), ptr(histB), histA.shape[0], ptr(ip), ptr(scal), ptr(nip),
                                        ptr(s2), ptr(s3), stream_ptr()), "ocn_cn_weights_cn6")
    _mark("cn_weights")
    return histA.view(torch.float32).view(-1, 4), histB.view(torch.float32).view(-1, 4), nip


@_on_device
def cn_gather3(rowptrA, colA, src, dst, off, flagsA, flagsB, wA: Tensor, wB: Tensor, nip: Tensor, h: Tensor,
               order: Optional[Tensor] = None):
    """(xcn1, xcn2, xcn3, x_i * x_j) of the 3-hop predictor."""
    _req(wA, torch.float32, "weightsA", 2); _req(wB, torch.float32, "weightsB", 2)
    _req(h, torch.float32, "h", 2)
    if wA.shape != (h.shape[0], 4) or wB.shape != (h.shape[0], 4):
        raise ValueError("weights must be [N,4] with N = h.shape[0]")
    B, H = src.numel(), h.shape[1]
    if H not in LN_WIDTHS:
        raise NotImplementedError(f"cn6 pooling supports hidden widths {LN_WIDTHS}, got {H}")
    out = torch.empty(4, B, H, dtype=torch.float32, device=h.device)
    check(_lib.lib().ocn_cn_gather3(ptr(rowptrA), ptr(colA), ptr(src), ptr(dst), ptr(order), B, ptr(off), ptr(flagsA),
                                    ptr(flagsB), ptr(wA), ptr(wB), ptr(nip), ptr(h), H, ptr(out[0]), ptr(out[1]),
                                    ptr(out[2]), ptr(out[3]), stream_ptr()), "ocn_cn_gather3")
    _mark("cn_gather")
    return out[0], out[1], out[2], out[3]


@_on_device
def cn_gather3_backward(rowptrA, colA, src, dst, off, flagsA, flagsB, wA: Tensor, wB: Tensor, nip: Tensor, h: Tensor,
                        g1: Tensor, g2: Tensor, g3: Tensor, g4: Tensor, order: Optional[Tensor] = None) -> Tensor:
    """Gradient of cn6's (xcn1, xcn2, xcn3, xij) with respect to h (ocn_hip.h: ocn_cn_gather3_backward; fp32 atomics)."""
    B, H = src.numel(), h.shape[1]
    for t, nm in ((g1, "g1"), (g2, "g2"), (g3, "g3"), (g4, "g4")):
        if _req(t, torch.float32, nm, 2).shape != (B, H):
            raise ValueError(f"{nm}: expected [{B}, {H}]")
    if H not in LN_WIDTHS:
        raise NotImplementedError(f"cn6 pooling backward supports hidden widths {LN_WIDTHS}, got {H}")
    dh = torch.zeros_like(h)
    check(_lib.lib().ocn_cn_gather3_backward(ptr(rowptrA), ptr(colA), ptr(src), ptr(dst), ptr(order), B, ptr(off), ptr(flagsA),
                                             ptr(flagsB), ptr(wA), ptr(wB), ptr(nip), ptr(h), H, ptr(g1), ptr(g2), ptr(g3), ptr(g4),
                                             ptr(dh), stream_ptr()), "ocn_cn_gather3_backward")
    return dh


@_on_device
def cn_gather_backward(rowptrA, colA, src, dst, off, flags, wc, weights: Tensor, h: Tensor, g1: Tensor, g2: Tensor,
                       g3: Tensor, order: Optional[Tensor] = None) -> Tensor:
    """Gradient of (xcn1, xcn2, xij) with respect to h: node by node in a fixed order (``deterministic_backward``, the
    default), or with fp32 atomics (``OCN_ATOMIC_BACKWARD=1``)."""
    B, H = src.numel(), h.shape[1]
    for t, nm in ((g1, "g1"), (g2, "g2"), (g3, "g3")):
        if _req(t, torch.float32, nm, 2).shape != (B, H):
            raise ValueError(f"{nm}: expected [{B}, {H}]")
    if H not in LN_WIDTHS:
        raise NotImplementedError(f"pooling backward supports hidden widths {LN_WIDTHS}, got {H}")
    dh = torch.zeros_like(h)
    if deterministic_backward:
        l = _lib.lib()
        cap, N = flags.numel(), h.shape[0]
        ws = torch.empty(int(l.ocn_cn_gather_backward_det_workspace_bytes(N, B, cap)), dtype=torch.uint8, device=h.device)
        check(l.ocn_cn_gather_backward_det(ptr(rowptrA), ptr(colA), ptr(src), ptr(dst), B, ptr(off), ptr(flags), ptr(wc), cap,
                                           ptr(weights), ptr(h), N, H, ptr(g1), ptr(g2), ptr(g3), ptr(dh), ptr(ws),
                                           stream_ptr()), "ocn_cn_gather_backward_det")
        return dh
    check(_lib.lib().ocn_cn_gather_backward(ptr(rowptrA), ptr(colA), ptr(src), ptr(dst), ptr(order), B, ptr(off),
                                            ptr(flags), ptr(wc), ptr(weights), ptr(h), H, ptr(g1), ptr(g2), ptr(g3),
                                            ptr(dh), stream_ptr()), "ocn_cn_gather_backward")
    return dh


@_on_device
def cn_gather_backward_lists(rowptrA, colA, src, dst, off, flags, N: int):
    """The per-node key lists of the deterministic pooling backward (ocn_hip.h: ocn_cn_gather_backward_det_lists), for
    inspection: (col_off int64 [N + 1], keys int32 [col_off[N]]).  One host sync (the list total)."""
    l = _lib.lib()
    B, cap = src.numel(), flags.numel()
    ws = torch.empty(int(l.ocn_cn_gather_backward_det_workspace_bytes(N, B, cap)), dtype=torch.uint8, device=src.device)
    check(l.ocn_cn_gather_backward_det_lists(ptr(rowptrA), ptr(colA), ptr(src), ptr(dst), B, ptr(off), ptr(flags), cap, N, ptr(ws),
                                             stream_ptr()), "ocn_cn_gather_backward_det_lists")
    col_off = ws[: (N + 1) * 8].view(torch.int64)
    k0 = int(l.ocn_cn_gather_backward_det_keys_offset(N))
    return col_off, ws[k0: k0 + 4 * _total(col_off[N])].view(torch.int32)


SPMM_MODES = {"sum": 0, "add": 0, "mean": 1, "max": 2}


@_on_device
def spmm_csr(rowptr: Tensor, col: Tensor, x: Tensor, pre: Optional[Tensor] = None,
             post: Optional[Tensor] = None, mode: str = "sum", edge_scale: bool = False,
             self_mode: int = 0, val: Optional[Tensor] = None) -> Tensor:
    _req(rowptr, torch.int64, "rowptr", 1); _req(col, torch.int32, "col", 1)
    _req(x, torch.float32, "x", 2)
    n = rowptr.numel() - 1
    if pre is not None and (_req(pre, torch.float32, "pre", 1).numel() != x.shape[0]):
        raise ValueError("pre must have one entry per row of x")
    if post is not None and (_req(post, torch.float32, "post", 1).numel() != n):
        raise ValueError("post must have one entry per output row")
    if self_mode and x.shape[0] != n:
        raise ValueError("self term needs a square operator")
    if val is not None and _req(val, torch.float32, "val", 1).numel() != col.numel():
        raise ValueError("val must have one entry per stored column")
    y = torch.empty(n, x.shape[1], dtype=torch.float32, device=x.device)
    check(_lib.lib().ocn_spmm_csr(ptr(rowptr), ptr(col), ptr(val), n, ptr(x), x.shape[1], ptr(pre), ptr(post),
                                  SPMM_MODES[mode], int(edge_scale), int(self_mode), ptr(y), stream_ptr()),
          "ocn_spmm_csr")
    return y


@_on_device
def deg_rsqrt(rowptr: Tensor, add: float = 1.0, val: Optional[Tensor] = None) -> Tensor:
    _req(rowptr, torch.int64, "rowptr", 1)
    n = rowptr.numel() - 1
    out = torch.empty(n, dtype=torch.float32, device=rowptr.device)
    check(_lib.lib().ocn_deg_rsqrt(ptr(rowptr), ptr(val), n, float(add), ptr(out), stream_ptr()), "ocn_deg_rsqrt")
    return out


@_on_device
def bitrows_from_csr(rowptr: Tensor, col: Tensor, n_cols: int) -> Tensor:
    """ocn_hip.h: ocn_bitrows_from_csr — the pattern as dense bit rows, int32 [n_rows, ceil(n_cols / 32)]."""
    _req(rowptr, torch.int64, "rowptr", 1); _req(col, torch.int32, "col", 1)
    n = rowptr.numel() - 1
    bits = torch.zeros(n, (int(n_cols) + 31) // 32, dtype=torch.int32, device=rowptr.device)
    check(_lib.lib().ocn_bitrows_from_csr(ptr(rowptr), ptr(col), n, ptr(bits), bits.shape[1], stream_ptr()), "ocn_bitrows_from_csr")
    return bits


@_on_device
def spgemm_pattern(rowptrA, colA, rowptrB, colB, n_cols_b: int, defer_fill: bool = False):
    """CSR pattern of A·B (columns ascending) and, when it fits ``a2_bitmap_max_bytes``, the same
    rows as dense bit rows.  One host sync for the output size — with ``defer_fill`` and bit rows, not before somebody calls
    the returned thunk: ``(rowptrC, fill, bitmap)`` where ``fill()`` -> colC runs the second pass (the counting pass, the scan
    of the row lengths and the bit rows are done)."""
    _req(rowptrA, torch.int64, "rowptrA", 1); _req(colA, torch.int32, "colA", 1)
    _req(rowptrB, torch.int64, "rowptrB", 1); _req(colB, torch.int32, "colB", 1)
    l = _lib.lib()
    if n_cols_b > l.ocn_spgemm_max_cols():
        raise NotImplementedError(f"A·B pattern with {n_cols_b} columns exceeds the LDS bitmap "
                                  f"({l.ocn_spgemm_max_cols()}); the reference does not form A² at "
                                  "that size either (it uses the per-batch walk-count route)")
    n = rowptrA.numel() - 1
    dev = colA.device
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    words = (n_cols_b + 31) // 32
    bitmap = None
    if n * words * 4 <= a2_bitmap_max_bytes and n > 0:
        bitmap = torch.empty(n, words, dtype=torch.int32, device=dev)       # every row is written in full
    check(l.ocn_spgemm_pattern_count(ptr(rowptrA), ptr(colA), n, ptr(rowptrB), ptr(colB), n_cols_b,
                                     ptr(cnt), ptr(bitmap), words, stream_ptr()), "ocn_spgemm_pattern_count")
    rowptrC = scan_i32(cnt)

    def fill() -> Tensor:
        nnz = _total(rowptrC[-1])
        colC = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
        if nnz:
            with torch.cuda.device(dev):
                check(l.ocn_spgemm_pattern_fill(ptr(rowptrA), ptr(colA), n, ptr(rowptrB), ptr(colB), n_cols_b,
                                                ptr(rowptrC), ptr(colC), stream_ptr()), "ocn_spgemm_pattern_fill")
        return colC
    if defer_fill and bitmap is not None:
        return rowptrC, fill, bitmap
    return rowptrC, fill(), bitmap


def small_graph_cols() -> int:
    """ocn_hip.h: ocn_cn_flags_small_graph_cols — up to this many columns the intersection pass reads T2's row lengths beside its
    bit rows (so a product whose rows are built on demand is not for such graphs)."""
    return int(_lib.lib().ocn_cn_flags_small_graph_cols())


lazy_product_rows = os.environ.get("OCN_LAZY_PRODUCT", "1") != "0"     # a product formed under autograd builds its bit rows on demand (sparse._lazy_product)


def spgemm_max_cols() -> int:
    return int(_lib.lib().ocn_spgemm_max_cols())


@_on_device
def spgemm_bit_rows(rowptrA, colA, rowptrB, colB, n_cols_b: int, rows: Tensor, done: Tensor, bitmap: Tensor) -> None:
    """ocn_hip.h: ocn_spgemm_bit_rows — the dense bit rows ``rows`` (int64 ids) of A·B that ``done`` (int32 per row) does not
    mark yet, into ``bitmap`` [n_rows, words]; on the current stream."""
    _req(rowptrA, torch.int64, "rowptrA", 1); _req(colA, torch.int32, "colA", 1)
    _req(rowptrB, torch.int64, "rowptrB", 1); _req(colB, torch.int32, "colB", 1)
    _req(rows, torch.int64, "rows", 1); _req(done, torch.int32, "done", 1); _req(bitmap, torch.int32, "bitmap", 2)
    n = rowptrA.numel() - 1
    if done.numel() != n or bitmap.shape[0] != n or bitmap.shape[1] * 32 < n_cols_b:
        raise ValueError("spgemm_bit_rows: done / bitmap do not match the product")
    _mark("begin")
    check(_lib.lib().ocn_spgemm_bit_rows(ptr(rowptrA), ptr(colA), n, ptr(rowptrB), ptr(colB), n_cols_b, ptr(rows), rows.numel(),
                                         ptr(done), ptr(bitmap), bitmap.shape[1], stream_ptr()), "ocn_spgemm_bit_rows")
    _mark("adj2_rows")


dense_adj2_max_nodes = 32768      # block route: A as a dense int8 matrix (n^2 bytes, twice) up to this many nodes
# utils.sparse_tensor_multiply (the drivers' --adj2byblock call, NeighborOverlap_large.py:116): False = the intended A²,
# True = the reference as written (utils.py:318-321, SURVEY Q7: block-local indices, all blocks folded onto one corner)
adj2_fold_quirk = os.environ.get("OCN_ADJ2_FOLD_QUIRK", "0") == "1"


@_on_device
def dense_block_adj2(rowptr: Tensor, col: Tensor, n: int, block_size: int, fold: bool = False):
    """Pattern of A·A by the reference's block loop on the integer matrix cores (ocn_hip.h: ocn_dense_block_mm_bits).
    Returns (rowptrC, colC, bit rows [n, words] int32)."""
    _req(rowptr, torch.int64, "rowptr", 1); _req(col, torch.int32, "col", 1)
    if block_size <= 0 or block_size % 32:
        raise ValueError("block_size must be a positive multiple of 32 for the dense block route")
    l = _lib.lib()
    dev = col.device
    ld = (n + 63) // 64 * 64
    dense = torch.zeros(ld, ld, dtype=torch.int8, device=dev)
    denseT = torch.zeros(ld, ld, dtype=torch.int8, device=dev)
    check(l.ocn_dense_from_csr(ptr(rowptr), ptr(col), n, ld, ptr(dense), ptr(denseT), stream_ptr()), "ocn_dense_from_csr")
    words = (n + 31) // 32
    bits = torch.zeros(n, words, dtype=torch.int32, device=dev)
    for r0 in range(0, n, block_size):                   # the tile loop of utils.py:305-321
        for c0 in range(0, n, block_size):
            check(l.ocn_dense_block_mm_bits(ptr(dense), ptr(denseT), ld, n, r0, min(r0 + block_size, n), c0, min(c0 + block_size, n),
                                            int(fold), ptr(bits), words, stream_ptr()), "ocn_dense_block_mm_bits")
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    check(l.ocn_bitrows_count(ptr(bits), words, n, n, ptr(cnt), stream_ptr()), "ocn_bitrows_count")
    rowptrC = scan_i32(cnt)
    nnz = _total(rowptrC[-1])
    colC = torch.empty(max(nnz, 1), dtype=torch.int32, device=dev)[:nnz]
    if nnz:
        check(l.ocn_bitrows_fill(ptr(bits), words, n, n, ptr(rowptrC), ptr(colC), stream_ptr()), "ocn_bitrows_fill")
    return rowptrC, colC, bits


@_on_device
def rows_ln_relu(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, relu: bool, inplace: bool = False) -> Tensor:
    """LayerNorm over the last dim of a [rows, H] fp32 matrix, optionally followed by ReLU."""
    _req(x, torch.float32, "x", 2)
    _req(gamma, torch.float32, "gamma", 1); _req(beta, torch.float32, "beta", 1)
    rows, H = x.shape
    if gamma.numel() != H or beta.numel() != H or H not in LN_WIDTHS:
        raise ValueError("rows_ln_relu: unsupported width")
    y = x if inplace else torch.empty_like(x)
    check(_lib.lib().ocn_rows_ln_relu(ptr(x), ptr(gamma), ptr(beta), float(eps), int(relu), rows, H, ptr(y),
                                      stream_ptr()), "ocn_rows_ln_relu")
    return y


@_on_device
def ln_drop_relu_forward(x: Tensor, gamma: Optional[Tensor], beta: Optional[Tensor], eps: float, p: float, seed: int, relu: bool):
    """y = relu?(dropout_p(LN?(x))) in one launch (ocn_hip.h: ocn_ln_drop_relu_forward); returns (y, stats | None)."""
    _req(x, torch.float32, "x", 2)
    rows, H = x.shape
    if H not in LN_WIDTHS:
        raise ValueError("ln_drop_relu: unsupported width")
    if gamma is not None and (_req(gamma, torch.float32, "gamma", 1).numel() != H or _req(beta, torch.float32, "beta", 1).numel() != H):
        raise ValueError("ln_drop_relu: gamma / beta width")
    y = torch.empty_like(x)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device) if gamma is not None else None
    check(_lib.lib().ocn_ln_drop_relu_forward(ptr(x), ptr(gamma), ptr(beta), float(eps), float(p), int(seed), int(relu), rows, H, ptr(y),
                                              ptr(stats), stream_ptr()), "ocn_ln_drop_relu_forward")
    return y, stats


@_on_device
def ln_drop_relu_backward(g: Tensor, x: Tensor, y: Tensor, stats: Optional[Tensor], gamma: Optional[Tensor], p: float, seed: int,
                          relu: bool):
    """(dx, dgamma | None, dbeta | None) of ``ln_drop_relu_forward`` (deterministic: ocn_hip.h)."""
    g = _req(g.contiguous(), torch.float32, "g", 2)
    rows, H = g.shape
    dx = torch.empty_like(g)
    dg = db = ws = None
    if gamma is not None:
        dg, db = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(int(_lib.lib().ocn_ln_drop_relu_workspace_bytes(H)), dtype=torch.uint8, device=g.device)
    check(_lib.lib().ocn_ln_drop_relu_backward(ptr(g), ptr(x), ptr(y), ptr(stats), ptr(gamma), float(p), int(seed), int(relu), rows, H,
                                               ptr(dx), ptr(dg), ptr(db), ptr(ws), stream_ptr()), "ocn_ln_drop_relu_backward")
    return dx, dg, db


@_on_device
def dropout_keep_mask(seed: int, p: float, n: int, device) -> Tensor:
    out = torch.empty(n, dtype=torch.uint8, device=device)
    with torch.cuda.device(out.device):
        check(_lib.lib().ocn_dropout_keep_mask(int(seed), float(p), n, ptr(out), stream_ptr()), "ocn_dropout_keep_mask")
    return out


@_on_device
def combine3(coef: Tensor, x1: Tensor, x2: Tensor, x3: Tensor) -> Tensor:
    """coef[0]*x1 + coef[1]*x2 + coef[2]*x3 with the coefficients read on the device."""
    _req(coef, torch.float32, "coef", 1)
    for t, nm in ((x1, "x1"), (x2, "x2"), (x3, "x3")):
        _req(t, torch.float32, nm)
    if coef.numel() != 3 or x1.shape != x2.shape or x1.shape != x3.shape or x1.numel() % 4:
        raise ValueError("combine3: shape mismatch")
    out = torch.empty_like(x1)
    check(_lib.lib().ocn_combine3(ptr(coef), ptr(x1), ptr(x2), ptr(x3), x1.numel(), ptr(out), stream_ptr()),
          "ocn_combine3")
    return out


@_on_device
def mix3_backward(coef: Tensor, g: Tensor, x1: Tensor, x2: Tensor, x3: Tensor):
    """(d1, d2, d3, dcoef) of z = coef[0] x1 + coef[1] x2 + coef[2] x3 (ocn_hip.h: ocn_mix3_backward)."""
    g = _req(g.contiguous(), torch.float32, "g")
    for t, nm in ((x1, "x1"), (x2, "x2"), (x3, "x3")):
        if _req(t, torch.float32, nm).shape != g.shape:
            raise ValueError("mix3_backward: shape mismatch")
    n = g.numel()
    if n % 4:
        raise ValueError("mix3_backward: n must be a multiple of 4")
    l = _lib.lib()
    d1, d2, d3 = torch.empty_like(g), torch.empty_like(g), torch.empty_like(g)
    dcoef = torch.empty(3, dtype=torch.float32, device=g.device)
    ws = torch.empty(int(l.ocn_mix3_workspace_bytes()), dtype=torch.uint8, device=g.device)
    check(l.ocn_mix3_backward(ptr(coef), ptr(g), ptr(x1), ptr(x2), ptr(x3), n, ptr(d1), ptr(d2), ptr(d3), ptr(dcoef), ptr(ws),
                              stream_ptr()), "ocn_mix3_backward")
    return d1, d2, d3, dcoef


@_on_device
def fill_rows(dst: Tensor, vec: Tensor, row_range: Tensor) -> None:
    """dst[rows of the device-side range] = vec (dst may be a column slice of a wider buffer)."""
    _req_strided(dst, "dst")
    _req(vec, torch.float32, "vec")
    _req(row_range, torch.int64, "row_range", 1)
    if vec.numel() != dst.shape[1]:
        raise ValueError("fill_rows: width mismatch")
    check(_lib.lib().ocn_fill_rows(ptr(dst), dst.stride(0), dst.shape[1], ptr(vec), ptr(row_range), dst.shape[0],
                                   stream_ptr()), "ocn_fill_rows")
    _mark("mlp_glue")


LINEAR_WIDTHS = (32, 64, 128, 256)
fast_linear = True               # route eligible nn.Linear layers of the heads through ocn_linear_bf16x6
_panels: dict = {}


@_on_device
def linear_panel(weight: Tensor) -> Tensor:
    """Pre-split, fragment-ordered bf16 panel of an nn.Linear weight.  Cached per live tensor object
    (weak reference) and rebuilt when the weight is modified in place (``_version``), re-pointed
    (``data_ptr``) or reshaped — a freed weight's address being reused by another module must not
    hit."""
    import weakref
    key = id(weight)
    hit = _panels.get(key)
    if (hit is not None and hit[0]() is weight and hit[1] == weight.data_ptr() and hit[2] == weight._version
            and hit[3] == tuple(weight.shape)):
        return hit[4]
    w = _req(weight.detach(), torch.float32, "weight", 2)
    N, K = w.shape
    panel = torch.empty(int(_lib.lib().ocn_linear_panel_bytes(N, K)), dtype=torch.uint8, device=w.device)
    check(_lib.lib().ocn_linear_split_weight(ptr(w), N, K, ptr(panel), stream_ptr()), "ocn_linear_split_weight")
    if len(_panels) > 512:
        for k in [k for k, v in _panels.items() if v[0]() is None]:
            del _panels[k]
    _panels[key] = (weakref.ref(weight), weight.data_ptr(), weight._version, (N, K), panel)
    return panel


@_on_device
def heads_panel(weight: Tensor):
    """The f16 hi/lo panel of a weight [N, K] for ocn_heads_fused (ocn_hip.h: ocn_heads_split_weight) and the inverse of
    the power of two it was scaled by (max |scale * W| in [2^13, 2^14): f16 has a 5-bit exponent).  Not cached here
    (the heads pack caches it); one host sync for the maximum."""
    import math
    w = _req(weight.detach(), torch.float32, "weight", 2)
    N, K = w.shape
    m = float(w.abs().max().item()) if w.numel() else 0.0
    if not math.isfinite(m):
        raise ValueError("heads_panel: non-finite weight")
    sw = 13 - math.frexp(m)[1] + 1 if m > 0.0 else 0          # frexp: m = f * 2^e, f in [0.5, 1)  ->  floor(log2 m) = e - 1
    sw = max(-100, min(100, sw))
    panel = torch.empty(int(_lib.lib().ocn_heads_panel_bytes(N, K)), dtype=torch.uint8, device=w.device)
    check(_lib.lib().ocn_heads_split_weight(ptr(w), N, K, float(2.0 ** sw), ptr(panel), stream_ptr()), "ocn_heads_split_weight")
    return panel, float(2.0 ** -sw)


HEADS_WIDTHS = (128, 256)
train_linear = True              # autograd on: the heads' Linear layers (forward and input gradient) on the MFMA kernel
train_tails = os.environ.get("OCN_TRAIN_TAILS", "1") != "0"      # ... and their LayerNorm -> Dropout -> ReLU tails as one launch each way (ocn_ln_drop_relu_*)
fused_heads = True               # cn5 / cn7 eval: the whole MLP head as one launch (ocn_heads_fused)
fused_heads_min_width = 128      # narrower heads (ppa / citation2 / ddi: H = 32..64) have a k-loop of 2-4 steps: the fused
                                 # kernel's per-tile epilogues dominate and the grouped launches are faster (ddi: 18 vs 31 us)


def heads_small_batch(max_rows: int = -1) -> int:
    """ocn_hip.h: ocn_heads_small_batch — batches up to ``max_rows`` candidates take the latency form of the fused head (same
    bits); returns the previous bound, a negative argument only queries."""
    return int(_lib.lib().ocn_heads_small_batch(int(max_rows)))


@_on_device
def heads_fused(x1: Tensor, x2: Tensor, xij: Tensor, pack: dict, ranges: Optional[Tensor], y_row_map: Optional[Tensor],
                b_on_union: bool, scratch: Tensor, dump: Optional[Tensor] = None) -> Optional[Tensor]:
    """ocn_hip.h: ocn_heads_fused.  ``pack``: panels / vectors prepared by ``model._CNPredictorBase._fused_pack``.
    Returns the [B, 1] scores (or None in constants mode, where ``dump`` receives what a skipped branch contributes;
    scoring mode reads it back as ``pack["cpark"]``)."""
    for t, nm in ((x1, "xcn1"), (x2, "xcn2"), (xij, "xij")):
        _req(t, torch.float32, nm, 2)
    B, H = xij.shape
    if x1.shape != (B, H) or x2.shape != (B, H) or H not in HEADS_WIDTHS:
        raise ValueError("heads_fused: shape mismatch")
    a = _lib.OcnHeadsArgs()
    a.x[0], a.x[1], a.x[2] = x1.data_ptr(), x2.data_ptr(), xij.data_ptr()
    a.ldx, a.B, a.H = H, B, H
    for i in range(3):
        a.p_first[i] = pack["first"][i].data_ptr()
        a.p_out[i] = pack["out"][i].data_ptr()
    a.p_mid[0], a.p_mid[1] = pack["mid"][0].data_ptr(), pack["mid"][1].data_ptr()
    a.vec = pack["vec"].data_ptr()
    a.ranges = 0 if ranges is None else _req(ranges, torch.int64, "ranges").data_ptr()
    a.y_row_map = 0 if y_row_map is None else _req(y_row_map, torch.int64, "y_row_map", 1).data_ptr()
    y = None
    nconst = int(_lib.lib().ocn_heads_const_bytes(H)) // 4
    if dump is None:
        y = torch.empty((B, 1), dtype=torch.float32, device=xij.device)
        a.y = y.data_ptr()
        if _req(pack["cpark"], torch.float32, "cpark", 1).numel() != nconst:
            raise ValueError("heads_fused: constants buffer of the wrong size")
        a.cpark = pack["cpark"].data_ptr()
    else:
        if B != 1 or _req(dump, torch.float32, "dump", 1).numel() != nconst:
            raise ValueError("heads_fused: constants mode takes one row and ocn_heads_const_bytes(H) bytes")
        a.dump = dump.data_ptr()
    if scratch.numel() * scratch.element_size() < int(_lib.lib().ocn_heads_scratch_bytes(H)):
        raise ValueError("heads_fused: scratch too small")
    a.scratch = scratch.data_ptr()
    a.eps, a.ln, a.b_on_union = float(pack["eps"]), int(pack["ln"]), int(bool(b_on_union))
    _mark("mlp_glue")
    check(_lib.lib().ocn_heads_fused(ctypes.byref(a), stream_ptr()), "ocn_heads_fused")
    _mark("linear", pack["flops_per_row"] * B)
    return y


def linear_ok(x: Tensor, weight: Tensor) -> bool:
    return (fast_linear and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous()
            and weight.shape[0] in LINEAR_WIDTHS and weight.shape[1] % 16 == 0 and weight.shape[1] == x.shape[1])


@_on_device
def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, ln=None, relu: bool = False,
           dot=None, y_row_map: Optional[Tensor] = None) -> Tensor:
    """epilogue(x @ weight.T + bias): optional LayerNorm ``ln=(gamma, beta, eps)``, ReLU, and a
    trailing Linear(N -> 1) ``dot=(w[1,N], b[1] | None)`` (then the result is [M, 1]; with
    ``y_row_map`` row r of x lands in row y_row_map[r] of it)."""
    _req(x, torch.float32, "x", 2)
    M, K = x.shape
    N = weight.shape[0]
    if y_row_map is not None:
        if dot is None:
            raise ValueError("y_row_map needs the dot epilogue")
        y = torch.empty((M, 1), dtype=torch.float32, device=x.device)
        linear_grouped([dict(x=x, weight=weight, bias=bias, ln=ln, relu=relu, dot=(dot[0].reshape(1, -1), dot[1]), y=y,
                             y_row_map=y_row_map)], K, N)
        return y
    panel = linear_panel(weight)
    _mark("mlp_glue")
    y = torch.empty((M, 1) if dot is not None else (M, N), dtype=torch.float32, device=x.device)
    g = b = None
    eps = 0.0
    if ln is not None:
        g, b, eps = ln
    dw = db = None
    if dot is not None:
        dw, db = dot
        dw = dw.reshape(-1)
    check(_lib.lib().ocn_linear_bf16x6(ptr(x), M, K, ptr(panel), N, ptr(bias), ptr(g), ptr(b), float(eps),
                                       int(relu), ptr(dw), ptr(db), ptr(y), stream_ptr()), "ocn_linear_bf16x6")
    _mark("linear", 2.0 * M * K * N)
    return y


def linear_ok_t(gy: Tensor, weight: Tensor) -> bool:
    """``linear_t`` takes these shapes (the product gy @ weight as a Linear whose weight is weight^T)."""
    return (fast_linear and gy.is_cuda and gy.dim() == 2 and gy.dtype == torch.float32
            and weight.shape[1] in LINEAR_WIDTHS and weight.shape[0] % 16 == 0 and weight.shape[0] == gy.shape[1])


@_on_device
def linear_t(gy: Tensor, weight: Tensor) -> Tensor:
    """gy @ weight ([M, N] x [N, K] -> [M, K]) on the bf16x6 MFMA kernel: the input gradient of y = x @ weight^T.
    The transposed panel is built per call (the weight changes with every optimiser step)."""
    gy = _req(gy.contiguous(), torch.float32, "gy", 2)
    wt = _req(weight.detach().t().contiguous(), torch.float32, "weight^T", 2)          # [K, N]: K outputs, N inputs
    K, N = wt.shape
    M = gy.shape[0]
    panel = torch.empty(int(_lib.lib().ocn_linear_panel_bytes(K, N)), dtype=torch.uint8, device=wt.device)
    check(_lib.lib().ocn_linear_split_weight(ptr(wt), K, N, ptr(panel), stream_ptr()), "ocn_linear_split_weight")
    y = torch.empty((M, K), dtype=torch.float32, device=gy.device)
    check(_lib.lib().ocn_linear_bf16x6(ptr(gy), M, N, ptr(panel), K, None, None, None, 0.0, 0, None, None, ptr(y),
                                       stream_ptr()), "ocn_linear_bf16x6")
    return y


@_on_device
def wgrad(gy: Tensor, x: Tensor, with_bias: bool = True):
    """Weight / bias gradient of ``y = x @ W^T + b``: ``(gy^T @ x, gy.sum(0))`` — ``ocn_wgrad`` (bf16x6 on the matrix
    cores, split over the batch, partial results added in slice order: deterministic).  gy [B, N], x [B, K] fp32 with
    unit column stride (row strides free)."""
    if gy.dim() == 2 and gy.stride(1) != 1:
        gy = gy.contiguous()
    if x.dim() == 2 and x.stride(1) != 1:
        x = x.contiguous()
    _req_strided(gy, "gy")
    _req_strided(x, "x")
    B, N = gy.shape
    K = x.shape[1]
    if x.shape[0] != B:
        raise ValueError(f"wgrad: {tuple(gy.shape)} against {tuple(x.shape)}")
    l = _lib.lib()
    ws = torch.empty(int(l.ocn_wgrad_workspace_bytes(B, N, K)), dtype=torch.uint8, device=gy.device)
    gw = torch.empty((N, K), dtype=torch.float32, device=gy.device)
    gb = torch.empty((N,), dtype=torch.float32, device=gy.device) if with_bias else None
    ldy = gy.stride(0) if B > 1 else max(N, gy.stride(0))
    ldx = x.stride(0) if B > 1 else max(K, x.stride(0))
    check(l.ocn_wgrad(ptr(gy), ldy, ptr(x), ldx, B, N, K, ptr(gw), ptr(gb) if with_bias else None, ptr(ws), stream_ptr()),
          "ocn_wgrad")
    return gw, gb


@_on_device
def coo_to_csr(row: Tensor, col: Tensor, n_rows: int, n_cols: int, symmetrize: bool = False, dedupe: bool = False,
               check_range: bool = True):
    """(rowptr int64[n_rows + 1], col int32[nnz']) of the pattern with entries (row[q], col[q]) — plus the transposed
    entries when ``symmetrize`` — columns ascending per row, duplicates removed when ``dedupe`` (``ocn_coo_to_csr``:
    counting pass, chained scan, fill, per-row sorts; no sort over the edge list).  ONE device -> host read (entries
    written and the range status)."""
    row = _req(row.contiguous(), torch.int64, "row", 1)
    col = _req(col.contiguous(), torch.int64, "col", 1)
    nnz = row.numel()
    if col.numel() != nnz:
        raise ValueError("coo_to_csr: row and col differ in length")
    l = _lib.lib()
    dev = row.device
    ws = torch.empty(int(l.ocn_coo_to_csr_workspace_bytes(nnz, n_rows, int(symmetrize), int(dedupe))), dtype=torch.uint8, device=dev)
    rowptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
    out = torch.empty(nnz * (2 if symmetrize else 1), dtype=torch.int32, device=dev)
    res = torch.empty(2, dtype=torch.int64, device=dev)
    check(l.ocn_coo_to_csr(ptr(row), ptr(col), nnz, n_rows, n_cols, int(symmetrize), int(dedupe), ptr(rowptr), ptr(out),
                           ptr(ws), ptr(res), stream_ptr()), "ocn_coo_to_csr")
    n_out, bad = (int(v) for v in res.tolist())
    if bad:          # (the status is on the host anyway: out-of-range entries are never dropped silently; `check_range` kept for callers)
        raise IndexError("SparseTensor: index out of range for sparse_sizes")
    if n_out < 0:
        raise _lib.OcnHipError("ocn_coo_to_csr: scan state was not zero")
    return rowptr, out[:n_out]


def linear_grouped(groups, K: int, N: int) -> None:
    """Up to five Linear layers of the same (K, N) in one launch.  Each group is a dict:
    x [M, >=K] (row stride may exceed K), weight [N, K], y (preallocated, row stride >= N, or [M, 1] with
    ``dot``), optional bias, ln=(gamma, beta, eps), relu, scale (device float[1]), addend [M, >=N] (or one
    row [1, N] with ``add_bcast``), dot=(w[1, N], b[1]), row_range (device int64[2]: only rows
    [begin, end) of the buffers), y_row_map (device int64[M]: destination row of the dot output)."""
    if not 1 <= len(groups) <= 5:
        raise ValueError("1..5 groups")
    idx = groups[0]["x"].device.index
    if idx is not None and idx != torch._C._cuda_getDevice():
        with torch.cuda.device(idx):
            return linear_grouped(groups, K, N)
    arr = (_lib.OcnLinearGroup * len(groups))()
    keep = []
    for a, g in zip(arr, groups):
        x, w, y = g["x"], g["weight"], g["y"]
        _req_strided(x, "x"); _req_strided(y, "y")
        if x.shape[1] != K or tuple(w.shape) != (N, K) or x.shape[0] != y.shape[0]:
            raise ValueError("linear_grouped: shape mismatch")
        panel = linear_panel(w)
        keep.append(panel)
        ln = g.get("ln")
        dot = g.get("dot")
        add = g.get("addend")
        a.X, a.ldX, a.M, a.Wp = x.data_ptr(), x.stride(0), x.shape[0], panel.data_ptr()
        a.bias = 0 if g.get("bias") is None else g["bias"].data_ptr()
        a.gamma, a.beta, a.eps = (0, 0, 0.0) if ln is None else (ln[0].data_ptr(), ln[1].data_ptr(), float(ln[2]))
        a.relu = int(bool(g.get("relu")))
        a.scale = 0 if g.get("scale") is None else g["scale"].data_ptr()
        if add is not None:
            _req_strided(add, "addend")
            a.addend, a.ldAdd = add.data_ptr(), add.stride(0)
            a.add_bcast = int(bool(g.get("add_bcast")))
            if a.add_bcast and add.shape != (1, N):
                raise ValueError("add_bcast: addend must be [1, N]")
        rr, rm = g.get("row_range"), g.get("y_row_map")
        if rr is not None:
            _req(rr, torch.int64, "row_range", 1)
            a.row_range = rr.data_ptr()
        if rm is not None:
            _req(rm, torch.int64, "y_row_map", 1)
            a.y_row_map = rm.data_ptr()
        if dot is not None:
            a.dotw = dot[0].data_ptr()
            a.dotb = 0 if dot[1] is None else dot[1].data_ptr()
            a.ldY = 1
        else:
            a.ldY = y.stride(0)
        a.Y = y.data_ptr()
    _mark("mlp_glue")
    check(_lib.lib().ocn_linear_grouped(ctypes.cast(arr, ctypes.c_void_p), len(groups), K, N, stream_ptr()),
          "ocn_linear_grouped")
    _mark("linear", sum(2.0 * g["x"].shape[0] * K * N for g in groups))


def _req_strided(t: Tensor, name: str) -> None:
    """2-d fp32 device tensor whose rows are contiguous (row stride may exceed the width)."""
    if not (isinstance(t, Tensor) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
        raise ValueError(f"{name}: expected a 2-d fp32 device tensor with contiguous rows")
