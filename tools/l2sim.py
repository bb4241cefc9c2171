"""CPU-side model of the pooling kernel's L2 behaviour at the collab shape (no GPU): which visiting orders / feature
slicings of cn_gather_kernel would cut its traffic past L2 (1.07 GB moved for 0.34 GB compulsory, TCC hit 53 %).

    python tools/l2sim.py [--scale 1.0] [--batch 65536]

Builds the bench's seeded graph and candidate batch, forms every candidate's CN column list with scipy (cn1 u cn2 of the
pattern route), and replays the row fetches of several schedules through per-XCD LRU caches (4 MiB each, 16-way-ish modelled as
fully associative LRU over 128-byte lines / whole rows).  Concurrency is modelled by interleaving W candidates per XCD round robin.
Prints hit rates and bytes past L2 per scheme.  A design aid, not a measurement: the numbers that count are the PMC passes."""
import argparse
import sys
import os
import time
from collections import OrderedDict

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocn_amd.synth import dataset_like, sample_edges  # noqa: E402


def lru_stream(stream, capacity):
    """hits, misses of an access stream (iterable of hashable line ids) through an LRU of `capacity` lines."""
    od = OrderedDict()
    hit = 0
    n = 0
    for a in stream:
        n += 1
        if a in od:
            od.move_to_end(a)
            hit += 1
        else:
            od[a] = None
            if len(od) > capacity:
                od.popitem(last=False)
    return hit, n - hit


def interleave(cands, lists, W):
    """Round-robin interleaving of the candidates' row lists with W candidates in flight (a model of the waves resident
    on one XCD): yields row ids."""
    active = []
    it = iter(cands)
    out = []
    def refill():
        while len(active) < W:
            c = next(it, None)
            if c is None:
                return
            l = lists[c]
            if len(l):
                active.append([l, 0])
    refill()
    while active:
        nxt = []
        for st in active:
            l, p = st
            # a wave fetches 8 rows per round (UNR of pool_range)
            out.extend(l[p:p + 8].tolist())
            st[1] = p + 8
            if st[1] < len(l):
                nxt.append(st)
        active = nxt
        refill()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--W", type=int, default=512, help="candidates in flight per XCD")
    ap.add_argument("--H", type=int, default=256)
    args = ap.parse_args()
    t0 = time.time()
    ei, n, shape = dataset_like("collab", seed=0, scale=args.scale)
    r = torch.cat([ei[0], ei[1]]).numpy()
    c = torch.cat([ei[1], ei[0]]).numpy()
    A = sp.csr_matrix((np.ones(r.size, dtype=np.int8), (r, c)), shape=(n, n))
    A.sum_duplicates()
    A.data[:] = 1
    A.sort_indices()
    print(f"graph n={n} nnz={A.nnz} ({time.time() - t0:.1f}s)")
    e = sample_edges(torch.from_numpy(A.tocoo().row.astype(np.int64)), torch.from_numpy(A.tocoo().col.astype(np.int64)), n,
                     args.batch, seed=1).numpy()
    src, dst = e[0], e[1]
    # cn1 u cn2 rows: N(i) ∩ (N(j) u N2(j)); N2(j) = pattern of (A A)[j]
    R1 = A[src]                                   # B x N
    Aj = A[dst]
    A2j = (Aj @ A)                                # B x N : 2-walk counts from j
    A2j.data[:] = 1
    U = (Aj + A2j)
    U.data[:] = 1
    CN = R1.multiply(U).tocsr()
    CN.sort_indices()
    B = args.batch
    lists = [CN.indices[CN.indptr[q]:CN.indptr[q + 1]] for q in range(B)]
    nent = CN.nnz
    distinct = np.unique(CN.indices).size
    rowb = args.H * 4
    print(f"entries {nent} ({nent / B:.1f}/cand), distinct rows {distinct}, formula bytes {nent * rowb / 1e9:.2f} GB, "
          f"distinct-row bytes {distinct * rowb / 1e6:.0f} MB ({time.time() - t0:.1f}s)")
    cost = np.array([len(l) for l in lists])
    order = np.argsort(src, kind="stable")        # the counting sort by source
    L2 = 4 << 20

    def run(name, per_xcd_cands, line_bytes, lines_per_row_in_xcd, cap_bytes=L2, W=args.W):
        hits = miss = 0
        for x, cands in enumerate(per_xcd_cands):
            st = interleave(cands, lists, W)
            h, m = lru_stream(st, int(cap_bytes * 0.75) // line_bytes)      # 3/4 of the L2 for the rows (flags, ids, outputs stream through)
            hits += h
            miss += m
        past = miss * line_bytes
        print(f"{name:55s} hit {hits / max(hits + miss, 1):.3f}  past-L2 {past / 1e9:.3f} GB")
        return past

    eighth = np.array_split(order, 8)
    run("source order, XCD eighths (round 2)", eighth, rowb, 1)

    def longest_first(cands):
        # groups of 4 slots, 64 cost buckets (bucket = max cost >> 5), stable
        g = cands[: len(cands) // 4 * 4].reshape(-1, 4)
        gc = np.minimum(cost[g].max(1) >> 5, 63)
        o = np.argsort(-gc, kind="stable")
        return g[o].reshape(-1)
    run("source order, eighths, longest first (shipped r3)", [longest_first(c_) for c_ in eighth], rowb, 1)
    for seg in (64, 128, 256, 512, 1024):
        def seg_first(cands, seg=seg):
            return np.concatenate([longest_first(cands[q:q + 4 * seg]) for q in range(0, len(cands), 4 * seg)])
        run(f"... longest first inside segments of {seg} groups", [seg_first(c_) for c_ in eighth], rowb, 1)
    q = np.percentile(cost, [50, 90, 99, 99.9, 100])
    print("entries per candidate: median %d, p90 %d, p99 %d, p99.9 %d, max %d; share of entries in candidates > 64: %.2f, > 128: %.2f" % (
        *q, cost[cost > 64].sum() / cost.sum(), cost[cost > 128].sum() / cost.sum()))
    # feature slicing: every XCD sees ALL candidates, 1/S of a row each
    for S in (2, 4, 8):
        groups = 8 // S                                           # XCDs per slice; candidates dealt over them in contiguous chunks
        parts = np.array_split(order, groups)
        per = [parts[x % groups] for x in range(8)]               # XCD x: slice x // groups ... each (slice, chunk) pair once
        # XCDs with the same chunk and different slices behave identically: simulate `groups` of them and scale
        run(f"{S} feature slices ({rowb // S} B per row and XCD), source order", [longest_first(p) for p in per], rowb // S, 1)
    # column-clustered orders within an eighth
    first = np.array([l[0] if len(l) else -1 for l in lists])
    med = np.array([l[len(l) // 2] if len(l) else -1 for l in lists])
    for nm, key in (("min CN column", first), ("median CN column", med)):
        o2 = np.lexsort((src, key))
        run(f"order by {nm}, eighths", np.array_split(o2, 8), rowb, 1)
        o3 = np.lexsort((src, key >> 10))
        run(f"order by {nm} >> 10 then source, eighths", np.array_split(o3, 8), rowb, 1)
    o4 = np.lexsort((src, src >> 10))
    run("source order (control: same as first)", np.array_split(o4, 8), rowb, 1)
    for W in (64, 2048):
        run(f"source order eighths, W={W}", eighth, rowb, 1, W=W)
    print(f"done ({time.time() - t0:.1f}s)")


if __name__ == "__main__":
    main()
