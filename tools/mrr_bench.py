"""Throughput of the citation2 MRR evaluation layout (NeighborOverlapCitation2.py:227-254): every positive
(source, target) comes with 1000 negatives that share its source.  Experiment / measurement tool."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ocn_amd import ops  # noqa: E402
from ocn_amd.evaluate import Evaluator  # noqa: E402
from ocn_amd.pipeline import score_mrr_split  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--positives", type=int, default=256)
ap.add_argument("--negatives", type=int, default=1000)
ap.add_argument("--batch", type=int, default=2048)          # the driver's --testbs 2048 (README.md:92)
a = ap.parse_args()
args = argparse.Namespace(dataset="citation2", scale=1.0, hiddim=None, predictor=None, batch=None, batches=1, innerprod=0.0)
dev = torch.device("cuda:0")
wl = bench.build_workload(args, dev, 0, 1)
pred, h, adj = wl["pred"], wl["h"], wl["adj"]
g = torch.Generator().manual_seed(3)
r, c, _ = adj.coo()
pick = torch.randint(0, r.numel(), (a.positives,), generator=g).to(dev)
source, target = r[pick], c[pick]                       # positives = existing edges (degree-biased sources)
target_neg = torch.randint(0, wl["n"], (a.positives, a.negatives), generator=g).to(dev)
ev = Evaluator("ogbl-citation2")
n = a.positives * (1 + a.negatives)
res = {}
modes = (int(os.environ["MRR_SHARE"]),) * 2 if "MRR_SHARE" in os.environ else (0, 2, 0, 2)
for share in modes:                                          # A/B in one process: per-candidate sweeps vs shared-source sweep
    ops.walk_share_min = share
    score_mrr_split(pred, h, adj, source[:8], target[:8], target_neg[:8], a.batch, wl["args"], ev)   # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mrr = score_mrr_split(pred, h, adj, source, target, target_neg, a.batch, wl["args"], ev)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res[share] = (dt, mrr)
    print(f"MRR layout, walk_share_min={share}: {n} candidates in {dt * 1e3:.1f} ms = {n / dt / 1e6:.2f} M edges/s (batch {a.batch}, "
          f"mrr {mrr:.6f}, mean deg(source) {float((adj._rowptr[source + 1] - adj._rowptr[source]).float().mean()):.0f})", flush=True)
if len(res) == 2:
    print(f"shared-source sweep: {res[0][0] / res[2][0]:.2f}x, mrr identical: {res[0][1] == res[2][1]}")
