"""Scratch measurement: sequential predictor calls vs the two-stream scorer on the bench workload."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    a = ap.parse_args()
    args = argparse.Namespace(dataset="collab", scale=1.0, hiddim=256, predictor="cn5", batch=65536)
    dev = torch.device("cuda:0")
    from ocn_amd import ops
    from ocn_amd.pipeline import TwoStreamScorer
    from ocn_amd.utils import adjoverlap
    wl = bench.build_workload(args, dev, 0, 1)
    pred, h, adj, adj2, e = wl["pred"], wl["h"], wl["adj"], wl["adj2"], wl["edges"]
    with torch.no_grad():
        ref = pred(h, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e)
    ops.validate_indices = False

    def seq():
        with torch.no_grad():
            return [pred(h, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e) for _ in range(a.steps)]

    sc = TwoStreamScorer(pred, dev)

    def pipe():
        with torch.no_grad():
            sc.begin()
            outs = [sc.submit(h, adj, adj2, e) for _ in range(a.steps)]
            sc.end(outs)
            return outs

    for name, fn in (("sequential", seq), ("two-stream", pipe), ("sequential", seq), ("two-stream", pipe)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok = all(torch.equal(o, ref) for o in outs)
        print(f"{name:11s} {dt / a.steps * 1e3:.3f} ms/step  {e.shape[1] * a.steps / dt / 1e6:.1f} M edges/s  identical={ok}")


if __name__ == "__main__":
    main()
