"""The pooling kernel alone on the bench workload, variants interleaved in ONE process (run-to-run and box-to-box noise is
larger than most effects worth keeping):  python tools/poolbench.py [--iters 30]
Variants: the visiting order of the slot groups (ops.heavy_first / ops.sched_segment).  Medians of HIP-event times of
`ocn_gather_schedule` + `ocn_cn_gather` on 4 distinct batches; results must agree bit for bit."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--config", default="collab")
    ap.add_argument("--segments", type=int, nargs="*", default=None)
    ap.add_argument("--no-check", action="store_true", help="variant builds whose results are wrong on purpose")
    a = ap.parse_args()
    import bench
    from ocn_amd import ops
    from ocn_amd.utils import CNState
    args = argparse.Namespace(dataset=a.config, scale=1.0, hiddim=None, predictor=None, batch=None, batches=4, innerprod=0.0)
    dev = torch.device("cuda:0")
    wl = bench.build_workload(args, dev, 0, 1)
    adj, adj2, h = wl["adj"], wl["adj2"], wl["h"]
    ops.validate_indices = False
    variants = [("longest first", True, 0), ("source order", False, 0)] + [(f"segments of {s}", True, s) for s in (a.segments or [])]
    times = {v[0]: [] for v in variants}
    ref = {}
    for rep in range(a.iters):
        for name, heavy, seg in variants:
            ops.heavy_first, ops.sched_segment = heavy, seg
            for bi, e in enumerate(wl["edges"]):
                st = CNState(adj, adj, adj2, e)                    # (sched allocated only when heavy_first)
                w = st.weights_cn5(torch.zeros(1, device=dev))
                torch.cuda.synchronize()
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                out = st.gather(w, h)
                t1.record()
                torch.cuda.synchronize()
                if rep >= 2:
                    times[name].append(t0.elapsed_time(t1) * 1e3)
                if rep == 0:
                    key = tuple(float(o.double().sum()) for o in out)
                    assert a.no_check or ref.setdefault(bi, key) == key, (name, bi)
    for name, _, _ in variants:
        t = sorted(times[name])
        print(f"{name:22s} median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}  p90 {t[int(len(t) * 0.9)]:7.1f}  (n={len(t)})")


if __name__ == "__main__":
    main()
