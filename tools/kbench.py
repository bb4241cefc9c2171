"""Per-kernel timing on the bench workload, optionally for experimental builds of the library.

    python tools/kbench.py                        # shipped library
    python tools/kbench.py -DOCN_X_NOATOMIC ...   # each flag set = one variant build in /tmp

Every variant runs in its own subprocess (OCN_LIB_PATH) and prints HIP-event times of the CN-stage
kernels.  Experiments only — nothing here is part of the product or the tests.
"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import argparse

    import torch

    import bench
    from ocn_amd import ops
    from ocn_amd.utils import CNState
    args = argparse.Namespace(dataset=os.environ.get("KB_DATASET", "collab"), scale=1.0,
                              hiddim=int(os.environ.get("KB_H", "256")), predictor="cn5",
                              batch=int(os.environ.get("KB_B", "65536")), batches=1, innerprod=0.0)
    dev = torch.device("cuda:0")
    wl = bench.build_workload(args, dev, 0, 1)
    adj, adj2, h, e = wl["adj"], wl["adj2"], wl["h"], wl["edges"][0]
    ops.validate_indices = False
    iters = int(os.environ.get("KB_ITERS", "20"))
    mode = os.environ.get("KB_SORT", "")
    if mode == "src":
        e = e[:, torch.argsort(e[0])].contiguous()
    elif mode == "src_xcd":       # sorted by src, then dealt so that blocks b, b+8, b+16.. (one XCD) get a contiguous run
        o = torch.argsort(e[0])
        nb = e.shape[1]
        b = torch.arange(nb, device=dev)
        pos = (b % 8) * (nb // 8) + b // 8
        e = e[:, o[pos]].contiguous()
    elif mode == "dst":
        e = e[:, torch.argsort(e[1])].contiguous()

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        t = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record()
            torch.cuda.synchronize()
            t.append(a.elapsed_time(b))
        t.sort()
        return t[len(t) // 2], t[0]

    tm = bench.StageTimer(pool=16 * (iters + 2), lean=False)
    ops.stage_timer = tm
    for _ in range(iters + 2):
        tm.mark("begin")
        st = CNState(adj, adj, adj2, e)
        w = st.weights_cn5(torch.zeros(1, device=dev))
        st.gather(w, h)
    torch.cuda.synchronize()
    ops.stage_timer = None
    tot = tm.totals()
    chk = int(st.cnt1.sum()) * 1000003 + int(st.cnt2.sum())
    print(" ".join(f"{k}={v[0]*1e3:.1f}us" for k, v in tot.items()), f"checksum={chk}", flush=True)


def main():
    variants = [()] + [tuple(a.split(",")) for a in sys.argv[1:]]
    from ocn_amd import _lib
    for v in variants:
        out = _lib.LIB_PATH if not v else f"/tmp/libocn_{abs(hash(v))}.so"
        if v:
            _lib.build(force=True, extra_flags=v, out=out)
        env = dict(os.environ, OCN_LIB_PATH=out, KB_CHILD="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if "checksum" in l]
        print(f"{' '.join(v) or 'shipped':40s} {line[-1] if line else r.stderr[-400:]}", flush=True)


if __name__ == "__main__":
    child() if os.environ.get("KB_CHILD") else main()
