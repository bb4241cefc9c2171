"""Timing of the walk-count intersection stage (ppa / citation2 route) on the bench workload, for the
shipped library and for experimental builds (compile flags as in tools/kbench.py).  Experiments only.

    python tools/walkbench.py citation2 [-DOCN_X_WALK_NOSWEEP ...]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import argparse
    import torch
    import bench
    from ocn_amd import ops
    from ocn_amd.utils import CNState
    args = argparse.Namespace(dataset=os.environ["KB_DATASET"], scale=1.0, hiddim=None, predictor=None, batch=None, batches=1, innerprod=0.0)
    dev = torch.device("cuda:0")
    wl = bench.build_workload(args, dev, 0, 1)
    adj, e = wl["adj"], wl["edges"][0]
    ops.validate_indices = False
    out = []
    for two in (False, True):
        ops.walk_two_sided = two
        ws = {}
        for _ in range(3):
            st = CNState(adj, None, None, e, walk=True, ws=ws)
        torch.cuda.synchronize()
        t = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); st = CNState(adj, None, None, e, walk=True, ws=ws); b.record()
            torch.cuda.synchronize()
            t.append(a.elapsed_time(b))
        t.sort()
        chk = int(st.cnt1.sum()) * 1000003 + int(st.cnt2.sum()) + int(st.wc[: int(st.off[-1])].sum()) * 7
        out.append(f"{'two-sided' if two else 'forward  '} {t[len(t) // 2] * 1e3:8.1f}us chk={chk}")
    print(" | ".join(out), "checksum", flush=True)


def main():
    from ocn_amd import _lib
    ds = sys.argv[1]
    for v in [()] + [tuple(a.split(",")) for a in sys.argv[2:]]:
        out = _lib.LIB_PATH if not v else f"/tmp/libocn_{abs(hash(v))}.so"
        if v:
            _lib.build(force=True, extra_flags=v, out=out)
        env = dict(os.environ, OCN_LIB_PATH=out, KB_CHILD="1", KB_DATASET=ds)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if "checksum" in l]
        print(f"{' '.join(v) or 'shipped':28s} {line[-1] if line else r.stderr[-800:]}", flush=True)


if __name__ == "__main__":
    child() if os.environ.get("KB_CHILD") else main()
