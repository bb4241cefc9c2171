"""The order-exact column sums of a trained cn5 (innerprod != 0) at the collab shape, kernel by kernel:
    python tools/colsumbench.py [-D flags ...]
HIP-event time of ocn_cn_colsum_exact + weights per batch, the number of ordered / long columns, S2 checksum."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = tuple(f for f in sys.argv[1:] if f.startswith("-D"))
if flags:
    os.environ["OCN_LIB_PATH"] = "/tmp/libocn_cs.so"
import argparse  # noqa: E402

import torch  # noqa: E402
from ocn_amd import _lib  # noqa: E402

if flags:
    _lib.build(force=True, extra_flags=flags, out="/tmp/libocn_cs.so")
import bench  # noqa: E402
from ocn_amd import ops  # noqa: E402
from ocn_amd.utils import CNState  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    args = argparse.Namespace(dataset="collab", scale=1.0, hiddim=256, predictor="cn5", batch=65536, batches=2, innerprod=0.37)
    wl = bench.build_workload(args, dev, 0, 1)
    adj, adj2 = wl["adj"], wl["adj2"]
    ops.validate_indices = False
    ip = torch.tensor([0.37], device=dev)
    for e in wl["edges"]:
        st = CNState(adj, adj, adj2, e)
        hc = st.hist_counts()
        n1, nu = hc[:, 0], hc[:, 2]
        ordered = (n1 >= 2) & (nu > 0)
        print(f"union entries {int(nu.sum())}, touched columns {int((nu > 0).sum())}, ordered columns {int(ordered.sum())}, "
              f"of them longer than 64 entries: {int((ordered & (nu > 64)).sum())}, longest {int(nu[ordered].max())}", flush=True)
        times = []
        for _ in range(12):
            st = CNState(adj, adj, adj2, e)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            w = st.weights_cn5(ip)
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        times.sort()
        print(f"{' '.join(flags) or 'product':24s} weights with order-exact sums: median {times[len(times) // 2] * 1e3:.1f} us, best {times[0] * 1e3:.1f} us, "
              f"checksum {float(w.double().sum()):.9f}", flush=True)


if __name__ == "__main__":
    main()
