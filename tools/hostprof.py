"""cProfile of the host side of one predictor step (where the 0.3 ms of Python per batch goes)."""
import argparse
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--config", default="collab"); ap.add_argument("--scale", type=float, default=1.0)
a = ap.parse_args()
args = argparse.Namespace(dataset=a.config, scale=a.scale, hiddim=None, predictor=None, batch=None)
dev = torch.device("cuda:0")
wl = bench.build_workload(args, dev, 0, 1)
from ocn_amd import ops  # noqa: E402
pred, h, adj, e = wl["pred"], wl["h"], wl["adj"], wl["edges"]


def step():
    with torch.no_grad():
        c1, c2 = bench.cn_handles(wl, e)
        return pred(h, adj, c1, c2, e, wl["args"])


step(); ops.validate_indices = False
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
