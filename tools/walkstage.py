"""Single-process loop over the walk-count intersection stage of one bench configuration — the target
for `rocprofv3 --pmc ... -- python tools/walkstage.py citation2` (no child processes)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ocn_amd import ops  # noqa: E402
from ocn_amd.utils import CNState  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "citation2"
ops.walk_two_sided = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
args = argparse.Namespace(dataset=cfg, scale=1.0, hiddim=None, predictor=None, batch=None)
dev = torch.device("cuda:0")
wl = bench.build_workload(args, dev, 0, 1)
ops.validate_indices = False
ws = {}
for _ in range(6):
    st = CNState(wl["adj"], None, None, wl["edges"], walk=True, ws=ws)
torch.cuda.synchronize()
print("done", int(st.cnt2.sum()))
