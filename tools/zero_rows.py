import argparse, os, sys
sys.path.insert(0, os.getcwd())
import torch, bench
from ocn_amd.utils import CNState
a = argparse.Namespace(dataset="collab", scale=1.0, hiddim=None, predictor=None, batch=None)
wl = bench.build_workload(a, torch.device("cuda:0"), 0, 1)
st = CNState(wl["adj"], wl["adj"], wl["adj2"], wl["edges"])
c1, c2 = st.cnt1, st.cnt2
B = c1.numel()
print("B", B, "cnt1==0:", (c1 == 0).float().mean().item(), "cnt2==0:", (c2 == 0).float().mean().item(),
      "both 0:", ((c1 == 0) & (c2 == 0)).float().mean().item())
w = st.weights_cn5(torch.zeros(1, device="cuda"))
x1, x2, _ = st.gather(w, wl["h"])
print("xcn1 zero rows:", (x1.abs().amax(1) == 0).float().mean().item(), "xcn2 zero rows:", (x2.abs().amax(1) == 0).float().mean().item())
