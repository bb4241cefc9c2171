import argparse, sys, os, torch
sys.path.insert(0, "/root/repo")
import bench
from ocn_amd import ops
from ocn_amd.utils import CNState
args = argparse.Namespace(dataset="collab", scale=1.0, hiddim=None, predictor=None, batch=None, batches=1, innerprod=0.0)
dev = torch.device("cuda:0")
wl = bench.build_workload(args, dev, 0, 1)
adj, adj2, h = wl["adj"], wl["adj2"], wl["h"]
ops.validate_indices = False
e = wl["edges"][0]
ops.heavy_first = False
st = CNState(adj, adj, adj2, e)
w = st.weights_cn5(torch.zeros(1, device=dev))
z = torch.zeros_like(st.cnt1)
out_row = torch.arange(e.shape[1], device=dev)
def run(cnt1, cnt2, out_row):
    ts = []
    for _ in range(12):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.cn_gather(adj._rowptr, adj._col, st.src, st.dst, st.off, st.flags, None, w, h, order=st.order, max_row_len=adj.max_rowcount(),
                      out_row=out_row, cnt1=cnt1, cnt2=cnt2)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts)//2]
print("all candidates, real counts (wave per candidate, source order):", run(st.cnt1, st.cnt2, out_row))
print("counts forced to zero (every wave: x_i * x_j only, no xcn stores):", run(z, z, out_row))
print("counts zero, batch order output rows None (stores zeros too):", run(z, z, None))
# an elementwise torch kernel of the same output volume for scale
x = torch.empty(e.shape[1], 256, device=dev)
torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); torch.mul(h[st.src], h[st.dst], out=x); b.record(); torch.cuda.synchronize(); print("torch h[src]*h[dst] (3 launches):", a.elapsed_time(b) * 1e3)
