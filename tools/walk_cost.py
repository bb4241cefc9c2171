"""Work of the walk-count route per candidate batch, from either endpoint:  wc[k] = |N(k) ∩ N(j)| for
k in N(i) can be enumerated from i's side (Σ_{k∈N(i)} d_k probes) or from j's side (Σ_{m∈N(j)} d_m)."""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser(); ap.add_argument("--config", default="citation2"); a = ap.parse_args()
args = argparse.Namespace(dataset=a.config, scale=1.0, hiddim=None, predictor=None, batch=None)
dev = torch.device("cuda:0")
wl = bench.build_workload(args, dev, 0, 1)
adj, e = wl["adj"], wl["edges"]
rowptr, col = adj._rowptr, adj._col.long()
deg = (rowptr[1:] - rowptr[:-1])
rows = torch.repeat_interleave(torch.arange(deg.numel(), device=dev), deg)
S = torch.zeros(deg.numel(), dtype=torch.int64, device=dev).index_add_(0, rows, deg[col])
i, j = e[0], e[1]
Si, Sj, di, dj = S[i].double(), S[j].double(), deg[i].double(), deg[j].double()
print("B", e.shape[1], "mean d_i", di.mean().item(), "mean d_j", dj.mean().item(), "max d_i", di.max().item())
print("i-side work  Σ S_i          ", Si.sum().item())
print("j-side work  Σ S_j          ", Sj.sum().item())
print("min side     Σ min(S_i,S_j) ", torch.minimum(Si, Sj).sum().item())
for logd in (8, 11, 13):
    cj = Sj * logd + di
    pick = cj < Si
    print(f"cost model j-side = S_j*{logd} + d_i: Σ min = {torch.where(pick, cj, Si).sum().item():.3e}  j-side edges {pick.float().mean().item():.2f}")
srt = torch.sort(Si, descending=True).values
print("share of i-side work in top 1% / 10% edges:", (srt[: max(1, len(srt)//100)].sum() / srt.sum()).item(), (srt[: len(srt)//10].sum() / srt.sum()).item())

# per-chunk element totals of the forward sweep (64 consecutive neighbours of i per work item)
a0 = rowptr[i]; d = deg[i]
nch = (d + 63) // 64
slot = torch.repeat_interleave(torch.arange(len(i), device=dev), nch)
first = torch.cumsum(nch, 0) - nch
c_in = torch.arange(int(nch.sum()), device=dev) - first[slot]
lo = a0[slot] + c_in * 64
hi = torch.minimum(lo + 64, a0[slot] + d[slot])
degc = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(deg[col], 0)])
T = degc[hi] - degc[lo]
srt = torch.sort(T, descending=True).values
print("chunks", len(T), "Σ T", T.sum().item(), "max T", srt[0].item(), "top-10", srt[:10].tolist())
print("chunks with T > 8192:", (T > 8192).sum().item(), " their share of Σ T:", (T[T > 8192].sum() / T.sum()).item())
print("items if split at 8192:", ((T + 8191) // 8192).sum().item())

from ocn_amd.utils import CNState
from ocn_amd import ops
ops.validate_indices = False
st = CNState(adj, None, None, e, walk=True)
n = int(st.off[-1])
wcs = st.wc[:n].long()
print("positions Σ d_i", n, " hits Σ wc", wcs.sum().item(), " max wc", wcs.max().item(), " nonzero", (wcs > 0).sum().item())
# hits per edge vs elements per edge
seg = torch.repeat_interleave(torch.arange(len(i), device=dev), deg[i])
hits_e = torch.zeros(len(i), dtype=torch.int64, device=dev).index_add_(0, seg, wcs)
print("hit rate overall", wcs.sum().item() / Si.sum().item(), " edges with hit rate > 10%:", ((hits_e.double() / Si.clamp(min=1)) > 0.1).sum().item())
top = torch.argsort(Si, descending=True)[:8]
for t in top.tolist():
    print("  edge", t, "d_i", int(di[t]), "d_j", int(dj[t]), "S_i", int(Si[t]), "S_j", int(Sj[t]), "hits", int(hits_e[t]))
