"""Which objects of a training step are only freed by Python's cycle collector (they hold GPU memory until it runs, and a gen-2
pass over them is a 0.1 s pause)?  Runs trainbench-like steps with the collector off, then collects with DEBUG_SAVEALL and
prints the types found.   python tools/cyclecheck.py"""
import gc
import os
import sys
from collections import Counter
from types import SimpleNamespace

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import ocn_amd.model as M  # noqa: E402
from ocn_amd.sparse import SparseTensor  # noqa: E402
from ocn_amd.synth import dataset_like  # noqa: E402
from ocn_amd.utils import adjoverlap  # noqa: E402

dev = torch.device("cuda:0")
cfg = dict(bench.CONFIGS["collab"])
H, B = cfg["H"], 8192
ei, n, shape = dataset_like("collab", seed=0, scale=0.1)
pos = ei.to(dev)
x = torch.randn(n, shape["feat"], device=dev)
model = M.GCN(shape["feat"], H, H, 1, 0.05, True, False, -1, "gin", True, 0.0, xdropout=0.7, taildropout=0.3).to(dev).train()
pred = M.predictor_dict["cn5"](H, H, 1, 3, 0.05, 0.4, True).to(dev).train()
opt = torch.optim.Adam(list(model.parameters()) + list(pred.parameters()), lr=1e-3)
args = SimpleNamespace(sum=1.0)
neg = torch.randint(0, n, pos.shape, device=dev)


def step(k):
    perm = torch.arange(k * B, (k + 1) * B, device=dev) % pos.shape[1]
    opt.zero_grad()
    mask = torch.ones(pos.shape[1], dtype=torch.bool, device=dev)
    mask[perm] = False
    adj = SparseTensor.from_edge_index(pos[:, mask], sparse_sizes=(n, n)).to_symmetric()
    h = model(x, adj)
    sp = adj.to_torch_sparse_coo_tensor()
    adj2 = SparseTensor.from_torch_sparse_coo_tensor(sp @ sp, False)
    e = pos[:, perm]
    p = pred.multidomainforward(h, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e, args, cndropprobs=[])
    e = neg[:, perm]
    q = pred.multidomainforward(h, adj, adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e), e, args, cndropprobs=[])
    loss = -F.logsigmoid(p).mean() - F.logsigmoid(-q).mean()
    loss.backward()
    opt.step()


for k in range(3):
    step(k)
torch.cuda.synchronize()
gc.collect()
gc.disable()
gc.set_debug(gc.DEBUG_SAVEALL)
for k in range(3, 5):
    step(k)
torch.cuda.synchronize()
found = gc.collect()
types = Counter(type(o).__module__ + "." + type(o).__qualname__ for o in gc.garbage)
print("objects only the cycle collector frees, after 2 steps:", found)
for t, c in types.most_common(25):
    print(f"  {c:6d}  {t}")
tens = [o for o in gc.garbage if isinstance(o, torch.Tensor)]
print("tensors among them:", len(tens), "bytes:", sum(t.numel() * t.element_size() for t in tens if t.is_cuda))
