"""One training step of the reference driver (NeighborOverlap_large.py:28-94: masked adjacency rebuilt per batch, encoder,
A² of the masked graph, positive and negative passes, logsigmoid loss, backward, Adam) on the collab-shaped synthetic
graph, timed per step and meant to be run under `rocprofv3 --kernel-trace --stats` (tools/README.md): the kernel list
of a training step, e.g. that no rocPRIM / torch sort kernel is in it.

    python tools/trainbench.py [--config collab] [--scale 1.0] [--steps 6] [--batch 65536]
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="collab", choices=["collab", "cora"])
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-gc", action="store_true", help="Python's cycle collector off during the timed steps (diagnosis of step-time outliers)")
    a = ap.parse_args()
    import ocn_amd.model as M
    from ocn_amd import ops
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import dataset_like
    from ocn_amd.utils import PermIterator, adjoverlap
    dev = torch.device("cuda:0")
    cfg = dict(bench.CONFIGS[a.config])
    H, B = cfg["H"], a.batch or cfg["batch"]
    ei, n, shape = dataset_like(a.config, seed=0, scale=a.scale)
    pos_train_edge = ei.to(dev)                                   # [2, E]: each undirected edge once
    fin = shape["feat"] or H
    torch.manual_seed(0)
    x = torch.randn(n, fin, device=dev)
    model = getattr(M, cfg["enc"])(fin, H, H, cfg["layers"], 0.05, cfg["ln"], cfg["res"], -1, cfg["conv"], cfg["jk"], 0.0,
                                   xdropout=0.7, taildropout=0.3).to(dev)
    predictor = M.predictor_dict[cfg["pred"]](H, H, 1, cfg["nnlayers"], cfg["predp"], cfg["preedp"], cfg["lnnn"]).to(dev)
    opt = torch.optim.Adam([{"params": model.parameters(), "lr": 0.004}, {"params": predictor.parameters(), "lr": 0.003}])
    args = SimpleNamespace(sum=cfg["sum"], adj2byblock=False)
    negedge = torch.randint(0, n, pos_train_edge.shape, device=dev)
    model.train(); predictor.train()
    adjmask = torch.ones_like(pos_train_edge[0], dtype=torch.bool)
    times, losses, mallocs = [], [], []
    it = iter(PermIterator(dev, adjmask.shape[0], B))
    for step in range(a.warmup + a.steps):
        if a.no_gc and step == a.warmup:
            import gc
            gc.collect(); gc.disable()
        try:
            perm = next(it)
        except StopIteration:
            it = iter(PermIterator(dev, adjmask.shape[0], B))
            perm = next(it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        opt.zero_grad()
        adjmask[perm] = 0
        tei = pos_train_edge[:, adjmask]
        adj = SparseTensor.from_edge_index(tei, sparse_sizes=(n, n)).to_device(dev, non_blocking=True)
        adjmask[perm] = 1
        adj = adj.to_symmetric()
        h = model(x, adj)
        spadj = adj.to_torch_sparse_coo_tensor()
        adj2 = SparseTensor.from_torch_sparse_coo_tensor(spadj @ spadj, False)
        edge = pos_train_edge[:, perm]
        pos = predictor.multidomainforward(h, adj, adjoverlap(adj, adj, edge, False), adjoverlap(adj, adj2, edge, False), edge, args,
                                           cndropprobs=[])
        edge = negedge[:, perm]
        neg = predictor.multidomainforward(h, adj, adjoverlap(adj, adj, edge, []), adjoverlap(adj, adj2, edge, []), edge, args,
                                           cndropprobs=[])
        loss = -F.logsigmoid(pos).mean() - F.logsigmoid(-neg).mean()
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        if step >= a.warmup:
            times.append(time.perf_counter() - t0)
            losses.append(float(loss))
            ms = torch.cuda.memory_stats()
            mallocs.append((ms.get("num_device_alloc", 0), ms.get("num_device_free", 0), ms.get("num_alloc_retries", 0),
                            round(ms.get("reserved_bytes.all.current", 0) / 2**30, 2)))
    print(json.dumps({"workload": f"{a.config}-shaped synthetic graph, training step of the reference driver (maskinput, A^2 per batch, "
                                  f"pos + neg pass, backward, Adam), batch {B}", "n": n, "train_edges": int(pos_train_edge.shape[1]),
                      "steps": a.steps, "ms_per_step": 1e3 * sum(times) / len(times), "ms_min": 1e3 * min(times),
                      "ms_median": 1e3 * sorted(times)[len(times) // 2], "ms_steps": [round(1e3 * t, 2) for t in times], "allocator": mallocs,
                      "edges_per_s": 2 * B * len(times) / sum(times), "loss_first": losses[0], "loss_last": losses[-1],
                      "deterministic_backward": bool(ops.deterministic_backward)}))


if __name__ == "__main__":
    main()
