"""The reference's training / evaluation driver (NeighborOverlap_large.py: train() :28-94, test()
:97-180, the epoch loop of main() :300-345) written against ocn_amd, on a synthetic dataset of the
named shape (no network: ogb / Planetoid downloads are replaced by ocn_amd.synth.loaddataset_like,
ogb's Evaluator by ocn_amd.evaluate.Evaluator).  Everything between the import block and the
argument parser is the reference's call sequence with the three import lines swapped
(INTEGRATION.md §2).

    python examples/run_like_reference.py --dataset cora --predictor cn5 --epochs 5
    python examples/run_like_reference.py --dataset collab --scale 0.05 --hiddim 64 --batch_size 8192
"""
import argparse
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocn_amd.evaluate import Evaluator                                              # noqa: E402
from ocn_amd.model import GCN, predictor_dict                                       # noqa: E402
from ocn_amd.sparse import SparseTensor                                             # noqa: E402
from ocn_amd.synth import loaddataset_like                                          # noqa: E402
from ocn_amd.utils import PermIterator, adjoverlap, sparse_tensor_multiply          # noqa: E402


def build_adj2(adj, args):
    if args.adj2byblock:
        return sparse_tensor_multiply(SparseTensor.from_torch_sparse_coo_tensor(adj.to_torch_sparse_coo_tensor()), 1024)
    spadj = adj.to_torch_sparse_coo_tensor()
    return SparseTensor.from_torch_sparse_coo_tensor(spadj @ spadj, False)


def train(model, predictor, data, split_edge, optimizer, batch_size, maskinput, args):
    model.train(); predictor.train()
    pos_train_edge = split_edge['train']['edge'].to(data.x.device).t()
    total_loss = []
    adjmask = torch.ones_like(pos_train_edge[0], dtype=torch.bool)
    negedge = torch.randint(0, data.num_nodes, pos_train_edge.shape, device=pos_train_edge.device)
    for perm in PermIterator(adjmask.device, adjmask.shape[0], batch_size):
        optimizer.zero_grad()
        if maskinput:
            adjmask[perm] = 0
            tei = pos_train_edge[:, adjmask]
            adj = SparseTensor.from_edge_index(tei, sparse_sizes=(data.num_nodes, data.num_nodes)).to_device(
                pos_train_edge.device, non_blocking=True)
            adjmask[perm] = 1
            adj = adj.to_symmetric()
        else:
            adj = data.adj_t
        h = model(data.x, adj)
        adj2 = build_adj2(adj, args)
        edge = pos_train_edge[:, perm]
        pos_outs = predictor.multidomainforward(h, adj, adjoverlap(adj, adj, edge), adjoverlap(adj, adj2, edge), edge,
                                                args, cndropprobs=[])
        pos_losss = -F.logsigmoid(pos_outs).mean()
        edge = negedge[:, perm]
        neg_outs = predictor.multidomainforward(h, adj, adjoverlap(adj, adj, edge), adjoverlap(adj, adj2, edge), edge,
                                                args, cndropprobs=[])
        neg_losss = -F.logsigmoid(-neg_outs).mean()
        loss = neg_losss + pos_losss
        loss.backward()
        optimizer.step()
        total_loss.append(loss.detach())
    return float(torch.stack(total_loss).mean())


@torch.no_grad()
def test(model, predictor, data, split_edge, evaluator, batch_size, use_valedges_as_input, args):
    model.eval(); predictor.eval()
    dev = data.x.device
    edges = {k: split_edge[s][f].to(dev) for k, (s, f) in dict(
        pos_train=('train', 'edge'), pos_valid=('valid', 'edge'), neg_valid=('valid', 'edge_neg'),
        pos_test=('test', 'edge'), neg_test=('test', 'edge_neg')).items()}
    adj = data.adj_t
    h = model(data.x, adj)
    adj2 = build_adj2(adj, args)

    def score(e, h, adj):
        return torch.cat([predictor(h, adj, adjoverlap(adj, adj, e[perm].t()), adjoverlap(adj, adj2, e[perm].t()),
                                    e[perm].t(), args).squeeze(-1)
                          for perm in PermIterator(e.device, e.shape[0], batch_size, False)], dim=0)

    pred = {k: score(edges[k], h, adj) for k in ("pos_train", "pos_valid", "neg_valid")}
    if use_valedges_as_input:
        adj = data.full_adj_t
        h = model(data.x, adj)
    pred.update({k: score(edges[k], h, adj) for k in ("pos_test", "neg_test")})
    results = {}
    for K in [20, 50, 100]:
        evaluator.K = K
        hits = [evaluator.eval({'y_pred_pos': pred[p], 'y_pred_neg': pred[n]})[f'hits@{K}']
                for p, n in (("pos_train", "neg_valid"), ("pos_valid", "neg_valid"), ("pos_test", "neg_test"))]
        results[f'Hits@{K}'] = tuple(hits)
    return results, h


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="cora")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--predictor", default="cn5", choices=sorted(predictor_dict))
    ap.add_argument("--model", default="puregcn")
    ap.add_argument("--hiddim", type=int, default=64)
    ap.add_argument("--mplayers", type=int, default=1)
    ap.add_argument("--nnlayers", type=int, default=3)
    ap.add_argument("--batch_size", type=int, default=1152)
    ap.add_argument("--testbs", type=int, default=8192)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--maskinput", action="store_true")
    ap.add_argument("--adj2byblock", action="store_true")
    ap.add_argument("--use_valedges_as_input", action="store_true")
    ap.add_argument("--sum", type=float, default=0.0)
    ap.add_argument("--gnnlr", type=float, default=0.0043)
    ap.add_argument("--prelr", type=float, default=0.0024)
    ap.add_argument("--feat", type=int, default=0, help="feature width override for the synthetic x")
    args = ap.parse_args(argv)
    dev = torch.device("cuda:0")
    evaluator = Evaluator(name='ogbl-ppa' if args.dataset in ("cora", "citeseer", "pubmed") else f'ogbl-{args.dataset}')
    data, split_edge = loaddataset_like(args.dataset, args.use_valedges_as_input, scale=args.scale, feat=args.feat)
    data.x = data.x.to(dev)
    data.adj_t = data.adj_t.to_device(dev)
    data.full_adj_t = data.full_adj_t.to_device(dev) if args.use_valedges_as_input else data.adj_t
    torch.manual_seed(0)
    fin = args.hiddim if data.max_x >= 0 else data.x.shape[1]
    model = GCN(fin, args.hiddim, args.hiddim, args.mplayers, 0.05, True, False, data.max_x, args.model, True, 0.0,
                xdropout=0.3, taildropout=0.1).to(dev)
    predictor = predictor_dict[args.predictor](args.hiddim, args.hiddim, 1, args.nnlayers, 0.05, 0.0, True,
                                               use_xlin=True, tailact=True).to(dev)
    optimizer = torch.optim.Adam([{'params': model.parameters(), "lr": args.gnnlr},
                                  {'params': predictor.parameters(), 'lr': args.prelr}])
    out = []
    for epoch in range(1, 1 + args.epochs):
        t1 = time.time()
        loss = train(model, predictor, data, split_edge, optimizer, args.batch_size, args.maskinput, args)
        t2 = time.time()
        results, _ = test(model, predictor, data, split_edge, evaluator, args.testbs, args.use_valedges_as_input, args)
        torch.cuda.synchronize()
        t3 = time.time()
        line = (f"epoch {epoch:3d} loss {loss:.4f} train {t2 - t1:.2f}s test {t3 - t2:.2f}s  " +
                "  ".join(f"{k} train/valid/test {v[0]:.3f}/{v[1]:.3f}/{v[2]:.3f}" for k, v in results.items()))
        print(line, flush=True)
        out.append((loss, results))
    return out


if __name__ == "__main__":
    main()
