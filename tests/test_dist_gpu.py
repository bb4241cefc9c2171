"""GPU rehearsal of the edge-sharded path: two ranks share the one GPU of the test box and talk over
gloo (the production backend is RCCL, which the driver exercises on the 8-GPU node).  The sharded
scores must equal the single-process scores of the same global batch bit for bit: the only thing
exchanged before pooling is an integer histogram."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _setup(seed=0):
    from ocn_amd.model import predictor_dict
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import dataset_like, sample_edges
    dev = torch.device("cuda:0")
    ei, n, _ = dataset_like("collab", seed=seed, scale=0.03)
    adj = SparseTensor.from_edge_index(ei.to(dev), sparse_sizes=(n, n)).to_symmetric()
    sp = adj.to_torch_sparse_coo_tensor()
    adj2 = SparseTensor.from_torch_sparse_coo_tensor(sp @ sp, False)
    r, c, _ = adj.coo()
    edges = sample_edges(r.cpu(), c.cpu(), n, 9001, seed=5).to(dev)          # does not divide by 2; each half is large enough for the class-major heads
    torch.manual_seed(3)
    h = torch.randn(n, 64, device=dev)
    preds = {k: predictor_dict[k](64, 64, 1, 3, 0.0, 0.0, True).to(dev).eval() for k in ("cn5", "cn7")}
    with torch.no_grad():
        preds["cn5"].innerprod.fill_(12.5)
    return dev, adj, adj2, edges, h, preds


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from ocn_amd.dist import sharded_predict
        dev, adj, adj2, edges, h, preds = _setup()
        args = SimpleNamespace(sum=2.74)
        with torch.no_grad():
            out = {k: sharded_predict(p, h, adj, adj2, edges, args).cpu().numpy() for k, p in preds.items()}
            # the scoring loop's form: two batches in flight (begin / finish), one gather per batch here
            from ocn_amd.dist import gather_scores, shard_bounds
            from ocn_amd.utils import adjoverlap
            batches = [edges, edges.flip(1).contiguous()]
            s, e = shard_bounds(edges.shape[1], world)[rank]
            for k, p in preds.items():
                p.set_edge_sharding(None, enabled=True)
                try:
                    mine = [b[:, s:e].contiguous() for b in batches]
                    tok = p.begin(h, adj, adjoverlap(adj, adj, mine[0]), adjoverlap(adj, adj2, mine[0]), mine[0], slot=0)
                    nxt = p.begin(h, adj, adjoverlap(adj, adj, mine[1]), adjoverlap(adj, adj2, mine[1]), mine[1], slot=1)
                    loc0 = p.finish(h, tok, args)
                    loc1 = p.finish(h, nxt, args)
                finally:
                    p.set_edge_sharding(None, enabled=False)
                out[k + "_pipe0"] = gather_scores(loc0, edges.shape[1]).cpu().numpy()
                out[k + "_pipe1"] = gather_scores(loc1, edges.shape[1]).cpu().numpy()
            # whole-batch dealing (VERDICT r3 #7a): the split's PermIterator batches dealt round robin, no histogram exchange
            from ocn_amd.pipeline import score_edges, score_mrr_split
            split = edges.t().contiguous()
            for k, p in preds.items():
                out[k + "_dealt"] = score_edges(p, h, adj, adj2, split, 1300, args, group=True).cpu().numpy()
            src, dst = edges[0, :700].contiguous(), edges[1, :700].contiguous()
            neg = torch.randint(0, adj.size(0), (700, 3), generator=torch.Generator().manual_seed(2)).to(dev)
            pos, negp = score_mrr_split(preds["cn7"], h, adj, src, dst, neg, 256, args, group=True)
            out["mrr_dealt"] = torch.cat([pos, negp.reshape(-1)]).cpu().numpy()
        q.put((rank, out))                       # numpy: pickled by value (a tensor would travel as a shared-memory handle of a process that exits)
    finally:
        dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_single_process(hiplib):
    from types import SimpleNamespace
    from ocn_amd.utils import adjoverlap
    dev, adj, adj2, edges, h, preds = _setup()
    args = SimpleNamespace(sum=2.74)
    with torch.no_grad():
        single = {k: p(h, adj, adjoverlap(adj, adj, edges), adjoverlap(adj, adj2, edges), edges, args).cpu()
                  for k, p in preds.items()}
        rev = edges.flip(1).contiguous()
        for k, p in preds.items():
            single[k + "_pipe0"] = single[k]
            single[k + "_pipe1"] = p(h, adj, adjoverlap(adj, adj, rev), adjoverlap(adj, adj2, rev), rev, args).cpu()
        from ocn_amd.pipeline import score_edges, score_mrr_split
        for k, p in preds.items():
            single[k + "_dealt"] = score_edges(p, h, adj, adj2, edges.t().contiguous(), 1300, args).cpu()
        src, dst = edges[0, :700].contiguous(), edges[1, :700].contiguous()
        neg = torch.randint(0, adj.size(0), (700, 3), generator=torch.Generator().manual_seed(2)).to(dev)
        pos, negp = score_mrr_split(preds["cn7"], h, adj, src, dst, neg, 256, args)
        single["mrr_dealt"] = torch.cat([pos, negp.reshape(-1)]).cpu()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out in res:
        for k in single:
            assert out[k].shape == tuple(single[k].shape)
            assert torch.equal(torch.from_numpy(out[k]), single[k]), f"rank {rank} {k}: sharded scores differ from the single-device batch"


def _rccl_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        from types import SimpleNamespace
        import ocn_amd.dist as D
        from ocn_amd.utils import adjoverlap
        dev, adj, adj2, edges, h, preds = _setup()
        args = SimpleNamespace(sum=2.74)
        D.force_collectives = True
        ok = {}
        with torch.no_grad():
            for k, p in preds.items():
                single = p(h, adj, adjoverlap(adj, adj, edges), adjoverlap(adj, adj2, edges), edges, args)
                p.set_edge_sharding(None, enabled=True)          # force the collectives: all-reduce, ring, all-gather on RCCL
                try:
                    from ocn_amd.dist import gather_scores
                    loc = p(h, adj, adjoverlap(adj, adj, edges), adjoverlap(adj, adj2, edges), edges, args)
                    out = gather_scores(loc, edges.shape[1])
                    # the scoring loop's form: the all-gather of batch t in flight beside batch t + 1
                    pend, outs = None, []
                    for _ in range(3):
                        loc = p(h, adj, adjoverlap(adj, adj, edges), adjoverlap(adj, adj2, edges), edges, args)
                        if pend is not None:
                            pend.wait()
                        o, pend = gather_scores(loc, edges.shape[1], async_op=True)
                        outs.append(o)
                    if pend is not None:
                        pend.wait()
                    torch.cuda.synchronize()
                finally:
                    p.set_edge_sharding(None, enabled=False)
                ok[k] = bool(torch.equal(out, single))
                ok[k + "_async"] = all(bool(torch.equal(o, single)) for o in outs)
        t = torch.ones(1 << 20, device=dev)
        dist.all_reduce(t)
        ok["allreduce"] = bool((t == 1).all())
        q.put(ok)
    finally:
        dist.destroy_process_group()


def test_rccl_backend_single_rank(hiplib):
    """The production backend ("nccl" = RCCL) with one rank on the test box's GPU: the library initialises, the
    histogram all-reduce / S2 ring / score all-gather code paths of ocn_amd.dist run on device tensors (no host
    staging) and leave the scores bit-identical.  (More ranks need more GPUs: the driver's scaling run.)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    ok = q.get(timeout=300)
    p.join(60)
    assert p.exitcode == 0
    assert all(ok.values()), ok
