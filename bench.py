#!/usr/bin/env python
"""Headline benchmark: candidate edges / second through the CN predictor forward on a synthetic graph
of an OGB dataset's shape.  Default = BASELINE.json configs[1]: ogbl-collab shape, gin, hiddim 256,
cn5, batch 65536 (README.md:42 of the reference).

A step = one candidate batch through the hot path exactly as ``test()`` runs it
(NeighborOverlap_large.py:121-159): the CN builder (adjoverlap(A, A, e) + adjoverlap(A, A², e), or
get_cn1_cn2(A, e) on the pygho route), then the predictor forward (intersection -> column weights ->
pooling -> MLP heads), with the encoder output h and A² computed once per graph outside the timed
region and everything resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config citation2          # other BASELINE configs: citation2 | ppa | ddi | cora

N > 1: one process per GPU; graph / h / weights replicated, each rank owns a ``batch``-edge slice of
a global batch of N x batch (weak scaling); per step one RCCL all-reduce of the packed column
histograms and one all-gather of the scores (ocn_amd/dist.py).
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12       # B/s, MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
F32_MFMA_PEAK = 157.3e12    # FLOP/s, dense f32-in/f32-acc MFMA (same guide, chip-level table)
BF16_MFMA_PEAK = 2.5e15     # FLOP/s, dense bf16 MFMA

# the reference's README commands (README.md:27,42,47,92,98): encoder class, conv, layers, hiddim,
# predictor, CN route, batch, ids-as-features, args.sum
CONFIGS = {
    "collab":    dict(enc="GCN",  conv="gin",     layers=1, H=256, pred="cn5", route="adj2",  batch=65536, ids=False, sum=0.0, res=False),
    "cora":      dict(enc="GCN",  conv="puregcn", layers=1, H=256, pred="cn5", route="adj2",  batch=1152,  ids=False, sum=0.0, res=False),
    "ppa":       dict(enc="GCN2", conv="gcn",     layers=1, H=64,  pred="cn5", route="walk",  batch=2048,  ids=True,  sum=0.0, res=False),
    "citation2": dict(enc="GCN3", conv="gcn",     layers=5, H=32,  pred="cn7", route="walk",  batch=2048,  ids=False, sum=1.0, res=True),
    "ddi":       dict(enc="GCN",  conv="puregcn", layers=3, H=64,  pred="cn7", route="block", batch=32768, ids=True,  sum=2.74, res=True),
}


class StageTimer:
    """HIP events on torch's current stream — the stream every ocn_* kernel is launched on."""

    def __init__(self, pool=0):
        self.events = []
        self.active = True
        self.flops = {}
        # hipEventCreate is the expensive part on a busy host: create the events before the timed
        # region, only record() inside it
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(pool)]
        for ev in self.pool:                   # torch creates the hipEvent lazily, at the first record()
            ev.record()

    def mark(self, name, flops=0.0):
        if not self.active:
            return
        self.flops[name] = self.flops.get(name, 0.0) + flops
        ev = self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)
        ev.record()
        self.events.append((name, ev))

    def totals(self):
        tot, prev = {}, None
        for name, ev in self.events:
            if name != "begin" and prev is not None:
                tot.setdefault(name, []).append(prev.elapsed_time(ev))
            prev = ev
        return {k: (sum(v) / len(v), len(v)) for k, v in tot.items()}


def build_workload(args, dev, rank, world):
    import ocn_amd.model as M
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import dataset_like, sample_edges
    from ocn_amd.utils import sparse_tensor_multiply

    cfg = dict(CONFIGS[args.dataset])
    cfg["H"] = args.hiddim or cfg["H"]
    cfg["pred"] = args.predictor or cfg["pred"]
    cfg["batch"] = args.batch or cfg["batch"]
    H = cfg["H"]
    t0 = time.time()
    ei, n, shape = dataset_like(args.dataset, seed=0, scale=args.scale)
    adj = SparseTensor.from_edge_index(ei.to(dev), sparse_sizes=(n, n), trust_data=True).to_symmetric()
    del ei
    torch.cuda.synchronize()
    t_graph = time.time() - t0
    torch.manual_seed(0)
    if cfg["ids"]:
        x, fin, max_x = torch.arange(n, device=dev), H, n
    else:
        fin, max_x = (shape["feat"] or H), -1
        x = torch.randn(n, fin, device=dev)
    enc = getattr(M, cfg["enc"])(fin, H, H, cfg["layers"], 0.05, True, cfg["res"], max_x, cfg["conv"], True, 0.0,
                                 xdropout=0.7, taildropout=0.3).to(dev).eval()
    pred = M.predictor_dict[cfg["pred"]](H, H, 1, 3, 0.05, 0.4, True, use_xlin=True, tailact=True).to(dev).eval()
    with torch.no_grad():
        h = enc(x, adj)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            h = enc(x, adj)
        torch.cuda.synchronize()
        t_enc = (time.time() - t0) / 3
    adj2, t_a2 = None, 0.0
    if cfg["route"] != "walk":
        t0 = time.time()
        if cfg["route"] == "block":
            adj2 = sparse_tensor_multiply(adj, 1024)
        else:
            sp = adj.to_torch_sparse_coo_tensor()
            adj2 = SparseTensor.from_torch_sparse_coo_tensor(sp @ sp, False)
        torch.cuda.synchronize()
        t_a2 = time.time() - t0
    r, c, _ = adj.coo()
    # one global batch of world x B edges, seeded; rank r owns slice r
    edges = sample_edges(r.cpu(), c.cpu(), n, cfg["batch"] * world, seed=1).to(dev)
    return dict(cfg=cfg, n=n, adj=adj, adj2=adj2, h=h.contiguous(), pred=pred, edges=edges, enc_s=t_enc, a2_s=t_a2,
                graph_s=t_graph, nnz=adj.nnz(), nnz2=adj2.nnz() if adj2 is not None else None,
                max_deg=adj.max_rowcount(), args=SimpleNamespace(sum=cfg["sum"]))


def cn_handles(wl, e):
    from ocn_amd.utils import adjoverlap, get_cn1_cn2
    if wl["cfg"]["route"] == "walk":
        return get_cn1_cn2(wl["adj"], e)
    return adjoverlap(wl["adj"], wl["adj"], e), adjoverlap(wl["adj"], wl["adj2"], e)


def algorithmic_bytes(wl, mine, cnt1, cnt2, H):
    """bytes(e) = 4(d_i+d_j) + 4 d2_j + 4H(c1+c2) + 8H + 28, split by the kernel that
    has to move them.  On the walk route there is no A² row; the builder reads the rows of the
    neighbours of i instead: 4(d_i + d_j + sum_{k in N(i)} d_k)."""
    adj = wl["adj"]
    rp = adj._rowptr
    deg = rp[1:] - rp[:-1]
    di = deg[mine[0]].sum().item()
    dj = deg[mine[1]].sum().item()
    B = mine.shape[1]
    if wl["adj2"] is not None:
        rp2 = wl["adj2"]._rowptr
        second = (rp2[mine[1] + 1] - rp2[mine[1]]).sum().item()
    else:
        r, c, _ = adj.coo()
        nbr_deg = torch.zeros(wl["n"], dtype=torch.int64, device=r.device).index_add_(0, r, deg[c])
        second = nbr_deg[mine[0]].sum().item()
    c12 = int(cnt1.sum().item()) + int(cnt2.sum().item())
    flags = 4 * (di + dj) + 4 * second + 24 * B       # CSR rows of i, j, A² row of j (or rows of N(i)), ids, counts
    gather = 4 * H * c12 + 8 * H * B + 4 * B          # one embedding row per CN entry, x_i, x_j, score
    return dict(cn_flags=flags, cn_gather=gather, total=flags + gather,
                mean_di=di / B, mean_dj=dj / B, mean_second=second / B, mean_c1=cnt1.float().mean().item(),
                mean_c2=cnt2.float().mean().item())


def cpu_baseline(wl, args):
    """The oracle (a port of the reference's op sequence, torch CPU ops, the host cores this process
    may use) on a bounded sample: the first ``b`` edges of the batch as a batch of their own."""
    from oracle import ocn_oracle as O
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    torch.set_num_threads(ncores)
    cfg, adj, adj2 = wl["cfg"], wl["adj"], wl["adj2"]
    r, c, _ = adj.coo()
    oadj = O.SpM(r.cpu(), c.cpu(), None, wl["n"], wl["n"])
    oadj2 = None
    if adj2 is not None:
        r2, c2, _ = adj2.coo()
        oadj2 = O.SpM(r2.cpu(), c2.cpu(), None, wl["n"], wl["n"])
    h = wl["h"].cpu()
    sd = {k: v.detach().cpu() for k, v in wl["pred"].state_dict().items()}

    def run(b):
        e = wl["edges"][:, :b].cpu()
        t0 = time.time()
        if oadj2 is None:
            cn1, cn2 = O.get_cn1_cn2(oadj, e)
        else:
            cn1, cn2 = O.adjoverlap(oadj, oadj, e), O.adjoverlap(oadj, oadj2, e)
        if cfg["pred"] == "cn5":
            out = O.cn5_forward(sd, h, cn1, cn2, e, ln=True, tailact=True)
        else:
            out = O.cn7_forward(sd, h, cn1, cn2, e, cfg["sum"], ln=True, tailact=True)
        return time.time() - t0, out

    b = min(args.cpu_sample, wl["edges"].shape[1])
    t, out = run(b)
    while t < 6.0 and b * 2 <= min(wl["edges"].shape[1], 16384):
        b *= 2
        t, out = run(b)
    builder = "get_cn1_cn2" if oadj2 is None else "adjoverlap x2"
    return dict(value=b / t, unit="edges/s", cores=ncores, kind="port",
                sample=f"first {b} edges of the step's batch as one batch, oracle/ocn_oracle.py {builder} + "
                       f"{cfg['pred']}_forward, {t:.2f} s, torch {torch.get_num_threads()} threads"), b, out


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of a kernel from the committed PMC passes (profiles/r*_pmc.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of the default command).  FETCH_SIZE is
    doubled (gfx950 counts 128-B requests as 64 B — MI355X_MICROARCH.md §HBM; calibrated on
    rows_ln_relu / combine3 whose byte counts are known exactly), both are in KiB."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None
    for k, v in json.load(open(files[-1])).items():
        if kernel_substr in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            return (2.0 * v["FETCH_SIZE"]["avg"] + v["WRITE_SIZE"]["avg"]) * 1024.0
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", "--dataset", dest="dataset", default="collab", choices=sorted(CONFIGS))
    ap.add_argument("--predictor", default=None)
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--hiddim", type=int, default=None)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timers", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a captured HIP graph (pipeline.GraphedScorer; one GPU, no stage events: "
                         "the roofline objects are then null); small-batch configurations only (cora, ppa, citation2)")
    ap.add_argument("--prewarm", type=int, default=64, help="untimed runtime pre-warm steps before --warmup")
    ap.add_argument("--run-ahead", type=int, default=6, help="steps the host may enqueue ahead of the GPU")
    ap.add_argument("--timer-every", type=int, default=8, help="record stage events on every n-th timed step")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from ocn_amd import _lib, ops
    from ocn_amd.dist import gather_scores, shard_bounds
    _lib.lib()                                     # no HIP extension -> no benchmark

    wl = build_workload(args, dev, rank, world)
    cfg, pred, h, adj = wl["cfg"], wl["pred"], wl["h"], wl["adj"]
    H = cfg["H"]
    is_default = args.dataset == "collab" and args.scale == 1.0 and not (args.predictor or args.batch or args.hiddim)
    pmc = pmc_traffic if is_default else (lambda k: None)
    B_total = wl["edges"].shape[1]
    s, e = shard_bounds(B_total, world)[rank]
    mine = wl["edges"][:, s:e].contiguous()
    pred.set_edge_sharding(None, enabled=world > 1)

    def step():
        with torch.no_grad():
            c1, c2 = cn_handles(wl, mine)
            loc = pred(h, adj, c1, c2, mine, wl["args"])
            return gather_scores(loc, B_total)

    out = step()                                   # validated once (bounds check + flag capacity)
    torch.cuda.synchronize()
    ops.validate_indices = False                   # same ids every step: no per-step host sync
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is a single-GPU option")
        from ocn_amd.pipeline import GraphedScorer
        scorer = GraphedScorer(pred, h, adj, wl.get("adj2"), mine.shape[1], wl["args"],
                               route="walk" if cfg["route"] == "walk" else "pattern")
        eager_step = step

        def step():
            return scorer(mine)
        assert torch.equal(step(), eager_step())
        args.no_stage_timers = True
    # Runtime pre-warm (untimed, before the W warm-up steps): the HIP runtime grows its internal
    # command/signal pools in ~35 ms host stalls during the first ~1000 launches of a process; a short
    # --warmup would otherwise put one of them inside the timed region.
    for i in range(args.prewarm):
        step()
        if i % 4 == 3:
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    timer = None if args.no_stage_timers else StageTimer(pool=32 * (args.steps // args.timer_every + 1))
    ops.stage_timer = timer
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # Flow control: the host enqueues a step several times faster than the GPU runs it.  Left alone, the
    # HIP runtime lets ~1000 commands pile up and then blocks the host until the queue has drained
    # completely (measured: 35 ms stalls, GPU idle at the end of each) — so the host waits on the event of
    # the step `run_ahead` steps back, as a real scoring loop that consumes its scores would.
    run_ahead = max(args.run_ahead, 1)
    ring = [torch.cuda.Event() for _ in range(run_ahead)]
    for ev in ring:
        ev.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_wait = 0.0
    for it in range(args.steps):
        tw = time.perf_counter()
        ring[it % run_ahead].synchronize()
        t_wait += time.perf_counter() - tw
        if timer:                                  # stage events on every `timer_every`-th step only: on a busy
            timer.active = it % args.timer_every == 0   # host each hipEventRecord costs tens of microseconds
            timer.mark("begin")
        out = step()
        if timer:
            timer.mark("mlp_glue")
        ring[it % run_ahead].record()
        if os.environ.get("OCN_BENCH_DEBUG"):
            print("step", it, round((time.perf_counter() - t0) * 1e3, 3), file=sys.stderr)
    t_launch = time.perf_counter() - t0 - t_wait    # host time spent enqueueing the K steps (flow-control waits excluded)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.stage_timer = None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    if rank == 0:
        from ocn_amd.utils import CNState
        st = (CNState(adj, None, None, mine, walk=True) if cfg["route"] == "walk"
              else CNState(adj, adj, wl["adj2"], mine))
        ab = algorithmic_bytes(wl, mine, st.cnt1, st.cnt2, H)
        stages = {k: dict(ms=v[0], launches=v[1]) for k, v in (timer.totals() if timer else {}).items()}
        roof, roof_hbm = None, None
        if stages:
            for k in ("cn_flags", "cn_gather"):
                if k in stages:
                    stages[k]["algorithmic_GBps"] = ab[k] / (stages[k]["ms"] * 1e-3) / 1e9
            # dominant kernel = largest total time per step
            sampled = len([1 for it in range(args.steps) if it % args.timer_every == 0])
            per_step = {k: v["ms"] * v["launches"] / sampled for k, v in stages.items()}
            dom = max((k for k in ("cn_flags", "cn_gather", "linear") if k in stages), key=lambda k: per_step[k])
            g = "cn_gather"
            roof_hbm = dict(bound="hbm", kernel="cn_gather_kernel", achieved=ab[g] / (stages[g]["ms"] * 1e-3) / 1e9,
                            peak=HBM_PEAK / 1e9, unit="GB/s", frac=ab[g] / (stages[g]["ms"] * 1e-3) / HBM_PEAK,
                            traffic=pmc("cn_gather_kernel"), algorithmic_bytes_per_launch=ab[g],
                            avg_launch_ms=stages[g]["ms"],
                            note="algorithmic bytes price one embedding row per CN entry; rows shared by "
                                 "candidates processed together are served by L2, so frac can exceed 1")
            if dom == "linear":
                # f32 FLOPs of the Linear layers per launch.  With the zero-row skipping of the heads the
                # launches cover only part of the batch (device-side row ranges): count the rows they do.
                fl = timer.flops.get("linear", 0.0) / stages["linear"]["launches"]
                skip = getattr(pred, "_skip_state", None)
                skipping = bool(ops.skip_zero_rows and skip is not None and skip["off"] == 0
                                and mine.shape[1] >= ops.skip_zero_min_batch and pred._heads_plan(H) is not None)
                if skipping:
                    Bm = mine.shape[1]
                    n_cn1 = int((st.cnt1 > 0).sum())
                    n_any = int(((st.cnt1 > 0) | (st.cnt2 > 0)).sum())
                    n_b = n_any if pred._xcn2_on_union else int((st.cnt2 > 0).sum())
                    sx_layers = len(pred._heads_plan(H)[2])
                    lin_layers = sum(1 for m in pred.lin if isinstance(m, torch.nn.Linear) and m.out_features == H)
                    executed = 2.0 * H * H * (2 * n_cn1 + 2 * n_b + sx_layers * Bm + lin_layers * Bm) + 2.0 * (2 * H) * H * n_any
                    fl = executed / (stages["linear"]["launches"] / sampled)
                t = stages["linear"]["ms"] * 1e-3
                roof = dict(bound="mfma", kernel="linear_bf16x6_kernel", achieved=fl / t / 1e12,
                            peak=F32_MFMA_PEAK / 1e12, unit="TFLOP/s", frac=fl / t / F32_MFMA_PEAK,
                            traffic=pmc("linear_bf16x6_kernel"),
                            algorithmic_flops_per_launch=fl, avg_launch_ms=stages["linear"]["ms"],
                            launches_per_step=stages["linear"]["launches"] / sampled,
                            executed_bf16_tflops=6 * fl / t / 1e12, executed_frac_of_bf16_peak=6 * fl / t / BF16_MFMA_PEAK,
                            skipped_zero_rows=skipping,
                            note="per-launch averages over the grouped launches of a step (3 + 2 + 2 + 1 Linear(H,H) "
                                 "equivalents, less the rows whose pooled input is all zero when skipped_zero_rows); "
                                 "f32 Linear evaluated as six bf16 MFMA cross terms: achieved/peak are the "
                                 "algorithmic f32 FLOPs against the dense f32 MFMA peak; executed_* count the "
                                 "bf16 MFMAs actually issued against the dense bf16 peak")
            else:
                kname = "cn_walk_kernel" if cfg["route"] == "walk" else "cn_flags_kernel"
                roof = dict(roof_hbm) if dom == g else dict(
                    bound="hbm", kernel=kname, achieved=ab[dom] / (stages[dom]["ms"] * 1e-3) / 1e9,
                    peak=HBM_PEAK / 1e9, unit="GB/s", frac=ab[dom] / (stages[dom]["ms"] * 1e-3) / HBM_PEAK,
                    traffic=pmc(kname), algorithmic_bytes_per_launch=ab[dom], avg_launch_ms=stages[dom]["ms"])
        cpu, err = None, None
        if world == 1 and not args.no_cpu_baseline:
            cpu, b, ref = cpu_baseline(wl, args)
            ops.validate_indices = False
            sub = mine[:, :b].contiguous()
            with torch.no_grad():
                c1, c2 = cn_handles(wl, sub)
                got = pred(h, adj, c1, c2, sub, wl["args"]).cpu()
            err = (got - ref).abs().max().item()
        idx = {"cora": 0, "collab": 1, "ppa": 2, "citation2": 3, "ddi": 4}[args.dataset]
        line = {
            "metric": f"candidate-edges/sec (CN predictor fwd), ogbl-{args.dataset} shape",
            "value": B_total * args.steps / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ogbl-{args.dataset}-shaped synthetic graph, {cfg['enc']} {cfg['conv']} x{cfg['layers']} "
                                   f"hiddim={H} predictor={cfg['pred']} CN route={cfg['route']} batch {cfg['batch']} per GPU "
                                   f"(BASELINE.json configs[{idx}])",
                       "nodes": wl["n"], "nnz": wl["nnz"], "nnz_A2": wl["nnz2"], "max_deg": wl["max_deg"],
                       "global_batch": B_total, "parallelism": f"edge-shard x{world}", "graph_scale": args.scale,
                       "mean_deg_src": ab["mean_di"], "mean_deg_dst": ab["mean_dj"],
                       "mean_second_operand_len": ab["mean_second"],
                       "mean_cn1": ab["mean_c1"], "mean_cn2": ab["mean_c2"]},
            "roofline": roof, "roofline_hbm_kernel": roof_hbm, "cpu_baseline": cpu,
            "stages": stages,
            "once_per_graph": {"encoder_ms": wl["enc_s"] * 1e3, "adj2_build_ms": wl["a2_s"] * 1e3,
                               "graph_build_s": wl["graph_s"]},
            "algorithmic_bytes_per_step": ab["total"],
            "host_enqueue_ms_per_step": t_launch / args.steps * 1e3,
            "parity_on_cpu_sample_max_abs_err": err,
            "score_checksum": float(out.double().sum().item()),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
