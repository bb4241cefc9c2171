#!/usr/bin/env python
"""Headline benchmark: candidate edges / second through the CN predictor forward (cn5) on an
ogbl-collab-shaped synthetic graph (BASELINE.json configs[1]: gin, hiddim 256, batch 65536).

A step = one candidate batch through the hot path exactly as ``test()`` runs it
(NeighborOverlap_large.py:121-159): adjoverlap(A, A, e), adjoverlap(A, A², e), predictor forward
(intersection -> column weights -> pooling -> MLP heads) with the encoder output h and A² computed
once per graph outside the timed region and everything resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; graph / h / weights replicated, each rank owns a 65536-edge slice of a
global batch of N x 65536 (weak scaling); per step one RCCL all-reduce of the int32 column
histograms and one all-gather of the scores (ocn_amd/dist.py).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12       # B/s, MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured copy)
F32_MFMA_PEAK = 157.3e12    # FLOP/s, dense f32-in/f32-acc MFMA (same guide, chip-level table)
BF16_MFMA_PEAK = 2.5e15     # FLOP/s, dense bf16 MFMA


class StageTimer:
    """HIP events on torch's current stream — the stream every ocn_* kernel is launched on."""

    def __init__(self):
        self.events = []

    def mark(self, name):
        ev = torch.cuda.Event(enable_timing=True)
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
    from ocn_amd.model import GCN, predictor_dict
    from ocn_amd.sparse import SparseTensor
    from ocn_amd.synth import dataset_like, sample_edges

    t0 = time.time()
    ei, n, shape = dataset_like(args.dataset, seed=0, scale=args.scale)
    adj = SparseTensor.from_edge_index(ei.to(dev), sparse_sizes=(n, n), trust_data=True).to_symmetric()
    torch.cuda.synchronize()
    t_graph = time.time() - t0
    H = args.hiddim
    torch.manual_seed(0)
    x = torch.randn(n, shape["feat"] or H, device=dev)
    # collab README command: --model gin --mplayers 1 --hiddim 256 --ln --jk (README.md:42)
    enc = GCN(x.shape[1], H, H, 1, 0.05, True, False, -1, "gin", True, 0.0, xdropout=0.7, taildropout=0.3).to(dev).eval()
    pred = predictor_dict[args.predictor](H, H, 1, 3, 0.05, 0.4, True, use_xlin=True, tailact=True).to(dev).eval()
    with torch.no_grad():
        h = enc(x, adj)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(3):
            h = enc(x, adj)
        torch.cuda.synchronize()
        t_enc = (time.time() - t0) / 3
    t0 = time.time()
    sp = adj.to_torch_sparse_coo_tensor()
    adj2 = SparseTensor.from_torch_sparse_coo_tensor(sp @ sp, False)
    torch.cuda.synchronize()
    t_a2 = time.time() - t0
    r, c, _ = adj.coo()
    # one global batch of world x B edges, seeded; rank r owns slice r
    edges = sample_edges(r.cpu(), c.cpu(), n, args.batch * world, seed=1).to(dev)
    return dict(n=n, adj=adj, adj2=adj2, h=h.contiguous(), pred=pred, edges=edges, enc_s=t_enc, a2_s=t_a2,
                graph_s=t_graph, nnz=adj.nnz(), nnz2=adj2.nnz(), max_deg=adj.max_rowcount())


def algorithmic_bytes(wl, mine, cnt1, cnt2, H):
    """bytes(e) = 4(d_i+d_j) + 4 d2_j + 4H(c1+c2) + 8H + 28, split by the kernel
    that has to move them."""
    rp, rp2 = wl["adj"]._rowptr, wl["adj2"]._rowptr
    di = (rp[mine[0] + 1] - rp[mine[0]]).sum().item()
    dj = (rp[mine[1] + 1] - rp[mine[1]]).sum().item()
    d2j = (rp2[mine[1] + 1] - rp2[mine[1]]).sum().item()
    c12 = int(cnt1.sum().item()) + int(cnt2.sum().item())
    B = mine.shape[1]
    flags = 4 * (di + dj) + 4 * d2j + 24 * B          # CSR rows of i, j, A² row of j, edge ids, two counts
    gather = 4 * H * c12 + 8 * H * B + 4 * B          # one embedding row per CN entry, x_i, x_j, score
    return dict(cn_flags=flags, cn_gather=gather, total=flags + gather,
                mean_di=di / B, mean_dj=dj / B, mean_d2j=d2j / B, mean_c1=cnt1.float().mean().item(),
                mean_c2=cnt2.float().mean().item())


def cpu_baseline(wl, args):
    """The oracle (a port of the reference's op sequence, torch CPU ops, all host cores) on a bounded
    sample: the first ``b`` edges of the batch as a batch of their own."""
    from oracle import ocn_oracle as O
    ncores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    torch.set_num_threads(ncores)
    adj, adj2 = wl["adj"], wl["adj2"]
    r, c, _ = adj.coo()
    r2, c2, _ = adj2.coo()
    oadj = O.SpM(r.cpu(), c.cpu(), None, wl["n"], wl["n"])
    oadj2 = O.SpM(r2.cpu(), c2.cpu(), None, wl["n"], wl["n"])
    h = wl["h"].cpu()
    sd = {k: v.detach().cpu() for k, v in wl["pred"].state_dict().items()}

    def run(b):
        e = wl["edges"][:, :b].cpu()
        t0 = time.time()
        cn1 = O.adjoverlap(oadj, oadj, e)
        cn2 = O.adjoverlap(oadj, oadj2, e)
        out = O.cn5_forward(sd, h, cn1, cn2, e, ln=True, tailact=True)
        return time.time() - t0, out

    b = args.cpu_sample
    t, out = run(b)
    while t < 6.0 and b * 2 <= wl["edges"].shape[1] and b < 16384:
        b *= 2
        t, out = run(b)
    return dict(value=b / t, unit="edges/s", cores=ncores, kind="port",
                sample=f"first {b} edges of the step's batch as one batch, oracle/ocn_oracle.py "
                       f"adjoverlap x2 + cn5_forward, {t:.2f} s, torch {torch.get_num_threads()} threads"), b, out


def pmc_traffic(kernel_substr):
    """HBM-side bytes per launch of a kernel from the committed PMC passes (profiles/r*_pmc.json:
    separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  FETCH_SIZE is
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
    ap.add_argument("--dataset", default="collab")
    ap.add_argument("--predictor", default="cn5")
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--hiddim", type=int, default=256)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--cpu-sample", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stage-timers", action="store_true")
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
    from ocn_amd.utils import adjoverlap
    _lib.lib()                                     # no HIP extension -> no benchmark

    wl = build_workload(args, dev, rank, world)
    pred, h, adj, adj2 = wl["pred"], wl["h"], wl["adj"], wl["adj2"]
    B_total = wl["edges"].shape[1]
    s, e = shard_bounds(B_total, world)[rank]
    mine = wl["edges"][:, s:e].contiguous()
    pred.set_edge_sharding(None, enabled=world > 1)

    def step():
        with torch.no_grad():
            loc = pred(h, adj, adjoverlap(adj, adj, mine), adjoverlap(adj, adj2, mine), mine)
            return gather_scores(loc, B_total)

    out = step()                                   # validated once (bounds check + flag capacity)
    torch.cuda.synchronize()
    ops.validate_indices = False                   # same ids every step: no per-step host sync
    for _ in range(args.warmup):
        step()
    timer = None if args.no_stage_timers else StageTimer()
    ops.stage_timer = timer
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if timer:
            timer.mark("begin")
        out = step()
        if timer:
            timer.mark("mlp_glue")
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
        st = CNState(adj, adj, adj2, mine)
        ab = algorithmic_bytes(wl, mine, st.cnt1, st.cnt2, args.hiddim)
        stages = {k: dict(ms=v[0], launches=v[1]) for k, v in (timer.totals() if timer else {}).items()}
        roof, roof_hbm = None, None
        if stages:
            H = args.hiddim
            for k in ("cn_flags", "cn_gather"):
                if k in stages:
                    stages[k]["algorithmic_GBps"] = ab[k] / (stages[k]["ms"] * 1e-3) / 1e9
            # dominant kernel = largest total time per step
            per_step = {k: v["ms"] * v["launches"] / args.steps for k, v in stages.items()}
            dom = max((k for k in ("cn_flags", "cn_gather", "linear") if k in stages), key=lambda k: per_step[k])
            g = "cn_gather"
            roof_hbm = dict(bound="hbm", kernel="cn_gather_kernel", achieved=ab[g] / (stages[g]["ms"] * 1e-3) / 1e9,
                            peak=HBM_PEAK / 1e9, unit="GB/s", frac=ab[g] / (stages[g]["ms"] * 1e-3) / HBM_PEAK,
                            traffic=pmc_traffic("cn_gather_kernel"), algorithmic_bytes_per_launch=ab[g],
                            avg_launch_ms=stages[g]["ms"],
                            note="algorithmic bytes price one embedding row per CN entry; rows shared by "
                                 "candidates processed together are served by L2, so frac can exceed 1")
            if dom == "linear":
                fl = 2.0 * mine.shape[1] * H * H          # one Linear(H,H) over the batch
                t = stages["linear"]["ms"] * 1e-3
                roof = dict(bound="mfma", kernel="linear_bf16x6_kernel", achieved=fl / t / 1e12,
                            peak=F32_MFMA_PEAK / 1e12, unit="TFLOP/s", frac=fl / t / F32_MFMA_PEAK,
                            traffic=pmc_traffic("linear_bf16x6_kernel"),
                            algorithmic_flops_per_launch=fl, avg_launch_ms=stages["linear"]["ms"],
                            launches_per_step=stages["linear"]["launches"] / args.steps,
                            executed_bf16_tflops=6 * fl / t / 1e12, executed_frac_of_bf16_peak=6 * fl / t / BF16_MFMA_PEAK,
                            note="f32 Linear evaluated as six bf16 MFMA cross terms: achieved/peak are the "
                                 "algorithmic f32 FLOPs against the dense f32 MFMA peak; executed_* count the "
                                 "bf16 MFMAs actually issued against the dense bf16 peak")
            else:
                roof = dict(roof_hbm) if dom == g else dict(
                    bound="hbm", kernel="cn_flags_kernel", achieved=ab[dom] / (stages[dom]["ms"] * 1e-3) / 1e9,
                    peak=HBM_PEAK / 1e9, unit="GB/s", frac=ab[dom] / (stages[dom]["ms"] * 1e-3) / HBM_PEAK,
                    traffic=None, algorithmic_bytes_per_launch=ab[dom], avg_launch_ms=stages[dom]["ms"])
        cpu, err = None, None
        if world == 1 and not args.no_cpu_baseline:
            cpu, b, ref = cpu_baseline(wl, args)
            ops.validate_indices = False
            sub = mine[:, :b].contiguous()
            with torch.no_grad():
                got = pred(h, adj, adjoverlap(adj, adj, sub), adjoverlap(adj, adj2, sub), sub).cpu()
            err = (got - ref).abs().max().item()
        line = {
            "metric": "candidate-edges/sec (CN predictor fwd), ogbl-collab shape",
            "value": B_total * args.steps / dt, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"ogbl-{args.dataset}-shaped synthetic graph, gin hiddim={args.hiddim} "
                                   f"predictor={args.predictor} batch {args.batch} per GPU "
                                   "(BASELINE.json configs[1])",
                       "nodes": wl["n"], "nnz": wl["nnz"], "nnz_A2": wl["nnz2"], "max_deg": wl["max_deg"],
                       "global_batch": B_total, "parallelism": f"edge-shard x{world}",
                       "mean_deg_src": ab["mean_di"], "mean_deg_dst": ab["mean_dj"], "mean_deg2_dst": ab["mean_d2j"],
                       "mean_cn1": ab["mean_c1"], "mean_cn2": ab["mean_c2"]},
            "roofline": roof, "roofline_hbm_kernel": roof_hbm, "cpu_baseline": cpu,
            "stages": stages,
            "once_per_graph": {"encoder_ms": wl["enc_s"] * 1e3, "adj2_build_ms": wl["a2_s"] * 1e3},
            "algorithmic_bytes_per_step": ab["total"],
            "parity_on_cpu_sample_max_abs_err": err,
            "score_checksum": float(out.double().sum().item()),
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
