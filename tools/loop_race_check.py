"""Stress of the two-stream scoring loop against the one-stream loop: the tool that found the H = 64 pooling kernel's packed-math
fault (one batch in a hundred with 64 bytes of one xcn1 row wrong, only beside another stream's heads).
    DBG_H=64 DBG_MODE=real DBG_REPS=250 python tools/loop_race_check.py [-DOCN_X_WAVE_CHECK]
n = 20 000, twelve ragged batches of DBG_B (8 192) candidates, trained cn5; every repetition runs the depth-2 loop and compares each
batch's scores with the one-stream loop's.  DBG_MODE: real (the product), noskip (no class-major rows), pool_only / dummy_writes /
heads_then_pool (what phase B does: nothing, unrelated work, the heads with the pooled rows compared — which buffer is hit).
-DOCN_X_WAVE_CHECK: the wave pooling kernel recomputes every sum straight from memory and counts disagreements in the kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
flags = tuple(f for f in sys.argv[1:] if f.startswith("-D"))
if flags:
    os.environ["OCN_LIB_PATH"] = "/tmp/libocn_dbg.so"
import torch
if flags:
    from ocn_amd import _lib
    _lib.build(force=True, extra_flags=flags, out="/tmp/libocn_dbg.so")
from ocn_amd.sparse import SparseTensor
from ocn_amd.synth import chung_lu_graph, sample_edges
from ocn_amd import ops
from ocn_amd.model import predictor_dict
from ocn_amd.pipeline import overlapped_steps
from ocn_amd.utils import adjoverlap
DEV = torch.device("cuda:0")
n, avg, mx, B, seed, iso = int(os.environ.get('DBG_N', '20000')), 10, 600, int(os.environ.get('DBG_B', '8192')), 3, 100
ei = chung_lu_graph(n - iso, avg_deg=avg, max_deg=min(mx, n - iso - 1), seed=seed, clique_frac=0.5)       # the last `iso` nodes keep degree 0
adj = SparseTensor.from_edge_index(ei.to(DEV), sparse_sizes=(n, n)).to_symmetric()
r_, c_, _ = adj.coo()
e0 = sample_edges(r_.cpu(), c_.cpu(), n, B, seed=seed + 50)
with torch.no_grad():
    sp_ = adj.to_torch_sparse_coo_tensor()
    adj2 = SparseTensor.from_torch_sparse_coo_tensor(sp_ @ sp_, False)
adj2.nnz()
H = int(os.environ.get("DBG_H", "64"))
torch.manual_seed(seed + 5)
x = torch.randn(n, H, device=DEV)
name = os.environ.get("DBG_PRED", "cn5")
pred = predictor_dict[name](H, H, 1, 3, 0.0, 0.0, True).to(DEV).eval()
if name == "cn5":
    pred.innerprod.fill_(float(os.environ.get("DBG_IP", "0.37")))
args = SimpleNamespace(sum=0.7)
g = torch.Generator().manual_seed(9)
batches = [e0.to(DEV)[:, torch.randperm(B, generator=g).to(DEV)][:, : max(B - 5 * q, 1)].contiguous() for q in range(12)]
route = os.environ.get("DBG_ROUTE", "pattern")            # pattern: adjoverlap on A and A.A; walk: the pygho route (get_cn1_cn2)
def handles(e):
    if route == "walk":
        from ocn_amd.utils import get_cn1_cn2
        return get_cn1_cn2(adj, e)
    return adjoverlap(adj, adj, e), adjoverlap(adj, adj2, e)
def begin(it):
    return pred.begin(x, adj, *handles(batches[it]), batches[it], slot=it, args=args if it % 2 else None)
with torch.no_grad():
    ref = [pred(x, adj, *handles(e), e, args).clone() for e in batches]
    ref2 = [pred(x, adj, *handles(e), e, args).clone() for e in batches]
    print("ref repeatable", all(torch.equal(a, b) for a, b in zip(ref, ref2)))
    mode = os.environ.get("DBG_MODE", "pool_only")
    if mode == "noskip":
        ops.skip_zero_rows = False
    if os.environ.get("DBG_NOFUSE") == "1":
        ops.fused_heads = False
    def fin(tok):
        st = tok[0]
        if mode == "pool_only":          # phase B does nothing on the device: is the side stream alone enough?
            return torch.stack([t.clone() for t in st.pooled]) if st.cls is None else torch.stack([t[st.cls[1]] for t in st.pooled])
        if mode == "dummy_writes":       # phase B = unrelated main-stream work (reads and writes of its own buffers), then the pooled rows
            for _ in range(6):
                torch.mm(dummy_a, dummy_b, out=dummy_c)
                dummy_c.mul_(0.5)
            return torch.stack([t.clone() for t in st.pooled]) if st.cls is None else torch.stack([t[st.cls[1]] for t in st.pooled])
        if mode == "heads_then_pool":    # the real heads, but the result compared is the pooled rows (who is hit?)
            y = pred.finish(x, tok, args)
            pl = torch.stack([t.clone() for t in st.pooled]) if st.cls is None else torch.stack([t[st.cls[1]] for t in st.pooled])
            live.append((st.pooled, None if st.cls is None else st.cls[1], pl))
            return torch.cat([pl.reshape(-1), y.reshape(-1)])
        return pred.finish(x, tok, args)
    dummy_a = torch.randn(8192, 256, device=DEV); dummy_b = torch.randn(256, 256, device=DEV); dummy_c = torch.empty(8192, 256, device=DEV)
    loop_batch = int(os.environ['DBG_LOOP_BATCH']) if os.environ.get('DBG_LOOP_BATCH') else None     # what the loop takes its depth from (ops.loop_depth)
    live = []
    def run(overlap):
        live.clear()
        return [o.clone() for o in overlapped_steps(begin, fin, len(batches), overlap=overlap, batch=loop_batch)]
    base = run(False)
    torch.cuda.synchronize()
    base2 = run(False)
    torch.cuda.synchronize()
    print(mode, "one-stream repeatable", all(torch.equal(a, b) for a, b in zip(base, base2)))
    nbad = 0
    for rep in range(int(os.environ.get("DBG_REPS", "80"))):
        outs = run(True)
        torch.cuda.synchronize()
        for i, (a, b) in enumerate(zip(outs, base)):
            if not torch.equal(a, b):
                nbad += 1
                d = (a != b)
                print(mode, "rep", rep, "batch", i, "differing elements", int(d.sum()), "where", d.nonzero()[:3].tolist())
                if live and i >= len(batches) - 8:       # the batch's scratch set has not been used again: what is in memory NOW?
                    pooled, inv, pl = live[i]
                    now = torch.stack([t.clone() for t in pooled]) if inv is None else torch.stack([t[inv] for t in pooled])
                    ref_pl = base[i][: pl.numel()].reshape(pl.shape)
                    print("   in-loop clone == memory now:", torch.equal(now, pl), "| memory now == reference:", torch.equal(now, ref_pl))
                    w = (now != ref_pl).nonzero()
                    pln, row, c0 = int(w[0, 0]), int(w[0, 1]), int(w[0, 2])
                    e = batches[i]
                    print("   plane", pln, "row", row, "cols", c0, "..", int(w[-1, 2]), "deg(src)", int(adj._rowptr[e[0, row] + 1] - adj._rowptr[e[0, row]]),
                          "deg(dst)", int(adj._rowptr[e[1, row] + 1] - adj._rowptr[e[1, row]]))
                    print("   wrong  ", [round(v, 4) for v in now[pln, row, c0:c0 + 16].tolist()])
                    print("   correct", [round(v, 4) for v in ref_pl[pln, row, c0:c0 + 16].tolist()])
                    print("   row before (correct)", [round(v, 4) for v in ref_pl[pln, row, c0 - 4:c0].tolist()], "xcn2 same cols", [round(v, 4) for v in ref_pl[1, row, c0:c0 + 4].tolist()])
    print(mode, "failures", nbad)
    if "-DOCN_X_WAVE_CHECK" in flags:
        import ctypes
        lib = ctypes.CDLL("/tmp/libocn_dbg.so")
        outb = (ctypes.c_float * (4096 * 6))()
        nn = ctypes.c_uint(0)
        print("wave check rc", lib.ocn_debug_wave_check(outb, ctypes.byref(nn)), "in-kernel mismatches (LDS path vs memory path):", nn.value)
        for q in range(min(nn.value, 24)):
            print("   e", int(outb[q * 6]), "lane", int(outb[q * 6 + 1]), "acc1", outb[q * 6 + 2], "direct", outb[q * 6 + 3], "acc2", outb[q * 6 + 4], "direct", outb[q * 6 + 5])
