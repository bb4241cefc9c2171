"""Latency of the fused head at small batches: the throughput form (128 candidates per workgroup, eight layers on one wave)
against the small-batch form (32 candidates per workgroup, the waves split every layer's output features) — same bits.
    python tools/headslat.py [--H 256] [--ln 1]
Per batch size: both forms alone on one stream (HIP events over 50 launches), bit-equality of the scores, with class ranges
(half of the rows without pooled inputs, as a Cora batch has) and without."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = tuple(f for f in sys.argv[1:] if f.startswith("-D"))
sys.argv = [v for v in sys.argv if not v.startswith("-D")]
if flags:
    os.environ["OCN_LIB_PATH"] = "/tmp/libocn_hl.so"
import torch  # noqa: E402
from ocn_amd import _lib  # noqa: E402

if flags:
    _lib.build(force=True, extra_flags=flags, out="/tmp/libocn_hl.so")
import ocn_amd.model as M  # noqa: E402
from ocn_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=256)
    ap.add_argument("--ln", type=int, default=1)
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H = a.H
    torch.manual_seed(0)
    pred = M.predictor_dict["cn5"](H, H, 1, 3, 0.0, 0.0, bool(a.ln)).to(dev).eval()
    pack = pred._fused_pack(H, dev)
    scratch = ops.buf(pred._ws, "heads_scratch", int(ops._lib.lib().ocn_heads_scratch_bytes(H)) // 4, torch.float32, dev)
    prev = ops.heads_small_batch()

    def timed(fn):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            y = fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps * 1e3, y

    if "-DOCN_X_HN_STAMPS" in flags:      # s_memtime at the phase boundaries of workgroup 0 (heads.hip: HN_STAMP)
        x1, x2, xij = (torch.randn(1152, H, device=dev) for _ in range(3))
        ops.heads_small_batch(1 << 40)
        with torch.no_grad():
            for _ in range(3):
                ops.heads_fused(x1, x2, xij, pack, None, None, True, scratch)
        torch.cuda.synchronize()
        st = scratch.view(torch.int64)[-512:][:24].cpu().tolist()
        names = ["vec+sync"] + [f"{b}.{p}" for b in "ab" for p in ("x operand", "L0", "bias+operand", "L3", "bias+LN", "operand", "Lout", "share")]
        names += ["c.x operand", "c.L0", "c.bias+LN", "c.operand", "c.Lout", "final"]
        print("  ".join(f"{n} {st[i + 1] - st[i]}" for i, n in enumerate(names)), flush=True)
        print(f"workgroup cycles (s_memtime, 100 MHz ticks x clock ratio) {st[23] - st[0]}", flush=True)
        ops.heads_small_batch(prev)
        return
    try:
        for B in (32, 256, 1152, 2048, 4096, 8192, 16384, 32768, 65536):
            x1, x2, xij = (torch.randn(B, H, device=dev) for _ in range(3))
            n3, n2, n1 = B // 4, B // 8, B // 8                      # both | cn1 only | cn2 only | none
            r = torch.tensor([[0, n3 + n2], [0, n3], [n3 + n2, n3 + n2 + n1], [0, n3 + n2 + n1], [n3 + n2 + n1, B], [n3, n3 + n2], [0, B]],
                             dtype=torch.int64, device=dev)
            for nm, ranges in (("all rows", None), ("class ranges", r)):
                out = {}
                for form, bound in (("throughput", 0), ("small", 1 << 40)):
                    ops.heads_small_batch(bound)
                    with torch.no_grad():
                        out[form] = timed(lambda: ops.heads_fused(x1, x2, xij, pack, ranges, None, True, scratch))
                same = torch.equal(out["throughput"][1], out["small"][1])
                print(f"H={H} ln={a.ln} B={B:6d} {nm:12s}: 128 per workgroup {out['throughput'][0]:8.1f} us   32 per workgroup {out['small'][0]:8.1f} us"
                      f"   bit-equal {same}", flush=True)
    finally:
        ops.heads_small_batch(prev)


if __name__ == "__main__":
    main()
