"""Times ocn_linear_bf16x6 against torch's fp32 GEMM on the head shapes; variants via compile flags
like tools/kbench.py.  Experiments only."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch
    from ocn_amd import ops
    dev = torch.device("cuda:0")
    M, K, N = int(os.environ.get("LB_M", 65536)), 256, 256
    x = torch.randn(M, K, device=dev)
    lin = torch.nn.Linear(K, N).to(dev)
    ln = torch.nn.LayerNorm(N).to(dev)

    def timed(fn, iters=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / iters * 1e3

    with torch.no_grad():
        t_plain = timed(lambda: ops.linear(x, lin.weight, lin.bias))
        t_ln = timed(lambda: ops.linear(x, lin.weight, lin.bias, (ln.weight, ln.bias, 1e-5), True))
        t_torch = timed(lambda: torch.nn.functional.linear(x, lin.weight, lin.bias))
        err = (ops.linear(x, lin.weight, lin.bias) - torch.nn.functional.linear(x, lin.weight, lin.bias)).abs().max().item()
    fl = 2.0 * M * K * N
    print(f"linear={t_plain:.1f}us ({fl / t_plain / 1e6:.0f} TF fp32-equiv) +ln_relu={t_ln:.1f}us torch={t_torch:.1f}us "
          f"maxdiff={err:.2e} checksum", flush=True)


def main():
    from ocn_amd import _lib
    for v in [()] + [tuple(a.split(",")) for a in sys.argv[1:]]:
        out = _lib.LIB_PATH if not v else f"/tmp/libocn_{abs(hash(v))}.so"
        if v:
            _lib.build(force=True, extra_flags=v, out=out)
        env = dict(os.environ, OCN_LIB_PATH=out, KB_CHILD="1")
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if "checksum" in l]
        print(f"{' '.join(v) or 'shipped':32s} {line[-1] if line else r.stderr[-600:]}", flush=True)


if __name__ == "__main__":
    child() if os.environ.get("KB_CHILD") else main()
