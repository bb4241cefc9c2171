"""Time ocn_heads_fused alone and its ablations (one process per variant, A/B on one box):
    python tools/headsbench.py [flags]      e.g. -DOCN_X_HD_NOGLDS | -DOCN_X_HD_NOMFMA
All rows run all branches (no class ranges): 8 panels per 128-row tile."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
flags = tuple(f for f in sys.argv[1:] if f.startswith("-D"))
if flags:
    os.environ["OCN_LIB_PATH"] = "/tmp/libocn_hb.so"
import torch  # noqa: E402
from ocn_amd import _lib  # noqa: E402

if flags:
    _lib.build(force=True, extra_flags=flags, out="/tmp/libocn_hb.so")
import ocn_amd.model as M  # noqa: E402

dev = torch.device("cuda:0")
for H, B in ((256, 65536), (128, 32768)):
    torch.manual_seed(0)
    pred = M.predictor_dict["cn5"](H, H, 1, 3, 0.0, 0.0, True).to(dev).eval()
    x1, x2, xij = (torch.randn(B, H, device=dev) for _ in range(3))
    with torch.no_grad():
        for _ in range(5):
            pred._heads_fused(x1, x2, xij, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = int(os.environ.get("HB_N", 20))
        for _ in range(n):
            pred._heads_fused(x1, x2, xij, None)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    fl = 2.0 * H * H * 8 * B
    if "-DOCN_X_HD_CLOCK" in flags:        # in-kernel clock: d(s_memtime) / d(s_memrealtime) x 100 MHz, median over workgroups
        sc = [v for v in pred._ws.values() if torch.is_tensor(v) and v.numel() * 4 == int(_lib.lib().ocn_heads_scratch_bytes(H))][0]
        per = 4 * 2 * (H // 32) * 4 * 64 * 16 // 8           # u64 words of one workgroup's park area
        st = sc.view(torch.int64).view(-1, per)[:, :2].double()
        ghz = (st[:, 0] / st[:, 1] * 0.1).median().item()
        print(f"in-kernel clock {ghz:.2f} GHz; kernel cycles (median WG) {st[:, 0].median().item():.0f}", flush=True)
    if "-DOCN_X_HD_STAMPS" in flags:       # s_memtime at the phase boundaries of workgroup 0's first tile (heads.hip: HD_STAMP)
        sc = [v for v in pred._ws.values() if torch.is_tensor(v) and v.numel() * 4 == int(_lib.lib().ocn_heads_scratch_bytes(H))][0]
        st = sc.view(torch.int64)[-512:][:21].cpu().tolist()
        names = ["ring prologue"] + [f"{b}.{p}" for b in "ab" for p in ("load x", "L0", "epi0", "L3", "epi3 LN", "L7", "park")] + ["c.load x", "c.L0", "c.epi LN", "c.Lout", "final"]
        print("  ".join(f"{n} {st[i + 1] - st[i]}" for i, n in enumerate(names)), flush=True)
        print(f"tile cycles {st[20] - st[0]}", flush=True)
    print(f"{' '.join(flags) or 'product':28s} H={H} B={B}: {dt * 1e6:8.1f} us  {fl / dt / 1e12:6.1f} TF f32-equivalent  {6 * fl / dt / 1e15:5.2f} PF bf16 issued", flush=True)
