"""Where a hub row's pooling time goes (GPU box): builds the library with -DOCN_X_LONG_STAMPS (s_memtime in thread 0 of every
cn_gather_long_kernel workgroup that has a hub row, each stamp behind s_waitcnt 0), runs the config's scoring step, prints
cycles per hub row and phase, the live entries per round and the slowest row.

    python tools/longstamps.py citation2|ppa
"""
import ctypes
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = "/tmp/libocn_ls.so"
os.environ["OCN_LIB_PATH"] = LIB
os.environ["OCN_ONE_STREAM"] = "1"
from ocn_amd import _lib  # noqa: E402

_lib.build(force=True, extra_flags=("-DOCN_X_LONG_STAMPS",), out=LIB)
import torch  # noqa: E402
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "citation2"
args = SimpleNamespace(dataset=cfg, hiddim=None, predictor=None, batch=None, scale=1.0, innerprod=0.0, batches=4)
dev = torch.device("cuda:0")
wl = bench.build_workload(args, dev, 0, 1)
_lib.lib()
dbg = ctypes.CDLL(LIB).ocn_debug_long_stamps
buf = (ctypes.c_ulonglong * 8)()
with torch.no_grad():
    for it in range(4):
        e = wl["edges"][it % 4]
        wl["pred"](wl["h"], wl["adj"], *bench.cn_handles(wl, e), e, wl["args"])
    torch.cuda.synchronize()
    dbg(buf, 1)
    n_it = 8
    for it in range(n_it):
        e = wl["edges"][it % 4]
        wl["pred"](wl["h"], wl["adj"], *bench.cn_handles(wl, e), e, wl["args"])
    torch.cuda.synchronize()
dbg(buf, 0)
t = list(buf)
rows, rounds = max(t[6], 1), max(t[5], 1)
names = ["ids / flags / values", "column weights", "compaction (2 barriers)", "sub-rounds (fetch, store, sum)"]
tot = sum(t[:4])
print(f"{cfg}: {rows / n_it:.0f} hub rows per batch, {rounds / rows:.1f} rounds per row, {t[4] / rounds:.0f} live entries per round")
for q in range(4):
    print(f"  {names[q]:32s} {t[q] / rows:9.0f} cycles/row  {t[q] / rounds:8.0f} /round  {100.0 * t[q] / tot:5.1f} %")
print(f"  mean {tot / rows:.0f} cycles per hub row, slowest row {t[7]} cycles; chain alone at 8 cycles per entry: {8 * t[4] / rows:.0f} per row")
