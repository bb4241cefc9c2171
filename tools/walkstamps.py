"""Where a forward walk item's time goes (GPU box): builds the library with -DOCN_X_WALK_STAMPS (s_memtime at the phase
boundaries of cn_walk_kernel, each behind s_waitcnt 0), runs the config's scoring step and prints the mean cycles per item
and phase over the first 256 workgroups.

    python tools/walkstamps.py ppa|citation2
"""
import ctypes
import os
import sys
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = "/tmp/libocn_ws.so"
os.environ["OCN_LIB_PATH"] = LIB
from ocn_amd import _lib  # noqa: E402

_lib.build(force=True, extra_flags=("-DOCN_X_WALK_STAMPS",), out=LIB)
import torch  # noqa: E402
import bench  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "ppa"
args = SimpleNamespace(dataset=cfg, hiddim=None, predictor=None, batch=None, scale=1.0, innerprod=0.0, batches=4)
dev = torch.device("cuda:0")
wl = bench.build_workload(args, dev, 0, 1)
l = _lib.lib()
dbg = ctypes.CDLL(LIB).ocn_debug_walk_stamps
buf = (ctypes.c_ulonglong * (256 * 8))()
with torch.no_grad():
    for it in range(4):
        e = wl["edges"][it % 4]
        c1, c2 = bench.cn_handles(wl, e)
        wl["pred"](wl["h"], wl["adj"], c1, c2, e, wl["args"])
    torch.cuda.synchronize()
    dbg(buf, 1)
    n_it = 8
    for it in range(n_it):
        e = wl["edges"][it % 4]
        c1, c2 = bench.cn_handles(wl, e)
        wl["pred"](wl["h"], wl["adj"], c1, c2, e, wl["args"])
    torch.cuda.synchronize()
dbg(buf, 0)
names = ["ticket+slot+endpoints", "bitmap + set + item rows", "sweep (+flushes)", "finalise", "end barrier", "loop top"]
tot = [0] * 8
for b in range(256):
    for q in range(8):
        tot[q] += buf[b * 8 + q]
items = max(tot[6], 1)
print(f"{cfg}: {items / n_it:.0f} items per batch in the first 256 workgroups, {items / n_it / 256:.1f} per workgroup")
cyc = sum(tot[:6])
for q in range(6):
    print(f"  {names[q]:28s} {tot[q] / items:9.0f} cycles/item  {100.0 * tot[q] / cyc:5.1f} %")
print(f"  total {cyc / items:.0f} cycles per item; per workgroup and batch {cyc / n_it / 256:.0f} cycles")
