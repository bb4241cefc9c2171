"""Where the pooling launch's time goes, per wave: builds the library with -DOCN_X_POOL_STAMPS (s_memtime at a wave's start and
end, its candidate's row length and entry counts, XCC / CU ids), runs the collab-shaped batch's pooling once warm and prints
the timeline (round 4: of the persistent cn_gather_loop_kernel — a "wave" below is one candidate's share of its wave's loop): launch length, wave durations by entry-count class, when the last wave of each class ends, waves alive over
time.  Diagnostic only (gpurun):  python tools/poolstamps.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import argparse
    import numpy as np
    import torch
    import bench
    from ocn_amd import _lib, ops
    from ocn_amd.utils import CNState
    args = argparse.Namespace(dataset="collab", scale=1.0, hiddim=None, predictor=None, batch=None, batches=2, innerprod=0.0)
    dev = torch.device("cuda:0")
    wl = bench.build_workload(args, dev, 0, 1)
    adj, adj2, h = wl["adj"], wl["adj2"], wl["h"]
    ops.validate_indices = False
    ops.heavy_first = os.environ.get("PS_HEAVY", "1") == "1"
    l = _lib.lib()
    l.ocn_debug_pool_stamps.restype = ctypes.c_int
    l.ocn_debug_pool_stamps.argtypes = [ctypes.c_void_p, ctypes.c_longlong]
    for rep in range(4):
        for e in wl["edges"]:
            st = CNState(adj, adj, adj2, e)
            w = st.weights_cn5(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record(); st.gather(w, h); t1.record()
            torch.cuda.synchronize()
    B = e.shape[1]
    buf = np.zeros((B, 4), dtype=np.uint64)
    assert l.ocn_debug_pool_stamps(buf.ctypes.data, B) == 0
    ms = t0.elapsed_time(t1)
    s, en = buf[:, 0].astype(np.int64), buf[:, 1].astype(np.int64)
    ok = en > 0
    da = (buf[:, 2] & np.uint64(0xfffff)).astype(np.int64)
    ent = (st.cnt1.long() + st.cnt2.long())[st.rec[:, 0]].cpu().numpy()          # entries of the candidate in each processing slot
    npass = (buf[:, 2] >> np.uint64(32)).astype(np.int64)
    xcc = (buf[:, 3] >> np.uint64(32)).astype(np.int64)
    # the counter has no common epoch across XCCs (and, as measured, not across the CUs of one either): times are taken relative
    # to the first wave start of the wave's own CU (HW_ID: cu_id bits 11:8, sh 12, se 15:13), the tick length from a CU's span
    hw = (buf[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
    cu = (xcc << 16) | ((hw >> 8) & 0xff)
    base = np.zeros(B, dtype=np.int64)
    spans = []
    for c in np.unique(cu[ok]):
        m = ok & (cu == c)
        base[m] = s[m].min()
        spans.append(int((en[m] - base[m]).max()))
    span = max(spans)
    raw = (en - s)[ok]
    print(f"{len(spans)} distinct (xcc, cu) ids; per-CU spans in ticks: min {min(spans)} median {int(np.median(spans))} max {span}; "
          f"wave durations in ticks: median {int(np.median(raw))} p99 {int(np.percentile(raw, 99))} max {int(raw.max())}")
    tick_us = ms * 1e3 / span                       # (launch length by events / longest per-CU span in ticks; includes launch overhead)
    print(f"launch {ms * 1e3:.1f} us by events ({tick_us * 1e3:.2f} ns per tick), stamped {ok.sum()} of {B} waves")
    dur = (en - s) * tick_us
    start, end = (s - base) * tick_us, (en - base) * tick_us
    for lo, hi in ((0, 0), (1, 8), (9, 64), (65, 128), (129, 256), (257, 100000)):
        m = ok & (ent >= lo) & (ent <= hi)
        if m.sum():
            print(f"entries {lo:4d}..{hi:6d}: {m.sum():6d} waves, duration median {np.median(dur[m]):7.1f} us p90 {np.percentile(dur[m], 90):7.1f} max {dur[m].max():7.1f}; "
                  f"start median {np.median(start[m]):6.1f} last start {start[m].max():6.1f}; last end {end[m].max():6.1f}; mean row length {da[m].mean():.0f}")
    edges = np.linspace(0, span * tick_us * 0.999, 21)
    alive = [int((ok & (start <= t) & (end > t)).sum()) for t in edges]
    print("waves alive at 5 % steps of the launch:", alive)
    alive_heavy = [int((ok & (ent > 128) & (start <= t) & (end > t)).sum()) for t in edges]
    print("... of them with > 128 entries:        ", alive_heavy)
    for x in range(16):
        m = ok & (xcc == x)
        if m.sum():
            print(f"XCC {x}: {m.sum():5d} waves, entries {ent[m].sum():8d}, last end {end[m].max():6.1f} us, busy wave-us {dur[m].sum():10.0f}, "
                  f"mean waves alive {dur[m].sum() / end[m].max():6.0f}")
    m = ok & (ent > 128)
    if m.sum() > 10:
        a, b = np.polyfit(ent[m], dur[m], 1)
        print(f"heavy waves: duration ~ {b:.1f} us + {a * 1e3:.1f} ns per entry")
    print("candidates per wave (passes):", int(npass[ok].max()), "; start of each wave's LAST candidate: median %.1f us, max %.1f us" % (
        np.median(start[ok & (npass == npass[ok].max())]), start[ok].max()))
    m = ok & (ent == 0)
    print(f"waves without entries: mean {dur[m].mean():.2f} us; all waves: total busy {dur[ok].sum() / 1e3:.1f} wave-ms")


def main():
    from ocn_amd import _lib
    out = "/tmp/libocn_poolstamps.so"
    _lib.build(force=True, extra_flags=("-DOCN_X_POOL_STAMPS",) + tuple(sys.argv[1:]), out=out)
    for heavy in ("1",) if len(sys.argv) > 1 else ("1", "0"):
        print(f"== heavy_first={heavy}", flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=dict(os.environ, OCN_LIB_PATH=out, PS_CHILD="1", PS_HEAVY=heavy),
                           capture_output=True, text=True)
        print(r.stdout[-4000:] if r.returncode == 0 else r.stderr[-3000:], flush=True)


if __name__ == "__main__":
    child() if os.environ.get("PS_CHILD") else main()
