"""A·A of the collab-shaped training graph (the per-batch product of NeighborOverlap_large.py:66-74), pass by pass:
    python tools/spgemmbench.py
counting pass with / without the dense bit rows, scan, fill pass; HIP events, 10 repetitions each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from ocn_amd import ops  # noqa: E402
from ocn_amd.sparse import SparseTensor  # noqa: E402
from ocn_amd.synth import dataset_like  # noqa: E402


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, out


def main():
    dev = torch.device("cuda:0")
    ei, n, shape = dataset_like("collab", seed=0, scale=1.0)
    adj = SparseTensor.from_edge_index(ei.to(dev), sparse_sizes=(n, n)).to_symmetric()
    rp, col = adj._rowptr, adj._col
    l = ops._lib.lib()
    words = (n + 31) // 32
    cnt = torch.empty(n, dtype=torch.int32, device=dev)
    bitmap = torch.empty(n, words, dtype=torch.int32, device=dev)
    ptr, sp = ops.ptr, ops.stream_ptr

    def count(bm):
        ops.check(l.ocn_spgemm_pattern_count(ptr(rp), ptr(col), n, ptr(rp), ptr(col), n, ptr(cnt), ptr(bm), words, sp()), "count")
    t_bm, _ = timed(lambda: count(bitmap))
    t_nb, _ = timed(lambda: count(None))
    t_zero, _ = timed(lambda: bitmap.zero_())
    t_scan, rowptrC = timed(lambda: ops.scan_i32(cnt))
    nnz = int(rowptrC[-1])
    colC = torch.empty(nnz, dtype=torch.int32, device=dev)
    t_fill, _ = timed(lambda: ops.check(l.ocn_spgemm_pattern_fill(ptr(rp), ptr(col), n, ptr(rp), ptr(col), n, ptr(rowptrC), ptr(colC), sp()), "fill"))
    print(f"n = {n}, nnz(A) = {col.numel()}, nnz(A.A) = {nnz} ({nnz / n:.0f} per row), bit rows {bitmap.numel() * 4 / 1e9:.2f} GB")
    print(f"counting pass with bit rows {t_bm:.3f} ms | without {t_nb:.3f} ms | zeroing the bit rows {t_zero:.3f} ms | scan {t_scan:.3f} ms | fill pass {t_fill:.3f} ms")


if __name__ == "__main__":
    main()
