"""Fold rocprofv3 --pmc passes (one directory per pass) into one JSON: per kernel and counter, the
average counter value over its dispatches (raw units: FETCH_SIZE / WRITE_SIZE are KiB per dispatch
before bench.py's gfx950 correction).

    python tools/pmc_summary.py out.json dir_fetch dir_write dir_l2
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def fold(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"][:60]
                    a = acc[k][row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: {c: {"avg": s / n, "dispatches": n} for c, (s, n) in v.items()} for k, v in acc.items()}


if __name__ == "__main__":
    out = fold(sys.argv[2:])
    with open(sys.argv[1], "w") as fh:
        json.dump(out, fh, indent=1)
    print(f"{len(out)} kernels -> {sys.argv[1]}")
