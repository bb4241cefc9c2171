#!/bin/bash
# Counters of the kernels matching a name, one rocprofv3 --pmc pass per counter group (GPU box):
#   tools/pmc_kernel.sh cn_walk "SQ_WAVES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" -- python tools/walkstage.py citation2
pat=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
export TMPDIR=/tmp
i=0
for g in "${groups[@]}"; do
  rm -rf gpurun_out/pmck_$i
  rocprofv3 --pmc $g --kernel-trace -d gpurun_out/pmck_$i -o run --output-format csv -- "$@" > /dev/null 2> gpurun_out/pmck_$i.err
  python - "$pat" gpurun_out/pmck_$i <<'PY'
import csv, glob, sys
from collections import defaultdict
pat, d = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f"{k:40s} {c:28s} avg {s / n:16.1f}  ({n} dispatches)")
PY
  rm -rf gpurun_out/pmck_$i
  i=$((i+1))
done
