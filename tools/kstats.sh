#!/bin/bash
# Kernel-trace summary of one bench configuration (run on the GPU box):  tools/kstats.sh <dataset> [bench args]
# -> gpurun_out/kstats_<dataset>.csv and the ten longest kernels on stdout.
C=$1; shift
export TMPDIR=/tmp
rm -rf gpurun_out/prof_k
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_k -o run --output-format csv -- python3 bench.py --dataset $C --no-validate-leg --no-cpu-baseline --steps 32 "$@" > gpurun_out/kstats_$C.log 2>&1
cp $(find gpurun_out/prof_k -name "*kernel_stats.csv" | head -1) gpurun_out/kstats_$C.csv
rm -rf gpurun_out/prof_k
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/kstats_$C.csv")))[:10]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
