#!/bin/bash
# MFMA-pipe counters of the fused heads (GPU box):  bash tools/heads_pmc.sh <tag>
# one rocprofv3 --pmc pass per counter group (never combined with other trace domains), program directly after `--`.
set -o pipefail
T=${1:-r03}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
i=0
for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS" \
         "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQ_IFETCH SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ"; do
  rm -rf $O/${T}_hpmc_$i
  HB_N=6 timeout -k 10 300 rocprofv3 --pmc $g --kernel-trace -d $O/${T}_hpmc_$i -o run --output-format csv -- python3 tools/headsbench.py > $O/${T}_hpmc_$i.out 2> $O/${T}_hpmc_$i.err || echo "pass $i failed"
  i=$((i+1))
done
python3 - $T <<'PY'
import csv, glob, json, sys
from collections import defaultdict
T = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for d in glob.glob(f"gpurun_out/{T}_hpmc_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "heads_fused" in r["Kernel_Name"]:
                a = acc[r["Kernel_Name"][:48]][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "heads_fused" in r["Kernel_Name"]:
                a = acc[r["Kernel_Name"][:48]]["duration_ns"]
                a[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); a[1] += 1
out = {k: {c: {"avg": s / n, "dispatches": n} for c, (s, n) in v.items()} for k, v in acc.items()}
json.dump(out, open(f"gpurun_out/{T}_heads_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/${T}_hpmc_[0-9]
