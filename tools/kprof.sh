#!/bin/bash
# Top kernels of one bench configuration (GPU box):  tools/kprof.sh citation2 [extra bench flags]
cfg=$1; shift
export TMPDIR=/tmp
if [ -n "$KPROF_FLAGS" ]; then   # experimental build:  KPROF_FLAGS="-DOCN_X_WALK_NOSWEEP" tools/kprof.sh citation2
  python -c "from ocn_amd import _lib; _lib.build(force=True, extra_flags='$KPROF_FLAGS'.split(','), out='/tmp/libocn_kprof.so')"
  export OCN_LIB_PATH=/tmp/libocn_kprof.so
fi
rm -rf gpurun_out/kprof_tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/kprof_tmp -o run --output-format csv -- python bench.py --config $cfg --steps 32 --warmup 5 --no-cpu-baseline --prewarm 8 "$@" > gpurun_out/kprof_$cfg.json 2> /dev/null
python - <<PY
import csv, glob, json
f = glob.glob("gpurun_out/kprof_tmp/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int("${KPROF_ROWS:-14}")]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(5), str(round(float(r["AverageNs"]) / 1e3, 1)).rjust(9), r["MinNs"], r["MaxNs"])
d = json.loads(open("gpurun_out/kprof_$cfg.json").read().strip().splitlines()[-1])
print("$cfg", round(d["value"]), d["ms_per_step"])
PY
rm -rf gpurun_out/kprof_tmp
