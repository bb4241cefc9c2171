set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/r03_kp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/r03_kp -o run --output-format csv -- python3 bench.py --config collab --innerprod 0.37 --no-cpu-baseline --no-validate-leg --steps 64 > $O/r03_kp.json 2> $O/r03_kp.err
cp $(find $O/r03_kp -name '*kernel_stats.csv' | head -1) $O/r03_kstats_collab_ip.csv
rm -rf $O/r03_kp
head -22 $O/r03_kstats_collab_ip.csv | cut -c1-150
for C in cora ddi ppa citation2; do
  timeout -k 10 300 python3 bench.py --config $C --no-cpu-baseline 2> $O/r03_b13_$C.err | grep "^{" > $O/r03_b13_$C.json
  python3 - $C <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/r03_b13_{sys.argv[1]}.json"))
print(sys.argv[1], round(d["value"]/1e6,2), "M edges/s", round(d["ms_per_step"],4), {k: round(v["ms"],4) for k,v in d["stages"].items()})
PY
done
