set -o pipefail
export TMPDIR=/tmp
O=gpurun_out
rm -rf $O/r03_kp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/r03_kp -o run --output-format csv -- python3 bench.py --config collab --innerprod 0.37 --no-cpu-baseline --no-validate-leg --steps 64 > $O/r03_kp.json 2> $O/r03_kp.err
cp $(find $O/r03_kp -name '*kernel_stats.csv' | head -1) $O/r03_kstats_collab_ip.csv
rm -rf $O/r03_kp
python3 - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r03_kstats_collab_ip.csv")))[:24]:
    if "colsum" in r["Name"] or "scan_chained<I32" in r["Name"] or "column" in r["Name"]:
        print(f"{r['Name'][:60]:60s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
