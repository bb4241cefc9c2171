set -o pipefail
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_configs_gpu.py -x -q -k "heads or fused or zero_row or skipping or end_to_end" > gpurun_out/t6.txt 2>&1; rc=$?; tail -3 gpurun_out/t6.txt
test $rc -eq 0 && for c in collab cora; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline > gpurun_out/r03c_bench_$c.json 2> gpurun_out/b.err && python - $c <<'PY'
import json,sys
c=sys.argv[1]
d=json.loads(open(f"gpurun_out/r03c_bench_{c}.json").read().strip().splitlines()[-1])
print(c, round(d["value"]/1e6,2), "M", round(d["ms_per_step"],4), {k:round(v["ms"],4) for k,v in d["stages"].items()}, "trained", d.get("value_trained_innerprod"))
PY
done
