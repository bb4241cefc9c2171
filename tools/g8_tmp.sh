set -o pipefail
cd /root/repo
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py tests/test_configs_gpu.py tests/test_property_gpu.py -x -q -k "walk or ppa or citation2 or pygho or cn7 or mrr or get_cn1 or ddi or flags or adjoverlap" > gpurun_out/t8.txt 2>&1; rc=$?; tail -3 gpurun_out/t8.txt
test $rc -eq 0 && for c in citation2 ppa collab; do timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-validate-leg --steps 64 > gpurun_out/r03f_bench_$c.json 2> gpurun_out/b.err && python - $c <<'PY'
import json,sys
c=sys.argv[1]
d=json.loads(open(f"gpurun_out/r03f_bench_{c}.json").read().strip().splitlines()[-1])
print(c, round(d["value"]/1e6,3), "M", round(d["ms_per_step"],4), {k:round(v["ms"],4) for k,v in d["stages"].items()})
PY
done
