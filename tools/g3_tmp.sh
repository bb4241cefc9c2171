set -o pipefail
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_fullsize_gpu.py tests/test_configs_gpu.py -x -q -k "full_rows or ddi or cn7 or hub" > gpurun_out/t5.txt 2>&1; rc=$?; tail -4 gpurun_out/t5.txt
test $rc -eq 0 && timeout -k 10 300 python bench.py --config ddi > gpurun_out/r03b_bench_ddi.json 2> gpurun_out/b.err && python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03b_bench_ddi.json").read().strip().splitlines()[-1])
print("ddi", round(d["value"]/1e6,2), "M", round(d["ms_per_step"],4), {k:round(v["ms"],4) for k,v in d["stages"].items()}, "parity", d.get("parity_on_cpu_sample_max_abs_err"), d.get("parity_on_cpu_sample_max_abs_ref"))
PY
