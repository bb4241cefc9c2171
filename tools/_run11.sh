set -o pipefail
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "collab or large_batch or graph_capture or class or zero_row" > gpurun_out/r03_t11.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t11.txt
tail -4 gpurun_out/r03_t11.txt
grep -q "rc=0" gpurun_out/r03_t11.txt || exit 1
timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline 2> gpurun_out/r03_bench11.err | grep "^{" > gpurun_out/r03_bench11.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03_bench11.json"))
print(round(d["value"]/1e6,1), "M edges/s", round(d["ms_per_step"],4), {k: round(v["ms"],4) for k,v in d["stages"].items()})
PY
