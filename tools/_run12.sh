set -o pipefail
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_dist_gpu.py tests/test_property_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "innerprod or colsum or cn5 or cn6 or trained or ring or appendix or order_sensitive" > gpurun_out/r03_t12.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t12.txt
tail -5 gpurun_out/r03_t12.txt
grep -q "rc=0" gpurun_out/r03_t12.txt || exit 1
timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline 2> gpurun_out/r03_bench12.err | grep "^{" > gpurun_out/r03_bench12.json
timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline --innerprod 0.37 2> gpurun_out/r03_bench12i.err | grep "^{" > gpurun_out/r03_bench12i.json
python3 - <<'PY'
import json
for f in ("r03_bench12", "r03_bench12i"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, round(d["value"]/1e6,1), "M edges/s", round(d["ms_per_step"],4), "trained:", d.get("value_trained_innerprod"), {k: round(v["ms"],4) for k,v in d["stages"].items()})
PY
