set -o pipefail
timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline > gpurun_out/r03_bench10.json 2> gpurun_out/r03_bench10.err || { tail -20 gpurun_out/r03_bench10.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline --rehearse-collectives > gpurun_out/r03_bench10r.json 2> gpurun_out/r03_bench10r.err || { tail -20 gpurun_out/r03_bench10r.err; exit 1; }
timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline --innerprod 0.37 > gpurun_out/r03_bench10i.json 2> gpurun_out/r03_bench10i.err || { tail -20 gpurun_out/r03_bench10i.err; exit 1; }
python3 - <<'PY'
import json
for f in ("r03_bench10", "r03_bench10r", "r03_bench10i"):
    d = json.load(open(f"gpurun_out/{f}.json"))
    print(f, round(d["value"]/1e6,1), "M edges/s", round(d["ms_per_step"],4), d.get("communication_pattern"), {k: round(v["ms"],4) for k,v in d["stages"].items()})
PY
timeout -k 10 600 python -m pytest tests/test_dist_gpu.py tests/test_parity_gpu.py -x -q -m gpu -k "dist or two_phase or sharded" > gpurun_out/r03_t10.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t10.txt
tail -4 gpurun_out/r03_t10.txt
