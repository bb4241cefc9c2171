set -o pipefail
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "fused_heads or heads_product or zero_row or smoke" > gpurun_out/r03_t6.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t6.txt
tail -5 gpurun_out/r03_t6.txt
grep -q "rc=0" gpurun_out/r03_t6.txt || exit 1
python3 tools/headsbench.py > gpurun_out/r03_hb6_base.txt 2>&1 &&
HB_N=10 python3 tools/headsbench.py -DOCN_X_HD_STAMPS > gpurun_out/r03_hb6_stamps.txt 2>&1 &&
timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline > gpurun_out/r03_bench6.json 2> gpurun_out/r03_bench6.err
tail -n 3 gpurun_out/r03_hb6_base.txt; tail -n 6 gpurun_out/r03_hb6_stamps.txt; head -c 300 gpurun_out/r03_bench6.json
