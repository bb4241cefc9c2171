set -o pipefail
mkdir -p gpurun_out
python3 tools/headsbench.py > gpurun_out/r03_hb2_base.txt 2>&1 &&
python3 tools/headsbench.py -DOCN_X_HD_NOGLDS > gpurun_out/r03_hb2_noglds.txt 2>&1 &&
timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err
tail -n 3 gpurun_out/r03_hb2_*.txt; head -c 1500 gpurun_out/r03_bench2.json
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t3.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t3.txt
tail -8 gpurun_out/r03_t3.txt
