set -o pipefail
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "fused_heads or heads_product or zero_row or smoke" > gpurun_out/r03_t8.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t8.txt
tail -3 gpurun_out/r03_t8.txt
grep -q "rc=0" gpurun_out/r03_t8.txt || exit 1
python3 tools/headsbench.py > gpurun_out/r03_hb8_base.txt 2>&1 &&
HB_N=10 python3 tools/headsbench.py -DOCN_X_HD_STAMPS > gpurun_out/r03_hb8_stamps.txt 2>&1
tail -n 3 gpurun_out/r03_hb8_base.txt; tail -n 6 gpurun_out/r03_hb8_stamps.txt
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t7.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t7.txt
tail -8 gpurun_out/r03_t7.txt
