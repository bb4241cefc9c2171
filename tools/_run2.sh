set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "fused_heads or heads_product or zero_row or smoke" > gpurun_out/r03_t2.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t2.txt
tail -15 gpurun_out/r03_t2.txt
