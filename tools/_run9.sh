set -o pipefail
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "pygho or fold_quirk or walk_route" > gpurun_out/r03_t9.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t9.txt
tail -12 gpurun_out/r03_t9.txt
