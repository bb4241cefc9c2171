set -o pipefail
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03_t7.txt 2>&1
echo "rc=$?" >> gpurun_out/r03_t7.txt
tail -8 gpurun_out/r03_t7.txt
