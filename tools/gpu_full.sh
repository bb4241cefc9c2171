#!/bin/bash
# The whole GPU suite on the GPU box (through gpurun): bash tools/gpu_full.sh  -> gpurun_out/gpu_full.txt
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_full.txt 2>&1
echo "rc=$?" >> gpurun_out/gpu_full.txt
tail -8 gpurun_out/gpu_full.txt
