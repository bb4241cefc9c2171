set -o pipefail
cd /root/repo; export TMPDIR=/tmp
bash tools/refresh_profiles.sh r03 collab citation2 ppa ddi cora &&
rm -rf gpurun_out/tp2 && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/tp2 -o run --output-format csv -- python3 bench.py --innerprod 0.37 --no-cpu-baseline --no-validate-leg --steps 64 > gpurun_out/r03_bench_prof_trained.json 2>> gpurun_out/train.err &&
cp $(find gpurun_out/tp2 -name "*kernel_stats.csv" | head -1) gpurun_out/r03_bench_trained_kernel_stats.csv && rm -rf gpurun_out/tp2
