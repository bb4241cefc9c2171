#!/bin/bash
# Round profile refresh (run on the GPU box through gpurun):  bash tools/refresh_profiles.sh r01
# tests -> default bench -> rocprofv3 kernel trace of the same command -> three separate PMC passes.
set -eo pipefail
R=${1:-r01}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x > $O/${R}_tests.log 2>&1
python bench.py > $O/${R}_bench_collab.json 2> $O/${R}_bench_collab.err
rm -rf $O/${R}_prof $O/${R}_pmc_fetch $O/${R}_pmc_write $O/${R}_pmc_l2
rocprofv3 --kernel-trace --stats -d $O/${R}_prof -o run --output-format csv -- python bench.py --no-cpu-baseline > $O/${R}_bench_prof.json 2> $O/${R}_prof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/${R}_pmc_fetch -o run --output-format csv -- python bench.py --no-cpu-baseline --steps 6 > /dev/null 2> $O/${R}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/${R}_pmc_write -o run --output-format csv -- python bench.py --no-cpu-baseline --steps 6 > /dev/null 2> $O/${R}_pmc_write.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/${R}_pmc_l2 -o run --output-format csv -- python bench.py --no-cpu-baseline --steps 6 > /dev/null 2> $O/${R}_pmc_l2.err
find $O/${R}_pmc_fetch -type f | head -5; tail -3 $O/${R}_pmc_fetch.err
python tools/pmc_summary.py $O/${R}_pmc.json $O/${R}_pmc_fetch $O/${R}_pmc_write $O/${R}_pmc_l2
cp $(find $O/${R}_prof -name '*kernel_stats.csv' | head -1) $O/${R}_bench_kernel_stats.csv
rm -rf $O/${R}_prof $O/${R}_pmc_fetch $O/${R}_pmc_write $O/${R}_pmc_l2      # raw traces: tens of MB, gpurun_out is capped
tail -3 $O/${R}_tests.log
cat $O/${R}_bench_collab.json
