#!/bin/bash
# Round profile refresh (run on the GPU box through gpurun):  bash tools/refresh_profiles.sh r02b [configs...]
# per config: separate PMC passes (FETCH_SIZE / WRITE_SIZE / TCC hit+miss / the matrix-pipe counters; never combined with
# other trace domains) -> their summary into profiles/ (so that the bench line's roofline.traffic is THIS build's) -> the bench line
# -> rocprofv3 kernel trace of the same command.  Everything lands in gpurun_out/; copy into profiles/ afterwards.
set -eo pipefail
R=${1:-r03}; shift || true
CFGS=${@:-collab citation2 ppa ddi cora}
O=gpurun_out
mkdir -p $O
export TMPDIR=/tmp
for C in $CFGS; do
  S=""; [ "$C" = collab ] || S="_$C"
  rm -rf $O/${R}_prof $O/${R}_pmc_fetch $O/${R}_pmc_write $O/${R}_pmc_l2 $O/${R}_pmc_mfma
  if [ -z "$SKIP_PMC" ]; then      # (SKIP_PMC=1: bench line + kernel trace only — the per-kernel counters do not depend on the loop's stream layout)
  for P in "fetch FETCH_SIZE" "write WRITE_SIZE" "l2 TCC_HIT_sum TCC_MISS_sum" \
           "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16"; do
    set -- $P; D=$1; shift
    timeout -k 10 400 rocprofv3 --pmc $@ --kernel-trace -d $O/${R}_pmc_$D -o run --output-format csv -- python3 bench.py --config $C --no-cpu-baseline --no-validate-leg --no-stage-timers --steps 8 --prewarm 8 > /dev/null 2> $O/${R}_pmc_$D.err
  done
  python3 tools/pmc_summary.py $O/${R}_pmc${S}.json $O/${R}_pmc_fetch $O/${R}_pmc_write $O/${R}_pmc_l2 $O/${R}_pmc_mfma
  cp $O/${R}_pmc${S}.json profiles/${R}_pmc${S}.json
  fi
  timeout -k 10 500 python3 bench.py --config $C > $O/${R}_bench_${C}.json 2> $O/${R}_bench_${C}.err
  # the kernel trace is taken on the DRIVER's command shape (20 steps, 5 warm-up) without the serialised-streams leg, so that
  # its per-kernel averages are over the overlapped loop the line's own stage events sample
  timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/${R}_prof -o run --output-format csv -- python3 bench.py --config $C --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-validate-leg --no-one-stream-leg > $O/${R}_bench_prof_${C}.json 2> $O/${R}_prof.err
  cp $(find $O/${R}_prof -name '*kernel_stats.csv' | head -1) $O/${R}_bench${S}_kernel_stats.csv
  rm -rf $O/${R}_prof $O/${R}_pmc_fetch $O/${R}_pmc_write $O/${R}_pmc_l2 $O/${R}_pmc_mfma      # raw traces: tens of MB, gpurun_out is capped
  echo "== $C"; head -c 400 $O/${R}_bench_${C}.json; echo
done
