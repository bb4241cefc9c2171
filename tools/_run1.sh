set -o pipefail
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/r03_counters.txt 2>&1 || true
python3 tools/headsbench.py > gpurun_out/r03_hb_base.txt 2>&1 &&
python3 tools/headsbench.py -DOCN_X_HD_NOEPI > gpurun_out/r03_hb_noepi.txt 2>&1 &&
python3 tools/headsbench.py -DOCN_X_HD_NOPARK > gpurun_out/r03_hb_nopark.txt 2>&1 &&
python3 tools/headsbench.py -DOCN_X_HD_NOGLDS > gpurun_out/r03_hb_noglds.txt 2>&1 &&
python3 tools/headsbench.py -DOCN_X_HD_NOGLDS -DOCN_X_HD_NOEPI -DOCN_X_HD_NOPARK > gpurun_out/r03_hb_noall.txt 2>&1 &&
python3 tools/headsbench.py -DOCN_X_HD_CLOCK > gpurun_out/r03_hb_clock.txt 2>&1 &&
bash tools/heads_pmc.sh r03 > gpurun_out/r03_heads_pmc.log 2>&1
tail -n 3 gpurun_out/r03_hb_*.txt
