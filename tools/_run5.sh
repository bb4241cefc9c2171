set -o pipefail
HB_N=10 python3 tools/headsbench.py -DOCN_X_HD_STAMPS > gpurun_out/r03_hb_stamps.txt 2>&1
tail -n 8 gpurun_out/r03_hb_stamps.txt
