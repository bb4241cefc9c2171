#!/bin/bash
# Register / LDS / spill figures of every kernel in one csrc file:  tools/kres.sh linear.hip [extra flags]
f=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -I/root/repo/include -S --cuda-device-only \
    /root/repo/ocn_amd/csrc/$f -o /tmp/kres.s "$@" 2>/dev/null
awk '/^\s+- \.agpr_count:/{a=$3} /\.group_segment_fixed_size:/{l=$2} /\.name:/{n=$2} /\.private_segment_fixed_size:/{p=$2} /\.sgpr_count:/{s=$2} /\.vgpr_count:/{v=$2} /\.vgpr_spill_count:/{print n, "vgpr", v, "agpr", a, "sgpr", s, "lds", l, "scratch", p, "spill", $2}' /tmp/kres.s
