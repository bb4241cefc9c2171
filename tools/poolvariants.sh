#!/bin/bash
# The pooling kernel under experiment builds, one box:  bash tools/poolvariants.sh "<-D flags>" ...   (gpurun; output on stdout)
set -o pipefail
export TMPDIR=/tmp
echo "== shipped"; timeout -k 10 300 python3 tools/poolbench.py --iters 12 2>/dev/null | tail -2
for V in "$@"; do
  python3 -c "
import sys; sys.path.insert(0,'.')
from ocn_amd import _lib
_lib.build(force=True, extra_flags=tuple('$V'.split()), out='/tmp/libocn_v.so')" > /dev/null 2>&1 || { echo "build failed: $V"; continue; }
  echo "== $V"; OCN_LIB_PATH=/tmp/libocn_v.so timeout -k 10 300 python3 tools/poolbench.py --iters 12 --no-check 2>/dev/null | tail -2
done
