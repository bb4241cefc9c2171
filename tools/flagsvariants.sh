#!/bin/bash
# The intersection kernel under experiment builds, one box, each variant twice in alternation:  bash tools/flagsvariants.sh "<-D flags>" ...
set -o pipefail
export TMPDIR=/tmp
i=0
for V in "$@"; do
  python3 -c "
import sys; sys.path.insert(0,'.')
from ocn_amd import _lib
_lib.build(force=True, extra_flags=tuple('$V'.split()), out='/tmp/libocn_f$i.so')" > /dev/null 2>&1 || echo "build failed: $V"
  i=$((i+1))
done
for rep in 1 2 3; do
  echo "shipped: $(KB_ITERS=40 timeout -k 10 200 python3 tools/kbench.py 2>/dev/null | tail -1 | cut -c1-150)"
  i=0
  for V in "$@"; do
    echo "$V: $(OCN_LIB_PATH=/tmp/libocn_f$i.so KB_CHILD=1 KB_ITERS=40 timeout -k 10 200 python3 tools/kbench.py 2>/dev/null | tail -1 | cut -c1-150)"
    i=$((i+1))
  done
done
