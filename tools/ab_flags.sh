#!/bin/bash
# A/B of library variants on ONE box: bash tools/ab_flags.sh "<-D flags variant 1>" "<variant 2>" ...  (bench stage times per variant)
set -o pipefail
mkdir -p gpurun_out
for V in "$@"; do
  OCN_LIB_PATH=/tmp/libocn_v.so python3 -c "
import sys; sys.path.insert(0,'.')
from ocn_amd import _lib
_lib.build(force=True, extra_flags=tuple('$V'.split()), out='/tmp/libocn_v.so')" > /dev/null 2>&1
  OCN_LIB_PATH=/tmp/libocn_v.so timeout -k 10 300 python3 bench.py --steps 64 --no-cpu-baseline --no-validate-leg ${AB_ARGS} 2> gpurun_out/ab.err | grep "^{" > gpurun_out/ab.json
  python3 - "$V" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(sys.argv[1] or "default", round(d["value"]/1e6,1), "M edges/s", {k: round(v["ms"],4) for k,v in d["stages"].items()})
PY
done
