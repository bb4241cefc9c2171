#!/bin/bash
# A/B of the product library against a build with extra -D flags, same box, same bench command:
#   tools/ab_bench.sh "-DOCN_X_NOSLICE" [bench.py args...]        (run through gpurun; output under gpurun_out/)
set -e
FLAGS="$1"; shift
mkdir -p gpurun_out
python -c "from ocn_amd import _lib; _lib.build(force=True, extra_flags=tuple('$FLAGS'.split()), out='/tmp/libocn_ab.so')"
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-validate-leg --no-cpu-baseline "$@" > gpurun_out/ab_A$i.log 2>gpurun_out/ab_A$i.err
  OCN_LIB_PATH=/tmp/libocn_ab.so timeout -k 10 300 python bench.py --no-validate-leg --no-cpu-baseline "$@" > gpurun_out/ab_B$i.log 2>gpurun_out/ab_B$i.err
done
python - <<'PY'
import json
for tag in ("A1", "B1", "A2", "B2"):
    l = [x for x in open(f"gpurun_out/ab_{tag}.log").read().splitlines() if x.startswith("{")][-1]
    d = json.loads(l)
    print(tag, "product" if tag[0] == "A" else "variant", f"{d['value'] / 1e6:.2f} M/s {d['ms_per_step']:.4f} ms", {k: round(v["ms"], 4) for k, v in d.get("stages", {}).items()}, "checksum", d["score_checksum"])
PY
