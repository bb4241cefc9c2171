set -o pipefail
bash tools/heads_pmc.sh r03b > gpurun_out/r03b_heads_pmc.log 2>&1
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r03b_heads_pmc.json"))
for k, v in d.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"   {c:34s} {x['avg']:16.1f}")
PY
