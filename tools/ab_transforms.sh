#!/bin/bash
# A/B of tuning settings on ONE box, reading the step AND the transforms leg: bash tools/ab_transforms.sh "16:1" "16:3"
set -u -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
mkdir -p gpurun_out/ab
for i in 1 2; do
for v in "$@"; do
  CLC_TUNING=$v timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reduced --no-reference-loop --no-parity > gpurun_out/ab/t.json 2> gpurun_out/ab/t.err || { tail -20 gpurun_out/ab/t.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab/t.json").read().strip().splitlines()[-1])
t=d["roofline"]["transforms"]
pk=d["roofline"]["per_kernel"]
att={k:(v["launches"],v["ms"]) for k,v in pk.items() if "winattn" in k}
print("CLC_TUNING=$v:", round(d["ms_per_step"],3), "ms/step; transforms", {k:t[k]["ms"] for k in ("g_a","g_s","total")}, att)
PY
done
done
