#!/bin/bash
# step-level A/B of an ENVIRONMENT knob on ONE box: bash tools/ab_env.sh rounds NAME v1 v2 ...
set -u -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
N=$1; NAME=$2; shift 2
mkdir -p gpurun_out/ab
for i in $(seq 1 $N); do
  for v in "$@"; do
    env $NAME=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-reduced --no-reference-loop --no-parity > gpurun_out/ab/step.json 2> gpurun_out/ab/step.err || { tail -20 gpurun_out/ab/step.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/step.json").read().strip().splitlines()[-1])
print("$NAME=$v round $i:", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms")
PY
  done
done
