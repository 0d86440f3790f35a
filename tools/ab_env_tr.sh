#!/bin/bash
# step-level A/B of an ENVIRONMENT knob on ONE box, with the transforms legs: bash tools/ab_env_tr.sh rounds NAME v1 v2 ...
set -u -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
N=$1; NAME=$2; shift 2
mkdir -p gpurun_out/ab
for i in $(seq 1 $N); do
  for v in "$@"; do
    env $NAME=$v timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-reduced --no-reference-loop --no-parity > gpurun_out/ab/step.json 2> gpurun_out/ab/step.err || { tail -20 gpurun_out/ab/step.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/ab/step.json").read().strip().splitlines()[-1])
t=d["roofline"]["transforms"]
print("$NAME=$v round $i:", round(d["value"],2), "img/s", round(d["ms_per_step"],3), "ms | transforms", t["total"]["ms"], "ms", t["total"]["frac_of_f32_mfma_peak"], "| g_a", t["g_a"]["ms"], "g_s", t["g_s"]["ms"], "ref", t["ref_encoder+adapter"]["ms"])
PY
  done
done
