#!/bin/bash
# step-level A/B of environment settings on ONE box (interleaved rounds), with the transforms legs:
#   bash tools/ab_envs.sh rounds "A=1,B=2" "A=0" ...        (each variant: comma-separated assignments; "-" = no extra environment)
# TR=0 skips the transforms legs (faster).
set -u -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
N=$1; shift
mkdir -p gpurun_out/ab
extra="--no-cpu-baseline --no-reduced --no-parity --no-reference-loop"
[ "${TR:-1}" = "0" ] && extra="$extra --no-roofline"
for i in $(seq 1 $N); do
  for v in "$@"; do
    envs=""; [ "$v" != "-" ] && envs=$(echo "$v" | tr ',' ' ')
    env $envs timeout -k 10 300 python bench.py --steps ${STEPS:-30} --warmup 5 $extra ${BENCH_ARGS:-} > gpurun_out/ab/step.json 2> gpurun_out/ab/step.err || { tail -20 gpurun_out/ab/step.err; exit 1; }
    python - "$v" "$i" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab/step.json").read().strip().splitlines()[-1])
line = f"[{sys.argv[1]}] round {sys.argv[2]}: {d['value']:.2f} img/s {d['ms_per_step']:.3f} ms"
t = d.get("roofline", {}).get("transforms")
if t:
    line += f" | transforms {t['total']['ms']} ms {t['total']['frac_of_f32_mfma_peak']} | g_a {t['g_a']['ms']} g_s {t['g_s']['ms']}" + (f" ref {t['ref_encoder+adapter']['ms']}" if 'ref_encoder+adapter' in t else "")
print(line)
PY
  done
done
