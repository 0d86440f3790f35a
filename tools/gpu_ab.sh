#!/bin/bash
set -u -o pipefail
# A/B of CLC_TUNING settings on the step rate:  bash tools/gpu_ab.sh "9:0" "9:1" ...   (each setting measured twice, interleaved)
mkdir -p gpurun_out
for rep in 1 2; do
for t in "$@"; do
  CLC_TUNING=$t timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err || { echo "bench [$t] failed"; tail -5 gpurun_out/bench_ab.err; exit 4; }
  python - "$t" <<'PY'
import json, sys
d = json.load(open('gpurun_out/bench_ab.json'))
print(f"tuning [{sys.argv[1]}]: {d['value']:.2f} img/s  {d['ms_per_step']:.3f} ms")
PY
done
done
