#!/bin/bash
set -u -o pipefail
# the other BASELINE configs' step rates (not the headline line): configs[2] = n_refs 3 @ 256^2 bs8; configs[4]'s shape = 512^2 bs4 n_refs 3
mkdir -p gpurun_out
for cfg in "--n-refs 3" "--n-refs 3 --size 512 --batch 4" "--n-refs 1 --size 512 --batch 4" "--no-graph"; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-parity $cfg > gpurun_out/bench_cfg.json 2> gpurun_out/bench_cfg.err || { echo "bench [$cfg] failed"; tail -5 gpurun_out/bench_cfg.err; continue; }
  python - "$cfg" <<'PY'
import json, sys
d = json.load(open('gpurun_out/bench_cfg.json'))
print(f"[{sys.argv[1]}]: {d['value']:.2f} img/s  {d['ms_per_step']:.3f} ms/step")
PY
done
