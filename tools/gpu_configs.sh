#!/bin/bash
set -u -o pipefail
# Every single-GPU BASELINE configuration at the current build, each a full bench line with roofline + transforms (+ parity), kept as JSON:
#   configs[2]  n_refs 3, lambda 0.025, 256x256 bs8         -> <tag>_bench_cfg2.json
#   configs[4]  512x512 bs4, n_refs 3, MS-SSIM, lambda 0.05 -> <tag>_bench_cfg4.json   (its 1-GPU half)
#   N = 128     train_CLC.py's default width, configs[1]'s shape -> <tag>_bench_n128.json
# usage: bash tools/gpu_configs.sh <tag>     (files land in gpurun_out/; copy them to profiles/)
tag=${1:-r4}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R; mkdir -p gpurun_out
run() {
  name=$1; shift
  timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-reduced --no-reference-loop --no-parity "$@" > gpurun_out/${tag}_bench_$name.json 2> gpurun_out/${tag}_bench_$name.err || { echo "bench [$name] failed"; tail -5 gpurun_out/${tag}_bench_$name.err; return 1; }
  python - gpurun_out/${tag}_bench_$name.json "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"[{sys.argv[2]}] {d['value']:.2f} img/s {d['ms_per_step']:.3f} ms/step | transforms {r.get('transforms_ms')} ms frac {r.get('transforms_frac')} | dominant {r['kernel']} frac {r['frac']} | {d['config']['workload'][:80]}")
PY
}
run cfg2 --n-refs 3 --lmbda 0.025 && run cfg4 --size 512 --batch 4 --n-refs 3 --loss ms_ssim --lmbda 0.05 && run n128 --N 128
