#!/bin/bash
# fused ResidualUnit: parity test, micro A/B, step A/B (one box)
set -u -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
cd $R
mkdir -p gpurun_out/ru
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "fused_residual" > gpurun_out/ru/test.log 2>&1 || { tail -30 gpurun_out/ru/test.log; exit 1; }
tail -2 gpurun_out/ru/test.log
timeout -k 10 200 python tools/bench_ru.py > gpurun_out/ru/micro.log 2>&1 || { tail -20 gpurun_out/ru/micro.log; exit 1; }
cat gpurun_out/ru/micro.log
for i in 1 2; do
  for v in 0 1; do
    CLC_FUSED_RU=$v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-reduced --no-reference-loop --no-parity > gpurun_out/ru/step_${v}_$i.json 2> gpurun_out/ru/step_${v}_$i.err || { tail -20 gpurun_out/ru/step_${v}_$i.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("gpurun_out/ru/step_${v}_$i.json").read().strip().splitlines()[-1])
print("FUSED_RU=$v round $i:", d["value"], "img/s", d["ms_per_step"], "ms")
PY
  done
done
