#!/bin/bash
mkdir -p gpurun_out
for t in "" "7:4" "7:8" "7:16" ""; do
  CLC_TUNING=$t timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_v13.json 2> gpurun_out/bench_v13.err || { echo "bench [$t] failed"; tail -5 gpurun_out/bench_v13.err; continue; }
  python -c "import json; d=json.load(open('gpurun_out/bench_v13.json')); print('tuning [$t]:', round(d['value'],2), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
