#!/bin/bash
mkdir -p gpurun_out
for cfg in "0|" "1|" "1|3:1"; do
  ws=${cfg%%|*}; t=${cfg##*|}
  CLC_WGRAD_STREAM=$ws CLC_TUNING=$t timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_v7.json 2> gpurun_out/bench_v7.err || { echo "bench ws=$ws t=$t failed"; tail -5 gpurun_out/bench_v7.err; continue; }
  python -c "import json; d=json.load(open('gpurun_out/bench_v7.json')); print('wgrad_stream=$ws tuning=[$t]:', round(d['value'],2), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
