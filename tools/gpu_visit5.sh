#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_harness_gpu.py tests/test_model_gpu.py -m gpu -q -rf --durations=4 > gpurun_out/pytest_v5.log 2>&1
rc=$?; tail -n 20 gpurun_out/pytest_v5.log
CLC_FORCE_SPLIT_GRAPHS=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_v5_split.json 2> gpurun_out/bench_v5_split.err && python -c "import json; d=json.load(open('gpurun_out/bench_v5_split.json')); print('forced split graphs', round(d['value'],2), 'img/s')"
exit $rc
