#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_harness_gpu.py tests/test_model_gpu.py tests/test_codec_golden.py -m gpu -q -rf -x > gpurun_out/pytest_p1.log 2>&1
rc=$?; tail -n 15 gpurun_out/pytest_p1.log
if [ $rc -gt 1 ]; then exit $rc; fi
for t in "" "0:1" "1:0" "0:1,1:0"; do
  CLC_TUNING=$t timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_p1_$t.json 2> gpurun_out/bench_p1_$t.err || { echo "bench $t failed"; tail -5 gpurun_out/bench_p1_$t.err; exit 3; }
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_p1_$t.json')); print('tuning [$t]', round(d['value'],2), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
AB=1 timeout -k 10 300 python tools/bench_conv.py 20 > gpurun_out/bench_conv_p1.log 2>&1; tail -n 30 gpurun_out/bench_conv_p1.log
exit $rc
