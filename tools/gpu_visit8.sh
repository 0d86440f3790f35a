#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_model_gpu.py -m gpu -q -rf -x -k "train_engine_steps" > gpurun_out/pytest_v8a.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_v8a.log | cut -c1-300
if [ $rc -ne 0 ]; then
  echo "--- retry with gates off"
  CLC_ACT_GATE=0 timeout -k 10 300 python -m pytest tests/test_model_gpu.py -m gpu -q -rf -x -k "train_engine_steps" > gpurun_out/pytest_v8b.log 2>&1
  echo "gates off rc=$?"; tail -n 3 gpurun_out/pytest_v8b.log | cut -c1-300
  exit $rc
fi
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_harness_gpu.py -m gpu -q -rf -x > gpurun_out/pytest_v8.log 2>&1
rc=$?; tail -n 12 gpurun_out/pytest_v8.log | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
for gte in 1 0; do
CLC_ACT_GATE=$gte timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_v8.json 2> gpurun_out/bench_v8.err || { echo "bench failed"; tail -5 gpurun_out/bench_v8.err; exit 3; }
python -c "import json; d=json.load(open('gpurun_out/bench_v8.json')); print('gates=$gte:', round(d['value'],2), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v8 -o r2 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-parity > $R/gpurun_out/prof_v8.log 2>&1 || { echo "prof failed"; tail -5 $R/gpurun_out/prof_v8.log; exit 4; }
python3 $R/tools/prof_db.py step $(find $R/gpurun_out/prof_v8 -name "*results.db" | head -1) > $R/gpurun_out/step_v8.txt 2>&1
head -30 $R/gpurun_out/step_v8.txt
