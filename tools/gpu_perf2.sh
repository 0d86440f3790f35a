#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_harness_gpu.py tests/test_kernels_gpu.py -m gpu -q -rf -x > gpurun_out/pytest_p2.log 2>&1
rc=$?; tail -n 8 gpurun_out/pytest_p2.log
if [ $rc -gt 1 ]; then exit $rc; fi
for t in "" "2:0" "0:1" "0:1,1:0"; do
  CLC_TUNING=$t timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_p2_$t.json 2> gpurun_out/bench_p2_$t.err || { echo "bench $t failed"; tail -5 gpurun_out/bench_p2_$t.err; exit 3; }
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_p2_$t.json')); print('tuning [$t]', round(d['value'],2), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
cd /tmp && export TMPDIR=/tmp
for t in "" "0:1,1:0"; do
  tag=$(echo "p2_$t" | tr ':,' '__')
  CLC_TUNING=$t timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -o r2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1 || { echo "prof $t failed"; tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log; exit 4; }
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -name "*results.db" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/prof_db.py step $f > $GRAFT_REPO_ROOT/gpurun_out/step_$tag.txt 2>&1
  head -45 $GRAFT_REPO_ROOT/gpurun_out/step_$tag.txt
  find $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -type f ! -name "*results.db" -delete
done
exit $rc
