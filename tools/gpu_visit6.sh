#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_harness_gpu.py -m gpu -q -rf -x -k "wgrad or gdn or backward or engine or train or split or two_phase" > gpurun_out/pytest_v6.log 2>&1
rc=$?; tail -n 8 gpurun_out/pytest_v6.log
if [ $rc -gt 1 ]; then exit $rc; fi
for g in 64; do
  CLC_WGRAD_GROUP=$g timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_v6_$g.json 2> gpurun_out/bench_v6_$g.err || { echo "bench failed"; tail -5 gpurun_out/bench_v6_$g.err; exit 3; }
  python -c "import json; d=json.load(open('gpurun_out/bench_v6_$g.json')); print('group $g:', round(d['value'],2), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_v6 -o r2 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-parity > $R/gpurun_out/prof_v6.log 2>&1 || { echo "prof failed"; tail -5 $R/gpurun_out/prof_v6.log; exit 4; }
python3 $R/tools/prof_db.py step $(find $R/gpurun_out/prof_v6 -name "*results.db" | head -1) > $R/gpurun_out/step_v6.txt 2>&1
grep -E "step wall|wgrad|fixup|compact" $R/gpurun_out/step_v6.txt | head -20
exit $rc
