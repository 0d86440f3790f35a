#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -rf -x -k "wgrad or conv_fwd" > gpurun_out/pytest_p3.log 2>&1
rc=$?; tail -n 5 gpurun_out/pytest_p3.log
if [ $rc -gt 1 ]; then exit $rc; fi
for t in "" "1:0"; do
  CLC_TUNING=$t timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_p3_$t.json 2> gpurun_out/bench_p3_$t.err || { echo "bench $t failed"; tail -5 gpurun_out/bench_p3_$t.err; exit 3; }
  python -c "import json,sys; d=json.load(open('gpurun_out/bench_p3_$t.json')); print('tuning [$t]', round(d['value'],2), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_p3 -o r2 -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof_p3.log 2>&1 || { echo "prof failed"; tail -5 $R/gpurun_out/prof_p3.log; exit 4; }
f=$(find $R/gpurun_out/prof_p3 -name "*results.db" | head -1)
python3 $R/tools/prof_db.py step $f > $R/gpurun_out/step_p3.txt 2>&1
head -30 $R/gpurun_out/step_p3.txt
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_p3 -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc_p3.log 2>&1 || { echo "pmc failed"; tail -5 $R/gpurun_out/pmc_p3.log; exit 5; }
c=$(find $R/gpurun_out/pmc_p3 -name "*counter_collection.csv" | head -1)
python3 $R/tools/pmc_step.py $c $R/gpurun_out/pmc_step_p3.md | head -40
find $R/gpurun_out/pmc_p3 $R/gpurun_out/prof_p3 -type f -size +20M -delete
exit $rc
