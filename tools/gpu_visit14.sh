#!/bin/bash
# kernel + model tests after the pinned-scalar / KS-specialised kernels, then the step rate and the per-kernel table
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_harness_gpu.py -m gpu -x -q > gpurun_out/pytest_v14.log 2>&1 || { echo "tests failed"; tail -30 gpurun_out/pytest_v14.log; exit 3; }
tail -3 gpurun_out/pytest_v14.log
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > gpurun_out/bench_v14.json 2> gpurun_out/bench_v14.err || { echo "bench failed"; tail -5 gpurun_out/bench_v14.err; exit 4; }
python -c "import json; d=json.load(open('gpurun_out/bench_v14.json')); print(round(d['value'],2), 'img/s', round(d['ms_per_step'],3), 'ms')"
done
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/v14
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o v14 -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > $O/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $O/kt.log; exit 5; }
db=$(find $O/kt -name "*results.db" | head -1)
python3 $R/tools/prof_db.py stats $db --md $O/kernel_stats.md --csv $O/kernel_stats.csv > /dev/null
head -30 $O/kernel_stats.md
find $O -type f \( -name "*.csv" -o -name "*.db" \) -size +8M -delete
