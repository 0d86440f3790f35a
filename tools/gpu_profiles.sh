#!/bin/bash
# Collects the round's judged profiles on the GPU box (scratch under gpurun_out/r2prof; summaries are copied to profiles/ afterwards).
# usage: bash tools/gpu_profiles.sh <tag>
tag=${1:-r2}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${tag}prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 3; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o $tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-parity > $O/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $O/kt.log; exit 4; }
db=$(find $O/kt -name "*results.db" | head -1)
python3 $R/tools/prof_db.py stats $db --md $O/kernel_stats.md --csv $O/kernel_stats.csv > /dev/null
python3 $R/tools/prof_db.py step $db > $O/step_timeline.txt
head -12 $O/step_timeline.txt
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmcA -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $O/pmcA.log 2>&1 || { echo "pmc A failed"; tail -5 $O/pmcA.log; exit 5; }
python3 $R/tools/pmc_step.py $(find $O/pmcA -name "*counter_collection.csv" | head -1) $O/pmc_clock_mfma.md | tail -3
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $O/pmcF.log 2>&1 || { echo "pmc F failed"; tail -5 $O/pmcF.log; exit 6; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity > $O/pmcW.log 2>&1 || { echo "pmc W failed"; tail -5 $O/pmcW.log; exit 7; }
python3 $R/tools/pmc_traffic.py $(find $O/pmcF -name "*counter_collection.csv" | head -1) $(find $O/pmcW -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json | head -12
find $O -type f \( -name "*.csv" -o -name "*.db" \) -size +8M -delete
ls -la $O
