#!/bin/bash
set -u -o pipefail
# Collects the round's judged profiles on the GPU box (scratch under gpurun_out/r2prof; summaries are copied to profiles/ afterwards).
# usage: bash tools/gpu_profiles.sh <tag>
tag=${1:-r2}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
O=$R/gpurun_out/${tag}prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 3; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o $tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --no-reduced --no-reference-loop > $O/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $O/kt.log; exit 4; }
db=$(find $O/kt -name "*results.db" | head -1)
python3 $R/tools/prof_db.py stats $db --md $O/kernel_stats.md --csv $O/kernel_stats.csv > /dev/null || { echo "kernel stats summary failed"; exit 4; }
python3 $R/tools/prof_db.py step $db --top 200 > $O/step_timeline.txt || { echo "timeline summary failed"; exit 4; }
head -12 $O/step_timeline.txt
timeout -k 10 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmcA -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-reduced --no-reference-loop > $O/pmcA.log 2>&1 || { echo "pmc A failed"; tail -5 $O/pmcA.log; exit 5; }
python3 $R/tools/pmc_step.py $(find $O/pmcA -name "*counter_collection.csv" | head -1) $O/pmc_clock_mfma.md | tail -3 || { echo "pmc clock summary failed"; exit 5; }
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-reduced --no-reference-loop > $O/pmcF.log 2>&1 || { echo "pmc F failed"; tail -5 $O/pmcF.log; exit 6; }
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-reduced --no-reference-loop > $O/pmcW.log 2>&1 || { echo "pmc W failed"; tail -5 $O/pmcW.log; exit 7; }
python3 $R/tools/pmc_traffic.py $(find $O/pmcF -name "*counter_collection.csv" | head -1) $(find $O/pmcW -name "*counter_collection.csv" | head -1) $O/pmc_traffic.json > $O/pmc_traffic.txt || { echo "pmc traffic summary failed"; exit 7; }
head -12 $O/pmc_traffic.txt
find $O -type f \( -name "*.csv" -o -name "*.db" \) -size +8M -delete
ls -la $O
