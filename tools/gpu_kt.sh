#!/bin/bash
# kernel-trace summary of a few steps (scratch under gpurun_out/<tag>kt): per-kernel totals and the step timeline
set -u -o pipefail
tag=${1:-kt}
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
O=$R/gpurun_out/${tag}kt
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/kt -o $tag -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --no-reduced --no-reference-loop > $O/kt.log 2>&1 || { echo "kernel-trace failed"; tail -5 $O/kt.log; exit 4; }
db=$(find $O/kt -name "*results.db" | head -1)
python3 $R/tools/prof_db.py stats $db --md $O/kernel_stats.md --csv $O/kernel_stats.csv > /dev/null || { echo "kernel stats summary failed"; exit 4; }
python3 $R/tools/prof_db.py step $db --top 120 > $O/step_timeline.txt || { echo "timeline summary failed"; exit 4; }
python3 $R/tools/prof_db.py seq $db > $O/step_seq.txt || { echo "sequence dump failed"; exit 4; }
head -12 $O/step_timeline.txt
find $O -type f \( -name "*.csv" -o -name "*.db" \) -size +8M -delete
