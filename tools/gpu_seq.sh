#!/bin/bash
set -u -o pipefail
# one hipGraph-replayed step as a dispatch sequence (gpurun_out/seq.txt)
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}
mkdir -p $R/gpurun_out/seq
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $R/gpurun_out/seq/kt -o seq -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-parity --no-reduced --no-reference-loop > $R/gpurun_out/seq/kt.log 2>&1 || { echo "trace failed"; tail -5 $R/gpurun_out/seq/kt.log; exit 3; }
db=$(find $R/gpurun_out/seq/kt -name "*results.db" | head -1)
python3 $R/tools/prof_db.py seq $db > $R/gpurun_out/seq.txt
tail -1 $R/gpurun_out/seq.txt
find $R/gpurun_out/seq -type f -size +8M -delete; rm -rf $R/gpurun_out/seq/kt
