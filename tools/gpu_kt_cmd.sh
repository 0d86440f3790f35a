#!/bin/bash
# rocprofv3 kernel-trace of an arbitrary python command on the GPU box, per-kernel table to stdout:
#   bash tools/gpu_kt_cmd.sh <tag> <python args...>      e.g.  bash tools/gpu_kt_cmd.sh mlp tools/bench_mlp.py 12
set -u -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
tag=$1; shift
O=$R/gpurun_out/kt_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${LIMIT:-300} rocprofv3 --kernel-trace --stats -d $O/kt -o $tag -- python3 "$@" > $O/run.log 2>&1 || { echo "profiled run failed"; tail -8 $O/run.log; exit 3; }
db=$(find $O/kt -name "*results.db" | head -1)
cd $R && python3 tools/prof_db.py stats $db --md $O/kernel_stats.md --top ${TOP:-25} | cut -c1-150
grep -v "^\[" $O/run.log | tail -${TAIL:-6}
find $O -type f \( -name "*.csv" -o -name "*.db" \) -size +8M -delete
