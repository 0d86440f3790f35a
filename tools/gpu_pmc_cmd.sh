#!/bin/bash
# rocprofv3 PMC pass (clock + MFMA-busy per kernel) of an arbitrary python command:  bash tools/gpu_pmc_cmd.sh <tag> <python args...>
# (counters in their own run, kernel-trace only: the pool refuses --pmc together with the runtime trace domains)
set -u -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
tag=$1; shift
O=$R/gpurun_out/pmc_$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 ${LIMIT:-300} rocprofv3 --pmc ${PMC:-GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES} --kernel-trace --output-format csv -d $O/pmc -o pmc -- python3 "$@" > $O/run.log 2>&1 || { echo "pmc run failed"; tail -8 $O/run.log; exit 3; }
cd $R && python3 tools/pmc_step.py $(find $O/pmc -name "*counter_collection.csv" | head -1) $O/clock_mfma.md | head -${TOP:-14} | cut -c1-150
find $O -type f \( -name "*.csv" -o -name "*.db" \) -size +8M -delete
