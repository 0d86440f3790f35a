#!/bin/bash
set -u -o pipefail
# wave-cycle breakdown (wait / issue / LDS) per kernel of one replayed step: two PMC passes, summarised by tools/pmc_waits.py
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}; O=$R/gpurun_out/pmcw; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/a -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-reduced --no-reference-loop > $O/a.log 2>&1 || { echo "pmc a failed"; tail -5 $O/a.log; exit 5; }
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-reduced --no-reference-loop > $O/b.log 2>&1 || { echo "pmc b failed"; tail -5 $O/b.log; exit 6; }
python3 $R/tools/pmc_waits.py $(find $O/a -name "*counter_collection.csv") $(find $O/b -name "*counter_collection.csv") > $O/waits.txt
cat $O/waits.txt
find $O -name "*.csv" -size +6M -delete
