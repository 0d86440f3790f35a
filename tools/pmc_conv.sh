#!/bin/bash
set -u -o pipefail
# PMC passes over the conv micro-benchmark (one shape): wave-cycle breakdown, MFMA busy, clock, LDS conflicts.
# usage (GPU box): bash tools/pmc_conv.sh <shape-name> <outdir>
set -e
shape=${1:-c128_128_3x3_128}; out=${2:-gpurun_out/pmc_conv}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out/a -o p -- python3 tools/bench_conv.py 3 $shape > $out/a.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $out/b -o p -- python3 tools/bench_conv.py 3 $shape > $out/b.log 2>&1
python3 - "$out" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
for sub in ("a", "b"):
    f = glob.glob(f"{out}/{sub}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv for", sub); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "conv_" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, c in agg.items():
        n = len(c["dur_ns"]) // max(1, len([x for x in c if x != "dur_ns"]))
        print(k, "dispatches", n)
        for name, v in c.items():
            print(f"   {name:28s} avg {sum(v)/len(v):16.1f}")
PY
