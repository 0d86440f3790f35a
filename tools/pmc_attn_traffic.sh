#!/bin/bash
# Is winattn_fwd_mfma_kernel<8,false>'s FETCH_SIZE (x2 "gfx950 correction") real traffic?  Raw L2 memory-side request counters of the window-attention
# micro-benchmark: TCC_EA0_RDREQ (all read requests) and TCC_EA0_RDREQ_32B (the 32-byte ones) -> bytes = 32B x 32 + (all - 32B) x 64 per the counter
# definitions; FETCH_SIZE for comparison in a second pass.   bash tools/pmc_attn_traffic.sh
set -u -o pipefail
R=${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT}
O=$R/gpurun_out/pmc_attn
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_WRREQ[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $O/avail.txt
ATTN_4B=3 timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/a -o pmc -- python3 $R/tools/bench_attn.py 2 > $O/a.log 2>&1 || { echo "pass A failed"; tail -5 $O/a.log; }
ATTN_4B=3 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o pmc -- python3 $R/tools/bench_attn.py 2 > $O/f.log 2>&1 || { echo "pass F failed"; tail -5 $O/f.log; }
ATTN_4B=3 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o pmc -- python3 $R/tools/bench_attn.py 2 > $O/w.log 2>&1 || { echo "pass W failed"; tail -5 $O/w.log; }
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
def load(d):
    f = glob.glob(f"{O}/{d}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not f: return agg
    for r in csv.DictReader(open(f[0])):
        n = r["Kernel_Name"]
        if "winattn" in n:
            import re
            m = re.search(r"(winattn_\w+<[^>]*>)", n)
            key = (m.group(1) if m else n[:40], r["Grid_Size"])
            agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
a, f, w = load("a"), load("f"), load("w")
print("avail:", open(f"{O}/avail.txt").read())
for key in sorted(set(a) | set(f)):
    med = lambda xs: sorted(xs)[len(xs) // 2] if xs else float("nan")
    g = lambda c: med(a[key].get(c, []))
    rd, rd32, rd64, rd128 = g("TCC_EA0_RDREQ_sum"), g("TCC_EA0_RDREQ_32B_sum"), g("TCC_EA0_RDREQ_64B_sum"), g("TCC_EA0_RDREQ_128B_sum")
    fs, ws = med(f[key].get("FETCH_SIZE", [])), med(w[key].get("WRITE_SIZE", []))
    print(f"{key[0]:34s} grid {key[1]:>8s}: RDREQ {rd:10.0f} = 32B {rd32:9.0f} + 64B {rd64:9.0f} + 128B {rd128:9.0f} -> {(rd32 * 32 + rd64 * 64 + rd128 * 128) / 1e6:7.1f} MB by request size | "
          f"FETCH_SIZE {fs * 1024 / 1e6:7.1f} MB (x2: {2 * fs * 1024 / 1e6:7.1f}) | WRITE_SIZE {ws * 1024 / 1e6:7.1f} MB")
PY
