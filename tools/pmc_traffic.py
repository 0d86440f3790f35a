"""HBM-side traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md "HBM"):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dirF> -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE ...                                   -d <dirW>
    python tools/pmc_traffic.py <dirF>/pmc_counter_collection.csv <dirW>/pmc_counter_collection.csv profiles/r1_pmc_traffic.json
Counters are in KiB.  gfx950 correction applied as the guide prescribes: FETCH_SIZE counts 128-B requests at 64 B for the
16-B-per-lane streaming reads all these kernels use -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
The window is a full hipGraph-replayed training step in the trace (between two grad_sqnorm_kernel launches)."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def step_rows(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "grad_sqnorm_kernel" in r[2]]   # one per training step
    return rows[marks[-3]:marks[-2]]


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*>)?)\(", name)
    return m.group(1) if m else name


FAMILIES = {}   # (per-kernel rows only; kernel names are the ones rocprofv3 / bench.py's roofline use)


def main():
    fpath, wpath, out = sys.argv[1:4]
    per = defaultdict(lambda: {"launches": 0, "fetch_kib": 0.0, "write_kib": 0.0})
    for _s, _e, n, v in step_rows(fpath):
        k = short(n)
        per[k]["launches"] += 1
        per[k]["fetch_kib"] += v
    for _s, _e, n, v in step_rows(wpath):
        per[short(n)]["write_kib"] += v
    kernels = {}
    for k, a in sorted(per.items(), key=lambda kv: -(2 * kv[1]["fetch_kib"] + kv[1]["write_kib"])):
        kernels[k] = {"launches_per_step": a["launches"], "fetch_bytes_per_step": round(2 * a["fetch_kib"] * 1024),
                      "write_bytes_per_step": round(a["write_kib"] * 1024)}
    fam = {}
    for key, pat in FAMILIES.items():
        ks = [k for k in kernels if re.search(pat, k)]
        fam[key] = {"kernels": ks, "hbm_bytes_per_step": sum(kernels[k]["fetch_bytes_per_step"] + kernels[k]["write_bytes_per_step"] for k in ks),
                    "kernel_launches_per_step": sum(kernels[k]["launches_per_step"] for k in ks)}
    total = sum(v["fetch_bytes_per_step"] + v["write_bytes_per_step"] for v in kernels.values())
    json.dump({"note": "bytes at the L2's memory side (Infinity-Cache hits included), one training step (bs 8, 256x256, n_refs 1); "
                       "FETCH_SIZE doubled per the gfx950 correction", "csrc_sha16": __import__("bench").csrc_digest(), "total_bytes_per_step": total, "families": fam, "kernels": kernels},
              open(out, "w"), indent=1)
    print(f"total {total / 1e9:.2f} GB/step")
    for k, v in list(kernels.items())[:14]:
        print(f"  {k[:60]:60s} n={v['launches_per_step']:4d}  fetch {v['fetch_bytes_per_step'] / 1e6:9.1f} MB  write {v['write_bytes_per_step'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
