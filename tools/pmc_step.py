#!/usr/bin/env python3
"""Per-kernel clock and MFMA utilisation of one hipGraph-replayed training step from a rocprofv3 PMC pass:
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d DIR -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-parity --no-reduced
    python tools/pmc_step.py DIR/.../pmc_counter_collection.csv [OUT.md]
clock        = GRBM_GUI_ACTIVE / 8 XCDs / duration          (MI355X_MICROARCH.md "DVFS give-back": reads high below ~0.3 ms)
MFMA busy %  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)   (fraction of the kernel's cycles its matrix pipes were busy)
Kernels whose average launch is shorter than 0.3 ms: GRBM_GUI_ACTIVE over-reads there (3-5 "GHz" on a 2.4 GHz part), and a busy % over the same
denominator is not evidence.  Their clock column is left empty and their busy % (marked ~) is taken over duration x the clock the LONG
kernels of the same step held (cycles / time summed over the launches of >= 0.3 ms).
(counter passes serialise and slow the kernels a little: never compare these durations with an un-profiled run)"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*>)?)\(", name)
    return (m.group(1) if m else name)[:90]


def main():
    path = sys.argv[1]
    disp = defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            d = disp[(int(r["Dispatch_Id"]))]
            d["name"], d["t0"], d["t1"] = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = sorted(disp.values(), key=lambda d: d["t0"])
    marks = [i for i, d in enumerate(rows) if "grad_sqnorm_kernel" in d["name"]]
    step = rows[marks[-2]:marks[-1]] if len(marks) >= 2 else rows
    agg = defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for d in step:
        a = agg[short(d["name"])]
        a[0] += 1
        a[1] += (d["t1"] - d["t0"]) * 1e-9
        a[2] += d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        a[3] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    tot_t = sum(a[1] for a in agg.values())
    tot_c = sum(a[2] for a in agg.values())
    tot_m = sum(a[3] for a in agg.values())
    SHORT = 0.3e-3
    long_ = [a for a in agg.values() if a[1] / a[0] >= SHORT]
    clock_ref = (sum(a[2] for a in long_) / sum(a[1] for a in long_)) if long_ else 2.4e9     # Hz held by the long kernels of this step
    lines = ["| kernel | launches | ms (profiled) | avg us | clock GHz | MFMA busy % |", "|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        if a[1] / a[0] >= SHORT:
            clock = f"{a[2] / a[1] / 1e9:.2f}"
            busy = f"{100.0 * a[3] / (a[2] * 1024):.1f}" if a[2] else "0.0"
        else:   # DVFS give-back: the counter's cycle count is not the kernel's
            clock = ""
            busy = f"~{100.0 * a[3] / (a[1] * clock_ref * 1024):.1f}" if a[1] else "0.0"
        lines.append(f"| `{k}` | {a[0]} | {a[1] * 1e3:.3f} | {a[1] / a[0] * 1e6:.1f} | {clock} | {busy} |")
    lines.append(f"| **whole step** | {len(step)} | {tot_t * 1e3:.3f} | | {clock_ref / 1e9:.2f} (launches >= 0.3 ms) | {100.0 * tot_m / (tot_t * clock_ref * 1024):.1f} (over time x that clock) |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(__doc__.split("\n")[0] + "\n\n" + out + "\n")


if __name__ == "__main__":
    main()
