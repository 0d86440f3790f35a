#!/usr/bin/env python3
"""Wave-cycle breakdown per kernel of one training step from a rocprofv3 PMC pass:
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE ...
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE ...
usage: python tools/pmc_waits.py CSV [CSV2] [filter-substring ...]"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(?:<[^(]*>)?)\(", name)
    return (m.group(1) if m else name)[:70]


def load(path):
    disp = defaultdict(dict)
    for r in csv.DictReader(open(path)):
        d = disp[int(r["Dispatch_Id"])]
        d["name"], d["t0"], d["t1"] = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = sorted(disp.values(), key=lambda d: d["t0"])
    marks = [i for i, d in enumerate(rows) if "grad_sqnorm_kernel" in d["name"]]
    return rows[marks[-2]:marks[-1]] if len(marks) >= 2 else rows


def main():
    files = [a for a in sys.argv[1:] if a.endswith(".csv")]
    filt = [a for a in sys.argv[1:] if not a.endswith(".csv")]
    agg = defaultdict(lambda: defaultdict(float))
    for f in files:
        for d in load(f):
            k = short(d["name"])
            if filt and not any(s in k for s in filt):
                continue
            a = agg[k]
            a["n@" + f] += 1
            a["ms@" + f] += (d["t1"] - d["t0"]) * 1e-6
            for c, v in d.items():
                if c not in ("name", "t0", "t1"):
                    a[c] += v
    for k, a in sorted(agg.items(), key=lambda kv: -max(v for c, v in kv[1].items() if c.startswith("ms@"))):
        ms = max(v for c, v in a.items() if c.startswith("ms@"))
        if ms < 0.25:
            continue
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        print(f"{k}: {ms:.3f} ms")
        if wc:
            print("   of wave cycles: wait_any %.1f %%  wait_inst_any %.1f %%  active_inst_any %.1f %%  active_inst_lds %.1f %%" % tuple(
                100 * a.get(c, 0.0) / wc for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS")))
        if a.get("SQ_LDS_IDX_ACTIVE"):
            print("   LDS bank-conflict cycles / LDS active cycles: %.1f %%   valu insts per wave %.0f" % (
                100 * a["SQ_LDS_BANK_CONFLICT"] / a["SQ_LDS_IDX_ACTIVE"], a.get("SQ_INSTS_VALU", 0) / max(a.get("SQ_WAVES", 1), 1)))


if __name__ == "__main__":
    main()
