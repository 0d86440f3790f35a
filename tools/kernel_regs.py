#!/usr/bin/env python3
"""Register / scratch / LDS metadata of the gfx950 kernels in a built object: python tools/kernel_regs.py clc_amd/csrc/fused_mlp.o [name-filter]
(reads the AMDGPU metadata note of the code object; a kernel with private_segment_fixed_size > 0 spills)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    obj, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    tmp = tempfile.mkdtemp(prefix="clc_regs_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f][0]
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    blocks = re.split(r"\n\s+- \.agpr_count:", notes)
    for b in blocks[1:]:
        b = ".agpr_count:" + b
        f = dict(re.findall(r"\.(\w+):\s+(\S+)", b))
        n = f.get("name", "")
        if flt not in n:
            continue
        dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
        print(f"{dem[:100]:100s} vgpr {f.get('vgpr_count', '?'):>4s} agpr {f.get('agpr_count', '?'):>4s} sgpr {f.get('sgpr_count', '?'):>4s} "
              f"scratch {f.get('private_segment_fixed_size', '?'):>5s} lds {f.get('group_segment_fixed_size', '?'):>6s} spill v{f.get('vgpr_spill_count', '?')} s{f.get('sgpr_spill_count', '?')}")


if __name__ == "__main__":
    main()
