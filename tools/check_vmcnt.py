#!/usr/bin/env python3
"""Build-time check of the COUNTED `s_waitcnt vmcnt(N)` waits (run by clc_amd/csrc/Makefile on the freshly built objects).

Two kernels let younger vector-memory requests stay in flight while they wait for older ones, with a literal count:
  ru_fused_kernel<BWD>        (fused_ru.hip)   waits for the LDS-DMA halo tile with vmcnt(15) / vmcnt(12): "the tile has landed, the 12
                                               filter-fragment loads (+ 3 bias loads forward) issued behind it may still be in flight"
  conv_igemm_p1x1_kernel<OP>  (conv_igemm.hip) wait_vm(3 + 16 k | 3 + 32 k): 3 DMA pieces per K-step, 16 dword stores per epilogue
The vector-memory counter is in ISSUE order, so the count is right only while the compiler emits exactly those instructions between the
requests waited for and the wait.  If a compiler version drops, merges or scalarises one of them the wait UNDER-counts and a tile is read
before it landed — silent corruption, in the codec's context model too.  This script disassembles the gfx950 code object and fails the
build when the instruction stream no longer matches the count (fewer younger requests than N => under-wait => error)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
VMEM = re.compile(r"^\s*(buffer_(load|store|atomic)|global_(load|store|atomic)|flat_(load|store|atomic)|scratch_(load|store))")


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="clc_vmcnt_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not co:
            raise SystemExit(f"check_vmcnt: no gfx950 code object in {obj}")
        out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    kernels, name = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.strip():
            kernels[name].append(line.split("//")[0].rstrip())
    return kernels


def check_fused_ru(kernels):
    errs = []
    for tag, n in (("ILb0E", 15), ("ILb1E", 12)):
        ks = [k for k in kernels if "ru_fused_kernel" in k and tag in k]
        if len(ks) != 1:
            errs.append(f"ru_fused_kernel<{tag}> not found")
            continue
        ins = kernels[ks[0]]
        wait = next((i for i, l in enumerate(ins) if re.search(rf"s_waitcnt vmcnt\({n}\)", l)), None)
        if wait is None:
            errs.append(f"{ks[0]}: no s_waitcnt vmcnt({n}) — the counted tile wait is gone")
            continue
        dmas = [i for i, l in enumerate(ins[:wait]) if "buffer_load_dwordx4" in l and " lds" in l]
        want_dma = 4 if n == 15 else 8
        if len(dmas) != want_dma:
            errs.append(f"{ks[0]}: {len(dmas)} LDS-DMA tile requests before the wait, expected {want_dma}")
            continue
        younger = sum(1 for l in ins[dmas[-1] + 1: wait] if VMEM.match(l))
        if younger < n:
            errs.append(f"{ks[0]}: only {younger} vector-memory requests between the last tile DMA and s_waitcnt vmcnt({n}): the wait would "
                        f"return with tile pieces still in flight (stage 1 reads a tile that has not landed)")
        elif younger > n:
            print(f"check_vmcnt: note: {ks[0]}: {younger} younger requests, wait counts {n} (over-waits: safe, slower)")
    return errs


def check_p1x1(kernels):
    errs = []
    ks = [k for k in kernels if "conv_igemm_p1x1_kernel" in k]
    if not ks:
        return ["conv_igemm_p1x1_kernel not found"]
    for k in ks:
        ins = kernels[k]
        wide = [l for l in ins if re.match(r"^\s*buffer_store_dwordx[234]", l)]
        if wide:
            errs.append(f"{k}: {len(wide)} widened result stores (buffer_store_dwordx*): wait_vm counts 16 dword stores per epilogue")
        st = sum(1 for l in ins if re.match(r"^\s*buffer_store_dword\b", l))
        if st == 0 or st % 16:
            errs.append(f"{k}: {st} buffer_store_dword instructions, expected a multiple of 16 (16 per epilogue path)")
        other = sum(1 for l in ins if re.match(r"^\s*(global|flat)_store", l))
        if other > 1:   # (one is the never-taken sentinel store of the CLC_TUNE_ABLATE timing diagnostic)
            errs.append(f"{k}: {other} result stores outside the SRD buffer path")
    return errs


def main():
    errs = []
    for obj in sys.argv[1:]:
        kernels = disassemble(obj)
        if "fused_ru" in os.path.basename(obj):
            errs += check_fused_ru(kernels)
        if "conv_igemm" in os.path.basename(obj):
            errs += check_p1x1(kernels)
    for e in errs:
        print("check_vmcnt: ERROR:", e, file=sys.stderr)
    if errs:
        raise SystemExit(1)
    print("check_vmcnt: counted vmcnt waits match the instruction stream:", ", ".join(os.path.basename(o) for o in sys.argv[1:]))


if __name__ == "__main__":
    main()
