"""The C-ABI library loads on a CPU-only box and exports every symbol include/clc_hip.h declares; the product path
refuses to run without a GPU (no CPU fallback) and never imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "clc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(clc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from clc_amd import lib

    L = lib.load()
    names = _declared_functions()
    assert len(names) >= 40, names
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/clc_hip.h but not exported by libclc_hip.so"
        assert n in lib.SIGNATURES, f"{n} has no ctypes signature in clc_amd/lib.py"
    assert set(lib.SIGNATURES) <= set(names), set(lib.SIGNATURES) - set(names)
    assert L.clc_version() >= 100


def test_gfx950_code_object_present():
    out = subprocess.run(["strings", "-a", os.path.join(ROOT, "clc_amd", "libclc_hip.so")], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from clc_amd import lib, models, ops

    m = models.TCM(N=64)
    with pytest.raises(lib.ClcError):
        m(torch.rand(1, 3, 256, 256))
    with pytest.raises(lib.ClcError):
        ops.conv2d(torch.rand(1, 64, 8, 8), torch.rand(64, 64, 3, 3))


def test_product_never_imports_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "clc_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "oracle/" in txt:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_missing_library_raises(tmp_path, monkeypatch):
    from clc_amd import lib

    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.ClcError):
        lib.load()
