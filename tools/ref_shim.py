"""Import the genuine reference model classes in the BUILD CONTAINER ONLY.

The reference (/root/reference, read-only, never shipped) cannot be imported as is:
its third-party leaves (compressai, timm) are not installed.  This shim registers the
oracle's restated leaves (oracle/leaves.py, oracle/rans_py.py) under those module names
and then imports /root/reference/models, so that the reference's OWN graph code
(CLC/TCM __init__/forward/compress/decompress, WMSA, Block, ConvTransBlock, SWAtten ...)
runs over them.  That pins the oracle's *wiring* against genuine reference code; it does
not pin the leaf arithmetic.

Used by tools/make_golden.py and by tests that are skipped when /root/reference is absent.
Nothing here is imported by the product or runs on the GPU box.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_fake_third_party():
    import torch
    from oracle import leaves, rans_py

    _mod("compressai")
    _mod("compressai.entropy_models", EntropyBottleneck=leaves.EntropyBottleneck,
         GaussianConditional=leaves.GaussianConditional)
    _mod("compressai.ans", BufferedRansEncoder=rans_py.BufferedRansEncoder, RansDecoder=rans_py.RansDecoder,
         RansEncoder=rans_py.RansEncoder)
    _mod("compressai.models", CompressionModel=leaves.CompressionModel)
    _mod("compressai.layers", AttentionBlock=leaves.AttentionBlock, ResidualBlock=leaves.ResidualBlock,
         ResidualBlockUpsample=leaves.ResidualBlockUpsample, ResidualBlockWithStride=leaves.ResidualBlockWithStride,
         conv3x3=leaves.conv3x3, subpel_conv3x3=leaves.subpel_conv3x3, GDN=leaves.GDN)

    class DropPath(torch.nn.Identity):
        def __init__(self, p=0.0):
            super().__init__()

    _mod("timm")
    _mod("timm.models")
    _mod("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_, DropPath=DropPath)


def import_reference_models():
    """Returns the reference's `models` package (TCM from tcm.py, CLC from CLC_run.py)."""
    if not available():
        raise RuntimeError("reference tree not present")
    sys.dont_write_bytecode = True  # never drop .pyc files into the read-only reference tree
    install_fake_third_party()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    return importlib.import_module("models")


def import_reference_file(relpath: str, name: str):
    """Import one reference file by path, bypassing models/__init__ (e.g. models/CLM.py)."""
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def extract_reference_functions(relpath: str, names, extra_ns=None):
    """AST-extract plain functions / classes from a reference file that cannot be imported as a module
    (models/Patch_Matching.py: `from turtle import shape`, cv2, compressai_local ...; train_CLC.py / eval_CLC.py: tensorboard, torchvision,
    the dataset modules) and exec them with torch/np/F/nn/time/math/optim in scope (+ extra_ns, e.g. the `ms_ssim` the file imports from
    pytorch_msssim).  `.cuda()` is patched to the identity by the caller (see tools/make_golden.py)."""
    import ast
    import math
    import time

    import numpy as np
    import torch
    import torch.nn as nn
    import torch.nn.functional as F

    src = open(os.path.join(REFERENCE_ROOT, relpath)).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "np": np, "nn": nn, "F": F, "time": time, "math": math, "optim": torch.optim}
    ns.update(extra_ns or {})
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.ClassDef)) and node.name in names:
            code = compile(ast.Module(body=[node], type_ignores=[]), relpath, "exec")
            exec(code, ns)
    missing = [n for n in names if n not in ns]
    if missing:
        raise RuntimeError(f"functions not found in {relpath}: {missing}")
    return ns
