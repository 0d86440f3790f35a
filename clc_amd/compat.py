"""Import-surface shim: lets the reference's scripts resolve every name they import to this package, unmodified.

`train_CLC.py:17-26` and `eval_CLC.py:1-17` import `models` (TCM, CLC), `compressai.datasets.ImageFolder`,
`compressai.zoo.models`, `pytorch_msssim.ms_ssim`; `models/CLC_run.py:1-20` (if it is imported at all) imports
`compressai.{entropy_models,ans,models,layers}` and `timm.models.layers`.  `install()` registers modules of those names in
`sys.modules`, each backed by the HIP implementation of this package, so

    python -c "import clc_amd.compat as c; c.install(); import runpy; runpy.run_path('train_CLC.py', run_name='__main__')" ...

(or a two-line `sitecustomize.py`: `import clc_amd.compat; clc_amd.compat.install()`) runs the reference loop on the MI355X
engine.  What this shim does NOT supply are the reference's non-path dependencies (torchvision, tensorboard, its
`dataloader_ref_cluster` — the retrieval pipeline of SURVEY.md §8(f)-2); `clc_amd.retrieval` covers the latter's numeric core.
Genuinely installed third-party packages are left alone unless `force=True`; `models` is always ours.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types


def _missing(name: str) -> bool:
    try:
        return importlib.util.find_spec(name) is None
    except (ImportError, ValueError, AttributeError):
        return True


def _module(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__clc_amd_shim__ = True
    sys.modules[name] = m
    return m


class ImageFolder:
    """compressai.datasets.ImageFolder surface (root/<split>/*.png|jpg, optional transform) — imported by train_CLC.py:21, never used
    there.  Decodes with PIL into float CHW tensors when no transform is given."""

    def __init__(self, root, transform=None, split="train"):
        splitdir = os.path.join(str(root), split)
        if not os.path.isdir(splitdir):
            raise RuntimeError(f'Missing directory "{splitdir}"')
        self.samples = sorted(os.path.join(splitdir, f) for f in os.listdir(splitdir) if os.path.isfile(os.path.join(splitdir, f)))
        self.transform = transform

    def __getitem__(self, index):
        import numpy as np
        import torch
        from PIL import Image

        img = Image.open(self.samples[index]).convert("RGB")
        if self.transform:
            return self.transform(img)
        return torch.from_numpy(np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0)

    def __len__(self):
        return len(self.samples)


def install(force: bool = False):
    """Register the shim modules; returns the list of module names that now resolve to this package."""
    import torch

    from . import ans, entropy_models, layers, models, train
    from .models import clc as _clc

    done = []

    def put(name, **attrs):
        if force or _missing(name) or getattr(sys.modules.get(name), "__clc_amd_shim__", False):
            _module(name, **attrs)
            done.append(name)
            return True
        return False

    sys.modules["models"] = models                      # `from models import TCM, CLC` (train_CLC.py:25, eval_CLC.py:4, eval.py:4)
    done.append("models")
    if put("compressai"):
        sys.modules["compressai.entropy_models"] = entropy_models
        sys.modules["compressai.ans"] = ans
        sys.modules["compressai.layers"] = layers
        done += ["compressai.entropy_models", "compressai.ans", "compressai.layers"]
        _module("compressai.models", CompressionModel=_clc.CompressionModel)
        _module("compressai.datasets", ImageFolder=ImageFolder)
        _module("compressai.zoo", models={"clc": models.CLC, "tcm": models.TCM})
        done += ["compressai.models", "compressai.datasets", "compressai.zoo"]
        c = sys.modules["compressai"]
        for sub in ("entropy_models", "ans", "layers", "models", "datasets", "zoo"):
            setattr(c, sub, sys.modules["compressai." + sub])

    class DropPath(torch.nn.Identity):
        """timm.models.layers.DropPath at drop_prob = 0 (the reference always passes 0, CLC_run.py:329-351)."""

        def __init__(self, drop_prob=0.0, *a, **k):
            super().__init__()
            if drop_prob:
                raise ValueError("DropPath(p > 0) is not supported (the reference never uses it)")

    if put("timm"):
        tm = _module("timm.models")
        tl = _module("timm.models.layers", trunc_normal_=torch.nn.init.trunc_normal_, DropPath=DropPath)
        sys.modules["timm"].models = tm
        tm.layers = tl
        done += ["timm.models", "timm.models.layers"]
    put("pytorch_msssim", ms_ssim=train.ms_ssim)   # train_CLC.py:23,33-34 / eval_CLC.py:16: ms_ssim(a, b, data_range=1.)
    return done
