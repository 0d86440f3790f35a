"""clc_amd — MI355X-native engine for the CLC encode/decode + RD-training hot path."""


def set_precision(mode: str = "f32") -> str:
    """Arithmetic of the MFMA convolution kernels of the analysis / synthesis transforms and the reference encoder (maps larger than
    16x16): "f32" (default: exact f32 MFMA — the mode every parity bar and the headline benchmark are stated in) or "bf16"
    (opt-in reduced precision, the counterpart of the reference's --use-mixed-precision branch, /root/reference/train_CLC.py:143-174:
    operands rounded to bf16 at fragment read, f32 accumulation, f32 tensors in HBM).  The entropy-parameter networks on the 16x16
    latents, the likelihoods, the codec and the optimizer stay f32 in both modes.  Process-wide (one clc_set_tuning switch); set it
    BEFORE a TrainEngine captures its hipGraph.  Returns the previous mode."""
    from . import lib as _lib

    if mode not in ("f32", "bf16"):
        raise ValueError(f"precision must be 'f32' or 'bf16', got {mode!r}")
    old = _lib.load().clc_set_tuning(14, 1 if mode == "bf16" else 0)
    return "bf16" if old else "f32"


def get_precision() -> str:
    from . import lib as _lib

    L = _lib.load()
    old = L.clc_set_tuning(14, 0)
    L.clc_set_tuning(14, old)
    return "bf16" if old else "f32"
