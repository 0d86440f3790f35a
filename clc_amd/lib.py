"""ctypes binding of libclc_hip.so (the C ABI declared in include/clc_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a kernel call
fails, this module raises — loudly — instead of routing anywhere else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLC_LIB_PATH") or os.path.join(_HERE, "libclc_hip.so")   # (CLC_LIB_PATH: A/B of two builds on one GPU box)

ACT_NONE, ACT_LRELU, ACT_RELU, ACT_GELU, ACT_HALFTANH, ACT_SIGMOID, ACT_SAVED_DERIV = 0, 1, 2, 3, 4, 5, 6
IN_NONE, IN_SQUARE = 0, 1
NORM_NONE, NORM_GDN, NORM_IGDN, NORM_MUL2 = 0, 1, 2, 3

fp = C.c_void_p  # device / host pointers are passed as integers (tensor.data_ptr())


class ConvDesc(C.Structure):
    _fields_ = [("x", fp), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("ldx", C.c_int),
                ("w", fp), ("bias", fp),
                ("y", fp), ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int), ("ldy", C.c_int),
                ("ks", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("transposed", C.c_int), ("in_op", C.c_int), ("act", C.c_int),
                ("norm", C.c_int), ("mul", fp), ("ldm", C.c_int),
                ("res", fp), ("ldr", C.c_int), ("res_scale", C.c_float),
                ("y_pre", fp), ("ldp", C.c_int),
                ("shuffle", C.c_int), ("res_first", C.c_int),
                ("xs", fp), ("ldxs", C.c_int), ("xs_act", C.c_int), ("xs_pre", C.c_int),
                ("w2", fp), ("bias2", fp), ("pre_deriv", C.c_int),
                ("res_gate", fp), ("ldg", C.c_int), ("res_gate_act", C.c_int), ("res_gate_pre", C.c_int),
                ("out_gate", fp), ("ldog", C.c_int), ("out_gate_act", C.c_int), ("out_gate_pre", C.c_int),
                ("w3", fp), ("bias3", fp), ("w4", fp), ("bias4", fp),
                ("workspace", fp), ("workspace_bytes", C.c_size_t), ("batch_variant_ok", C.c_int), ("w_packed", fp), ("w_wino", fp)]


class WgradDesc(C.Structure):
    _fields_ = [("x", fp), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("ldx", C.c_int),
                ("dy", fp), ("OH", C.c_int), ("OW", C.c_int), ("Cout", C.c_int), ("lddy", C.c_int),
                ("dw", fp), ("dbias", fp),
                ("ks", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("in_op", C.c_int), ("accumulate", C.c_int),
                ("workspace", fp), ("workspace_bytes", C.c_size_t),
                ("dys", fp), ("lddys", C.c_int), ("dys_act", C.c_int), ("dys_pre", C.c_int)]


class RUDesc(C.Structure):
    _fields_ = [("x", fp), ("ldx", C.c_int), ("t1", fp), ("t2", fp), ("y", fp),
                ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("sets", C.c_int),
                ("w1", fp * 4), ("b1", fp * 4), ("w2", fp * 4), ("b2", fp * 4), ("w3", fp * 4), ("b3", fp * 4),
                ("saved_y", fp), ("saved_t2", fp), ("saved_t1", fp)]


class MlpDesc(C.Structure):
    _fields_ = [("x", fp), ("ldx", C.c_int), ("w1", fp), ("b1", fp), ("w2", fp), ("b2", fp), ("res", fp), ("ldr", C.c_int),
                ("y", fp), ("ldy", C.c_int), ("M", C.c_long), ("Cin", C.c_int), ("Chid", C.c_int), ("Cout", C.c_int),
                ("dy", fp), ("lddy", C.c_int), ("w2t", fp), ("dx", fp), ("lddx", C.c_int), ("dh", fp), ("g", fp), ("h", fp),
                ("ln_gamma", fp), ("ln_beta", fp), ("ln_out", fp), ("ln_ws", fp)]


class LnLinDesc(C.Structure):
    _fields_ = [("x", fp), ("ldx", C.c_int), ("ln_gamma", fp), ("ln_beta", fp), ("w", fp), ("b", fp), ("y", fp), ("ln_out", fp),
                ("M", C.c_long), ("Cin", C.c_int), ("Cout", C.c_int), ("dy", fp), ("wt", fp), ("dx", fp), ("lddx", C.c_int),
                ("dadd", fp), ("ldadd", C.c_int), ("ln_ws", fp)]


class GDNEntry(C.Structure):
    _fields_ = [("gamma", fp), ("beta", fp), ("gamma_eff", fp), ("gamma_eff_t", fp), ("beta_eff", fp), ("C", C.c_int), ("first_block", C.c_int),
                ("gamma_bound", C.c_float), ("beta_bound", C.c_float), ("pedestal", C.c_float)]


class TransposeEntry(C.Structure):
    _fields_ = [("w", fp), ("wt", fp), ("Cout", C.c_int), ("T", C.c_int), ("Cin", C.c_int), ("tile_begin", C.c_int)]


class WinoEntry(C.Structure):
    _fields_ = [("w", fp), ("out", fp), ("N", C.c_int), ("K", C.c_int), ("flip", C.c_int), ("block_begin", C.c_int)]


class HaloPackEntry(C.Structure):
    _fields_ = [("w", fp), ("out", fp), ("N", C.c_int), ("K", C.c_int), ("block_begin", C.c_int)]


class ReduceEntry(C.Structure):
    _fields_ = [("partial", fp), ("nblocks", C.c_int), ("n", C.c_int), ("out0", fp), ("out1", fp), ("split", C.c_int), ("accumulate", C.c_int)]


class ParamEntry(C.Structure):
    _fields_ = [("p", fp), ("g", fp), ("m", fp), ("v", fp), ("n", C.c_long)]


_i, _l, _f, _d, _sz = C.c_int, C.c_long, C.c_float, C.c_double, C.c_size_t
_pp = C.POINTER(fp)

# name -> (restype, argtypes); every symbol include/clc_hip.h declares must appear here
SIGNATURES = {
    "clc_last_error": (C.c_char_p, []),
    "clc_version": (_i, []),
    "clc_set_tuning": (_i, [_i, _i]),
    "clc_get_tuning": (_i, [_i]),
    "clc_kernel_config_tag": (_i, []),
    "clc_kernel_config_hash": (C.c_uint, []),
    "clc_conv2d": (_i, [C.POINTER(ConvDesc), fp]),
    "clc_conv2d_workspace_bytes": (_sz, [C.POINTER(ConvDesc)]),
    "clc_conv2d_wgrad_workspace_bytes": (_sz, [C.POINTER(WgradDesc)]),
    "clc_conv2d_wgrad": (_i, [C.POINTER(WgradDesc), fp]),
    "clc_conv2d_wgrad_batched": (_i, [C.POINTER(WgradDesc), _i, fp]),
    "clc_conv2d_wgrad_group_workspace_bytes": (_sz, []),
    "clc_conv2d_wgrad_sk_workspace_bytes": (_sz, [C.POINTER(WgradDesc)]),
    "clc_conv2d_wgrad_variant": (_i, [C.POINTER(WgradDesc)]),
    "clc_conv2d_wgrad_batched_sk": (_i, [C.POINTER(WgradDesc), _i, fp, _sz, fp]),
    "clc_filter_transpose": (_i, [fp, fp, _i, _i, _i, fp]),
    "clc_filter_pack_halo": (_i, [fp, fp, _i, _i, fp]),
    "clc_filter_pack_halo_batched": (_i, [fp, _i, _i, fp]),
    "clc_filter_wino": (_i, [fp, fp, _i, _i, _i, fp]),
    "clc_filter_wino_batched": (_i, [fp, _i, _i, fp]),
    "clc_filter_transpose_batched": (_i, [fp, _i, _i, fp]),
    "clc_partial_reduce_batched": (_i, [C.POINTER(ReduceEntry), _i, fp]),
    "clc_act_bwd": (_i, [fp, _i, fp, _i, _i, _i, fp, _i, _l, _i, fp]),
    "clc_colsum_workspace_bytes": (_sz, [_l, _i]),
    "clc_colsum": (_i, [fp, _i, _l, _i, fp, _i, fp, _sz, fp]),
    "clc_layernorm_fwd": (_i, [fp, _i, fp, fp, fp, _i, fp, fp, _l, _i, fp]),
    "clc_layernorm_bwd_workspace_bytes": (_sz, [_l, _i]),
    "clc_layernorm_bwd_blocks": (_i, [_l, _i]),
    "clc_layernorm_fwd_pair": (_i, [fp, _i, fp, fp, fp, fp, _l, fp, _i, fp, fp, _l, _i, fp]),
    "clc_layernorm_bwd_pair": (_i, [fp, _i, fp, _i, fp, fp, _l, fp, fp, fp, _i, fp, _i, fp, fp, fp, fp, _i, _l, _i, fp, _sz, fp]),
    "clc_layernorm_bwd": (_i, [fp, _i, fp, _i, fp, fp, fp, fp, _i, fp, _i, fp, fp, _i, _l, _i, fp, _sz, fp]),
    "clc_gdn_bwd_elem": (_i, [fp, fp, fp, fp, fp, _l, _i, fp]),
    "clc_gdn_bwd_combine": (_i, [fp, fp, fp, fp, _l, fp]),
    "clc_gdn_reparam_fwd": (_i, [fp, fp, _i, _f, _f, _f, fp, fp, fp, fp]),
    "clc_gdn_reparam_fwd_batched": (_i, [fp, _i, _i, fp]),
    "clc_gdn_reparam_bwd": (_i, [fp, fp, _i, _f, _f, fp, fp, fp, fp, _i, fp]),
    "clc_unshuffle_act_bwd": (_i, [fp, _i, fp, _i, _i, _i, fp, _i, _i, _i, _i, fp]),
    "clc_gate_fwd": (_i, [fp, fp, fp, fp, _l, fp]),
    "clc_gate_bwd": (_i, [fp, fp, fp, fp, fp, _l, fp]),
    "clc_axpby": (_i, [fp, _f, fp, _f, fp, _l, fp]),
    "clc_sum_n": (_i, [_pp, C.POINTER(C.c_int), _i, fp, _i, _l, _i, fp]),
    "clc_copy2d": (_i, [fp, _i, fp, _i, _l, _i, fp]),
    "clc_stem_pack": (_i, [fp, fp, fp, fp, _i, _i, fp]),
    "clc_stem_unpack_add": (_i, [fp, fp, fp, fp, _i, _i, fp]),
    "clc_im2col_small": (_i, [fp, _i, _i, _i, _i, _i, _i, _i, _i, fp, _i, _i, _i, fp]),
    "clc_winattn_fwd": (_i, [fp, _i, fp, fp, _i, fp, _i, _i, _i, _i, _i, _i, _i, fp]),
    "clc_winattn_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "clc_winattn_bwd_blocks": (_i, [_i, _i, _i, _i, _i, _i]),
    "clc_winattn_fwd_pair": (_i, [fp, _i, fp, fp, fp, _i, fp, _i, _i, _i, _i, _i, _i, _i, fp]),
    "clc_winattn_bwd_pair": (_i, [fp, _i, fp, _i, fp, fp, fp, _i, fp, fp, _i, fp, fp, _i, _i, _i, _i, _i, _i, _i, _i, fp, _sz, fp]),
    "clc_winattn_bwd": (_i, [fp, _i, fp, _i, fp, fp, _i, fp, fp, _i, fp, _i, _i, _i, _i, _i, _i, _i, _i, fp, _sz, fp]),
    "clc_gauss_lik_partials": (_i, [_l, _i]),
    "clc_gauss_lik_fwd": (_i, [fp, _i, fp, _i, fp, _i, fp, _i, fp, _i, fp, _i, _l, _i, _i, fp, _i, fp]),
    "clc_gauss_lik_bwd": (_i, [fp, _i, fp, _i, fp, _i, fp, _i, fp, _i, fp, _i, fp, _i, fp, _i, _l, _i, _i, fp, _i, fp]),
    "clc_eb_lik_fwd": (_i, [fp, _i, fp, _i, fp, _pp, _pp, _pp, fp, _i, fp, _i, _l, _i, _i, fp]),
    "clc_eb_lik_bwd": (_i, [fp, _i, fp, _i, fp, _i, fp, _pp, _pp, _pp, _pp, _pp, _pp, fp, _i, _l, _i, _i, fp]),
    "clc_eb_aux": (_i, [fp, _pp, _pp, _pp, fp, fp, fp, _i, fp]),
    "clc_quantize_build_indexes": (_i, [fp, _i, fp, _i, fp, _i, fp, _i, fp, fp, fp, _i, _l, _i, fp]),
    "clc_log2_sum_partials": (_i, [fp, _i, _l, _i, fp, _i, fp]),
    "clc_scaled_recip": (_i, [fp, _i, _l, _i, fp, _f, fp, _i, fp]),
    "clc_scaled_diff": (_i, [fp, fp, _l, fp, _f, fp, fp]),
    "clc_sum_partials": (_i, [fp, _i, _f, fp, _i, fp]),
    "clc_rd_combine": (_i, [fp, _i, fp, _i, fp, _i, _f, _f, _f, fp, fp, fp, fp]),
    "clc_rd_grad_scalars": (_i, [fp, fp, fp, _f, _f, _f, fp, fp, fp]),
    "clc_sqdiff_partials": (_i, [fp, fp, _l, fp, _i, fp]),
    "clc_clm_sim_colsum_workspace_bytes": (_sz, [_i, _i]),
    "clc_clm_sim_colsum": (_i, [fp, _i, fp, _i, _i, _i, _i, _f, fp, fp, _sz, fp]),
    "clc_clm_scale_rows": (_i, [fp, _i, fp, fp, _i, _l, _i, fp]),
    "clc_clm_deform": (_i, [fp, _i, fp, _i, fp, _i, fp, _i, _i, _i, _i, _i, fp]),
    "clc_clm_fuse": (_i, [_pp, _pp, _i, _i, _i, fp, _i, fp, _i, _l, _i, _i, fp]),
    "clc_pm_prep": (_i, [fp, fp, _i, _i, _i, _f, fp]),
    "clc_pm_gauss_mask": (_i, [fp, _i, _i, _i, _i, fp]),
    "clc_pm_pearson_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "clc_pm_pearson": (_i, [fp, _i, fp, _i, _i, _i, _i, _i, fp, fp, fp, _sz, fp]),
    "clc_pm_topk": (_i, [fp, _i, _i, _i, fp, fp, fp]),
    "clc_pm_gather": (_i, [fp, _i, _i, _i, _i, _i, fp, fp, _i, _f, fp, fp]),
    "clc_ssim_init": (_i, []),
    "clc_ssim_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "clc_ssim_scale_fwd": (_i, [fp, _i, fp, _i, _i, _i, _i, _i, _f, fp, fp, _sz, fp]),
    "clc_ssim_scale_bwd": (_i, [fp, _i, fp, _i, _i, _i, _i, _i, _f, fp, fp, fp, _i, fp, _sz, fp]),
    "clc_avgpool2": (_i, [fp, _i, fp, _i, _i, _i, _i, fp]),
    "clc_residual_unit_fwd": (_i, [C.POINTER(RUDesc), fp]),
    "clc_residual_unit_dgrad": (_i, [C.POINTER(RUDesc), fp]),
    "clc_mlp_fwd": (_i, [C.POINTER(MlpDesc), fp]),
    "clc_mlp_bwd": (_i, [C.POINTER(MlpDesc), fp]),
    "clc_mlp_blocks": (_i, [_l]),
    "clc_lnlin_fwd": (_i, [C.POINTER(LnLinDesc), fp]),
    "clc_lnlin_bwd": (_i, [C.POINTER(LnLinDesc), fp]),
    "clc_gdn_bwd_fused": (_i, [fp, fp, fp, fp, fp, fp, _l, _i, _i, fp]),
    "clc_maxpool2d": (_i, [fp, _i, fp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, fp]),
    "clc_adaptive_pool2d": (_i, [fp, _i, fp, _i, _i, _i, _i, _i, _i, fp]),
    "clc_optim_chunk_elems": (_i, []),
    "clc_grad_sqnorm_partials": (_i, [fp, fp, _i, fp, _f, fp]),
    "clc_adamw_step": (_i, [fp, fp, _i, fp, _f, fp, _d, _d, _f, _f, fp, _f, fp]),
    "clc_adam_tick": (_i, [fp, _d, _d, fp]),
    "clc_scalar_add": (_i, [fp, _f, fp]),
    "clc_rans_encode_bound": (_l, [_l]),
    "clc_rans_encode": (_l, [fp, fp, _l, fp, _i, fp, fp, fp, _l]),
    "clc_rans_decoder_create": (fp, [fp, _l]),
    "clc_rans_decoder_decode": (_l, [fp, fp, _l, fp, _i, fp, fp, fp]),
    "clc_rans_decoder_destroy": (None, [fp]),
    "clc_pmf_to_quantized_cdf": (_i, [fp, _i, _i, fp]),
}


class ClcError(RuntimeError):
    pass


_lib = None


def load():
    """Load libclc_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ClcError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(make -C clc_amd/csrc). There is no CPU fallback for the clc_amd product path.")
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    # CLC_TUNING="key:value,..." -> clc_set_tuning (A/B switches between kernel variants that compute the same results)
    for item in filter(None, os.environ.get("CLC_TUNING", "").split(",")):
        k, v = item.split(":")
        L.clc_set_tuning(int(k), int(v))
    if os.environ.get("CLC_WINO"):   # Winograd F(2x2, 3x3) for the 3x3 layers of the transforms (tuning key 23): bit 0 forward, bit 1 data gradients, bit 2 the 64-wide kernel
        L.clc_set_tuning(23, int(os.environ["CLC_WINO"]))
    _lib = L
    return L


def check(rc, what=""):
    if rc is None:
        return
    if rc < 0:
        msg = load().clc_last_error()
        raise ClcError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
    return rc


def ptr_array(ptrs):
    arr = (fp * len(ptrs))(*ptrs)
    return arr
