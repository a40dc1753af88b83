"""ctypes binding of libvitgan_hip.so (C ABI: include/vitgan_hip.h).

There is no CPU fallback: if the library is missing every operator raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITGAN_HIP_LIB", os.path.join(_HERE, "libvitgan_hip.so"))  # override: kernel experiments only
CSRC = os.path.join(_HERE, "csrc")

c_void_p, c_int, c_float, c_ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong
ABI_VERSION = 9  # VG_ABI_VERSION of include/vitgan_hip.h this binding was written against


class VgVitDims(C.Structure):
    _fields_ = [(n, c_int) for n in ("C", "IH", "P", "E", "H", "L", "R", "Kc")]


class VgVitLayout(C.Structure):
    _fields_ = [(n, c_ll) for n in (
        "conv_w", "conv_b", "pos", "cls", "layer0", "layer_stride", "wqkv", "wo", "w1", "w2",
        "ln1_w", "ln1_b", "bqkv", "bo", "ln2_w", "ln2_b", "b1", "b2", "layer_weights",
        "lnf_w", "lnf_b", "hw1", "hb1", "hw2", "hb2", "total")]


class VgVitNet(C.Structure):
    _fields_ = [("d", VgVitDims), ("P", c_void_p), ("Pb", c_void_p), ("G", c_void_p),
                ("dropout_p", c_float), ("dropout_seed", C.c_ulonglong), ("dropout_step", c_void_p), ("ctx", c_void_p),
                ("attn_fp8", c_int), ("dense_top", c_int)]


class VgVitWsMap(C.Structure):
    _fields_ = ([(n, c_ll) for n in ("X", "xn1", "qkv", "ao", "xmid", "xn2", "z1", "a1", "lse", "mean1", "rstd1", "mean2", "rstd2")]
                + [(n, c_ll * 2) for n in ("gin", "gmid", "dqkv", "dz1")] + [("total", c_ll), ("xtop", c_ll), ("dxtop", c_ll)])


class VgGenWsMap(C.Structure):
    _fields_ = ([(n, c_ll) for n in ("wmod", "s1", "qkv", "cat", "htmp", "s2", "hout", "sf", "y1", "zf1", "zf2")]
                + [("g", c_ll * 3), ("dw_acc", c_ll), ("total", c_ll)])


class VgGenDims(C.Structure):
    _fields_ = ([(n, c_int) for n in ("Z", "T", "E", "H", "L", "O", "CW")] + [("omega0", c_float)]
                + [(n, c_int) for n in ("patch", "C", "IH")])


class VgGenLayout(C.Structure):
    _fields_ = [(n, c_ll) for n in (
        "emb", "map_w", "map_b", "layer0", "layer_stride", "wqkv", "wo", "wm",
        "sln1_w", "sln1_b", "sln1_s", "sln2_w", "sln2_b", "sln2_s", "bo", "bm", "layer_weights",
        "slnf_w", "slnf_b", "slnf_s", "s1_w", "s1_b", "s2_w", "s2_b", "total")]


class VgGenNet(C.Structure):
    _fields_ = [("d", VgGenDims), ("P", c_void_p), ("Pb", c_void_p), ("G", c_void_p),
                ("dropout_p", c_float), ("dropout_seed", C.c_ulonglong), ("dropout_step", c_void_p), ("pos_table", c_void_p)]


P = c_void_p
_SIGNATURES = {
    "vg_abi_version": (c_int, []),
    "vg_linear_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_linear_gelu_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P]),
    "vg_linear_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P, c_float, P]),
    "vg_linear_wgrad_slab_floats": (c_ll, [c_int, c_int, c_int]),
    "vg_linear_wgrad": (c_int, [P, P, P, P, c_ll, c_int, c_int, c_int, c_int, c_int, P]),
    "vg_linear_wgrad_group": (c_int, [c_int, P, P, P, P, P, c_int, c_int, P, c_ll, P, c_ll, c_int, P]),
    "vg_layernorm_fwd": (c_int, [P, c_ll, P, P, P, c_ll, P, P, c_int, c_int, c_float, P]),
    "vg_layernorm_bwd_parts": (c_int, [c_int]),
    "vg_layernorm_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "vg_sln_fwd": (c_int, [P, c_int, P, P, P, P, P, P, P, P, c_int, c_int, c_float, P]),
    "vg_sln_bwd": (c_int, [P, P, c_int, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int, c_int, P]),
    "vg_row_pack_elems": (c_ll, [c_int]),
    "vg_row_pack_weight": (c_int, [P, c_int, c_int, c_int, P, P]),
    "vg_row_parts": (c_int, [c_int]),
    "vg_linear_ln_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_linear_dgrad_ln_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_row_pack_elems_e": (c_ll, [c_int, c_int]),
    "vg_row_pack_weight_e": (c_int, [c_int, P, c_int, c_int, c_int, P, P]),
    "vg_linear_ln_fwd_e": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_linear_dgrad_ln_bwd_e": (c_int, [c_int, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_linear_sln_fwd_e": (c_int, [c_int, P, P, P, P, P, c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_linear_dgrad_sln_bwd_e": (c_int, [c_int, P, P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int, c_int, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_encoder_mlp_image_elems": (c_ll, []),
    "vg_encoder_mlp_pack": (c_int, [P, P, P, P]),
    "vg_encoder_mlp_fwd": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_float, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_encoder_post_attention_image_elems": (c_ll, []),
    "vg_encoder_post_attention_pack": (c_int, [P, P, P, P, P]),
    "vg_encoder_post_attention_fwd": (c_int, [P] * 20 + [c_int, c_float, c_float, C.c_ulonglong, c_int, c_int, P, P]),
    "vg_linear_sln_fwd": (c_int, [P, P, P, P, P, c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_linear_dgrad_sln_bwd": (c_int, [P, P, P, c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, P, c_int, c_int, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_colsum_f32": (c_int, [P, c_int, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, P]),
    "vg_colsum_bf16_parts": (c_int, [c_int]),
    "vg_colsum_bf16": (c_int, [P, c_ll, c_int, c_int, P, P, c_int, P]),
    "vg_dropout_apply": (c_int, [P, P, c_ll, c_float, C.c_ulonglong, c_int, P, P]),
    "vg_attention_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_attention_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_attention_cls_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_attention_cls_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_unfold_tokens_fwd": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vg_unfold_tokens_bwd": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "vg_attention_fp8_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_attention_fp8_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_attention_l2_fwd": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_attention_l2_bwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_act_fwd": (c_int, [P, P, c_ll, c_int, P]),
    "vg_act_bwd": (c_int, [P, P, P, c_ll, c_int, P]),
    "vg_act_bwd_bwd": (c_int, [P, P, P, P, P, c_ll, c_int, P]),
    "vg_layernorm_bwd_bwd_parts": (c_int, [c_int]),
    "vg_layernorm_bwd_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "vg_attention_bwd_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_zero_tick": (c_int, [P, c_ll, P, P]),
    "vg_step_inputs": (c_int, [P, P, c_ll, P, c_ll, C.c_ulonglong, P, P]),
    "vg_gan_loss": (c_int, [P, P, P, c_int, c_int, c_int, c_float, P]),
    "vg_gan_loss_pair": (c_int, [P, P, P, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    "vg_adamw_step": (c_int, [P, P, P, P, P, c_ll, c_float, c_float, c_float, c_float, c_float, c_int, P, c_float, P]),
    "vg_diversity_loss": (c_int, [P, P, P, P, c_int, c_int, c_float, P]),
    "vg_grad_clip": (c_int, [P, c_ll, c_float, c_float, P, P]),
    "vg_cast_f32_bf16": (c_int, [P, P, c_ll, P]),
    "vg_ctx_create": (c_int, [C.POINTER(c_void_p)]),
    "vg_ctx_destroy": (c_int, [c_void_p]),
    "vg_vit_layout": (c_int, [C.POINTER(VgVitDims), C.POINTER(VgVitLayout)]),
    "vg_vit_ws_bytes": (c_ll, [C.POINTER(VgVitDims), c_int]),
    "vg_vit_ws_map": (c_int, [C.POINTER(VgVitDims), c_int, C.POINTER(VgVitWsMap)]),
    "vg_gen_ws_map": (c_int, [C.POINTER(VgGenDims), c_int, C.POINTER(VgGenWsMap)]),
    "vg_gen_backward_stages": (c_int, [C.POINTER(VgGenNet), c_int, P, P, c_int, c_int, P]),
    "vg_vit_forward": (c_int, [C.POINTER(VgVitNet), c_int, P, c_int, P, P, P]),
    "vg_vit_backward": (c_int, [C.POINTER(VgVitNet), c_int, P, P, P, c_int, P]),
    "vg_vit_backward_stages": (c_int, [C.POINTER(VgVitNet), c_int, P, P, P, c_int, c_int, c_int, P]),
    "vg_vit_penalty_ws_bytes": (c_ll, [C.POINTER(VgVitDims), c_int]),
    "vg_vit_penalty": (c_int, [C.POINTER(VgVitNet), c_int, P, P, P, c_float, P, P, P, P]),
    "vg_gen_layout": (c_int, [C.POINTER(VgGenDims), C.POINTER(VgGenLayout)]),
    "vg_gen_ws_bytes": (c_ll, [C.POINTER(VgGenDims), c_int]),
    "vg_gen_forward": (c_int, [C.POINTER(VgGenNet), c_int, P, P, P, P]),
    "vg_gen_backward": (c_int, [C.POINTER(VgGenNet), c_int, P, P, P]),
}

_lib: Optional[C.CDLL] = None


def build(verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into libvitgan_hip.so (hipcc cross-compiles without a GPU)."""
    jobs = str(min(6, os.cpu_count() or 1))
    r = subprocess.run(["make", "-C", CSRC, "-j", jobs], capture_output=True, text=True)
    if verbose or r.returncode != 0:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode != 0:
        raise RuntimeError("libvitgan_hip.so build failed")
    return LIB_PATH


def lib() -> C.CDLL:
    """The loaded library; raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        # torch bundles its own libamdhip64; it must be loaded FIRST so that this library binds to the
        # same HIP runtime instance (same SONAME) - otherwise torch's device pointers are foreign to it.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the only compute path of this package. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc).")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.vg_abi_version() != ABI_VERSION:
            raise RuntimeError("libvitgan_hip.so ABI version mismatch; rebuild")
        _lib = handle
    return _lib


_ctx = None


def context() -> c_void_p:
    """Process-wide execution context (second stream + events) for concurrent weight-gradient work."""
    global _ctx
    if _ctx is None:
        h = c_void_p()
        check(lib().vg_ctx_create(C.byref(h)), "vg_ctx_create")
        _ctx = h
    return _ctx


class HipError(RuntimeError):
    pass


def check(rc: int, what: str) -> None:
    if rc != 0:
        kind = "argument validation" if rc < 0 else "hipError_t"
        raise HipError(f"{what} failed: {kind} {rc}")
