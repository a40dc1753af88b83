"""Helpers for the -m gpu parity tests: raw C-ABI calls on torch device tensors."""
import ctypes as C

import torch

import vit_gan_amd  # noqa: F401
from vit_gan_amd import _lib

BF = torch.bfloat16


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(t, dtype=None):
    t = torch.as_tensor(t)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def rbf(t):
    """round to bf16 and back: the fp32 value the kernel actually sees"""
    return t.to(BF).float()


def call(name, *args):
    _lib.check(getattr(_lib.lib(), name)(*args), name)


def sync():
    torch.cuda.synchronize()


def assert_close(got, ref, rel, what="", floor=1e-6):
    """max|got-ref| <= rel * max|ref|  (per-tensor tolerance)"""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{what}: non-finite values"
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= rel * scale + floor, f"{what}: max err {err:.3e} > {rel:g} * max|ref| {scale:.3e}"
    return err / max(scale, floor)
