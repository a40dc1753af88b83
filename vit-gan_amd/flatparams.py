"""Flat parameter storage shared by the nn.Modules and the HIP engine.

All parameters of one network live in ONE fp32 buffer (layout from the C library); the
nn.Parameters are views into it (so ``state_dict`` keys/shapes are the reference's), their
``.grad`` are views into one fp32 gradient buffer, and a bf16 shadow of the whole buffer feeds
the MFMA GEMMs.  One buffer = one fused AdamW launch and one all-reduce per network.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Tuple

import torch
from torch import nn

from . import _lib
from .flat import numel


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class FlatParams:
    def __init__(self, named: "Dict[str, nn.Parameter]", slots: "Dict[str, Tuple[int, tuple]]", total: int):
        self.slots = slots
        self.total = int(total)
        self.named = named
        missing = set(slots) - set(named)
        extra = set(named) - set(slots)
        if missing or extra:
            raise RuntimeError(f"parameter/layout mismatch: missing {sorted(missing)[:3]} extra {sorted(extra)[:3]}")
        self.flat = None
        self.grad = None
        self.shadow = None
        self.rebuild()

    # ------------------------------------------------------------------ storage
    def rebuild(self) -> None:
        """(Re)allocate the flat buffers on the parameters' current device and re-alias the views."""
        some = next(iter(self.named.values()))
        device = some.device
        flat = torch.zeros(self.total, dtype=torch.float32, device=device)
        grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        with torch.no_grad():
            for name, (off, shape) in self.slots.items():
                p = self.named[name]
                n = numel(shape)
                if tuple(p.shape) != tuple(shape):
                    raise RuntimeError(f"{name}: shape {tuple(p.shape)} != layout {tuple(shape)}")
                flat[off:off + n].copy_(p.detach().reshape(-1).to(torch.float32))
                if p.grad is not None:
                    grad[off:off + n].copy_(p.grad.detach().reshape(-1).to(torch.float32))
                p.data = flat[off:off + n].view(shape)
                p.grad = grad[off:off + n].view(shape)
        self.flat, self.grad = flat, grad
        self.shadow = torch.empty(self.total, dtype=torch.bfloat16, device=device) if device.type == "cuda" else None

    def aliased(self) -> bool:
        base = self.flat.data_ptr()
        for name, (off, _) in self.slots.items():
            if self.named[name].data_ptr() != base + 4 * off:
                return False
        return True

    # ------------------------------------------------------------------ device ops
    def refresh_shadow(self) -> None:
        """bf16 copy of the master weights for the GEMMs (one streaming kernel)."""
        if self.shadow is None:
            raise RuntimeError("the HIP path needs the parameters on a cuda device")
        _lib.check(_lib.lib().vg_cast_f32_bf16(self.flat.data_ptr(), self.shadow.data_ptr(), self.total, _stream()),
                   "vg_cast_f32_bf16")

    def attach_grads(self) -> None:
        """Make every p.grad a view of the flat gradient buffer, preserving accumulate semantics:
        a parameter whose grad is None starts from zero, a foreign grad tensor is copied in."""
        base = self.grad.data_ptr()
        none_count, foreign = 0, []
        for name, (off, shape) in self.slots.items():
            g = self.named[name].grad
            if g is None:
                none_count += 1
            elif g.data_ptr() != base + 4 * off:
                foreign.append(name)
        if none_count == 0 and not foreign:
            return
        with torch.no_grad():
            if none_count == len(self.slots):
                self.grad.zero_()
            for name, (off, shape) in self.slots.items():
                p = self.named[name]
                view = self.grad[off:off + numel(shape)].view(shape)
                if p.grad is None:
                    if none_count != len(self.slots):
                        view.zero_()
                elif name in foreign:
                    view.copy_(p.grad)
                else:
                    continue
                p.grad = view

    def zero_grad(self) -> None:
        self.grad.zero_()
        self.attach_grads()
