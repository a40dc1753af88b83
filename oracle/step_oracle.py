"""fp32 CPU restatement of the alternating G/D step.

TEST INFRASTRUCTURE - see ``oracle/__init__.py``.

Step ordering follows src/v2/training.py:176-211 ( == src/v1/gan.py:221-252 ):
  D.zero_grad; D(real)->loss->backward; fake=G(noise); D(fake.detach())->loss->backward;
  disc_optimizer.step(); G.zero_grad; D(fake)->loss(label=real)->backward; gen_optimizer.step().
Optimizer: AdamW(lr, weight_decay=1e-3), default betas/eps (training.py:150-157).
Loss: the reference's v2 criterion call raises; the executable loss is
v1's BCE on sigmoid outputs with labels 1/0 and 1 for G (gan.py:16-20,227,238,250)
== BCE-with-logits here ("ns").  "hinge" is the optional loss north_star names;
it is not in the reference and is pinned only against torch.nn.functional.
"""
from __future__ import annotations

from typing import Callable, Dict, Mapping, Tuple

import torch
import torch.nn.functional as F

from . import bf16_model
from .gen_oracle import GenDims, gen_forward
from .vit_oracle import VitDims, vit_forward

Tensor = torch.Tensor


def d_loss_real(logits: Tensor, kind: str) -> Tensor:
    if kind == "ns":
        return F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    if kind == "hinge":
        return F.relu(1.0 - logits).mean()
    if kind == "wasserstein":  # src/v2/training.py:97: -(mean(real_output) - mean(fake_output)), the real half
        return -logits.mean()
    raise ValueError(kind)


def d_loss_fake(logits: Tensor, kind: str) -> Tensor:
    if kind == "ns":
        return F.binary_cross_entropy_with_logits(logits, torch.zeros_like(logits))
    if kind == "hinge":
        return F.relu(1.0 + logits).mean()
    if kind == "wasserstein":
        return logits.mean()
    raise ValueError(kind)


def g_loss(logits: Tensor, kind: str) -> Tensor:
    if kind == "ns":
        return F.binary_cross_entropy_with_logits(logits, torch.ones_like(logits))
    if kind in ("hinge", "wasserstein"):  # training.py:72: -torch.mean(output)
        return -logits.mean()
    raise ValueError(kind)


def diversity_loss(fake: Tensor) -> Tensor:
    """src/v2/utils.py:147-152: sum of all pairwise L1 distances between the flattened images / (B (B-1))."""
    B = fake.shape[0]
    flat = fake.reshape(B, -1)
    return torch.cdist(flat, flat, p=1).sum() / (B * (B - 1))


def gradient_penalty(d_fn: Callable[[Tensor], Tensor], real: Tensor, fake: Tensor, epsilon: Tensor) -> Tensor:
    """src/v2/utils.py:124-144: interpolate with a per-sample epsilon [B,1,1,1] (drawn by the caller: the reference draws
    it with torch.rand inside), differentiate the discriminator's output sum w.r.t. the interpolated images with
    create_graph=True, penalise (||grad||_2 - 1)^2 per sample, mean over the batch."""
    x = (epsilon * real + (1 - epsilon) * fake).detach().requires_grad_(True)
    out = d_fn(x)
    (g,) = torch.autograd.grad(outputs=out, inputs=x, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True,
                               only_inputs=True)
    norm = g.reshape(g.shape[0], -1).norm(2, dim=1)
    return ((norm - 1) ** 2).mean()


class GanStepOracle:
    """Holds leaf tensors for D (v2 ViT) and G (v1 SLN/SIREN) and runs reference steps."""

    def __init__(self, d_state: Mapping[str, Tensor], g_state: Mapping[str, Tensor],
                 ddims: VitDims, gdims: GenDims, lr_d: float = 5e-4, lr_g: float = 5e-4,
                 weight_decay: float = 1e-3, loss: str = "ns", clip_d: float = None, clip_g: float = None,
                 faithful: bool = False):
        """faithful: run both networks through ``bf16_model`` (bf16 roundings exactly where the HIP engine stores bf16,
        fp32 master weights and optimizer) instead of the pure-fp32 restatement - the tight parity tier."""
        self.ddims, self.gdims, self.loss = ddims, gdims, loss
        self.faithful = bool(faithful)
        self.clip_d, self.clip_g = clip_d, clip_g  # utils.clip_grad_norm_ max norms (training.py:78,104), None = off
        self.diversity_weight = 0.0                # weight of diversity_loss(fake) in the G loss (0.1 at training.py:73-74)
        self.gp_weight = 0.0                       # c.lambda_gp of training.py:106 (the field is missing from the reference's Config)
        self.d = {k: v.detach().clone().float().requires_grad_(True) for k, v in d_state.items()}
        self.g = {k: v.detach().clone().float().requires_grad_(True) for k, v in g_state.items()}
        self.opt_d = torch.optim.AdamW(list(self.d.values()), lr=lr_d, weight_decay=weight_decay)
        self.opt_g = torch.optim.AdamW(list(self.g.values()), lr=lr_g, weight_decay=weight_decay)

    def D(self, x: Tensor, masks=None) -> Tensor:
        if self.faithful:
            return bf16_model.vit_forward(self.d, x, self.ddims, masks=masks)
        return vit_forward(self.d, x, self.ddims, masks=masks)

    def G(self, z: Tensor, masks=None) -> Tensor:
        if self.faithful:
            return bf16_model.gen_forward(self.g, z, self.gdims, masks=masks)
        return gen_forward(self.g, z, self.gdims, masks=masks)

    def step(self, real: Tensor, z: Tensor, noisy_inputs=None, masks=None, gp_epsilon: Tensor = None) -> Dict[str, float]:
        """noisy_inputs: optional (noisy_real, noisy_fake) the discriminator sees in ITS step (training.py:83-90:
        real / fake + 0.1 randn); the generator's pass through D always uses the clean fake.
        masks (test hook): explicit dropout multipliers standing in for nn.Dropout's RNG, a dict with the keys "d_real",
        "d_fake", "d_gen" (the three discriminator passes) and "g", each a mask mapping of vit_forward / gen_forward."""
        mk = masks or {}
        for p in self.d.values():
            p.grad = None
        loss_real = d_loss_real(self.D(real if noisy_inputs is None else noisy_inputs[0], mk.get("d_real")), self.loss)
        loss_real.backward()
        fake = self.G(z, mk.get("g"))
        loss_fake = d_loss_fake(self.D(fake.detach() if noisy_inputs is None else noisy_inputs[1], mk.get("d_fake")), self.loss)
        loss_fake.backward()
        self.last_gp = None
        if self.gp_weight:  # loss += c.lambda_gp * gradient_penalty(D, noisy_real, noisy_fake), training.py:101-106 (fp32 path)
            r_in = real if noisy_inputs is None else noisy_inputs[0]
            f_in = fake.detach() if noisy_inputs is None else noisy_inputs[1]
            gp = gradient_penalty(lambda t: vit_forward(self.d, t, self.ddims), r_in, f_in, gp_epsilon)
            (self.gp_weight * gp).backward()
            self.last_gp = float(gp.detach())
        if self.clip_d is not None:
            torch.nn.utils.clip_grad_norm_(list(self.d.values()), max_norm=self.clip_d)
        self.opt_d.step()
        for p in self.g.values():
            p.grad = None
        loss_g = g_loss(self.D(fake, mk.get("d_gen")), self.loss)
        total_g = loss_g + self.diversity_weight * diversity_loss(fake) if self.diversity_weight else loss_g
        total_g.backward()
        if self.clip_g is not None:
            torch.nn.utils.clip_grad_norm_(list(self.g.values()), max_norm=self.clip_g)
        self.opt_g.step()
        return {"d_real": float(loss_real.detach()), "d_fake": float(loss_fake.detach()), "g": float(loss_g.detach())}
