"""``gradient_penalty`` and ``diversity_loss`` of the reference (src/v2/utils.py:124-152) on the HIP engine.

The penalty is the WGAN-GP term of the reference's (unreached) Wasserstein step, training.py:101-106:
``loss += c.lambda_gp * gradient_penalty(gan.discriminator, noisy_real_images, noisy_fake_images, device)``.
It needs the derivative of the discriminator's input gradient with respect to its parameters - a double backward through
every operator.  The discriminator is therefore run through ``twice_differentiable_forward`` (ops2.py): forward kernels,
backward kernels and the backward kernels' own backward (csrc/second_order.hip; GEMMs for the Linear layers).
"""
from __future__ import annotations

from typing import Optional

import torch


def gradient_penalty(discriminator, real_images: torch.Tensor, fake_images: torch.Tensor, device=None,
                     epsilon: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Same signature and arithmetic as src/v2/utils.py:124-144; ``epsilon`` [B,1,1,1] may be supplied (tests), else it
    is drawn with ``torch.rand`` like the reference does."""
    batch_size = real_images.size(0)
    if epsilon is None:
        epsilon = torch.rand(batch_size, 1, 1, 1, device=real_images.device if device is None else device)
    interpolated = (epsilon * real_images.float() + (1 - epsilon) * fake_images.float()).detach().requires_grad_(True)
    vit = discriminator.vit if hasattr(discriminator, "vit") else discriminator
    out = vit.twice_differentiable_forward(interpolated)
    from . import ops2
    with ops2.input_grad_only():  # this backward is for d out / d interpolated alone: no parameter gradients
        (gradients,) = torch.autograd.grad(outputs=out, inputs=interpolated, grad_outputs=torch.ones_like(out), create_graph=True,
                                           retain_graph=True, only_inputs=True)
    gradient_norm = gradients.reshape(batch_size, -1).norm(2, dim=1)
    return ((gradient_norm - 1) ** 2).mean()
