"""``train_model(config=None)``: the entry point of the reference's ``main-v2.py`` (src/v2/training.py:34)
on the HIP engine, restricted to the hot path.

What is kept: ``Config`` handling (``Config() if not config else Config(**config)``, training.py:130), the
two AdamW optimizers (lr from the config, weight_decay 1e-3, :150-157), the alternating step of
:176-211 and the per-epoch loss log line (:227-230).
What is not (out of scope): CIFAR-10 download, PNG dumps, FID, plots, Ray Tune,
checkpoint files - the batches here are synthetic (uniform [-1,1] images, the range of the reference's
Normalize(0.5, 0.5)).  Two documented substitutions for pieces that cannot execute in the reference
(SURVEY 0.2): the generator is the v1 SLN/SIREN network (the v2 tail ``view`` raises) and the loss is
BCE-with-logits on a 1-logit discriminator (the v2 ``criterion`` call raises).
"""
from __future__ import annotations

import datetime
from typing import Any, Dict, Optional

import torch

from .config import Config
from .engine import GanEngine
from .generator import SirenGenerator
from .modules import ViTDiscriminator


def log(message: str) -> None:
    stamp = datetime.datetime.now().strftime("[%F %T.%f")[:-3] + "]"
    print(f"{stamp} {message}", flush=True)


def train_model(config: Optional[Dict[str, Any]] = None, steps_per_epoch: int = 50, max_epochs: Optional[int] = None,
                loss: str = "ns", device: str = "cuda:0", seed: int = 0):
    c = Config() if not config else Config(**config)
    if not torch.cuda.is_available():
        raise RuntimeError("train_model needs an MI355X: the HIP engine has no CPU path")
    dev = torch.device(device)
    torch.manual_seed(seed)
    d_cfg = c.model_copy(update={"classes_count": 1, "dropout_rate": 0.0})
    D = ViTDiscriminator(d_cfg).to(dev)
    G = SirenGenerator(image_size=c.image_size, channels=c.input_channels).to(dev)
    eng = GanEngine(D, G, batch=c.batch_size, loss=loss, lr_d=c.discriminator_learning_rate, lr_g=c.generator_learning_rate,
                    weight_decay=1e-3)
    log(f"Starting training at: {datetime.datetime.now()}")
    log("Parameters:\n" + str(c))
    gen = torch.Generator(device=dev).manual_seed(1234)
    epochs = c.epochs if max_epochs is None else min(c.epochs, max_epochs)
    history = []
    for epoch in range(epochs):
        for _ in range(steps_per_epoch):
            real = torch.rand(c.batch_size, c.input_channels, c.image_size, c.image_size, device=dev, generator=gen) * 2 - 1
            losses = eng.step(real)
        d_real, d_fake, g = losses.tolist()  # the only host sync of the epoch (training.py:228)
        history.append((d_real + d_fake, g))
        log(f"Epoch [{epoch}/{epochs}] | Disc Loss: {d_real + d_fake:.8f}, Gen Loss: {g:.4f}")
    return {"discriminator": D, "generator": G, "engine": eng, "history": history}
