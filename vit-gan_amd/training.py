"""``train_model(config=None)``: the entry point of the reference's ``main-v2.py`` (src/v2/training.py:34) on the
HIP engine, with the reference's trainer shell around the hot path.

Kept from the reference:
  * ``Config`` handling (``Config() if not config else Config(**config)``, training.py:130);
  * two AdamW optimizers (lr from the config, weight_decay 1e-3, :150-157) and the alternating step (:176-211) - both
    inside ``GanEngine``;
  * the output tree of src/v2/utils.py:13-20,176-182: ``$SCRATCH/output/<start time>/{images,input,noise,checkpoints}``;
  * ``log`` (utils.py:185-189): ``[YYYY-mm-dd HH:MM:SS.mmm] message`` on the console and appended to ``training.log``;
  * per epoch (:163-166,171-172): a fixed-noise sample grid ``images/samples_epoch_<e>.png``, the noise itself
    ``noise/noise_epoch_<e>.png`` and the first input batch ``input/input_epoch_<e>.png``, written as PNG grids with
    ``nrow = floor(sqrt(batch_size))`` and min-max normalisation (what ``vutils.save_image(..., normalize=True)`` does);
  * the epoch line ``Epoch [e/E] | Disc Loss: ..., Gen Loss: ... | FID: ...`` (:227-230), ``best_model_epoch_<e>_fid_<n>.pth``
    checkpoints when the FID improves (:216-226) and, in ``finally``, ``final_model.ckpt`` + a last sample grid +
    the run-time line (:252-268);
  * exceptions raised inside the loop are logged, not re-raised (:248-251) - EXCEPT errors of the HIP engine itself
    (``_lib.HipError``: a failed launch or rejected kernel arguments), which are logged and then re-raised without
    writing a checkpoint: a broken kernel must not look like a finished run.

Substitutions (documented in DESIGN.md):
  * data: CIFAR-10 needs a download (utils.py:109-114); pass ``data_loader`` (any iterable of ``(images, labels)``
    batches, e.g. the reference's own DataLoader) or get synthetic uniform [-1,1] batches, the range of its
    ``Normalize(0.5, 0.5)``;
  * FID needs Inception weights that cannot be fetched offline (utils.py:155-175): pass ``fid_fn(gan, epoch) -> float``
    to enable it; without it the score is NaN and no "best" checkpoint is written;
  * the generator is the v1 SLN/SIREN network (the v2 ``ViTGenerator`` tail raises) and the loss is
    BCE-with-logits on a 1-logit discriminator (the v2 ``criterion`` call raises).
"""
from __future__ import annotations

import datetime
import math
import os
import struct
import traceback
import zlib
from typing import Any, Callable, Dict, Iterable, Optional, Union

import torch
from torch import nn

from . import _lib
from .config import Config
from .engine import GanEngine
from .modules import ViTGAN

START_TIME = datetime.datetime.now()


class RunDirs:
    """The reference's module-level path constants (src/v2/utils.py:13-20), bound to one run."""

    def __init__(self, base: Optional[str] = None, start: Optional[datetime.datetime] = None):
        self.base = base if base is not None else os.getenv("SCRATCH", ".")
        self.start = start or START_TIME
        self.output = os.path.join(self.base, "output")
        self.save = os.path.join(self.output, self.start.strftime("%Y%m%d-%H%M%S"))
        self.images = os.path.join(self.save, "images")
        self.input = os.path.join(self.save, "input")
        self.noise = os.path.join(self.save, "noise")
        self.checkpoints = os.path.join(self.save, "checkpoints")

    def construct(self) -> None:  # utils.py:176-182
        for d in (self.output, self.save, self.images, self.input, self.noise, self.checkpoints):
            os.makedirs(d, exist_ok=True)


_log_file: Optional[str] = None


def log(message: str) -> None:
    """utils.py:185-189 without the rich markup pass."""
    stamp = datetime.datetime.now().strftime("[%F %T.%f")[:-3] + "]"
    line = f"{stamp} {message}"
    print(line, flush=True)
    if _log_file is not None:
        with open(_log_file, "a", encoding="utf-8") as handle:
            handle.write(line + "\n")


# ---- PNG grids (torchvision is not a dependency) ---------------------------------------------------------------
def make_grid(images: torch.Tensor, nrow: int, padding: int = 2, normalize: bool = True) -> torch.Tensor:
    """[B,C,H,W] -> [C, rows*(H+p)+p, cols*(W+p)+p] like ``torchvision.utils.make_grid``: ``nrow`` images per row,
    ``padding`` zero pixels around each, and with ``normalize`` the WHOLE batch shifted/scaled by its min / max."""
    x = images.detach().float().cpu()
    if x.dim() != 4:
        raise ValueError("expected a [B,C,H,W] batch")
    if x.shape[1] == 1:
        x = x.expand(-1, 3, -1, -1)
    if normalize:
        lo, hi = float(x.min()), float(x.max())
        x = ((x - lo) / max(hi - lo, 1e-5)).clamp(0, 1)
    B, C, H, W = x.shape
    cols = min(max(nrow, 1), B)
    rows = int(math.ceil(B / cols))
    grid = torch.zeros(C, rows * (H + padding) + padding, cols * (W + padding) + padding)
    for i in range(B):
        r, c = divmod(i, cols)
        y0, x0 = r * (H + padding) + padding, c * (W + padding) + padding
        grid[:, y0:y0 + H, x0:x0 + W] = x[i]
    return grid


def write_png(path: str, chw: torch.Tensor) -> None:
    """8-bit RGB PNG of a [3,H,W] tensor in [0,1] (round-half-up like ``mul(255).add_(0.5).clamp_(0,255)``)."""
    img = chw.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8).contiguous()
    H, W, C = img.shape
    if C != 3:
        raise ValueError("RGB only")
    raw = bytearray()
    data = img.numpy().tobytes()
    for y in range(H):
        raw.append(0)  # filter type 0 per scanline
        raw += data[y * W * 3:(y + 1) * W * 3]

    def chunk(tag: bytes, payload: bytes) -> bytes:
        return struct.pack(">I", len(payload)) + tag + payload + struct.pack(">I", zlib.crc32(tag + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))


def save_images(save_path: str, images: torch.Tensor, batch_size: int) -> None:
    """training.py:47-50."""
    write_png(save_path, make_grid(images, nrow=max(1, math.floor(math.sqrt(batch_size))), normalize=True))


def save_figures(save_dir: str, **series) -> None:
    """utils.py:46-98: loss / FID curves, written only when matplotlib is importable and the series are non-empty."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot as plt
    except Exception:  # plotting is optional
        return
    if series.get("gen_losses") and series.get("disc_losses"):
        plt.figure(figsize=(10, 5))
        plt.title("Generator and Discriminator Loss During Training")
        plt.plot(series["gen_losses"], label="G Loss")
        plt.plot(series["disc_losses"], label="D Loss")
        plt.xlabel("Iterations"); plt.ylabel("Loss"); plt.legend()
        plt.savefig(os.path.join(save_dir, "losses.png")); plt.close()
    fids = [f for f in series.get("fid_scores", []) if f == f]
    if fids:
        plt.figure(figsize=(10, 5))
        plt.title("FID Score During Training")
        plt.plot(fids, label="FID Score")
        plt.xlabel("Iterations"); plt.ylabel("FID"); plt.legend()
        plt.savefig(os.path.join(save_dir, "fid_score.png")); plt.close()


def get_data_loader(c: Config):
    """utils.py:100-121 (CIFAR-10, Resize/ToTensor/Normalize(0.5), shuffle, 4 workers, drop_last).  Needs torchvision
    and a reachable / pre-downloaded dataset, neither of which this build depends on."""
    try:
        import torchvision.datasets as datasets
        import torchvision.transforms as transforms
    except ImportError as e:
        raise ImportError("get_data_loader needs torchvision; pass train_model(data_loader=...) or use the synthetic default") from e
    from torch.utils.data import DataLoader
    tf = transforms.Compose([transforms.Resize(c.image_size), transforms.ToTensor(),
                             transforms.Normalize([0.5] * c.input_channels, [0.5] * c.input_channels)])
    ds = datasets.CIFAR10(root=os.path.expanduser("~/rep/me/vit-gan/data/cifar-10-python/"), train=True, download=True, transform=tf)
    return DataLoader(ds, batch_size=c.batch_size, shuffle=True, num_workers=4, drop_last=True)


class SyntheticLoader:
    """``steps`` batches of uniform [-1,1] images (the range of Normalize(0.5, 0.5)) generated on the device."""

    def __init__(self, c: Config, steps: int, device: torch.device, seed: int = 1234):
        self.c, self.steps, self.device = c, steps, device
        self.gen = torch.Generator(device=device).manual_seed(seed)

    def __len__(self) -> int:
        return self.steps

    def __iter__(self):
        c = self.c
        for _ in range(self.steps):
            x = torch.rand(c.batch_size, c.input_channels, c.image_size, c.image_size, device=self.device, generator=self.gen)
            yield x * 2 - 1, None


def trainable_config(c: Config) -> Config:
    """The configuration ``train_model`` builds ``ViTGAN`` from: a 1-logit discriminator (the executable loss, SURVEY 8
    row a12) and a generator that can produce an image - the reference's default "v2" tail raises, so it is
    replaced by the SLN/SIREN network (row-token layout at 32x32, patch grid beyond)."""
    kind = c.generator_kind
    if kind == "v2":
        kind = "sln_siren" if c.image_size <= 32 else "sln_siren_patch"
    return c.model_copy(update={"classes_count": 1, "generator_kind": kind})


def discriminator_state(gan_state: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """``ViTDiscriminator`` keys (``vit.*``) out of a ``ViTGAN`` checkpoint (``discriminator.vit.*``, ``generator.*``) as
    written by ``train_model`` / src/v2/training.py:220-226,263 - for ``D.load_state_dict(..., strict=True)``."""
    pre = "discriminator."
    return {k[len(pre):]: v for k, v in gan_state.items() if k.startswith(pre)}


def train_model(config: Optional[Dict[str, Any]] = None, steps_per_epoch: int = 50, max_epochs: Optional[int] = None,
                loss: str = "ns", device: str = "cuda:0", seed: int = 0, data_loader: Optional[Iterable] = None,
                fid_fn: Optional[Callable[[nn.Module, int], float]] = None, output_base: Optional[str] = None,
                save_artifacts: bool = True, clip_d: Optional[float] = None, clip_g: Optional[float] = None,
                diversity_weight: float = 0.0, instance_noise: float = 0.0, gp_weight: float = 0.0):
    """``loss``: "ns" (default: the executable v1 loss), "hinge", or "wasserstein" - the critic losses of the reference's
    unreached step (training.py:67-125); ``clip_d`` / ``clip_g``: its clip_grad_norm_ limits (5.0 / 0.5 there);
    ``diversity_weight``: its diversity term (0.1 there); ``instance_noise``: sigma of the noise on D's inputs (0.1
    there); ``gp_weight``: the weight of its gradient penalty (``c.lambda_gp``, a field the reference's Config lacks)."""
    global _log_file
    c = Config() if not config else Config(**config)
    if not torch.cuda.is_available():
        raise RuntimeError("train_model needs an MI355X: the HIP engine has no CPU path")
    dev = torch.device(device)
    torch.manual_seed(seed)
    dirs = RunDirs(output_base, datetime.datetime.now())
    if save_artifacts:
        dirs.construct()
        _log_file = os.path.join(dirs.save, "training.log")
    gan = ViTGAN(trainable_config(c)).to(dev).train()  # modules.ViTGAN(c).to(device); gan.train(), training.py:145,148
    D, G = gan.discriminator, gan.generator
    eng = GanEngine(D, G, batch=c.batch_size, loss=loss, lr_d=c.discriminator_learning_rate, lr_g=c.generator_learning_rate,
                    weight_decay=1e-3, seed=seed, clip_d=clip_d, clip_g=clip_g, diversity_weight=diversity_weight,
                    instance_noise=instance_noise, gp_weight=gp_weight)
    loader = data_loader if data_loader is not None else SyntheticLoader(c, steps_per_epoch, dev)
    epochs = c.epochs if max_epochs is None else min(c.epochs, max_epochs)

    def construct_noise():  # the v1 generator's latent, gan.py:231-232 (the v2 noise is image-shaped, training.py:35-42)
        return torch.randn(c.batch_size, G.latent, device=dev)

    def save_samples(label: Union[str, int], noise: torch.Tensor):  # training.py:52-57
        if not save_artifacts:
            return
        was = G.training
        G.eval()
        with torch.no_grad():
            samples = G(noise).detach().float().cpu() * 0.5 + 0.5
        G.train(was)
        save_images(os.path.join(dirs.images, f"samples_epoch_{label}.png"), samples, c.batch_size)

    def noise_as_image(noise: torch.Tensor) -> torch.Tensor:  # the latent is a vector here: show it as 1x32x32 tiles
        side = int(math.isqrt(noise.shape[1]))
        return noise[:, :side * side].reshape(noise.shape[0], 1, side, side)

    best_fid = float("inf")
    disc_losses, gen_losses, fid_scores, history = [], [], [], []
    epoch = 0
    fatal: Optional[BaseException] = None
    try:
        log(f"Starting training at: {datetime.datetime.now()}")
        log("Parameters:\n" + str(c))
        for epoch in range(epochs):
            noise = construct_noise()
            if save_artifacts:
                save_images(os.path.join(dirs.noise, f"noise_epoch_{epoch}.png"), noise_as_image(noise), c.batch_size)
            save_samples(epoch, noise)
            losses = None
            for i, (real_images, _) in enumerate(loader):
                if i == 0 and save_artifacts:
                    save_images(os.path.join(dirs.input, f"input_epoch_{epoch}.png"), real_images, c.batch_size)
                losses = eng.step(real_images.to(dev))
            if losses is None:
                raise RuntimeError("the data loader produced no batch")
            d_real, d_fake, g = losses.tolist()  # the only host sync of the epoch (training.py:228)
            disc_losses.append(d_real + d_fake)
            gen_losses.append(g)
            history.append((d_real + d_fake, g))
            fid_score = float(fid_fn(gan, epoch)) if fid_fn is not None else float("nan")
            fid_scores.append(fid_score)
            if fid_score < best_fid:
                best_fid = fid_score
                if save_artifacts:
                    eng.gather_master()  # (a sharded update keeps each rank's fp32 master current on its share only)
                    torch.save(gan.state_dict(), os.path.join(dirs.checkpoints, f"best_model_epoch_{epoch}_fid_{int(fid_score)}.pth"))
            log(f"Epoch [{epoch}/{epochs}] | Disc Loss: {d_real + d_fake:.8f}, Gen Loss: {g:.4f} | FID: {fid_score:.4f}")
            if save_artifacts:
                save_figures(dirs.save, disc_losses=disc_losses, gen_losses=gen_losses, fid_scores=fid_scores)
    except KeyboardInterrupt as ke:
        log(f"{ke} raised!")
    except _lib.HipError as e:  # a failed launch / rejected kernel arguments is never a "successful" run
        log(f"HIP engine error: {e}\n{traceback.format_exc()}")
        fatal = e
    except Exception as e:  # the reference logs and carries on to `finally` (training.py:250-251)
        log(f"Exception: {e}\n{traceback.format_exc()}")
    finally:
        model_path = os.path.join(dirs.save, "final_model.ckpt")
        if save_artifacts and fatal is None:  # no further GPU work, no checkpoint of a broken run
            save_figures(dirs.save, disc_losses=disc_losses, gen_losses=gen_losses, fid_scores=fid_scores)
            eng.gather_master()
            torch.save(gan.state_dict(), model_path)
            save_samples(epoch, construct_noise())
        took = datetime.datetime.now() - dirs.start
        if fatal is None and save_artifacts:
            log(f"Run took {took}. Saving the model to: {model_path}")
        elif fatal is not None:  # the log of a failed run must not claim a checkpoint that was never written
            log(f"Run took {took}. NO checkpoint was written: the run ended on {type(fatal).__name__}")
        else:
            log(f"Run took {took}.")
        _log_file = None
    if fatal is not None:
        raise fatal
    return {"discriminator": D, "generator": G, "gan": gan, "engine": eng, "history": history, "dirs": dirs}
