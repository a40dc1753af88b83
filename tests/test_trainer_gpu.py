"""train_model end to end on the GPU: artefacts of the reference's trainer shell."""
import glob
import os
import re

import pytest
import torch

import vit_gan_amd  # noqa: F401
from vit_gan_amd.config import Config
from vit_gan_amd.modules import ViTDiscriminator, ViTGAN
from vit_gan_amd.training import discriminator_state, train_model, trainable_config

pytestmark = pytest.mark.gpu


def test_train_model_writes_the_reference_artefacts(tmp_path):
    cfg = {"epochs": 2, "batch_size": 16, "embeddings_dimension": 128, "attention_heads_count": 4, "transformer_blocks_count": 2}
    fids = iter([30.7, 12.2])
    out = train_model(cfg, steps_per_epoch=3, output_base=str(tmp_path), fid_fn=lambda gan, epoch: next(fids))
    d = out["dirs"]
    assert len(out["history"]) == 2 and all(torch.isfinite(torch.tensor(h)).all() for h in out["history"])
    for e in (0, 1):
        for sub, stem in ((d.images, "samples"), (d.noise, "noise"), (d.input, "input")):
            assert os.path.getsize(os.path.join(sub, f"{stem}_epoch_{e}.png")) > 100
    assert sorted(os.path.basename(p) for p in glob.glob(os.path.join(d.checkpoints, "*.pth"))) == [
        "best_model_epoch_0_fid_30.pth", "best_model_epoch_1_fid_12.pth"]
    text = open(os.path.join(d.save, "training.log")).read()
    assert "Starting training at:" in text and "Parameters:" in text and "Run took" in text
    lines = re.findall(r"Epoch \[(\d)/2\] \| Disc Loss: [-\d.]+, Gen Loss: [-\d.]+ \| FID: ([\d.]+)", text)
    assert lines == [("0", "30.7000"), ("1", "12.2000")]
    # the final checkpoint loads strict=True into freshly built modules and reproduces the trained weights
    state = torch.load(os.path.join(d.save, "final_model.ckpt"), map_location="cpu")
    c = trainable_config(Config(**cfg))
    assert isinstance(out["gan"], ViTGAN) and c.generator_kind == "sln_siren"
    fresh = ViTGAN(c)
    fresh.load_state_dict(state, strict=True)
    for k, v in out["gan"].state_dict().items():
        assert torch.equal(v.cpu(), state[k]), k
    # the discriminator alone out of a ViTGAN checkpoint, strict=True, same outputs
    D = ViTDiscriminator(c)
    D.load_state_dict(discriminator_state(state), strict=True)
    D = D.cuda().eval()
    x = torch.rand(4, 3, 32, 32, device="cuda") * 2 - 1
    with torch.no_grad():
        assert torch.equal(D(x), out["discriminator"].eval()(x))


def test_engine_errors_are_raised_not_swallowed(tmp_path, monkeypatch):
    """A HIP-side failure must not end as a log line plus a checkpoint."""
    from vit_gan_amd import _lib
    from vit_gan_amd.engine import GanEngine

    def broken_step(self, real):
        _lib.check(-3, "vg_vit_forward")
    monkeypatch.setattr(GanEngine, "step", broken_step)
    with pytest.raises(_lib.HipError):
        train_model({"epochs": 1, "batch_size": 8, "embeddings_dimension": 128, "transformer_blocks_count": 1},
                    steps_per_epoch=1, output_base=str(tmp_path))
    runs = glob.glob(os.path.join(str(tmp_path), "output", "*"))
    assert runs and not os.path.exists(os.path.join(runs[0], "final_model.ckpt"))
    text = open(os.path.join(runs[0], "training.log")).read()
    assert "HIP engine error" in text
    assert "Saving the model" not in text and "NO checkpoint was written" in text  # the log must not claim a file that does not exist


def test_exceptions_inside_the_loop_are_logged_not_raised(tmp_path):
    def boom(gan, epoch):
        raise ValueError("fid backend missing")
    out = train_model({"epochs": 1, "batch_size": 8, "embeddings_dimension": 128, "transformer_blocks_count": 1},
                      steps_per_epoch=1, output_base=str(tmp_path), fid_fn=boom)
    text = open(os.path.join(out["dirs"].save, "training.log")).read()
    assert "Exception: fid backend missing" in text and "Run took" in text
    assert os.path.exists(os.path.join(out["dirs"].save, "final_model.ckpt"))
