"""Trainer shell: output tree, log format, PNG grids - host logic only, no GPU."""
import os
import re

import pytest
import struct
import zlib

import torch

import vit_gan_amd  # noqa: F401
from vit_gan_amd import training as T
from vit_gan_amd.config import Config


def _decode_png(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, chunks = 8, []
    while pos < len(raw):
        (n,) = struct.unpack(">I", raw[pos:pos + 4])
        tag, payload = raw[pos + 4:pos + 8], raw[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
        assert crc == zlib.crc32(tag + payload) & 0xFFFFFFFF
        chunks.append((tag, payload))
        pos += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    W, H, depth, ctype, _, _, _ = struct.unpack(">IIBBBBB", chunks[0][1])
    assert (depth, ctype) == (8, 2)
    data = zlib.decompress(chunks[1][1])
    rows = []
    for y in range(H):
        line = data[y * (1 + 3 * W):(y + 1) * (1 + 3 * W)]
        assert line[0] == 0
        rows.append(torch.tensor(list(line[1:]), dtype=torch.uint8).view(W, 3))
    return torch.stack(rows)  # [H, W, 3]


def test_run_dirs_follow_the_reference_tree(tmp_path):
    d = T.RunDirs(str(tmp_path))
    d.construct()
    assert os.path.basename(d.output) == "output" and re.fullmatch(r"\d{8}-\d{6}", os.path.basename(d.save))
    for sub in ("images", "input", "noise", "checkpoints"):
        assert os.path.isdir(os.path.join(d.save, sub))


def test_grid_layout_and_normalisation():
    imgs = torch.zeros(5, 3, 4, 4)
    for i in range(5):
        imgs[i] = i  # min 0, max 4 -> image i becomes i/4
    g = T.make_grid(imgs, nrow=2, padding=2, normalize=True)
    assert g.shape == (3, 3 * 6 + 2, 2 * 6 + 2)  # ceil(5/2) = 3 rows of 2
    assert torch.all(g[:, :2, :] == 0) and torch.all(g[:, :, :2] == 0)  # padding
    assert torch.allclose(g[:, 2:6, 2:6], torch.zeros(3, 4, 4))
    assert torch.allclose(g[:, 2:6, 8:12], torch.full((3, 4, 4), 0.25))
    assert torch.allclose(g[:, 14:18, 2:6], torch.full((3, 4, 4), 1.0))
    assert torch.all(g[:, 14:18, 8:12] == 0)  # the empty sixth cell


def test_png_round_trip(tmp_path):
    torch.manual_seed(0)
    imgs = torch.rand(16, 3, 8, 8) * 2 - 1
    p = str(tmp_path / "grid.png")
    T.save_images(p, imgs, batch_size=16)  # nrow = floor(sqrt(16)) = 4
    px = _decode_png(p)
    g = T.make_grid(imgs, nrow=4)
    want = g.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to(torch.uint8)
    assert px.shape == want.shape == (4 * 10 + 2, 4 * 10 + 2, 3)
    assert torch.equal(px, want)


def test_log_line_format(tmp_path, capsys):
    T._log_file = str(tmp_path / "training.log")
    try:
        T.log("hello")
    finally:
        T._log_file = None
    out = capsys.readouterr().out.strip()
    assert re.fullmatch(r"\[\d{4}-\d\d-\d\d \d\d:\d\d:\d\d\.\d{3}\] hello", out)
    assert open(tmp_path / "training.log").read().strip() == out


def test_config_extra_field_keeps_the_reference_str_and_routes_the_generator():
    """``Config()`` prints exactly the reference's 15 lines (src/v2/utils.py:42-43); the extra field
    only shows in what ``ViTGAN(config).generator`` is."""
    from vit_gan_amd.generator import SirenGenerator
    from vit_gan_amd.modules import ViTGAN, ViTGenerator
    lines = str(Config()).split("\n")
    assert len(lines) == 15 and lines[0] == "attention_heads_count=4" and lines[-1] == "transformer_blocks_count=6"
    assert "generator_kind" not in str(Config(generator_kind="sln_siren")) and "generator_kind" not in repr(Config())
    small = dict(embeddings_dimension=128, transformer_blocks_count=1)
    assert isinstance(ViTGAN(Config(**small)).generator, ViTGenerator)           # default: reference behaviour
    g = ViTGAN(Config(generator_kind="sln_siren", **small))
    assert isinstance(g.generator, SirenGenerator) and g.generator.patch_size == 0 and g.generator.latent == 1024
    keys = sorted(g.state_dict())
    assert keys[0].startswith("discriminator.vit.") and any(k.startswith("generator.mapping_mlp.") for k in keys)
    gp = ViTGAN(Config(generator_kind="sln_siren_patch", image_size=64, patch_size=8, **small)).generator
    assert isinstance(gp, SirenGenerator) and gp.patch_size == 8 and gp.embedding.shape == (64, 128)
    with pytest.raises(ValueError):
        ViTGAN(Config(generator_kind="dcgan", **small))
    tc = T.trainable_config(Config(**small))
    assert tc.classes_count == 1 and tc.generator_kind == "sln_siren" and T.trainable_config(Config(image_size=64, patch_size=8)).generator_kind == "sln_siren_patch"


def test_discriminator_state_strips_the_container_prefix():
    from vit_gan_amd.modules import ViTDiscriminator, ViTGAN
    c = Config(embeddings_dimension=128, transformer_blocks_count=1, generator_kind="sln_siren")
    gan = ViTGAN(c)
    sd = T.discriminator_state(gan.state_dict())
    D = ViTDiscriminator(c)
    assert sorted(sd) == sorted(D.state_dict())
    D.load_state_dict(sd, strict=True)
    assert all(torch.equal(D.state_dict()[k], v) for k, v in sd.items())
    # what INTEGRATION.md used to show loads NOTHING: every key is unexpected
    res = ViTDiscriminator(c).load_state_dict(gan.state_dict(), strict=False)
    assert len(res.missing_keys) == len(sd)


def test_train_model_refuses_to_run_without_the_gpu():
    if torch.cuda.is_available():
        return
    try:
        T.train_model({"epochs": 1}, steps_per_epoch=1, save_artifacts=False)
    except RuntimeError as e:
        assert "no CPU path" in str(e)
    else:
        raise AssertionError("train_model ran without a GPU")


def test_data_loader_is_gated_on_torchvision():
    try:
        import torchvision  # noqa: F401
    except ImportError:
        try:
            T.get_data_loader(Config())
        except ImportError as e:
            assert "torchvision" in str(e)
        else:
            raise AssertionError("expected ImportError")
