"""The CPU oracle must reproduce the reference's own outputs (tests/golden/*.npz).

Fixtures were produced by tests/golden/make_golden.py, which runs the unmodified
reference modules on parameters drawn by tests/golden/weights.py; the same
parameters are rebuilt here from the seed, so no reference code is needed.
Tolerance: fp32 vs fp32 on CPU, different op grouping -> rtol 2e-4 on norms,
atol 2e-5*scale on samples.
"""
import os

import numpy as np
import pytest
import torch

from cases import GEN_CASES, VIT_CASES
from weights import make_input, make_state, summarize

from oracle import gen_oracle as go
from oracle import vit_oracle as vo

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _check_summary(npz, key, arr, rtol=3e-4):
    s = summarize(arr)
    ref_norm = float(npz[f"{key}/norm"])
    assert list(s["shape"]) == list(npz[f"{key}/shape"]), key
    if ref_norm < 1e-5:  # mathematically-zero gradient (e.g. keys.bias: softmax is shift invariant)
        assert float(s["norm"]) < 1e-5, key
        return
    scale = max(ref_norm / max(1.0, np.sqrt(arr.size)), 1e-12)  # rms of the tensor
    np.testing.assert_allclose(float(s["norm"]), ref_norm, rtol=rtol, atol=1e-7, err_msg=key)
    np.testing.assert_allclose(s["sample"], npz[f"{key}/sample"], rtol=rtol, atol=20 * rtol * scale, err_msg=key)


def _dims(c):
    return vo.VitDims(channels=c["channels"], image=c["image"], patch=c["patch"], embed=c["embed"],
                      heads=c["heads"], layers=c["layers"], mlp_ratio=c["mlp_ratio"], classes=c["classes"])


@pytest.mark.parametrize("name", list(VIT_CASES))
def test_vit_oracle_matches_reference(name):
    c = VIT_CASES[name]
    npz = np.load(os.path.join(GOLD, f"vit_{name}.npz"))
    d = _dims(c)
    shapes = vo.vit_param_shapes(d)
    # state_dict contract: same names, same order, same shapes as the reference module
    assert list(shapes.keys()) == [str(s) for s in npz["param_names"]]
    assert [str(s) for s in shapes.values()] == [str(s) for s in npz["param_shapes"]]
    if name == "c1":
        assert len(shapes) == 106 and sum(int(np.prod(s)) for s in shapes.values()) == 7299466 - 384 * 9 - 9
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in make_state(shapes, c["seed"], "vit").items()}
    x = torch.from_numpy(make_input((c["batch"], c["channels"], c["image"], c["image"]), c["seed"], "uniform"))
    x.requires_grad_(True)
    taps = {}
    out = vo.vit_forward(st, x, d, taps=taps)
    np.testing.assert_allclose(out.detach().numpy(), npz["out"], rtol=2e-4, atol=2e-5)
    R = torch.from_numpy(make_input(tuple(out.shape), c["seed"] + 1))
    (out * R).sum().backward()
    _check_summary(npz, "dx", x.grad.numpy())
    _check_summary(npz, "tap/embed", taps["embed"].detach().numpy())
    for i, t in enumerate(taps["blocks"]):
        _check_summary(npz, f"tap/block{i}", t.detach().numpy())
    for k, p in st.items():
        _check_summary(npz, f"grad/{k}", p.grad.numpy())
    # second functional L = out.sum()
    x.grad = None
    vo.vit_forward(st, x, d).sum().backward()
    _check_summary(npz, "dx_sum", x.grad.numpy())


def test_param_count_matches_survey():
    # 7 299 466 params at E=384, K=10; 827 530 at the Config() default
    d = vo.VitDims(classes=10)
    assert sum(int(np.prod(s)) for s in vo.vit_param_shapes(d).values()) == 7299466
    d = vo.VitDims(embed=128, classes=10)
    assert sum(int(np.prod(s)) for s in vo.vit_param_shapes(d).values()) == 827530
    assert abs(vo.matmul_flops_per_image(vo.VitDims(classes=10)) / 1e6 - 961.72) < 0.01
    assert abs(go.matmul_flops_per_image(go.GenDims()) / 1e6 - 243.79) < 0.01
    assert sum(int(np.prod(s)) for s in go.gen_param_shapes(go.GenDims()).values()) == 15936114


def test_vitgenerator_v2_tail_matches_reference():
    """trunk + Linear(K, batch_size) + flat view, including its failure."""
    c = VIT_CASES["c1k10"]
    npz = np.load(os.path.join(GOLD, "vitgen_v2.npz"))
    d = _dims(c)
    rng = np.random.Generator(np.random.PCG64(99))
    for bs, tag in ((c["batch"], "illegal"), (96, "legal")):
        st = {k: torch.from_numpy(v) for k, v in make_state(vo.vit_param_shapes(d), c["seed"], "vit").items()}
        rng = np.random.Generator(np.random.PCG64(99))
        st["linear.weight"] = torch.from_numpy((rng.standard_normal(size=(bs, c["classes"])) * 0.3).astype(np.float32))
        st["linear.bias"] = torch.from_numpy((rng.standard_normal(size=(bs,)) * 0.1).astype(np.float32))
        assert list(st.keys()) == [str(s) for s in npz[f"{tag}/state_keys"]]
        x = torch.from_numpy(make_input((bs, c["channels"], c["image"], c["image"]), c["seed"]))
        np.testing.assert_allclose(vo.vit_forward(st, x, d).numpy(), npz[f"{tag}/vit_out"], rtol=2e-4, atol=2e-5)
        err = str(npz[f"{tag}/error"])
        if err:
            with pytest.raises(RuntimeError) as ei:
                vo.vit_generator_v2_forward(st, x, d)
            assert f"RuntimeError: {ei.value}" == err
        else:
            y = vo.vit_generator_v2_forward(st, x, d)
            assert list(y.shape) == list(npz[f"{tag}/out_shape"])
            _check_summary(npz, f"{tag}/out", y.numpy())


@pytest.mark.parametrize("name", list(GEN_CASES))
def test_gen_oracle_matches_reference(name):
    c = GEN_CASES[name]
    npz = np.load(os.path.join(GOLD, f"gen_{name}.npz"))
    d = go.GenDims()
    shapes = go.gen_param_shapes(d)
    assert list(shapes.keys()) == [str(s) for s in npz["param_names"]]
    assert [str(s) for s in shapes.values()] == [str(s) for s in npz["param_shapes"]]
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in make_state(shapes, c["seed"], "gen").items()}
    z = torch.from_numpy(make_input((c["batch"], d.latent), c["seed"]))
    taps = {}
    out = go.gen_forward(st, z, d, taps=taps)
    # sin(30 * .) amplifies fp32 reassociation noise: atol 2e-4 on values in [-1, 1]
    np.testing.assert_allclose(out.detach().numpy(), npz["out"], rtol=0, atol=3e-4)
    R = torch.from_numpy(make_input(tuple(out.shape), c["seed"] + 1))
    (out * R).sum().backward()
    _check_summary(npz, "tap/w", taps["w"].detach().numpy().reshape(c["batch"], -1))
    for i, t in enumerate(taps["blocks"]):
        _check_summary(npz, f"tap/block{i}", t.detach().numpy())
    for k, p in st.items():
        _check_summary(npz, f"grad/{k}", p.grad.numpy(), rtol=2e-3)


@pytest.mark.parametrize("name", ["l2", "l2spec", "l2e128"])
def test_v1_l2_attention_oracle_matches_reference(name):
    """v1 MultiHeadSelfAttention with cdist scores (and the spectral rescale) vs the reference."""
    from cases import V1ATT_CASES
    from oracle import v1att_oracle as ao
    c = V1ATT_CASES[name]
    npz = np.load(os.path.join(GOLD, f"v1att_{name}.npz"))
    shapes = ao.v1att_param_shapes(c["embed"], c["heads"], c["head_dim"], c["embed"])
    assert list(shapes.keys()) == [str(s) for s in npz["param_names"]]
    assert [str(s) for s in shapes.values()] == [str(s) for s in npz["param_shapes"]]
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in make_state(shapes, c["seed"], "v1att").items()}
    x = torch.from_numpy(make_input((c["batch"], c["seq"], c["embed"]), c["seed"])).requires_grad_(True)
    spec = npz["init_spectrum"] if c["spectral"] else None
    taps = {}
    out, used = ao.mhsa_l2_forward(st, x, c["heads"], c["head_dim"], spec, taps)
    np.testing.assert_allclose(out.detach().numpy(), npz["out"], rtol=3e-4, atol=3e-5)
    _check_summary(npz, "tap/dist_h0", taps["dist_h0"].detach().numpy())
    R = torch.from_numpy(make_input(tuple(out.shape), c["seed"] + 1))
    (out * R).sum().backward()
    _check_summary(npz, "dx", x.grad.numpy(), rtol=1e-3)
    for k, w in used.items():
        _check_summary(npz, f"grad/{k}", w.grad.numpy(), rtol=1e-3)


def test_v1_overlapping_tokeniser_oracle_matches_reference():
    """PatchEncoder._get_tokens (flat view of the double unfold) - exact, it only moves data."""
    from oracle import v1att_oracle as ao
    npz = np.load(os.path.join(GOLD, "v1tokens.npz"))
    for tag in ("a", "b", "c"):
        B, C, IH, P, ov = (int(v) for v in npz[f"{tag}/geometry"])
        x = torch.from_numpy(make_input((B, C, IH, IH), 40 + B + IH, "uniform"))
        np.testing.assert_array_equal(ao.unfold_tokens(x, P, ov).numpy(), npz[f"{tag}/tokens"])


def test_gradient_penalty_oracle_matches_reference():
    """the reference's own ``gradient_penalty`` (src/v2/utils.py:124-144) run on its ViTDiscriminator
    produced tests/golden/gp_v2.npz (penalty, the epsilon it drew, d penalty / d theta); the oracle's restatement on the
    fp32 ViT oracle must reproduce all of it - the double backward through LayerNorm, softmax attention, GELU and tanh."""
    from make_golden import GP_CASE as c
    from oracle import step_oracle as so
    npz = np.load(os.path.join(GOLD, "gp_v2.npz"))
    d = vo.VitDims(channels=c["channels"], image=c["image"], patch=c["patch"], embed=c["embed"], heads=c["heads"],
                   layers=c["layers"], mlp_ratio=c["mlp_ratio"], classes=c["classes"])
    shapes = vo.vit_param_shapes(d)
    assert list(shapes) == [str(s) for s in npz["param_names"]]
    st = {k: torch.from_numpy(v).requires_grad_(True) for k, v in make_state(shapes, c["seed"], "vit").items()}
    real = torch.from_numpy(make_input((c["batch"], c["channels"], c["image"], c["image"]), c["seed"], "uniform"))
    fake = torch.from_numpy(make_input((c["batch"], c["channels"], c["image"], c["image"]), c["seed"] + 1, "uniform"))
    pen = so.gradient_penalty(lambda t: vo.vit_forward(st, t, d), real, fake, torch.from_numpy(npz["epsilon"]))
    np.testing.assert_allclose(float(pen), float(npz["penalty"]), rtol=2e-5)
    pen.backward()
    assert [str(s) for s in npz["no_grad"]] == [k for k, p in st.items() if p.grad is None] == ["vit.classifier.fc2.bias"]
    for k, p in st.items():
        if p.grad is not None:
            _check_summary(npz, f"grad/{k}", p.grad.numpy(), rtol=1e-3)
    for k in ("vit.norm.weight", "vit.encoder.0.norm1.bias", "vit.classifier.fc2.weight"):
        ref = npz[f"full/{k}"]
        np.testing.assert_allclose(st[k].grad.numpy(), ref, rtol=2e-3, atol=2e-4 * float(np.abs(ref).max()))
