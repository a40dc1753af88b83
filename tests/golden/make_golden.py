#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own modules (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference (/root/reference, Python) is imported unmodified.  Its off-path
imports (torchvision / torchmetrics / ray: datasets, FID, HPO - SURVEY 8c) are
absent from this image, so empty stand-in modules are registered for them first;
none of them is touched by the modules exercised here.  Parameters are NOT taken
from the reference's RNG: tests/golden/weights.py draws them from numpy PCG64
and they are loaded into the reference modules with load_state_dict(strict=True),
so a fixture stores only seeds and fingerprints of outputs / gradients.
Families: vit_*.npz (src/v2 ViT, 5 shapes), vitgen_v2.npz (the v2 generator tail and its exception), gen_*.npz (src/v1
generator), v1att_*.npz (src/v1 MultiHeadSelfAttention with L2-distance scores, with and without the spectral rescale),
v1tokens.npz (src/v1 PatchEncoder._get_tokens).
The reference never travels: only this script and the .npz files are committed.
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from cases import GEN_CASES, V1ATT_CASES, VIT_CASES  # noqa: E402
from weights import make_input, make_state, summarize  # noqa: E402

REF = os.environ.get("VITGAN_REFERENCE", "/root/reference")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class _Absent:  # any use would be an error: these are off the hot path
        def __init__(self, *a, **k):
            raise RuntimeError("off-path dependency stub")

    _stub("torchmetrics"); _stub("torchmetrics.image")
    _stub("torchmetrics.image.fid", FrechetInceptionDistance=_Absent)
    tv = _stub("torchvision")
    tv.datasets = _stub("torchvision.datasets")
    tv.transforms = _stub("torchvision.transforms")
    tv.utils = _stub("torchvision.utils")
    tv.models = _stub("torchvision.models", vit_b_16=_Absent, ViT_B_16_Weights=_Absent)
    ray = _stub("ray"); ray.tune = _stub("ray.tune")
    os.environ.setdefault("SCRATCH", "/tmp/vitgan_scratch")  # src/v1/config.py:9 needs it at import
    sys.path.insert(0, REF)
    import src.v2.modules as v2m
    import src.v2.utils as v2u
    import src.v1.generator as v1g
    return v2m, v2u, v1g


def flat(prefix, d, out):
    for k, v in d.items():
        out[f"{prefix}/{k}"] = v


def vit_case(v2m, v2u, name, c):
    torch.manual_seed(0)
    cfg = v2u.Config(attention_heads_count=c["heads"], batch_size=c["batch"], classes_count=c["classes"],
                     dropout_rate=0.0, embeddings_dimension=c["embed"], image_size=c["image"],
                     input_channels=c["channels"], mlp_ratio=c["mlp_ratio"], patch_size=c["patch"],
                     transformer_blocks_count=c["layers"])
    D = v2m.ViTDiscriminator(cfg).double().float()
    shapes = {k: tuple(v.shape) for k, v in D.state_dict().items()}
    st = make_state(shapes, c["seed"], "vit")
    D.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    D.train()  # dropout_rate = 0.0 -> identity
    x = torch.from_numpy(make_input((c["batch"], c["channels"], c["image"], c["image"]), c["seed"], "uniform"))
    x.requires_grad_(True)
    taps = {}
    hooks = []
    hooks.append(D.vit.embedding.register_forward_hook(lambda m, i, o: taps.__setitem__("embed", o.detach())))
    for i, blk in enumerate(D.vit.encoder):
        hooks.append(blk.register_forward_hook(lambda m, i_, o, i=i: taps.__setitem__(f"block{i}", o.detach())))
    out = D(x)
    R = torch.from_numpy(make_input(tuple(out.shape), c["seed"] + 1))
    (out * R).sum().backward()
    for h in hooks:
        h.remove()
    rec = {"torch_version": np.asarray(torch.__version__), "param_names": np.asarray(list(shapes.keys()))}
    rec["param_shapes"] = np.asarray([str(s) for s in shapes.values()])
    rec["out"] = out.detach().numpy()
    flat("dx", summarize(x.grad.numpy()), rec)
    for k, v in taps.items():
        flat(f"tap/{k}", summarize(v.numpy()), rec)
    for k, p in D.named_parameters():
        flat(f"grad/{k}", summarize(p.grad.numpy()), rec)
    # second functional: L = out.sum()
    D.zero_grad(); x.grad = None
    D(x).sum().backward()
    flat("dx_sum", summarize(x.grad.numpy()), rec)
    np.savez_compressed(os.path.join(HERE, f"vit_{name}.npz"), **rec)
    print(f"vit_{name}: out {tuple(out.shape)} max|out| {float(out.abs().max()):.4f}")


def vitgen_v2_case(v2m, v2u):
    """ViTGenerator: trunk output, linear tail, and the view's exception text."""
    c = VIT_CASES["c1k10"]
    rec = {}
    for bs, tag in ((c["batch"], "illegal"), (96, "legal")):
        cfg = v2u.Config(attention_heads_count=c["heads"], batch_size=bs, classes_count=c["classes"],
                         dropout_rate=0.0, embeddings_dimension=c["embed"], image_size=c["image"],
                         input_channels=c["channels"], mlp_ratio=c["mlp_ratio"], patch_size=c["patch"],
                         transformer_blocks_count=c["layers"])
        G = v2m.ViTGenerator(cfg)
        shapes = {k: tuple(v.shape) for k, v in G.vit.state_dict(prefix="vit.").items()}
        st = make_state(shapes, c["seed"], "vit")
        G.vit.load_state_dict({k[4:]: torch.from_numpy(v) for k, v in st.items()}, strict=True)
        rng = np.random.Generator(np.random.PCG64(99))
        lw = (rng.standard_normal(size=tuple(G.linear.weight.shape)) * 0.3).astype(np.float32)
        lb = (rng.standard_normal(size=tuple(G.linear.bias.shape)) * 0.1).astype(np.float32)
        G.linear.load_state_dict({"weight": torch.from_numpy(lw), "bias": torch.from_numpy(lb)})
        x = torch.from_numpy(make_input((bs, c["channels"], c["image"], c["image"]), c["seed"]))
        rec[f"{tag}/state_keys"] = np.asarray(list(G.state_dict().keys()))
        with torch.no_grad():
            rec[f"{tag}/vit_out"] = G.vit(x).numpy()
            try:
                y = G(x)
                rec[f"{tag}/out_shape"] = np.asarray(y.shape)
                flat(f"{tag}/out", summarize(y.numpy()), rec)
                rec[f"{tag}/error"] = np.asarray("")
            except Exception as e:  # the reference's own failure mode
                rec[f"{tag}/error"] = np.asarray(f"{type(e).__name__}: {e}")
    np.savez_compressed(os.path.join(HERE, "vitgen_v2.npz"), **rec)
    print("vitgen_v2:", str(rec["illegal/error"]), "| legal shape", rec["legal/out_shape"])


def gen_case(v1g, name, c):
    torch.manual_seed(0)
    G = v1g.Generator()
    G.eval()  # attention/mlp dropout (0.2) -> identity; parity is defined without dropout
    shapes = {k: tuple(v.shape) for k, v in G.state_dict().items()}
    st = make_state(shapes, c["seed"], "gen")
    G.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    z = torch.from_numpy(make_input((c["batch"], 1024), c["seed"]))
    taps = {}
    hooks = [G.mapping_mlp.register_forward_hook(lambda m, i, o: taps.__setitem__("w", o.detach()))]
    for i, blk in enumerate(G.transformer_layers):
        hooks.append(blk.register_forward_hook(lambda m, i_, o, i=i: taps.__setitem__(f"block{i}", o[1].detach())))
    hooks.append(G.sln.register_forward_hook(lambda m, i, o: taps.__setitem__("sln_out", o.detach())))
    hooks.append(G.output_network[0].register_forward_hook(lambda m, i, o: taps.__setitem__("siren0", o.detach())))
    out = G(z)
    R = torch.from_numpy(make_input(tuple(out.shape), c["seed"] + 1))
    (out * R).sum().backward()
    for h in hooks:
        h.remove()
    rec = {"torch_version": np.asarray(torch.__version__), "param_names": np.asarray(list(shapes.keys()))}
    rec["param_shapes"] = np.asarray([str(s) for s in shapes.values()])
    rec["out"] = out.detach().numpy()
    for k, v in taps.items():
        flat(f"tap/{k}", summarize(v.numpy()), rec)
    for k, p in G.named_parameters():
        flat(f"grad/{k}", summarize(p.grad.numpy()), rec)
    np.savez_compressed(os.path.join(HERE, f"gen_{name}.npz"), **rec)
    print(f"gen_{name}: out {tuple(out.shape)} range [{float(out.min()):.3f}, {float(out.max()):.3f}]")


def v1att_case(name, c):
    """src/v1/attention.py MultiHeadSelfAttention with lp = 2 (cdist scores), optionally with the per-forward
    spectral rescale.  init_spectrum is fixed at construction from the module's OWN random init (attention.py:37-39),
    so it is recorded in the fixture and handed to the oracle as data."""
    import src.v1.attention as v1a
    from src.v1.config import TransformerParameters
    torch.manual_seed(0)
    tp = TransformerParameters(number_of_heads=c["heads"], input_features=c["embed"], lp=2, spectral_scaling=c["spectral"])
    M = v1a.MultiHeadSelfAttention(tp, output_size=c["embed"], head_dimension=c["head_dim"])
    shapes = {k: tuple(v.shape) for k, v in M.state_dict().items()}
    st = make_state(shapes, c["seed"], "v1att")
    M.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    x = torch.from_numpy(make_input((c["batch"], c["seq"], c["embed"]), c["seed"])).requires_grad_(True)
    rec = {"torch_version": np.asarray(torch.__version__), "param_names": np.asarray(list(shapes.keys())),
           "param_shapes": np.asarray([str(s) for s in shapes.values()])}
    if c["spectral"]:
        rec["init_spectrum"] = np.asarray([[float(v) for v in h.init_spectrum] for h in M.attention_heads], dtype=np.float64)
    # head 0's score matrix before the softmax (the cdist) as a tap
    h0 = M.attention_heads[0]
    params_before = {k: p for k, p in M.named_parameters()}
    out = M(x)
    R = torch.from_numpy(make_input(tuple(out.shape), c["seed"] + 1))
    (out * R).sum().backward()
    rec["out"] = out.detach().numpy()
    flat("dx", summarize(x.grad.numpy()), rec)
    # with spectral scaling the forward REPLACES q/k/v weights by new Parameters (attention.py:60-64): gradients land
    # on the replacements, which is what an optimizer built afterwards would see
    for k, p in M.named_parameters():
        flat(f"grad/{k}", summarize(p.grad.numpy()), rec)
    with torch.no_grad():
        q = h0.q(x); k_ = h0.k(x)
        flat("tap/dist_h0", summarize(torch.cdist(q, k_, p=2).numpy()), rec)
    np.savez_compressed(os.path.join(HERE, f"v1att_{name}.npz"), **rec)
    print(f"v1att_{name}: out {tuple(out.shape)} |out| {float(out.abs().max()):.3f}")


def v1tokens_case():
    """src/v1/patch_encoder.py:54-73 ``_get_tokens`` (the module itself cannot be constructed: its __init__ reads an
    attribute it never sets), called unbound on a stand-in carrying the three attributes it uses."""
    import src.v1.patch_encoder as pe
    rec = {"torch_version": np.asarray(torch.__version__)}
    for tag, (B, C, IH, P, ov) in {"a": (2, 3, 32, 8, 2), "b": (1, 3, 16, 4, 1), "c": (2, 1, 28, 4, 0)}.items():
        stride = (IH - P - 2 * ov) // P + 1  # patch_encoder.py:20-22
        me = types.SimpleNamespace(patch_size=P, overlap=ov, stride=stride)
        x = torch.from_numpy(make_input((B, C, IH, IH), 40 + B + IH, "uniform"))
        tok = pe.PatchEncoder._get_tokens(me, x)
        rec[f"{tag}/geometry"] = np.asarray([B, C, IH, P, ov], dtype=np.int64)
        rec[f"{tag}/tokens"] = tok.numpy()
    np.savez_compressed(os.path.join(HERE, "v1tokens.npz"), **rec)
    print("v1tokens:", {k: rec[k].shape for k in rec if k.endswith("tokens")})


GP_CASE = dict(image=32, patch=4, embed=128, heads=4, layers=2, mlp_ratio=2, classes=1, channels=3, batch=3, seed=41, eps_seed=5)


def gp_case(v2m, v2u, only: bool = False):
    """The reference's ``gradient_penalty`` (src/v2/utils.py:124-144) on its own ViTDiscriminator: the
    penalty value, d penalty / d theta for every parameter, and the epsilon it drew (replayed from the seed)."""
    c = GP_CASE
    cfg = v2u.Config(attention_heads_count=c["heads"], batch_size=c["batch"], classes_count=c["classes"], dropout_rate=0.0,
                     embeddings_dimension=c["embed"], image_size=c["image"], input_channels=c["channels"], mlp_ratio=c["mlp_ratio"],
                     patch_size=c["patch"], transformer_blocks_count=c["layers"])
    D = v2m.ViTDiscriminator(cfg)
    shapes = {k: tuple(v.shape) for k, v in D.state_dict().items()}
    st = make_state(shapes, c["seed"], "vit")
    D.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    D.train()
    real = torch.from_numpy(make_input((c["batch"], c["channels"], c["image"], c["image"]), c["seed"], "uniform"))
    fake = torch.from_numpy(make_input((c["batch"], c["channels"], c["image"], c["image"]), c["seed"] + 1, "uniform"))
    torch.manual_seed(c["eps_seed"])
    eps = torch.rand(c["batch"], 1, 1, 1)          # what gradient_penalty draws first (utils.py:126)
    torch.manual_seed(c["eps_seed"])
    pen = v2u.gradient_penalty(D, real, fake, torch.device("cpu"))
    pen.backward()
    rec = {"torch_version": np.asarray(torch.__version__), "param_names": np.asarray(list(shapes.keys())), "epsilon": eps.numpy(),
           "penalty": np.asarray(float(pen.detach()), dtype=np.float64)}
    untouched = []
    for k, p in D.named_parameters():
        if p.grad is None:   # the input gradient does not depend on it at all (the output bias)
            untouched.append(k)
            continue
        flat(f"grad/{k}", summarize(p.grad.numpy()), rec)
    rec["no_grad"] = np.asarray(untouched)
    for k in ("vit.norm.weight", "vit.encoder.0.norm1.bias", "vit.classifier.fc2.weight", "vit.encoder.1.attention.keys.bias"):
        rec[f"full/{k}"] = dict(D.named_parameters())[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, "gp_v2.npz"), **rec)
    print(f"gp_v2: penalty {float(pen):.6f}  eps {eps.reshape(-1).tolist()}")


def main():
    v2m, v2u, v1g = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "gp":
        gp_case(v2m, v2u)
        return
    gp_case(v2m, v2u)
    v1tokens_case()
    for name, c in V1ATT_CASES.items():
        v1att_case(name, c)
    for name, c in VIT_CASES.items():
        vit_case(v2m, v2u, name, c)
    vitgen_v2_case(v2m, v2u)
    for name, c in GEN_CASES.items():
        gen_case(v1g, name, c)


if __name__ == "__main__":
    main()
