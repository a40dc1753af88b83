"""CPU-side checks (no GPU): the C-ABI library loads and exports everything include/vitgan_hip.h
declares, layouts are consistent, and the nn.Module surface matches the reference's contract."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

import vit_gan_amd  # noqa: F401
from vit_gan_amd import _lib, flat
from vit_gan_amd.config import Config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _header_functions():
    text = open(os.path.join(ROOT, "include", "vitgan_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    names = _header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vitgan_hip.h but not exported"
        assert n in _lib._SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert set(_lib._SIGNATURES) <= set(names), set(_lib._SIGNATURES) - set(names)
    assert lib.vg_abi_version() == _lib.ABI_VERSION == 9


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (negative codes)."""
    lib = _lib.lib()
    assert lib.vg_attention_fwd(None, None, None, 1, 1, 1, 32, 1.0, None) == -1
    assert lib.vg_linear_fwd(None, None, None, None, None, None, None, 8, 8, 8, 0, 0.0, None) == -1
    assert lib.vg_linear_gelu_fwd(None, None, None, None, None, 8, 8, 8, None) == -1
    assert lib.vg_gan_loss_pair(None, None, None, 4, 0, 4, 1, 0, 1.0, None) == -1
    assert lib.vg_zero_tick(None, 4, None, None) == -1 and lib.vg_step_inputs(None, None, 0, None, 0, 1, None, None) == -1
    # full-row Linear + LayerNorm entry points (csrc/gemm_row.hip): host-side shape queries and null checks
    assert lib.vg_row_parts(33280) == 256 and lib.vg_row_parts(16640) == 256 and lib.vg_row_parts(2080) == 65 and lib.vg_row_parts(130) == 0
    assert lib.vg_row_pack_elems(384) == 384 * 384 and lib.vg_row_pack_elems(40) == -2
    assert lib.vg_row_pack_weight(None, 384, 384, 0, None, None) == -1
    assert lib.vg_linear_ln_fwd(None, None, None, None, None, None, None, None, None, None, 32, 384, 1e-5, 0.0, 0, 0, None, None) == -1
    assert lib.vg_linear_dgrad_ln_bwd(None, None, None, None, None, None, None, None, None, None, 32, 384, 0.0, 0, 0, None, None) == -1
    d = _lib.VgVitDims(3, 32, 4, 100, 4, 6, 2, 1)  # E not a multiple of 128
    assert lib.vg_vit_layout(C.byref(d), C.byref(_lib.VgVitLayout())) == -3
    assert lib.vg_vit_ws_bytes(C.byref(d), 4) == -1
    d = _lib.VgVitDims(3, 64, 4, 384, 4, 6, 2, 1)  # 257 tokens > 80
    assert lib.vg_vit_layout(C.byref(d), C.byref(_lib.VgVitLayout())) == -3
    # the gradient penalty as one call (ABI 9): null checks, then the shapes the full-row kernels do not take - all before any launch
    d = _lib.VgVitDims(3, 32, 4, 384, 4, 6, 2, 1)
    assert lib.vg_vit_penalty_ws_bytes(C.byref(d), 16) > 0 and lib.vg_vit_penalty_ws_bytes(None, 16) == -1
    net = _lib.VgVitNet(d, 16, 16, 16, 0.1, 1, None, None, 0, 0)  # (non-null dummies: never dereferenced on these paths)
    assert lib.vg_vit_penalty(C.byref(net), 16, None, None, None, 1.0, None, None, None, None) == -1
    p16 = C.c_void_p(16)
    net8 = _lib.VgVitNet(d, 16, 16, 16, 0.1, 1, None, None, 1, 0)
    assert lib.vg_vit_penalty(C.byref(net8), 16, p16, p16, p16, 1.0, p16, p16, p16, None) == -3  # fp8 attention
    nog = _lib.VgVitNet(d, 16, 16, None, 0.1, 1, None, None, 0, 0)
    assert lib.vg_vit_penalty(C.byref(nog), 16, p16, p16, p16, 1.0, p16, p16, p16, None) == -1   # no gradient buffer to accumulate into


def test_unsupported_depths_are_rejected_not_overflowed():
    """the backward queues 3 folds per ViT block (2L+1 for the generator) into a 42-entry host array (+ 2 for the head and the final LayerNorm, + 2 for the SIREN biases) and uses
    one event pair per block; depths beyond that used to write past the array.  The layouts now refuse them."""
    lib = _lib.lib()
    ok = _lib.VgVitDims(3, 32, 4, 128, 4, 13, 2, 1)
    assert lib.vg_vit_layout(C.byref(ok), C.byref(_lib.VgVitLayout())) == 0
    for L in (14, 20, 71, 500):
        d = _lib.VgVitDims(3, 32, 4, 128, 4, L, 2, 1)
        assert lib.vg_vit_layout(C.byref(d), C.byref(_lib.VgVitLayout())) == -3, L
        assert lib.vg_vit_ws_bytes(C.byref(d), 2) == -1
    g_ok = _lib.VgGenDims(128, 32, 128, 4, 19, 128, 96, 30.0, 0, 3, 32)
    assert lib.vg_gen_layout(C.byref(g_ok), C.byref(_lib.VgGenLayout())) == 0
    g_bad = _lib.VgGenDims(128, 32, 128, 4, 20, 128, 96, 30.0, 0, 3, 32)
    assert lib.vg_gen_layout(C.byref(g_bad), C.byref(_lib.VgGenLayout())) == -3
    with pytest.raises(_lib.HipError):
        from vit_gan_amd.modules import ViTDiscriminator
        ViTDiscriminator(Config(transformer_blocks_count=14))


def test_wgrad_slab_validation_without_gpu():
    """The split-K slab can no longer be overrun through the public entry point: size and split count are checked on the
    host before anything is launched (round-1 fault: slices written past a slab carved for 8)."""
    lib = _lib.lib()
    assert lib.vg_linear_wgrad_slab_floats(384, 384, 8) == 8 * 384 * 384
    assert lib.vg_linear_wgrad_slab_floats(384, 384, 65) == -2
    fake = C.c_void_p(4096)  # never dereferenced: validation fails first
    assert lib.vg_linear_wgrad(fake, fake, fake, fake, 8 * 384 * 384 - 1, 1024, 384, 384, 8, 0, None) == -2
    assert lib.vg_linear_wgrad(fake, fake, fake, fake, 1 << 40, 1024, 384, 384, 65, 0, None) == -2
    assert lib.vg_linear_wgrad(fake, fake, fake, None, 1 << 40, 1024, 384, 384, 2, 0, None) == -1
    # grouped form: the problems' regions must tile the fold region exactly (no gap, no overlap), the slab must hold every slice
    ptrs, two = (C.c_void_p * 2)(0x1000, 0x1000), (C.c_int * 2)(384, 384)
    ok_off, gap_off, lap_off = (C.c_longlong * 2)(384 * 384, 0), (C.c_longlong * 2)(0, 384 * 384 + 8), (C.c_longlong * 2)(0, 384 * 383)
    grp = lambda off, slab, region=2 * 384 * 384, splits=4: lib.vg_linear_wgrad_group(2, ptrs, ptrs, two, two, off, 1024, splits, fake, slab, fake, region, 1, None)  # noqa: E731
    assert grp(gap_off, 1 << 40) == -2 and grp(lap_off, 1 << 40) == -2 and grp(ok_off, 4 * 2 * 384 * 384 - 1) == -2
    assert grp(ok_off, 1 << 40, splits=65) == -2 and grp(ok_off, 1 << 40, region=2 * 384 * 384 + 1) == -2
    assert lib.vg_linear_wgrad_group(9, ptrs, ptrs, two, two, ok_off, 1024, 4, fake, 1 << 40, fake, 2 * 384 * 384, 1, None) == -1


def test_product_library_reads_no_environment():
    """Tuning knobs (VG_VIT_SPLITS / VG_GEMM_T4MIN / VG_GEMM_WM) exist only in `make var` builds (-DVG_TUNING)."""
    blob = open(_lib.LIB_PATH, "rb").read()
    for name in (b"VG_VIT_SPLITS", b"VG_GEMM_T4MIN", b"VG_GEMM_WM", b"VG_STAMP_PTR"):
        assert name not in blob, name
    import subprocess
    nm = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in nm


@pytest.mark.parametrize("dims", [(3, 32, 4, 384, 4, 6, 2, 1), (3, 32, 4, 128, 4, 6, 2, 10), (3, 64, 8, 512, 8, 6, 2, 1),
                                  (3, 128, 16, 768, 12, 6, 2, 1)])
def test_vit_layout_is_a_partition(dims):
    d = _lib.VgVitDims(*dims)
    lay = flat.vit_layout(d)
    slots = flat.vit_slots(d)
    iv = sorted((o, o + flat.numel(s)) for o, s in slots.values())
    assert iv[0][0] >= 0 and iv[-1][1] <= lay.total
    assert all(a[1] <= b[0] for a, b in zip(iv, iv[1:])), "overlapping parameter slots"
    assert all(o % 8 == 0 for o, _ in slots.values() if True), "16-byte alignment of every bf16 slot"
    # q|k|v weights and biases are adjacent (one fused [3E,E] GEMM operand)
    E = d.E
    q, k, v = (slots[f"vit.encoder.0.attention.{n}.weight"][0] for n in ("queries", "keys", "values"))
    assert k == q + E * E and v == k + E * E
    assert lay.total % 4 == 0 and lib_ws(d) > 0


def lib_ws(d):
    return _lib.lib().vg_vit_ws_bytes(C.byref(d), 8)


def test_config_is_the_references():
    c = Config()
    # the reference's 15 fields in its order, then the build's one extra field (hidden from repr / str)
    assert list(Config.model_fields)[-1] == "generator_kind" and c.generator_kind == "v2"
    assert list(Config.model_fields)[:15] == ["attention_heads_count", "batch_size", "classes_count", "discriminator_learning_rate",
                                         "dropout_rate", "embeddings_dimension", "epochs", "generator_learning_rate", "image_size",
                                         "input_channels", "mlp_ratio", "optimizer_beta1", "optimizer_beta2", "patch_size",
                                         "transformer_blocks_count"]
    assert (c.embeddings_dimension, c.batch_size, c.classes_count, c.dropout_rate, c.epochs) == (128, 64, 10, 0.1, 500)
    assert str(c).splitlines()[0] == "attention_heads_count=4" and len(str(c).splitlines()) == 15


def test_module_state_dict_contract_and_flat_storage():
    from vit_gan_amd.generator import SirenGenerator
    from vit_gan_amd.modules import ViTDiscriminator, ViTGAN, ViTGenerator
    npz = np.load(os.path.join(GOLD, "vit_c1.npz"))
    D = ViTDiscriminator(Config(embeddings_dimension=384, classes_count=1))
    sd = D.state_dict()
    assert list(sd.keys()) == [str(s) for s in npz["param_names"]]  # the reference module's own key list
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in npz["param_shapes"]]
    fp = D.vit._flat
    assert fp.aliased()
    # parameters are views of the flat buffer; load_state_dict writes through
    new = {k: torch.full_like(v, 0.25) for k, v in sd.items()}
    D.load_state_dict(new, strict=True)
    assert float(fp.flat.sum()) == pytest.approx(0.25 * sum(v.numel() for v in sd.values()))
    # grads are views of one buffer, zero_grad keeps them attached
    D.zero_grad()
    base = fp.grad.data_ptr()
    assert all(p.grad is not None and base <= p.grad.data_ptr() < base + 4 * fp.total for p in D.parameters())
    g = np.load(os.path.join(GOLD, "gen_g1.npz"))
    G = SirenGenerator()
    assert list(G.state_dict().keys()) == [str(s) for s in g["param_names"]]
    assert [str(tuple(v.shape)) for v in G.state_dict().values()] == [str(s) for s in g["param_shapes"]]
    v2 = np.load(os.path.join(GOLD, "vitgen_v2.npz"))
    assert list(ViTGenerator(Config(classes_count=10, batch_size=3, embeddings_dimension=384, transformer_blocks_count=2)).state_dict().keys()) == \
        [str(s) for s in v2["illegal/state_keys"]]
    assert len(ViTGAN(Config()).state_dict()) == 214


def test_no_cpu_fallback():
    from vit_gan_amd import ops
    from vit_gan_amd.generator import SirenGenerator
    from vit_gan_amd.modules import ViTDiscriminator
    D = ViTDiscriminator(Config())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        D(torch.zeros(2, 3, 32, 32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        SirenGenerator(layers=1)(torch.zeros(2, 1024))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(torch.zeros(4, 8), torch.zeros(8, 8))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vit-gan_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), f"{fn} mentions the oracle"


def test_oracle_is_only_reachable_from_the_allowed_places():
    """The oracle is test infrastructure: besides tests/ it may only be imported inside bench.py's cpu_baseline() and
    __graft_entry__.smoke(); the package and the tools never touch it."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        found = []

        def visit(node, fn):
            for child in ast.iter_child_nodes(node):
                name = child.name if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef)) else fn
                if isinstance(child, ast.ImportFrom) and (child.module or "").split(".")[0] == "oracle":
                    found.append(fn)
                if isinstance(child, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in child.names):
                    found.append(fn)
                visit(child, name)
        visit(tree, None)
        return found

    assert set(oracle_imports(os.path.join(root, "bench.py"))) == {"cpu_baseline"}
    assert set(oracle_imports(os.path.join(root, "__graft_entry__.py"))) == {"smoke"}
    for sub in ("vit-gan_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(root, sub)):
            for f in files:
                if f.endswith(".py"):
                    assert oracle_imports(os.path.join(dirpath, f)) == [], f


def test_gemm_wr_seam_wait_matches_the_compiled_store_count(tmp_path):
    """csrc/gemm_wr.hip's seam wait `s_waitcnt vmcnt(4 + EST)` hard-codes how many vector-memory stores the epilogue of the previous
    (always full, 8 m-tile) tile has in flight: EST = 8 per output.  If a compiler ever emitted FEWER store instructions for that
    epilogue, the wait would stop covering the next stage's LDS-DMA pieces (rare wrong tiles).  Build-time check on the
    ISA hipcc actually generates: in every instantiation the full tile's epilogue - the stores between the 192 MFMAs of the first
    tile variant and the next MFMA - is exactly 8 `global_store_dwordx4` per bf16 output (+ 8 `global_store_dwordx2` for the byte codes).  (A heuristic on the
    listing's layout, not a proof: a toolchain that moves the blocks around fails it and asks for a human look, which is the point.)"""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this machine")
    src = os.path.join(ROOT, "vit-gan_amd", "csrc", "gemm_wr.hip")
    out = tmp_path / "gemm_wr.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I", os.path.dirname(src),
                    "-I", os.path.join(ROOT, "include"), src, "-o", str(out)], check=True, capture_output=True)
    kernels, name = {}, None
    for line in out.read_text().splitlines():
        m = re.match(r"^(_Z\d+vg_gemm_wr_kernelILi(\d)ELi(\d)ELi(\d)E\w*):", line)
        if m:
            name = tuple(int(x) for x in m.groups()[1:])
            kernels[name] = {"mfma": 0, "x4": 0, "x2": 0, "done": False}
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            name = None
        if name is None:
            continue
        k = kernels[name]
        ins = line.split()[0] if line.split() else ""
        if ins.startswith("v_mfma"):
            if k["mfma"] == 192 and (k["x4"] or k["x2"]):
                k["done"] = True   # first MFMA behind the full tile's epilogue: stop counting
            k["mfma"] += 1
        elif k["mfma"] <= 192 and not k["done"]:  # (block layout may rotate the tile loop: a store of the epilogue can sit above the MFMAs)
            if ins == "global_store_dwordx4":
                k["x4"] += 1
            elif ins == "global_store_dwordx2":
                k["x2"] += 1
            elif ins.startswith(("global_store", "flat_store", "buffer_store", "scratch_store")):
                raise AssertionError(f"{name}: unexpected store form {ins} in the full tile's epilogue")
    assert len(kernels) >= 6, sorted(kernels)
    for (wtr, act, feat), k in sorted(kernels.items()):
        has_c2 = bool(feat & 2)
        assert k["mfma"] >= 192, (wtr, act, feat, k)
        # with a second output every slot has its bf16 form (dwordx4) and its byte-code form (dwordx2) behind a wave-uniform branch:
        # one of the two executes, so 8 + 8 stores are in flight for 16 + 8 in the listing
        want4, want2 = (16, 8) if has_c2 else (8, 0)
        assert k["x4"] == want4, f"<{wtr},{act},{feat}>: {k['x4']} global_store_dwordx4 in the full tile's epilogue, the seam wait assumes {want4}"
        assert k["x2"] == want2, f"<{wtr},{act},{feat}>: {k['x2']} global_store_dwordx2 (byte codes), the seam wait assumes {want2}"
