"""Race screen for the hand-synchronised GEMM kernels (counted vmcnt / lgkmcnt waits, raw barriers, LDS rings).

A wrong count or a missing barrier shows up as rare wrong tiles that come and go with timing, so each shape is launched many
times back to back - with other work in flight on the device - and every result is compared with an fp32 matmul of the same
bf16 inputs (computed once by torch on the GPU: a checker, not the product path).  Sizes cover the weights-in-registers kernel
(runs of one short tile, full tiles + short tile), the 128x384-tile weight gradients (1, 2, many stages per K slice, ragged last
slice) and the tiled kernel."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
REPS = 40


def _u():
    import gpu_util
    return gpu_util


@pytest.mark.parametrize("M,N,K", [(1312, 1152, 384), (16640, 384, 384), (8320, 768, 384), (4160, 384, 768), (2600, 384, 1152)])
def test_forward_and_input_gradient_repeatable(M, N, K):
    u = _u()
    g = torch.Generator().manual_seed(M + N + K)
    A = u.dev(u.rbf(torch.randn(M, K, generator=g)), u.BF)
    W = u.dev(u.rbf(torch.randn(N, K, generator=g) / math.sqrt(K)), u.BF)
    b = u.dev(torch.randn(N, generator=g) * 0.1)
    dY = u.dev(u.rbf(torch.randn(M, N, generator=g)), u.BF)
    ref_f = A.float() @ W.float().t() + b
    ref_d = dY.float() @ W.float()
    tol_f, tol_d = 2.0 ** -7 * float(ref_f.abs().max()), 2.0 ** -7 * float(ref_d.abs().max())
    outs = [torch.empty(M, N, dtype=u.BF, device="cuda") for _ in range(REPS)]
    dxs = [torch.empty(M, K, dtype=u.BF, device="cuda") for _ in range(REPS)]
    noise = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for i in range(REPS):
        u.call("vg_linear_fwd", u.ptr(A), u.ptr(W), u.ptr(b), None, u.ptr(outs[i]), None, None, M, N, K, 0, 0.0, u.stream())
        if i % 3 == 0:
            noise.add_(1)  # an unrelated streaming kernel between launches changes what is resident in L2 / in flight
        u.call("vg_linear_dgrad", u.ptr(dY), u.ptr(W), u.ptr(dxs[i]), M, N, K, 0, None, None, 0.0, u.stream())
    u.sync()
    for i in range(REPS):
        assert float((outs[i].float() - ref_f).abs().max()) <= tol_f, f"forward, launch {i}"
        assert float((dxs[i].float() - ref_d).abs().max()) <= tol_d, f"input gradient, launch {i}"
        assert torch.equal(outs[i], outs[0]) and torch.equal(dxs[i], dxs[0]), f"launch {i} differs bitwise from launch 0"


@pytest.mark.parametrize("M", [1312, 16640])
def test_gelu_and_byte_derivative_epilogues_repeatable(M):
    """The weights-in-registers kernel's seam wait (vmcnt(4 + EST), csrc/gemm_wr.hip) counts the stores of the previous tile's
    epilogue, and the epilogues differ per instantiation: the two the step runs beside the plain ones - fc1 + GELU with the
    one-byte derivative as second output (<0, 1, 2>) and the fc2 input gradient times the decoded byte (<1, 8, 0>) - get the same
    screen: many launches, other traffic in flight, every result bit-equal to the first and right against fp32."""
    u = _u()
    E, H = 384, 768
    g = torch.Generator().manual_seed(M)
    A = u.dev(u.rbf(torch.randn(M, E, generator=g)), u.BF)
    W1 = u.dev(u.rbf(torch.randn(H, E, generator=g) / math.sqrt(E)), u.BF)
    b1 = u.dev(torch.randn(H, generator=g) * 0.1)
    dY = u.dev(u.rbf(torch.randn(M, E, generator=g)), u.BF)
    W2 = u.dev(u.rbf(torch.randn(E, H, generator=g) / math.sqrt(H)), u.BF)
    pre = A.float() @ W1.float().t() + b1
    ref_g = torch.nn.functional.gelu(pre)
    outs = [torch.empty(M, H, dtype=u.BF, device="cuda") for _ in range(REPS)]
    codes = [torch.empty(M, H, dtype=torch.uint8, device="cuda") for _ in range(REPS)]
    dxs = [torch.empty(M, H, dtype=u.BF, device="cuda") for _ in range(REPS)]
    noise = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    for i in range(REPS):
        u.call("vg_linear_gelu_fwd", u.ptr(A), u.ptr(W1), u.ptr(b1), u.ptr(outs[i]), u.ptr(codes[i]), M, H, E, u.stream())
        if i % 3 == 0:
            noise.add_(1)
        u.call("vg_linear_dgrad", u.ptr(dY), u.ptr(W2), u.ptr(dxs[i]), M, E, H, 8, u.ptr(codes[i]), None, 0.0, u.stream())
    u.sync()
    ref_d = (dY.float() @ W2.float()) * ((codes[0].float() - 27.0) * 0.005)
    tol_g, tol_d = 2.0 ** -7 * float(ref_g.abs().max()), 2.0 ** -7 * float(ref_d.abs().max())
    for i in range(REPS):
        assert float((outs[i].float() - ref_g).abs().max()) <= tol_g, f"gelu, launch {i}"
        assert float((dxs[i].float() - ref_d).abs().max()) <= tol_d, f"input gradient x byte derivative, launch {i}"
        assert torch.equal(outs[i], outs[0]) and torch.equal(codes[i], codes[0]) and torch.equal(dxs[i], dxs[0]), f"launch {i} differs bitwise"


# stages per K slice: 22/22/21, 26, 52, 1, 2, 3, 4 and 5 (the ring holds four: prologue-only, one refill, steady state)
@pytest.mark.parametrize("M,N,K,splits", [(2080, 1152, 384, 3), (4160, 768, 1536, 5), (16640, 384, 768, 10), (96, 384, 384, 3),
                                          (256, 384, 384, 4), (288, 128, 384, 3), (512, 256, 768, 4), (800, 384, 384, 5),
                                          (4160, 512, 1024, 8)])  # last: the 128 x 512-tile variant
def test_weight_gradient_repeatable(M, N, K, splits):
    u = _u()
    g = torch.Generator().manual_seed(M + N + K + splits)
    dY = u.dev(u.rbf(torch.randn(M, N, generator=g)), u.BF)
    X = u.dev(u.rbf(torch.randn(M, K, generator=g)), u.BF)
    ref = dY.float().t() @ X.float()
    tol = 3e-5 * float(ref.abs().max())
    dWs = [torch.empty(N, K, device="cuda") for _ in range(REPS)]
    slab = torch.empty(splits * N * K, device="cuda")
    for i in range(REPS):
        u.call("vg_linear_wgrad", u.ptr(dY), u.ptr(X), u.ptr(dWs[i]), u.ptr(slab), slab.numel(), M, N, K, splits, 0, u.stream())
    u.sync()
    for i in range(REPS):
        assert float((dWs[i] - ref).abs().max()) <= tol, f"launch {i}"
        assert torch.equal(dWs[i], dWs[0]), f"launch {i} differs bitwise from launch 0"
