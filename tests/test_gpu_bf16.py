"""GPU parity of the bf16 kernels (bfloat16 activations, float32 accumulation on
v_mfma_f32_32x32x16_bf16) against the float64 oracle evaluated on the SAME bf16-rounded inputs and
weights, so that what is measured is the kernels' arithmetic and their single output rounding.

Tolerances (stated here, used below):
  * bf16 outputs (activations, saved derivatives, data-gradients): relative L2 <= 3e-3 and max abs
    <= 2^-7 * max|ref| (one round-to-nearest to 8 significant bits is <= 2^-9 relative per element);
  * float32 outputs (weight / bias gradients, 2-channel data-gradients, alpha gradient): relative
    L2 <= 2e-5 (exact bf16 products, float32 accumulation, fixed-order reductions).
"""
import numpy as np
import pytest
import torch

import music_generator_amd as M
from music_generator_amd import layers as L
from music_generator_amd import ops
from music_generator_amd.graph import Input
from oracle import ref_step as R

from helpers import rel_l2

pytestmark = pytest.mark.gpu

TOL_B, TOL_F = 3e-3, 2e-5


def _bf(a):
    """Round a float array to bf16 (nearest even) and return it as float32 numpy."""
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch.bfloat16).float().numpy()


def _dev_b(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda").to(torch.bfloat16).contiguous()


def _dev_f(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to("cuda").contiguous()


def _close_b(a, b, what):
    a = a.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(a).all(), what
    err = rel_l2(a, b)
    assert err <= TOL_B, "%s: rel-L2 %.3e" % (what, err)
    assert np.abs(a - b).max() <= 2.0 ** -7 * np.abs(b).max() + 1e-30, "%s: max abs %.3e" % (what, np.abs(a - b).max())


def _close_f(a, b, what, tol=TOL_F):
    a = a.float().cpu().numpy().astype(np.float64)
    assert np.isfinite(a).all(), what
    err = rel_l2(a, b)
    assert err <= tol, "%s: rel-L2 %.3e" % (what, err)


BCONV_CASES = [
    # (H, W, Cin, Cout, ksize, strides, padding, batch)
    (8, 64, 32, 32, (3, 3), (1, 1), "same", 1),
    (16, 70, 32, 32, (3, 3), (1, 1), "same", 2),       # ragged right edge, two tile rows
    (12, 45, 32, 32, (3, 3), (1, 1), "same", 3),       # 4 valid rows in the last tile row
    (16, 150, 128, 32, (3, 3), (1, 1), "same", 2),     # four input-channel chunks per tile
    (8, 50, 32, 32, (2, 2), (1, 1), "same", 2),        # pad 0 before / 1 after
    (24, 95, 32, 32, (2, 2), (1, 1), "same", 1),
    (16, 52, 32, 32, (2, 2), (2, 2), "valid", 3),      # stride-2 gather forward, depth-to-space data-gradient
    (32, 260, 32, 32, (2, 2), (2, 2), "valid", 2),
    (8, 41, 32, 128, (1, 1), (1, 1), "same", 2),       # four output-channel chunks per tile
    (4, 75, 32, 128, (1, 1), (1, 1), "same", 2),       # fewer rows than a tile
    (32, 700, 32, 32, (3, 3), (1, 1), "same", 12),     # 528 tiles: workgroups walk several tiles
]


@pytest.mark.parametrize("case", BCONV_CASES)
def test_bf16_conv_forward_dgrad_wgrad(case):
    h, w, ci, co, ks, st, pad, n = case
    M.clear_session()
    layer = L.Conv2D(co, ks, strides=st, padding=pad, kernel_initializer="he_normal")
    layer(Input(shape=(h, w, ci)))
    rng = np.random.default_rng(h * 1000 + w + ci)
    alpha = 7.0
    x = _bf(rng.uniform(-1, 1, (n, h, w, ci)))
    kern = rng.standard_normal(layer.kernel.shape).astype(np.float32) / np.sqrt(ks[0] * ks[1] * ci)
    bias = rng.uniform(-0.3, 0.3, (co,)).astype(np.float32)
    ho, wo = layer._out_hw(h, w)
    dy = _bf(rng.standard_normal((n, ho, wo, co)))
    mul = _bf(rng.uniform(0.1, 1.0, (n, h, w, ci)))

    # oracle on the rounded operands
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    kt = torch.tensor(_bf(kern), dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(bias, dtype=torch.float64, requires_grad=True)
    pre = R.conv2d(xt, kt, bt, strides=st, padding=pad)
    yo = R.soft_rectifier(pre, alpha)
    der = (1.0 - torch.tanh(pre) ** 2 + 1.0 / alpha).detach().numpy()
    # weight gradient / data-gradient of the bare contraction for the gradient g = dy
    gx, gk, gb = torch.autograd.grad((pre * torch.tensor(dy, dtype=torch.float64)).sum(), [xt, kt, bt])

    xd, kd, bd = _dev_b(x), _dev_f(kern), _dev_f(bias)
    fd = layer._desc("f", n, h, w)
    assert ops.conv_bf16_supported(fd)
    y = torch.empty((n, ho, wo, co), dtype=torch.bfloat16, device="cuda")
    dact = torch.empty_like(y)
    ops.conv_gemm_bf16(fd, xd, kd, bd, y, dact=dact, act_alpha=alpha)
    _close_b(y, yo.detach().numpy(), "output")
    _close_b(dact, der, "saved derivative")

    # no activation, no bias
    fd0 = layer._fwd_desc(n, h, w)
    fd0.nbias = 0
    y0 = torch.empty_like(y)
    ops.conv_gemm_bf16(fd0, xd, kd, None, y0)
    _close_b(y0, (pre - bt).detach().numpy(), "plain output")

    # data-gradient with the multiplier epilogue, and without
    dd = layer._desc("d", n, h, w)
    assert ops.conv_bf16_supported(dd)
    dyd = _dev_b(dy)
    dx = torch.empty((n, h, w, ci), dtype=torch.bfloat16, device="cuda")
    ops.conv_gemm_bf16(dd, dyd, kd, None, dx, ymul=_dev_b(mul))
    _close_b(dx, gx.numpy() * mul, "data-gradient x multiplier")
    ops.conv_gemm_bf16(dd, dyd, kd, None, dx)
    _close_b(dx, gx.numpy(), "data-gradient")

    # weight / bias gradient (added to what the buffers hold)
    dw = torch.full(layer.kernel.shape, 0.5, dtype=torch.float32, device="cuda").view(-1)
    db = torch.full((co,), -0.25, dtype=torch.float32, device="cuda")
    ops.conv_wgrad_bf16(fd, xd, dyd, dw, db)
    _close_f(dw.view(layer.kernel.shape) - 0.5, gk.numpy(), "weight gradient", 5e-5)
    _close_f(db + 0.25, gb.numpy(), "bias gradient", 5e-5)
    torch.cuda.synchronize()


def test_bf16_from_audio_conv():
    """1x1 convolution 2 float32 channels -> 128 bf16 channels, its gradients."""
    rng = np.random.default_rng(5)
    npos, alpha = 3 * 8 * 77, 9.0
    x = rng.uniform(-1, 1, (npos, 2)).astype(np.float32)
    w = rng.standard_normal((2, 128)).astype(np.float32)
    b = rng.uniform(-0.3, 0.3, (128,)).astype(np.float32)
    g = _bf(rng.standard_normal((npos, 128)))
    pre = x.astype(np.float64) @ w.astype(np.float64) + b
    yo = np.tanh(pre) + pre / alpha
    do = 1 - np.tanh(pre) ** 2 + 1 / alpha
    xd, wd, bd, gd = _dev_f(x), _dev_f(w), _dev_f(b), _dev_b(g)
    y = torch.empty((npos, 128), dtype=torch.bfloat16, device="cuda")
    dact = torch.empty_like(y)
    ops.pointwise2_bf16_fwd(xd, wd, bd, y, dact, npos, alpha)
    _close_b(y, yo, "output")
    _close_b(dact, do, "derivative")
    ops.pointwise2_bf16_fwd(xd, wd, bd, y, None, npos, 0.0)
    _close_b(y, pre, "plain output")
    dw = torch.zeros((256,), dtype=torch.float32, device="cuda")
    db = torch.zeros((128,), dtype=torch.float32, device="cuda")
    ops.pointwise2_bf16_wgrad(xd, gd, dw, db, npos)
    _close_f(dw.view(2, 128), x.astype(np.float64).T @ g.astype(np.float64), "weight gradient")
    _close_f(db, g.astype(np.float64).sum(0), "bias gradient")
    dx = torch.full((npos, 2), 1.0, dtype=torch.float32, device="cuda")
    ops.pointwise_to2_bf16(gd, wd, dx, npos, True)
    _close_f(dx - 1.0, g.astype(np.float64) @ w.astype(np.float64).T, "data-gradient (accumulated)")
    ops.pointwise_to2_bf16(gd, wd, dx, npos, False)
    _close_f(dx, g.astype(np.float64) @ w.astype(np.float64).T, "data-gradient")


@pytest.mark.parametrize("alpha", [0.0, 0.37, 1.0])
def test_bf16_weighted_sum(alpha):
    rng = np.random.default_rng(11)
    shape = (3, 8, 50, 128)
    a, b, dy = (_bf(rng.standard_normal(shape)) for _ in range(3))
    ad, bd, dyd = _dev_b(a), _dev_b(b), _dev_b(dy)
    al = torch.tensor([alpha], dtype=torch.float32, device="cuda")
    y = torch.empty(shape, dtype=torch.bfloat16, device="cuda")
    ops.weighted_sum_bf16_fwd(ad, bd, al, y)
    _close_b(y, (1 - alpha) * a.astype(np.float64) + alpha * b, "output")
    if alpha == 0.0:
        assert torch.equal(y, ad)         # models.py:34 at alpha = 0 returns input 0 exactly
    if alpha == 1.0:
        assert torch.equal(y, bd)
    da, db = torch.empty_like(y), torch.empty_like(y)
    dal = torch.zeros((1,), dtype=torch.float32, device="cuda")
    ops.weighted_sum_bf16_bwd(dyd, ad, bd, al, da, db, dal)
    _close_b(da, (1 - alpha) * dy.astype(np.float64), "da")
    _close_b(db, alpha * dy.astype(np.float64), "db")
    _close_f(dal, np.array([(dy.astype(np.float64) * (b.astype(np.float64) - a)).sum()]), "dalpha", 1e-5)


def test_bf16_casts_round_trip():
    x = torch.randn((5, 1003), device="cuda")
    b = ops.cast_to_bf16(x)
    assert torch.equal(b, x.to(torch.bfloat16))
    assert torch.equal(ops.cast_to_f32(b), b.float())
