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
    (64, 1500, 32, 32, (3, 3), (1, 1), "same", 4),     # 768 tiles of the fused backward: three per workgroup
    (64, 1500, 32, 32, (2, 2), (1, 1), "same", 4),
    (18, 333, 128, 32, (3, 3), (1, 1), "same", 3),     # fused backward from 128 channels: ragged 4 x 32 tiles
    (64, 1000, 128, 32, (3, 3), (1, 1), "same", 1),    # ... 512 tiles, two per workgroup
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
    # ... and with the factor recovered from a stored activation OUTPUT (no derivative tensor): values of tanh(v) +
    # v/alpha as bf16, some of them rounded up past 1, where the recovered derivative is held at 1/alpha
    act_out = _bf(np.tanh(3.0 * mul - 1.5) + (3.0 * mul - 1.5) / alpha)
    act_out.reshape(-1)[::97] = _bf(np.float32(1.0 + 2.0 ** -7))
    ops.conv_gemm_bf16(dd, dyd, kd, None, dx, ymul=_dev_b(act_out), act_alpha=alpha)
    rec = np.maximum(1.0 + 1.0 / alpha - act_out.astype(np.float64) ** 2, 1.0 / alpha)
    assert rec.min() == 1.0 / alpha
    _close_b(dx, gx.numpy() * rec, "data-gradient x recovered derivative")

    # weight / bias gradient (added to what the buffers hold)
    dw = torch.full(layer.kernel.shape, 0.5, dtype=torch.float32, device="cuda").view(-1)
    db = torch.full((co,), -0.25, dtype=torch.float32, device="cuda")
    ops.conv_wgrad_bf16(fd, xd, dyd, dw, db)
    _close_f(dw.view(layer.kernel.shape) - 0.5, gk.numpy(), "weight gradient", 5e-5)
    _close_f(db + 0.25, gb.numpy(), "bias gradient", 5e-5)

    # both gradients in one pass (32 -> 32 'same' layers and the 3x3 one from 128 channels): against the oracle and against the two kernels above, with
    # the derivative recovered from the layer input as the stored output of the layer before, and without it
    if ops.conv_bwd_bf16_supported(fd):
        xact = act_out if act_out.shape == x.shape else x
        xa = _dev_b(xact)
        xta = torch.tensor(xact, dtype=torch.float64, requires_grad=True)
        pre_a = R.conv2d(xta, kt, bt, strides=st, padding=pad)
        gxa, gka, gba = torch.autograd.grad((pre_a * torch.tensor(dy, dtype=torch.float64)).sum(), [xta, kt, bt])
        rec_a = np.maximum(1.0 + 1.0 / alpha - xact.astype(np.float64) ** 2, 1.0 / alpha)
        for a_alpha, want_dx in ((alpha, gxa.numpy() * rec_a), (0.0, gxa.numpy())):
            dw2 = torch.full(layer.kernel.shape, 0.5, dtype=torch.float32, device="cuda").view(-1)
            db2 = torch.full((co,), -0.25, dtype=torch.float32, device="cuda")
            dx2 = torch.empty((n, h, w, ci), dtype=torch.bfloat16, device="cuda")
            ops.conv_bwd_bf16(fd, xa, dyd, kd, dw2, db2, dx2, a_alpha)
            _close_b(dx2, want_dx, "fused: data gradient (act_alpha %g)" % a_alpha)
            _close_f(dw2.view(layer.kernel.shape) - 0.5, gka.numpy(), "fused: weight gradient", 5e-5)
            _close_f(db2 + 0.25, gba.numpy(), "fused: bias gradient", 5e-5)
            dx1 = torch.empty_like(dx2)
            ops.conv_gemm_bf16(dd, dyd, kd, None, dx1, ymul=xa if a_alpha else None, act_alpha=a_alpha)
            # same products, another order of the float32 sums: a rounding of the bf16 result may flip
            d12 = (dx1.float() - dx2.float()).abs()
            assert float(d12.max()) <= 2.0 ** -7 * float(dx1.float().abs().max()), float(d12.max())
            assert float((d12 > 0).float().mean()) < 0.02
    else:
        assert not (co == 32 and st == (1, 1) and (ci == 32 or ks == (3, 3)))
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,h,w", [(2, 16, 150), (3, 18, 333), (1, 64, 1000), (2, 8, 75)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_bf16_conv_dgrad_to_audio_equals_two_launches(n, h, w, accumulate):
    """mgx_conv_dgrad_to_audio_bf16 (data gradient of the 3x3 convolution 128 -> 32 times the derivative recovered from
    the from-audio layer's stored output, contracted with the from-audio kernel in registers) against
    mgx_conv_gemm_bf16 (ymul) followed by mgx_pointwise_to2_bf16: the same bf16-rounded 128-channel values, another
    order of the float32 sums over the 128 channels."""
    M.clear_session()
    conv = L.Conv2D(32, (3, 3), padding="same", kernel_initializer="he_normal")
    conv(Input(shape=(h, w, 128)))
    rng = np.random.default_rng(n * 100 + h + w)
    alpha = 20000.0
    g = _dev_b(rng.standard_normal((n, h, w, 32)))
    y128 = _dev_b(np.tanh(rng.uniform(-3, 3, (n, h, w, 128))))
    fa_w = _dev_f(rng.standard_normal((2, 128)).astype(np.float32))
    dd = conv._desc("d", n, h, w)
    g128 = torch.empty((n, h, w, 128), dtype=torch.bfloat16, device="cuda")
    ops.conv_gemm_bf16(dd, g, conv.kernel.value(), None, g128, ymul=y128, act_alpha=alpha)
    start = torch.from_numpy(rng.standard_normal((n, h, w, 2)).astype(np.float32)).cuda()
    want = start.clone()
    ops.pointwise_to2_bf16(g128, fa_w, want, n * h * w, accumulate)
    got = start.clone() if accumulate else torch.full_like(start, float("nan"))
    ops.conv_dgrad_to_audio_bf16(dd, g, conv.kernel.value(), y128, alpha, fa_w, got, accumulate)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got).all()), "positions left unwritten"
    scale = float((want - (start if accumulate else 0)).abs().max())
    assert float((got - want).abs().max()) <= 2e-6 * scale + 1e-6, (float((got - want).abs().max()), scale)


def test_bf16_discriminator_input_gradient_with_and_without_fused_audio_dgrad(bf16_precision):
    """The backward sweep of the generator step through a fade-in discriminator (no weight gradients, gradient of
    the input wanted) with the 3x3 + from-audio data gradients as one launch per grown block, against the
    layer-by-layer sweep."""
    from music_generator_amd.graph import Model
    d, g, comp, W = _build(4)
    model = d[3][1]
    for layer in model.layers:
        if isinstance(layer, M.WeightedSum):
            M.backend.set_value(layer.alpha, 0.37)
    real, _ = R.synthetic_inputs(3, 4, seed=5)
    x = torch.from_numpy(real).cuda()
    calls = []
    orig = ops.conv_dgrad_to_audio_bf16
    ops.conv_dgrad_to_audio_bf16 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    chains = Model.use_conv_chains
    try:
        outs = []
        for fused in (True, False):
            Model.use_conv_chains = fused
            y, tape = model.forward(x, save=True)
            dy = torch.ones_like(y)
            outs.append(model.backward(tape, dy, need_input_grad=True, need_weight_grads=False).clone())
            if fused:
                n_fused = len(calls)
        assert n_fused >= 1 and len(calls) == n_fused
    finally:
        Model.use_conv_chains = chains
        ops.conv_dgrad_to_audio_bf16 = orig
    a, b = outs
    assert float((a - b).norm()) <= 1e-5 * float(b.norm()), (float((a - b).norm()), float(b.norm()))


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
    # the same with the producers' activation derivatives recovered from a and b (their stored outputs)
    ya, yb = _bf(np.tanh(a)), _bf(np.tanh(b) + b / 11000.0)
    yad, ybd = _dev_b(ya), _dev_b(yb)
    dal2 = torch.zeros((1,), dtype=torch.float32, device="cuda")
    ops.weighted_sum_bf16_bwd(dyd, yad, ybd, al, da, db, dal2, act_alpha_a=40000.0, act_alpha_b=11000.0)
    fa = np.maximum(1 + 1 / 40000.0 - ya.astype(np.float64) ** 2, 1 / 40000.0)
    fb = np.maximum(1 + 1 / 11000.0 - yb.astype(np.float64) ** 2, 1 / 11000.0)
    _close_b(da, _bf((1 - alpha) * dy).astype(np.float64) * fa, "da x recovered derivative")
    _close_b(db, _bf(alpha * dy).astype(np.float64) * fb, "db x recovered derivative")
    _close_f(dal2, np.array([(dy.astype(np.float64) * (yb.astype(np.float64) - ya)).sum()]), "dalpha", 1e-5)
    da0 = torch.empty_like(da)
    ops.weighted_sum_bf16_bwd(dyd, yad, ybd, al, da0, None, None)
    assert torch.equal(da, ops.mul_dact_bf16(da0, yad, 40000.0))      # rounded where a multiply launch of its own rounds


def test_bf16_casts_round_trip():
    x = torch.randn((5, 1003), device="cuda")
    b = ops.cast_to_bf16(x)
    assert torch.equal(b, x.to(torch.bfloat16))
    assert torch.equal(ops.cast_to_f32(b), b.float())


# --------------------------------------------------------------------------------------------
# whole models / the train step with the grow blocks of the discriminator on the bf16 kernels
# --------------------------------------------------------------------------------------------
@pytest.fixture
def bf16_precision():
    M.backend.set_precision("bfloat16")
    yield
    M.backend.set_precision("float32")


def _build(n_blocks, trainable_alpha=True):
    from test_gpu_step import build
    return build(n_blocks, trainable_alpha)


def _d_pass(model, x, need_input_grad=True):
    y, tape = model.forward(x, save=True)
    flat = model.bind_grads()
    flat.zero_()
    dx = model.backward(tape, torch.ones_like(y, dtype=torch.float32), need_input_grad=need_input_grad)
    torch.cuda.synchronize()
    grads = {v.name: v.grad.detach().cpu().numpy().copy() for v in model.trainable_variables}
    return y.float().cpu().numpy(), dx.float().cpu().numpy(), grads


def _oracle_blocks(W, n_blocks, stage, fade, x, alpha, dy, emulate):
    """The grow blocks of the discriminator (everything in front of the 4 x 750 base block): output,
    input gradient and weight gradients for the output gradient dy, in the float64 oracle, optionally
    with bf16 storage emulated (oracle/bf16_emulation.py)."""
    import contextlib
    from oracle import bf16_emulation as E
    names = R.layer_names(n_blocks)
    Wt = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in W.items()}
    al = None
    if fade:
        al = torch.tensor(float(alpha), dtype=torch.float64, requires_grad=True)
        Wt[names["d"][stage]["ws"] + "/alpha"] = al
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    with (E.bf16_storage() if emulate else contextlib.nullcontext()):
        h = R._d_block(Wt, names, stage, R._from_audio(Wt, names, stage, xt))      # models.py:256-281
        if fade:
            h = R.weighted_sum(R._from_audio(Wt, names, stage - 1, R.avgpool2(xt)), h, al)   # models.py:292-297
        for s in range(stage - 1, 0, -1):
            h = R._d_block(Wt, names, s, h)
        d0 = names["d"][0]
        base = d0["convs"] + [d0["dense"]] + ([] if (fade and stage == 1) else [d0["from_audio"]])
        excl = {n + kind for n in base for kind in ("/kernel", "/bias")}
        keys = [k for k in R.trainable_names(names, stage, fade, "d") if k not in excl]
        gs = torch.autograd.grad((h * torch.tensor(dy, dtype=torch.float64)).sum(), [xt] + [Wt[k] for k in keys],
                                 allow_unused=True)
    return h.detach().numpy(), gs[0].numpy(), {k: g.numpy() for k, g in zip(keys, gs[1:]) if g is not None}


def _blocks_model(model):
    """Model from the discriminator's input to the tensor that enters its 4 x 750 base block."""
    from music_generator_amd.graph import Model
    for node in model._nodes:
        if type(node.layer).__name__ == "Conv2D" and node.inputs and tuple(node.inputs[0].shape[1:]) == (4, 750, 128):
            return Model(model.input, node.inputs[0])
    raise AssertionError("no base block found")


def test_bf16_chain_layer_by_layer_is_exact_to_rounding():
    """Every layer of a grow block on the bf16 kernels, fed with the DEVICE's own input, against the
    float64 oracle of that one layer on the same input: the stored bf16 value equals the correctly
    rounded oracle value except where float32 arithmetic lands on the other side of a rounding boundary
    (bound: 0.1 % of the elements, each off by one bf16 step; measured ~1e-4)."""
    from oracle import bf16_emulation as E
    stage = 1
    d, g, comp, W = _build(stage + 1)
    model = _blocks_model(d[stage][0])
    real, _ = R.synthetic_inputs(stage, 4, seed=0)
    names = R.layer_names(stage + 1)
    Wt = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in W.items()}
    blk = names["d"][stage]
    _, soft = R.activation_alphas(stage)
    seq = [(blk["from_audio"], {})] + [(c, ({"strides": (2, 2), "padding": "valid"} if i == 8 else {}))
                                        for i, c in enumerate(blk["convs"])]
    M.backend.set_precision("bfloat16")
    try:
        x = torch.from_numpy(real).cuda()
        for op, (name, kw) in zip(model._plan, seq):
            assert op.node.layer.name == name
            xin = torch.tensor(x.float().cpu().numpy(), dtype=torch.float64)
            x, _ = op.node.layer.forward([x], False, act_alpha=op.act_node.layer.alpha_value)
            assert x.dtype == torch.bfloat16
            want = _bf(E._conv_sr(Wt, name, xin, soft, **kw).numpy())
            got = x.float().cpu().numpy()
            off = got != want
            assert off.mean() <= 1e-3, (name, off.mean())
            assert np.abs(got - want).max() <= 2.0 ** -7 * np.abs(want).max(), name
    finally:
        M.backend.set_precision("float32")


@pytest.mark.parametrize("stage,twin", [(1, 0), (2, 0), (2, 1), (3, 1)])
def test_bf16_grow_blocks_match_oracle(stage, twin):
    """The chain of grow blocks (up to 3 blocks, 35 convolutions + the fade-in blend) on the bf16 kernels,
    forward and backward for a random output gradient, against the float64 oracle, plain and with bf16
    STORAGE emulated at the same points (oracle/bf16_emulation.py).  A rounding that falls on the other
    side of a bf16 boundary perturbs ~10^4 products of the next layer by far more than float32 arithmetic
    does, so two evaluations with bf16 storage decorrelate layer by layer (measured: 1e-4 of the elements
    differ after the first layer, 27 % after eleven) and agree no better than each agrees with the exact
    result: the distance to both oracles is the storage noise, ~3e-3 per block on the output and ~1e-2 per
    block on gradients.  Stated bound for output, input gradient and every weight gradient: relative L2
    <= 2e-2 x number of blocks (3e-2 x blocks for bias gradients, sums of signed values over all
    positions; 3e-1 for the scalar fade-in alpha, a cancelling sum).  The kernels'
    own arithmetic is pinned exactly by test_bf16_conv_forward_dgrad_wgrad and
    test_bf16_chain_layer_by_layer_is_exact_to_rounding."""
    d, g, comp, W = _build(stage + 1)
    full = d[stage][twin]
    for layer in full.layers:
        if isinstance(layer, M.WeightedSum):
            M.backend.set_value(layer.alpha, 0.37)
    model = _blocks_model(full)
    real, _ = R.synthetic_inputs(stage, 4, seed=0)
    rng = np.random.default_rng(3)
    dy = rng.standard_normal((4, 4, 750, 128)).astype(np.float32)
    M.backend.set_precision("bfloat16")
    try:
        y, tape = model.forward(torch.from_numpy(real).cuda(), save=True)
        model.bind_grads().zero_()
        dx = model.backward(tape, torch.from_numpy(dy).cuda(), need_input_grad=True)
        torch.cuda.synchronize()
    finally:
        M.backend.set_precision("float32")
    yb, dxb = y.float().cpu().numpy(), dx.float().cpu().numpy()
    gb = {v.name: v.grad.cpu().numpy().reshape(v.shape).copy() for v in model.trainable_variables}
    tol = 2e-2 * stage
    for emulate in (True, False):
        yo, dxo, go = _oracle_blocks(W, stage + 1, stage, bool(twin), real, 0.37, dy, emulate)
        assert set(go) == set(gb), (sorted(set(go) ^ set(gb)))
        errs = {k: rel_l2(gb[k], go[k]) for k in go}
        print("emulated" if emulate else "plain", "oracle: output", rel_l2(yb, yo), "dx", rel_l2(dxb, dxo),
              "worst weight gradient", max(errs.items(), key=lambda kv: kv[1]))
        assert rel_l2(yb, yo) <= tol and rel_l2(dxb, dxo) <= tol
        for k, err in errs.items():
            bound = 3e-1 if k.endswith("/alpha") else (1.5 * tol if k.endswith("/bias") else tol)
            assert err <= bound, "%s: rel-L2 %.3e" % (k, err)


def _oracle_d_pass(W, n_blocks, stage, fade, x, alpha, emulate):
    """Discriminator forward + gradients of sum(logits) in the float64 oracle, optionally with bf16
    storage emulated (oracle/bf16_emulation.py)."""
    import contextlib
    from oracle import bf16_emulation as E
    names = R.layer_names(n_blocks)
    Wt = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in W.items()}
    al = None
    if fade:
        al = torch.tensor(float(alpha), dtype=torch.float64, requires_grad=True)
        Wt[names["d"][stage]["ws"] + "/alpha"] = al
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    with (E.bf16_storage() if emulate else contextlib.nullcontext()):
        y = R.discriminator(Wt, names, stage, fade, xt, al)
        keys = R.trainable_names(names, stage, fade, "d")
        gs = torch.autograd.grad(y.sum(), [xt] + [Wt[k] for k in keys])
    return y.detach().numpy(), gs[0].numpy(), {k: g.numpy() for k, g in zip(keys, gs[1:])}


@pytest.mark.parametrize("stage,twin", [(1, 0), (2, 1), (3, 1)])
def test_bf16_discriminator_vs_plain_oracle(stage, twin):
    """The whole discriminator (bf16 grow blocks + float32 base block and head) against the plain float64
    oracle.  Its output activation tanh(z)+z/alpha is saturated for these weights, so every gradient of
    sample n carries the factor sech^2(z_n), whose relative error is 2*dz_n: the bf16 storage error of the
    logit pre-activation (~4e-3 abs) alone moves all gradients by ~1e-2, on top of the per-layer cost seen
    in test_bf16_grow_blocks_match_emulated_oracle.  Stated bounds: logits abs 2e-2, gradients relative L2
    <= 8e-2 (measured up to 5.5e-2 with 3 grow blocks), 1.2e-1 for bias gradients (per-channel sums over the
    whole activation that cancel to a fraction of their terms: measured 5e-2 .. 9e-2 depending on nothing
    but the float32 summation order inside the convolution kernels), 3e-1 for the scalar fade-in alpha."""
    d, g, comp, W = _build(stage + 1)
    model = d[stage][twin]
    for layer in model.layers:
        if isinstance(layer, M.WeightedSum):
            M.backend.set_value(layer.alpha, 0.37)
    real, _ = R.synthetic_inputs(stage, 4, seed=0)
    M.backend.set_precision("bfloat16")
    try:
        yb, dxb, gb = _d_pass(model, torch.from_numpy(real).cuda())
    finally:
        M.backend.set_precision("float32")
    yo, dxo, go = _oracle_d_pass(W, stage + 1, stage, bool(twin), real, 0.37, False)
    errs = {k: rel_l2(gb[k], go[k]) for k in go}
    print("logits", np.abs(yb - yo).max(), "dx", rel_l2(dxb, dxo), "worst weight gradient",
          max(errs.items(), key=lambda kv: kv[1]))
    assert np.abs(yb - yo).max() <= 2e-2
    assert rel_l2(dxb, dxo) <= 8e-2
    for k, err in errs.items():
        bound = 3e-1 if k.endswith("/alpha") else (1.2e-1 if k.endswith("/bias") else 8e-2)
        assert err <= bound, "%s: rel-L2 %.3e" % (k, err)


@pytest.mark.parametrize("stage,twin", [(2, 1)])
def test_bf16_derivative_recovered_from_output_against_stored_derivative(stage, twin):
    """backend.BF16_SAVE_DERIVATIVE: the default (False) keeps no derivative tensor and recovers 1 - tanh^2 + 1/alpha
    from the stored bf16 output; True stores it as a second bf16 tensor.  Same forward bit for bit; the gradients
    of the two settings differ by the rounding of y that enters 1 - y^2 (absolute 2 |y| 2^-9 per element against
    the 2^-9 relative of a stored derivative).  Stated bound between the two settings, and of each against the
    oracle that emulates it: relative L2 <= 2e-2 x blocks (3e-2 x blocks for bias gradients, 3e-1 for alpha), the
    bound of test_bf16_grow_blocks_match_oracle."""
    from oracle import bf16_emulation as E
    from music_generator_amd import backend as K
    real, _ = R.synthetic_inputs(stage, 4, seed=0)
    rng = np.random.default_rng(3)
    dy = rng.standard_normal((4, 4, 750, 128)).astype(np.float32)
    runs = {}
    for keep in (False, True):
        d, g, comp, W = _build(stage + 1)
        full = d[stage][twin]
        for layer in full.layers:
            if isinstance(layer, M.WeightedSum):
                M.backend.set_value(layer.alpha, 0.37)
        model = _blocks_model(full)
        M.backend.set_precision("bfloat16")
        K.BF16_SAVE_DERIVATIVE = keep
        try:
            torch.cuda.reset_peak_memory_stats()
            base = torch.cuda.memory_allocated()
            y, tape = model.forward(torch.from_numpy(real).cuda(), save=True)
            held = torch.cuda.memory_allocated() - base
            model.bind_grads().zero_()
            dx = model.backward(tape, torch.from_numpy(dy).cuda(), need_input_grad=True)
            torch.cuda.synchronize()
        finally:
            M.backend.set_precision("float32")
            K.BF16_SAVE_DERIVATIVE = False
        grads = {v.name: v.grad.cpu().numpy().reshape(v.shape).copy() for v in model.trainable_variables}
        E.SAVE_DERIVATIVE = keep
        try:
            yo, dxo, go = _oracle_blocks(W, stage + 1, stage, bool(twin), real, 0.37, dy, True)
        finally:
            E.SAVE_DERIVATIVE = False
        runs[keep] = (y.float().cpu().numpy(), dx.float().cpu().numpy(), grads, held)
        tol = 2e-2 * stage
        assert rel_l2(runs[keep][0], yo) <= tol and rel_l2(runs[keep][1], dxo) <= tol
        for k in go:
            bound = 3e-1 if k.endswith("/alpha") else (1.5 * tol if k.endswith("/bias") else tol)
            assert rel_l2(grads[k], go[k]) <= bound, "%s (stored derivative: %s): %.3e" % (k, keep, rel_l2(grads[k], go[k]))
    (y0, dx0, g0, held0), (y1, dx1, g1, held1) = runs[False], runs[True]
    assert np.array_equal(y0, y1)
    worst = max((rel_l2(g0[k], g1[k]), k) for k in g0 if not k.endswith("/alpha"))
    print("recovered against stored derivative: dx", rel_l2(dx0, dx1), "worst weight gradient", worst,
          "tape bytes", held0, "against", held1)
    tol = 2e-2 * stage
    assert rel_l2(dx0, dx1) <= tol and worst[0] <= 1.5 * tol
    assert held0 <= 0.62 * held1          # the tape of the taped forward: outputs only against outputs + derivatives


@pytest.mark.parametrize("name", ["s1_straight_n4", "s2_fade_a037_n4"])
def test_bf16_train_step_matches_emulated_oracle(name, bf16_precision):
    """The whole G+D step with the bf16 discriminator blocks against the oracle's train step evaluated
    with bf16 storage emulated.  Metrics and losses abs 5e-3; discriminator-step gradients relative L2
    <= 8e-2 (the saturated output activation turns the logit difference into a common factor of every
    gradient, see test_bf16_discriminator_vs_plain_oracle; the scalar alpha is not compared).  The generator-step gradients
    additionally follow the discriminator's sign-like first Adam update (see test_gpu_step.py: a weight
    whose gradient is ~0 moves by +-lr according to the sign of rounding noise) and are not compared here;
    test_gpu_step.py::test_generator_step_with_device_discriminator pins the generator kernels."""
    from oracle import bf16_emulation as E
    from test_gpu_step import CASES, device_step
    n_blocks, stage, fade, batch, ks, tr_alpha = CASES[name]
    gan, metrics, fake = device_step(name)
    W = R.synthetic_weights(n_blocks, seed=100)
    real, z4 = R.synthetic_inputs(stage, batch, seed=0)
    with E.bf16_storage():
        res = R.train_step(W, n_blocks, stage, fade, real, z4, alpha=(ks[0] / float(ks[1] - 1)) if ks else None,
                           trainable_alpha=tr_alpha)
    for k, v in res.metrics.items():
        assert abs(float(metrics[k]) - v) < 5e-3, (k, float(metrics[k]), v)
    assert abs(float(gan.last_losses["d_loss"]) - res.d_loss) < 5e-3
    assert abs(float(gan.last_losses["g_loss"]) - res.g_loss) < 5e-3
    worst = {}
    for which, model, grads, tol0 in (("d", gan.discriminator, res.d_grads, 8e-2), ("g", gan.generator, res.g_grads, None)):
        for v in model.trainable_variables:
            want = grads[v.name]
            if np.abs(want).max() < 1e-12 or tol0 is None or v.name.endswith("/alpha"):
                continue            # alpha: sum(dY (new - old)) cancels to ~1e-3 of its terms; test_bf16_weighted_sum pins it
            err = rel_l2(v.grad.cpu().numpy().reshape(want.shape), want)
            worst[which] = max(worst.get(which, 0.0), err)
            assert err < tol0, "%s: rel-L2 %.3e" % (v.name, err)
    print("worst relative L2:", worst)


@pytest.mark.parametrize("stage,batch", [(1, 3), (2, 2), (3, 1)])
def test_bf16_generator_forward_matches_oracle(stage, batch):
    """Tape-free generator forward with the grow blocks on the bf16 fused kernel against the float64
    oracle.  The chain is linear (no activation), every intermediate tensor is rounded to bf16 once:
    26 layers per block (the 2-channel block input enters as two bf16 parts).  Tolerance on the
    LayerNormalised audio: relative L2 <= 2e-2, max abs <= 8e-2 x max|reference| (measured ~1e-2 / 6e-2)."""
    d, g, comp, W = _build(stage + 1)
    _, z4 = R.synthetic_inputs(stage, batch, seed=0)
    M.backend.set_precision("bfloat16")
    try:
        got = g[stage][0](z4[0]).cpu().numpy()
    finally:
        M.backend.set_precision("float32")
    names = R.layer_names(stage + 1)
    Wt = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in W.items()}
    want = R.generator(Wt, names, stage, False, torch.tensor(z4[0], dtype=torch.float64)).numpy()
    err = rel_l2(got, want)
    print("generator audio: rel-L2", err, "max abs", np.abs(got - want).max())
    assert err <= 2e-2, err
    assert np.abs(got - want).max() <= 8e-2 * np.abs(want).max()


@pytest.mark.parametrize("stage,twin,budget", [(1, 0, None), (2, 1, None), (2, 0, 64 << 20)])
def test_bf16_generator_backward_matches_oracle(stage, twin, budget):
    """Taped generator forward + backward with the grow blocks on the bf16 fused kernels (tape and gradient
    side buffers in bf16; with `budget` the re-computing backward over position ranges) against float64
    autograd of the oracle for a random output gradient.  The chain is linear: every intermediate
    activation / gradient is rounded to bf16 once.  Tolerance: audio relative L2 <= 2e-2; weight and bias
    gradients of the grow blocks relative L2 <= 3.5e-2 per tensor (5e-2 for the few tensors with < 64
    elements); gradients of the float32 stage-0 layers and the LayerNormalization <= 3.5e-2.  (Measured:
    the worst tensors, 64-element biases, sit at 2.8e-2 .. 3.0e-2; which of them is worst moves with any
    change of the float32 summation order upstream, because that flips bf16 roundings downstream.)"""
    from music_generator_amd.graph import FusedGroup
    d, g, comp, W = _build(stage + 1)
    model = g[stage][twin]
    for layer in model.layers:
        if isinstance(layer, M.WeightedSum):
            M.backend.set_value(layer.alpha, 0.37)
    batch = 2
    _, z4 = R.synthetic_inputs(stage, batch, seed=0)
    h, w = 4 << stage, 750 << stage
    rng = np.random.default_rng(9)
    dy = rng.standard_normal((batch, h, w, 2)).astype(np.float32)
    M.backend.set_precision("bfloat16")
    FusedGroup.side_budget_bytes = budget
    try:
        y, tape = model.forward(torch.from_numpy(z4[0]).cuda(), save=True)
        model.bind_grads().zero_()
        model.backward(tape, torch.from_numpy(dy).cuda(), need_input_grad=False)
        torch.cuda.synchronize()
    finally:
        M.backend.set_precision("float32")
        FusedGroup.side_budget_bytes = None
    got = {v.name: v.grad.cpu().numpy().reshape(v.shape).copy() for v in model.trainable_variables}
    names = R.layer_names(stage + 1)
    Wt = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in W.items()}
    al = None
    if twin:
        al = torch.tensor(0.37, dtype=torch.float64, requires_grad=True)
        Wt[names["g"][stage]["ws"] + "/alpha"] = al
    yo = R.generator(Wt, names, stage, bool(twin), torch.tensor(z4[0], dtype=torch.float64), al)
    keys = R.trainable_names(names, stage, bool(twin), "g")
    gs = torch.autograd.grad((yo * torch.tensor(dy, dtype=torch.float64)).sum(), [Wt[k] for k in keys])
    assert rel_l2(y.cpu().numpy(), yo.detach().numpy()) <= 2e-2
    assert set(keys) == set(got), sorted(set(keys) ^ set(got))
    errs = {k: rel_l2(got[k], gr.numpy()) for k, gr in zip(keys, gs)}
    print("audio", rel_l2(y.cpu().numpy(), yo.detach().numpy()), "worst gradients",
          sorted(errs.items(), key=lambda kv: -kv[1])[:4])
    for k, err in errs.items():
        tol = 2e-1 if k.endswith("/alpha") else (5e-2 if got[k].size < 64 else 3.5e-2)
        assert err <= tol, "%s: rel-L2 %.3e" % (k, err)


@pytest.mark.parametrize("budget", [None, 64 << 20])
def test_bf16_generator_lean_tape_matches_full_tape(budget):
    """The grow blocks' backward with the four widest tensors recomputed by the weight-gradient kernels
    (backend.GBLOCK_LEAN_TAPE, the default) against the backward that reads every operand from a full tape, same
    weights, latent points and output gradient, stage-2 fade-in generator (two grow blocks).  The recomputed
    operands are the same bf16 roundings of the same float32 sums up to the order in which an MFMA adds its 16
    products, so the two differ by a few flipped bf16 roundings: every gradient tensor relative L2 <= 2e-3, the
    audio bit-identical (the forward arithmetic does not change)."""
    from music_generator_amd.graph import FusedGroup
    stage = 2
    d, g, comp, W = _build(stage + 1)
    model = g[stage][1]
    for layer in model.layers:
        if isinstance(layer, M.WeightedSum):
            M.backend.set_value(layer.alpha, 0.37)
    batch = 2
    _, z4 = R.synthetic_inputs(stage, batch, seed=0)
    h, w = 4 << stage, 750 << stage
    dy = torch.from_numpy(np.random.default_rng(9).standard_normal((batch, h, w, 2)).astype(np.float32)).cuda()
    out = {}
    M.backend.set_precision("bfloat16")
    FusedGroup.side_budget_bytes = budget
    try:
        for lean in (False, True):
            M.backend.GBLOCK_LEAN_TAPE = lean
            y, tape = model.forward(torch.from_numpy(z4[0]).cuda(), save=True)
            model.bind_grads().zero_()
            model.backward(tape, dy, need_input_grad=False)
            torch.cuda.synchronize()
            out[lean] = (y.clone(), {v.name: v.grad.clone() for v in model.trainable_variables})
    finally:
        M.backend.GBLOCK_LEAN_TAPE = True
        M.backend.set_precision("float32")
        FusedGroup.side_budget_bytes = None
    assert torch.equal(out[False][0], out[True][0])
    errs = {k: rel_l2(out[True][1][k].cpu().numpy(), out[False][1][k].cpu().numpy()) for k in out[False][1]}
    print("lean against full tape, worst:", sorted(errs.items(), key=lambda kv: -kv[1])[:5])
    for k, err in errs.items():
        assert err <= 2e-3, "%s: rel-L2 %.3e" % (k, err)


def _fresh_gan(stage, micro_batch, budget_gib=None, seed=11):
    from music_generator_amd.graph import FusedGroup
    M.clear_session()
    d = M.define_discriminator(stage + 1)
    g = M.define_generator(stage + 1)
    gan = M.define_composite(d, g, (1, 100, 2))[stage][1]
    gan.micro_batch = micro_batch
    FusedGroup.side_budget_bytes = int(budget_gib * 2 ** 30) if budget_gib else None
    gan.set_train_steps(9)
    gan.actual_step = 3
    gan.seed(seed)
    return gan


def test_bf16_upper_stage_micro_batches_equal_whole_batch(bf16_precision):
    """Stage 5 (128 x 24000), batch 8, bf16: the step in micro-batches of one MinibatchStdDev group against the
    whole batch at once.  Every sample's bf16 arithmetic is the same in both; only the float32 order in
    which per-micro-batch gradient sums are added differs.  Both learning rates are set to 0 for this
    comparison: a first Adam update is sign-like (beta_1 = 0, v = 0), so weights whose gradient is ~0 would
    move by +-lr according to the last bit of a sum and everything evaluated after an update (the metric
    blocks, the generator step) would compare two slightly different networks (measured: metrics 6e-5 after
    the discriminator's update and 2e-2 after the generator's, generator gradients up to 9e-2) instead of
    two summation orders.
    Metrics abs 1e-5, discriminator gradients rel-L2 5e-5 (5e-4 alpha), generator gradients 2e-3."""
    from music_generator_amd.graph import FusedGroup
    stage, n = 5, 8
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    real = torch.rand((n, 4 << stage, 750 << stage, 2), device="cuda", generator=gen) * 2 - 1
    runs = []
    try:
        for mb in (None, 4):
            gan = _fresh_gan(stage, mb)     # initial values are a function of the variable names: both builds equal
            gan.d_optimizer.lr = gan.g_optimizer.lr = 0.0
            m = {k: float(v) for k, v in gan.train_step(real).items()}
            grads = {v.name: v.grad.cpu().numpy().copy() for model in (gan.discriminator, gan.generator)
                     for v in model.trainable_variables}
            d_names = {v.name for v in gan.discriminator.trainable_variables}
            runs.append((m, grads))
            del gan
            torch.cuda.empty_cache()
    finally:
        FusedGroup.side_budget_bytes = None
    (m0, g0), (m1, g1) = runs
    for k in m0:
        assert np.isfinite(m0[k]) and abs(m0[k] - m1[k]) < 1e-5, (k, m0[k], m1[k])
    worst = {}
    for k, v in g0.items():
        assert np.isfinite(g1[k]).all(), k
        if np.abs(v).max() < 1e-12:
            continue
        which = "d" if k in d_names else "g"
        err = rel_l2(g1[k], v)
        worst[which] = max(worst.get(which, (0.0, "")), (err, k))
        tol = (5e-4 if k.endswith("/alpha") else 5e-5) if which == "d" else 2e-3
        assert err < tol, "%s: %.3e" % (k, err)
    print("worst relative L2:", worst)


def test_generator_forward_beside_a_taped_discriminator_forward_is_repeatable(bf16_precision):
    """The base generator's forward on the caller's stream while a taped bf16 stage-3 discriminator forward runs on
    lane 1 (as in the small-batch step), 300 rounds: the 8 x 4 x 750 x 2 output is bit-identical to the one computed
    alone every time.  (The Dense(8 -> 2) kernel of the generator's tail once summed its lane pairs through a
    run-time-offset shuffle loop; beside the discriminator's kernels about one round in ten came back with one
    output channel of eight neighbouring positions wrong, always lanes 48..63 of a wave, never when run alone.)"""
    from music_generator_amd import backend as K
    stage, n = 3, 8
    disc = M.define_discriminator(stage + 1)[stage][1]
    g0 = M.define_generator(1)[0][0]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(3)
    real = torch.rand((n, 4 << stage, 750 << stage, 2), device="cuda", generator=gen) * 2 - 1
    z = torch.randn((n, 1, 100, 2), device="cuda", generator=gen)
    alone = g0.forward(z, save=False)[0].clone()
    torch.cuda.synchronize()
    bad = []
    for it in range(300):
        lane = K.fork()
        with lane:
            _, tape = disc.forward(real, save=True)
            del tape
        f = g0.forward(z, save=False)[0]
        lane.join()
        torch.cuda.synchronize()
        if not torch.equal(f, alone):
            bad.append((it, int((f != alone).sum())))
    assert not bad, "rounds with differing elements: %s" % bad[:10]


def test_bf16_full_resolution_step_properties(bf16_precision):
    """Stage 6 (256 x 48000), batch 8 in micro-batches of 4, generator tape bounded to 8 GiB (re-computing
    backward), bf16: the configuration of the full-resolution bench step at a batch pytest can afford.
    Size-independent properties: all six metrics and every gradient finite and not all zero; the reported
    distance is |mean fake logit - mean real logit| (models.py:469-472); generated clips obey the
    LayerNormalization identity (per-clip mean 0, mean square just under 1 for gamma = 1, beta = 0)."""
    from music_generator_amd.graph import FusedGroup
    from music_generator_amd.inference import ChecksumSink, generate_clips
    stage, n = 6, 8
    try:
        gan = _fresh_gan(stage, 4, 8.0)
        chk = ChecksumSink()
        generate_clips(gan.generator, 2, 2, chk, seed=1)
        res = chk.result()
        size = 256 * 48000 * 2
        assert np.abs(res[:, 0] / size).max() < 1e-4 and ((res[:, 1] / size) < 1 + 1e-4).all()
        gen = torch.Generator(device="cuda")
        gen.manual_seed(4)
        real = torch.rand((n, 256, 48000, 2), device="cuda", generator=gen) * 2 - 1
        m = {k: float(v) for k, v in gan.train_step(real).items()}
        assert all(np.isfinite(v) for v in m.values()), m
        for i in ("0", "1"):
            assert abs(m["delta_" + i] - abs(m["cu_" + i] - m["ci_" + i])) < 1e-5, m
        for model in (gan.discriminator, gan.generator):
            flat = model.bind_grads(0)
            assert bool(torch.isfinite(flat).all()) and float(flat.abs().max()) > 0.0
        assert gan.actual_step == 4 and gan.d_optimizer.iterations == 1 and gan.g_optimizer.iterations == 1
    finally:
        FusedGroup.side_budget_bytes = None


@pytest.mark.parametrize("lean", [False, True])
def test_bf16_generator_tape_bound_is_chosen_without_any_setting(bf16_precision, lean):
    """Stage 5 (128 x 24000), batch 16, bf16, whole batch, no micro-batch and no FusedGroup.side_budget_bytes.
    With every tensor on the tape (lean False) the generator's tape (94 GiB for the top block) and its gradient
    buffer would fit the memory that is free at the generator's forward, but not beside the discriminator tape that
    follows in the generator update.  The train step announces that tape (Model.tape_bytes_estimate), so the
    automatic bound is chosen for the top block and the step runs without the allocator having to give memory back
    and ask again (num_alloc_retries stays 0 over the second step).  The lean tape (35 GiB for the top block) fits
    beside it: no bound, no recomputing forward."""
    from music_generator_amd.graph import FusedGroup
    stage, n = 5, 16
    FusedGroup.decisions.clear()
    M.backend.GBLOCK_LEAN_TAPE = lean
    try:
        _tape_bound_case(stage, n, lean)
    finally:
        M.backend.GBLOCK_LEAN_TAPE = True


def _tape_bound_case(stage, n, lean):
    from music_generator_amd.graph import FusedGroup
    gan = _fresh_gan(stage, None)
    assert FusedGroup.side_budget_bytes is None
    est = gan.discriminator.tape_bytes_estimate(n)
    assert 40e9 < est < 120e9, est                     # 128 x 24000 x 896 B a position and sample, plus the lower blocks
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    real = torch.rand((n, 4 << stage, 750 << stage, 2), device="cuda", generator=gen) * 2 - 1
    m = {k: float(v) for k, v in gan.train_step(real).items()}
    assert all(np.isfinite(v) for v in m.values()), m
    top = max(FusedGroup.decisions, key=lambda d: d[0])
    if lean:
        assert 30e9 < top[0] < 45e9 and top[1] is None, FusedGroup.decisions
    else:
        assert top[0] > 60e9 and top[1] is not None and top[1] <= 20 * 2 ** 30, FusedGroup.decisions
    retries = torch.cuda.memory_stats()["num_alloc_retries"]
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    m = {k: float(v) for k, v in gan.train_step(real).items()}
    print("D tape estimate %.1f GB, decisions %s, second step %.2f s, peak %.1f GiB" % (
        est / 1e9, FusedGroup.decisions, time.perf_counter() - t0, torch.cuda.max_memory_allocated() / 2 ** 30))
    assert all(np.isfinite(v) for v in m.values()), m
    assert torch.cuda.memory_stats()["num_alloc_retries"] == retries
    del gan, real
    torch.cuda.empty_cache()


def test_bf16_full_resolution_step_at_batch_256(bf16_precision):
    """The full-resolution bench configuration itself: stage 6 (256 x 48000) fade-in step, batch 256 in 64
    micro-batches of one MinibatchStdDev group, no bound set for the generator tape (the executor keeps the lean
    tape of a micro-batch: 47 GiB over the four grow blocks), bf16.  Properties that do not depend on the size: the
    six metrics finite, delta = |cu - ci| (models.py:469-472), every gradient finite and not all zero, exactly one
    update of each optimiser, no tape bound chosen, and the step's peak HBM within 200 GiB."""
    from music_generator_amd.graph import FusedGroup
    stage, n = 6, 256
    FusedGroup.decisions.clear()
    try:
        gan = _fresh_gan(stage, 4, None)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(4)
        real = torch.rand((n, 256, 48000, 2), device="cuda", generator=gen).mul_(2.0).sub_(1.0)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        m = {k: float(v) for k, v in gan.train_step(real).items()}
        peak = torch.cuda.max_memory_allocated() / 2.0 ** 30
        print("batch-256 full-resolution step: peak %.1f GiB" % peak, m)
        assert all(np.isfinite(v) for v in m.values()), m
        for i in ("0", "1"):
            assert abs(m["delta_" + i] - abs(m["cu_" + i] - m["ci_" + i])) < 1e-5, m
        for model in (gan.discriminator, gan.generator):
            flat = model.bind_grads(0)
            assert bool(torch.isfinite(flat).all()) and float(flat.abs().max()) > 0.0
        assert gan.actual_step == 4 and gan.d_optimizer.iterations == 1 and gan.g_optimizer.iterations == 1
        assert peak <= 200.0, peak
        assert FusedGroup.decisions and all(bound is None for _, bound in FusedGroup.decisions), FusedGroup.decisions
    finally:
        FusedGroup.side_budget_bytes = None
        del real
        torch.cuda.empty_cache()


def test_bf16_third_step_with_warmed_adam_matches_emulated_oracle(bf16_precision):
    """Adam at t = 3 on the bf16 path, generator-step gradients included.  The float64 oracle advances weights
    and both optimisers' second moments two steps on the host; the device takes that state over and runs the
    third step with bf16 precision, which is compared with the oracle's third step with bf16 storage emulated.
    With warmed second moments the discriminator update in front of the generator step is no longer sign-like
    (see test_bf16_train_step_matches_emulated_oracle), so the generator-step gradients are comparable:
    relative L2 <= 8e-2 for discriminator-step AND generator-step gradients (1.2e-1 for bias / beta gradients;
    the scalar alphas, cancelling sums over a whole activation, are pinned by test_bf16_weighted_sum), metrics
    and losses abs 5e-3, updated weights abs 2e-5 + 8e-2 of the oracle's own update."""
    import os
    import sys
    from oracle import bf16_emulation as E
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import make_golden
    _, n_blocks, stage, batch, n_steps = make_golden.T3
    cur, vd, vg = make_golden.warmed_state(n_blocks, stage, batch, n_steps, 2)
    d, g, comp, _ = _build(n_blocks)
    gan = comp[stage][1]
    for model in (gan.discriminator, gan.generator):
        for v in model.trainable_variables:
            v.assign(cur[v.name])
    gan.d_optimizer.set_second_moments(gan.discriminator, vd, 2)
    gan.g_optimizer.set_second_moments(gan.generator, vg, 2)
    gan.set_train_steps(n_steps)
    gan.actual_step = 2
    real, z4 = R.synthetic_inputs(stage, batch, seed=12)
    gan.inject_latents(z4)
    metrics = {k: float(v) for k, v in gan.train_step(real).items()}
    with E.bf16_storage():
        res = R.train_step(cur, n_blocks, stage, True, real, z4, alpha=2 / float(n_steps - 1), v_d=vd, v_g=vg, t=3)
    for k, v in res.metrics.items():
        assert abs(metrics[k] - v) < 5e-3, (k, metrics[k], v)
    assert abs(float(gan.last_losses["d_loss"]) - res.d_loss) < 5e-3
    assert abs(float(gan.last_losses["g_loss"]) - res.g_loss) < 5e-3
    assert gan.d_optimizer.iterations == 3 and gan.g_optimizer.iterations == 3
    worst, worst_u = {}, {}
    UPDATE_TOL = 2.5e-1
    for which, model, grads in (("d", gan.discriminator, res.d_grads), ("g", gan.generator, res.g_grads)):
        for v in model.trainable_variables:
            want = grads[v.name]
            if np.abs(want).max() < 1e-12 or v.name.endswith("/alpha"):
                continue
            err = rel_l2(v.grad.cpu().numpy().reshape(want.shape), want)
            worst[which] = max(worst.get(which, (0.0, "")), (err, v.name))
            tol = 1.2e-1 if v.name.endswith(("/bias", "/beta")) else 8e-2
            assert err < tol, "%s-step gradient of %s: rel-L2 %.3e" % (which, v.name, err)
            start = np.asarray(cur[v.name], np.float64)
            uerr = rel_l2(v.numpy() - start, res.weights[v.name] - start)
            if v.size >= 64:
                worst_u[which] = max(worst_u.get(which, (0.0, "")), (uerr, v.name))
    print("worst relative L2 of gradients (d-step, g-step):", worst, "of updates:", worst_u)
    for which, (uerr, name) in worst_u.items():
        assert uerr < UPDATE_TOL, "%s-step update of %s: rel-L2 %.3e" % (which, name, uerr)


@pytest.mark.parametrize("h,w,n,kb", [(16, 64, 1, 3), (16, 31, 2, 2), (20, 45, 2, 3), (32, 100, 3, 2), (48, 333, 1, 3),
                                      (17, 30, 1, 3), (64, 500, 2, 2)])
def test_bf16_conv_pair_equals_two_launches(h, w, n, kb):
    """mgx_conv_pair_bf16 (two convolution + activation layers in one launch, the intermediate in LDS) against two
    launches of mgx_conv_gemm_bf16: the same MFMAs on the same bf16 values in the same order, so the outputs
    must be equal bit for bit, on ragged tiles (rows not a multiple of 17 - kb, columns not of 33 - kb), at
    the image borders (zero padding of BOTH layers) and across images."""
    M.clear_session()
    a = L.Conv2D(32, (3, 3), padding="same", kernel_initializer="he_normal")
    b = L.Conv2D(32, (kb, kb), padding="same", kernel_initializer="he_normal")
    a(Input(shape=(h, w, 32)))
    b(Input(shape=(h, w, 32)))
    rng = np.random.default_rng(h * 100 + w + kb)
    for layer in (a, b):
        layer.bias.assign(rng.uniform(-0.3, 0.3, (32,)).astype(np.float32))
    x = _dev_b(rng.uniform(-1, 1, (n, h, w, 32)))
    alpha_a, alpha_b = 7.0, 80000.0
    mid = torch.empty((n, h, w, 32), dtype=torch.bfloat16, device="cuda")
    want = torch.empty_like(mid)
    ops.conv_gemm_bf16(a._desc("f", n, h, w), x, a.kernel.value(), a.bias.value(), mid, act_alpha=alpha_a)
    ops.conv_gemm_bf16(b._desc("f", n, h, w), mid, b.kernel.value(), b.bias.value(), want, act_alpha=alpha_b)
    assert ops.conv_pair_bf16_supported(n, h, w, kb)
    got = torch.full_like(want, float("nan"))
    ops.conv_pair_bf16(x, a.kernel.value(), a.bias.value(), alpha_a, b.kernel.value(), b.bias.value(), alpha_b, kb, got)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got.float()).all()), "positions left unwritten"
    diff = (got.float() - want.float()).abs()
    where = np.unravel_index(int(diff.argmax()), tuple(diff.shape))
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), "max abs difference %.3e at %s" % (float(diff.max()), where)


def test_bf16_discriminator_forward_with_and_without_pairs(bf16_precision):
    """A tape-free discriminator forward (the metric passes of the step) with the pair kernel (Model.use_conv_pairs,
    off by default: it is slower than the launches it replaces) against the same forward layer by layer:
    bit-identical logits; with a tape the pair kernel is not used at all."""
    from music_generator_amd.graph import Model
    d, g, comp, W = _build(4)
    model = d[3][1]
    real, _ = R.synthetic_inputs(3, 4, seed=2)
    x = torch.from_numpy(real).cuda()
    calls = []
    orig = ops.conv_pair_bf16
    ops.conv_pair_bf16 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    chains = Model.use_conv_chains
    try:
        Model.use_conv_chains = False          # the chain kernel would take the same layers first
        Model.use_conv_pairs = True
        y1 = model(x).clone()
        n_pairs = len(calls)
        Model.use_conv_pairs = False
        y0 = model(x).clone()
        assert len(calls) == n_pairs
        Model.use_conv_pairs = True
        model.forward(x, save=True)
        assert len(calls) == n_pairs
    finally:
        Model.use_conv_pairs = False
        Model.use_conv_chains = chains
        ops.conv_pair_bf16 = orig
    assert n_pairs >= 3, n_pairs
    assert torch.equal(y0, y1), (y0, y1)


@pytest.mark.parametrize("h,w,n,ks", [(16, 64, 1, (3, 3, 2)), (8, 29, 2, (3, 3, 2)), (9, 30, 1, (3, 3, 2)),
                                      (20, 45, 2, (3, 3)), (32, 100, 3, (3, 2)), (48, 333, 1, (3, 3, 2)),
                                      (17, 31, 1, (3, 2)), (64, 500, 2, (3, 3)), (2, 8, 1, (3, 3, 2)),
                                      (33, 61, 5, (3, 3, 2)), (50, 40, 1, (3, 3, 2)), (100, 64, 2, (3, 2)),
                                      (130, 33, 1, (3, 3))])
def test_bf16_conv_chain_equals_separate_launches(h, w, n, ks):
    """mgx_conv_chain_bf16 (two or three convolution + activation layers in one launch, a wave streaming a strip of
    columns down the rows with the inner layers' rows in LDS rings) against one launch of mgx_conv_gemm_bf16 per
    layer: the same MFMAs on the same bf16 values in the same order, so the outputs must be equal bit for bit,
    on ragged strips (widths that are not a multiple of 29 / 30 / 31), at the image borders (zero padding of
    EVERY layer, rows and columns), across images (more strips than one workgroup holds) and where a strip is
    walked in several pieces of rows (heights of 32 and more; pieces of unequal length at 33, 50, 100, 130)."""
    M.clear_session()
    layers = [L.Conv2D(32, (k, k), padding="same", kernel_initializer="he_normal") for k in ks]
    rng = np.random.default_rng(h * 100 + w + sum(ks))
    for layer in layers:
        layer(Input(shape=(h, w, 32)))
        layer.bias.assign(rng.uniform(-0.3, 0.3, (32,)).astype(np.float32))
    x = _dev_b(rng.uniform(-1, 1, (n, h, w, 32)))
    alphas = [7.0, 80000.0, 11000.0][:len(ks)]
    cur, inner = x, []
    for layer, alpha in zip(layers, alphas):
        nxt = torch.empty((n, h, w, 32), dtype=torch.bfloat16, device="cuda")
        ops.conv_gemm_bf16(layer._desc("f", n, h, w), cur, layer.kernel.value(), layer.bias.value(), nxt, act_alpha=alpha)
        cur = nxt
        inner.append(nxt)
    want = cur
    assert ops.conv_chain_bf16_supported(n, h, w, ks)
    got = torch.full_like(want, float("nan"))
    ops.conv_chain_bf16(x, [l.kernel.value() for l in layers], [l.bias.value() for l in layers], alphas, ks, got)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got.float()).all()), "positions left unwritten"
    diff = (got.float() - want.float()).abs()
    where = np.unravel_index(int(diff.argmax()), tuple(diff.shape))
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), "max abs difference %.3e at %s" % (float(diff.max()), where)
    # the taped launch: every layer's output, each position written by the one wave that owns it
    outs = [torch.full_like(want, float("nan")) for _ in ks]
    ops.conv_chain_bf16(x, [l.kernel.value() for l in layers], [l.bias.value() for l in layers], alphas, ks, outs)
    torch.cuda.synchronize()
    for i, (o, ref) in enumerate(zip(outs, inner)):
        assert bool(torch.isfinite(o.float()).all()), "layer %d: positions left unwritten" % i
        assert torch.equal(o.view(torch.int16), ref.view(torch.int16)), "layer %d: %.3e" % (i, float((o.float() - ref.float()).abs().max()))


@pytest.mark.parametrize("h,w,n,ks", [(48, 333, 1, (3, 3, 2)), (64, 500, 2, (3, 3)), (100, 64, 2, (3, 2)), (9, 30, 1, (3, 3, 2))])
def test_bf16_conv_chain_pipelined_variant(h, w, n, ks, monkeypatch):
    """The opt-in software-pipelined chain kernel (MGX_CHAIN_PIPELINED=1: a layer row's MFMAs interleaved with another
    row's epilogue, one wave per SIMD) gives the bits of the default kernel, tape-free and taped."""
    monkeypatch.setenv("MGX_CHAIN_PIPELINED", "1")
    test_bf16_conv_chain_equals_separate_launches(h, w, n, ks)


def test_bf16_conv_chain_rejects_what_it_does_not_serve():
    assert not ops.conv_chain_bf16_supported(1, 16, 64, (3, 3, 3))
    assert not ops.conv_chain_bf16_supported(1, 16, 64, (2, 3))
    assert not ops.conv_chain_bf16_supported(1, 16, 64, (3,))
    assert not ops.conv_chain_bf16_supported(1, 16, 4, (3, 3))
    x = torch.zeros((1, 16, 64, 32), dtype=torch.bfloat16, device="cuda")
    w3 = torch.zeros((3, 3, 32, 32), device="cuda")
    b = torch.zeros((32,), device="cuda")
    with pytest.raises(ops.MgxError):
        ops.conv_chain_bf16(x, [w3, w3], [b, b], [7.0, 0.0], (3, 3), torch.empty_like(x))     # alpha must be > 0
    with pytest.raises(ops.MgxError):
        ops.conv_chain_bf16(x, [w3, w3], [b, b], [7.0, 7.0], (3, 2), torch.empty_like(x))     # kernel shape mismatch


def test_bf16_discriminator_forward_with_and_without_chains(bf16_precision):
    """A tape-free discriminator forward (the metric passes of the step) through the chain kernel
    (Model.use_conv_chains, the default) against the same forward layer by layer: bit-identical logits, three chains
    per grown block that is tall enough.  With a tape the same groups run through the taped launch
    (Model.use_taped_chains): the tape holds the tensors the layer-by-layer forward leaves on it, bit for bit."""
    from music_generator_amd.graph import Model
    from music_generator_amd.layers import _BSaved
    d, g, comp, W = _build(4)
    model = d[3][1]
    for layer in model.layers:
        if isinstance(layer, M.WeightedSum):
            M.backend.set_value(layer.alpha, 0.37)
    real, _ = R.synthetic_inputs(3, 4, seed=2)
    x = torch.from_numpy(real).cuda()
    calls, blends = [], []
    orig, orig_blend = ops.conv_chain_bf16, ops.conv1x1_blend_audio_bf16
    ops.conv_chain_bf16 = lambda *a, **k: (calls.append(tuple(a[4])), orig(*a, **k))[1]
    ops.conv1x1_blend_audio_bf16 = lambda *a, **k: (blends.append(1), orig_blend(*a, **k))[1]
    chains, taped_chains = Model.use_conv_chains, Model.use_taped_chains
    try:
        Model.use_conv_chains = True
        y1 = model(x).clone()
        n_chains = len(calls)
        Model.use_conv_chains = False
        y0 = model(x).clone()
        assert len(calls) == n_chains
        Model.use_conv_chains = True
        Model.use_taped_chains = True
        yt, tape_t = model.forward(x, save=True)
        assert len(calls) == 2 * n_chains and calls[n_chains:] == calls[:n_chains]
        Model.use_taped_chains = False
        yl, tape_l = model.forward(x, save=True)
        assert len(calls) == 2 * n_chains
        assert torch.equal(yt, yl) and len(tape_t) == len(tape_l)
        n_entries = 0
        for a, b in zip(tape_t, tape_l):
            assert type(a) is type(b)
            if isinstance(a, _BSaved):
                n_entries += 1
                assert a.mode == b.mode and a.alpha == b.alpha and a.dact is None and b.dact is None
                assert torch.equal(a.x, b.x) and (a.yact is None) == (b.yact is None)
                assert a.yact is None or torch.equal(a.yact, b.yact)
        assert n_entries >= 14, n_entries
    finally:
        Model.use_conv_chains = chains
        Model.use_taped_chains = taped_chains
        ops.conv_chain_bf16 = orig
        ops.conv1x1_blend_audio_bf16 = orig_blend
    assert n_chains >= 6 and set(calls) == {(3, 2), (3, 3, 2), (3, 3)}, calls
    # the fade-in blend of the top block ran in the 1x1 convolution's epilogue, its other input (the old stage's
    # from-audio layer on the pooled audio) re-computed there
    assert len(blends) == 1
    assert torch.equal(y0, y1), (y0, y1)


@pytest.mark.parametrize("n,h,w", [(2, 8, 100), (3, 16, 70), (1, 32, 45), (5, 16, 1500)])
@pytest.mark.parametrize("alpha", [0.0, 0.37, 1.0])
def test_bf16_conv1x1_blend_audio_equals_three_launches(n, h, w, alpha):
    """mgx_conv1x1_blend_audio_bf16 (the blend's first input re-computed from the pooled audio in the epilogue) against
    mgx_pointwise2_bf16_fwd + mgx_conv_gemm_bf16 + mgx_weighted_sum_bf16_fwd: bit for bit, both tile shapes, ragged
    widths, audio values large enough to saturate the activation."""
    M.clear_session()
    conv = L.Conv2D(128, (1, 1), padding="same", kernel_initializer="he_normal")
    conv(Input(shape=(h, w, 32)))
    fa = L.Conv2D(128, (1, 1), padding="same", kernel_initializer="he_normal")
    fa(Input(shape=(h, w, 2)))
    rng = np.random.default_rng(n * 1000 + h + w + 7)
    conv.bias.assign(rng.uniform(-0.5, 0.5, (128,)).astype(np.float32))
    fa.bias.assign(rng.uniform(-0.5, 0.5, (128,)).astype(np.float32))
    x = _dev_b(rng.uniform(-1, 1, (n, h, w, 32)))
    audio = torch.from_numpy(rng.uniform(-3, 3, (n, h, w, 2)).astype(np.float32)).cuda()
    al = torch.tensor([alpha], device="cuda")
    old = torch.empty((n, h, w, 128), dtype=torch.bfloat16, device="cuda")
    mid = torch.empty_like(old)
    want = torch.empty_like(old)
    ops.pointwise2_bf16_fwd(audio, fa.kernel.value(), fa.bias.value(), old, None, n * h * w, 20000.0)
    ops.conv_gemm_bf16(conv._desc("f", n, h, w), x, conv.kernel.value(), conv.bias.value(), mid, act_alpha=90000.0)
    ops.weighted_sum_bf16_fwd(old, mid, al, want)
    got = torch.full_like(want, float("nan"))
    ops.conv1x1_blend_audio_bf16(conv._desc("f", n, h, w), x, conv.kernel.value(), conv.bias.value(), audio,
                                 fa.kernel.value(), fa.bias.value(), 20000.0, al, got, 90000.0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got.float()).all()), "positions left unwritten"
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), float((got.float() - want.float()).abs().max())


@pytest.mark.parametrize("n,h,w", [(2, 8, 100), (3, 16, 70), (1, 32, 45), (5, 16, 1500)])
@pytest.mark.parametrize("alpha", [0.0, 0.37, 1.0])
def test_bf16_conv1x1_blend_equals_two_launches(n, h, w, alpha):
    """mgx_conv1x1_blend_bf16 (1x1 convolution 32 -> 128 + activation with the fade-in blend in its epilogue) against
    mgx_conv_gemm_bf16 followed by mgx_weighted_sum_bf16_fwd: bit for bit, on both tile shapes and ragged widths."""
    M.clear_session()
    conv = L.Conv2D(128, (1, 1), padding="same", kernel_initializer="he_normal")
    conv(Input(shape=(h, w, 32)))
    rng = np.random.default_rng(n * 1000 + h + w)
    conv.bias.assign(rng.uniform(-0.5, 0.5, (128,)).astype(np.float32))
    x = _dev_b(rng.uniform(-1, 1, (n, h, w, 32)))
    old = _dev_b(rng.uniform(-2, 2, (n, h, w, 128)))
    al = torch.tensor([alpha], device="cuda")
    mid = torch.empty((n, h, w, 128), dtype=torch.bfloat16, device="cuda")
    want = torch.empty_like(mid)
    ops.conv_gemm_bf16(conv._desc("f", n, h, w), x, conv.kernel.value(), conv.bias.value(), mid, act_alpha=90000.0)
    ops.weighted_sum_bf16_fwd(old, mid, al, want)
    got = torch.full_like(want, float("nan"))
    ops.conv1x1_blend_bf16(conv._desc("f", n, h, w), x, conv.kernel.value(), conv.bias.value(), old, al, got, 90000.0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got.float()).all()), "positions left unwritten"
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), float((got.float() - want.float()).abs().max())


def test_bf16_weighted_sum_backward_applies_producer_derivatives():
    """mgx_weighted_sum_bf16_bwd_mul against mgx_weighted_sum_bf16_bwd followed by one multiply launch per input
    (what the executor did before): bit-identical gradients, the same alpha gradient, either multiplier optional."""
    rng = np.random.default_rng(11)
    shape = (2, 8, 100, 128)
    dy, a, b, ma, mb = (_dev_b(rng.uniform(-1, 1, shape)) for _ in range(5))
    alpha = torch.tensor([0.37], device="cuda")
    da0, db0 = torch.empty_like(a), torch.empty_like(b)
    dal0 = torch.zeros((1,), device="cuda")
    ops.weighted_sum_bf16_bwd(dy, a, b, alpha, da0, db0, dal0)
    want_a, want_b = ops.mul_bf16(da0, ma), ops.mul_bf16(db0, mb)
    for use_a, use_b in ((True, True), (True, False), (False, True)):
        da1, db1 = torch.empty_like(a), torch.empty_like(b)
        dal1 = torch.zeros((1,), device="cuda")
        ops.weighted_sum_bf16_bwd(dy, a, b, alpha, da1, db1, dal1, ma=ma if use_a else None, mb=mb if use_b else None)
        torch.cuda.synchronize()
        assert torch.equal(da1, want_a if use_a else da0)
        assert torch.equal(db1, want_b if use_b else db0)
        assert torch.equal(dal1, dal0)


@pytest.mark.parametrize("h,w,n", [(16, 64, 1), (16, 31, 2), (32, 100, 3), (48, 333, 1), (64, 500, 2), (16, 8, 1)])
def test_bf16_audio_conv_equals_separate_launches(h, w, n):
    """mgx_audio_conv3_bf16 (from-audio 1x1 convolution + activation and the 3x3 convolution + activation behind it
    in one launch, the 128-channel tensor re-computed per tile) against mgx_pointwise2_bf16_fwd followed by
    mgx_conv_gemm_bf16: the same arithmetic on the same values in the same order, so the outputs must be equal bit
    for bit, on ragged tiles, at the image borders (the convolution pads the ACTIVATION with zeros, not the audio)
    and across images."""
    M.clear_session()
    a = L.Conv2D(128, (1, 1), padding="same", kernel_initializer="he_normal")
    b = L.Conv2D(32, (3, 3), padding="same", kernel_initializer="he_normal")
    a(Input(shape=(h, w, 2)))
    b(Input(shape=(h, w, 128)))
    rng = np.random.default_rng(h * 100 + w)
    a.bias.assign(rng.uniform(-0.5, 0.5, (128,)).astype(np.float32))
    b.bias.assign(rng.uniform(-0.3, 0.3, (32,)).astype(np.float32))
    x = torch.from_numpy(rng.uniform(-1, 1, (n, h, w, 2)).astype(np.float32)).cuda()
    alpha_a, alpha_b = 11000.0, 7.0
    mid = torch.empty((n, h, w, 128), dtype=torch.bfloat16, device="cuda")
    want = torch.empty((n, h, w, 32), dtype=torch.bfloat16, device="cuda")
    ops.pointwise2_bf16_fwd(x, a.kernel.value(), a.bias.value(), mid, None, n * h * w, alpha_a)
    ops.conv_gemm_bf16(b._desc("f", n, h, w), mid, b.kernel.value(), b.bias.value(), want, act_alpha=alpha_b)
    assert ops.audio_conv3_bf16_supported(n, h, w)
    assert not ops.audio_conv3_bf16_supported(n, 24, w) and not ops.audio_conv3_bf16_supported(n, 8, w)
    got = torch.full_like(want, float("nan"))
    ops.audio_conv3_bf16(x, a.kernel.value(), a.bias.value(), alpha_a, b.kernel.value(), b.bias.value(), alpha_b, got)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got.float()).all()), "positions left unwritten"
    diff = (got.float() - want.float()).abs()
    where = np.unravel_index(int(diff.argmax()), tuple(diff.shape))
    assert torch.equal(got.view(torch.int16), want.view(torch.int16)), "max abs difference %.3e at %s" % (float(diff.max()), where)
