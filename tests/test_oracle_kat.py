"""CPU tests that pin the ORACLE: hand-derived known answers for every layer/loss of
trainer/models.py, the independent naive numpy restatement (oracle/naive.py), parameter counts,
and the committed golden vectors.  The reference holds no tests or fixtures for this path, so
these are the only anchors ("parity unpinned" with respect to the reference's own outputs).
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import naive as N
from oracle import ref_step as R

T = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)  # noqa: E731


def test_param_counts_match_layer_arithmetic():
    sh = R.weight_shapes(3)
    names = R.layer_names(3)
    count = lambda s, f, w: sum(int(np.prod(sh[n])) for n in R.trainable_names(names, s, f, w))  # noqa: E731
    assert count(0, False, "g") == 226602 and count(0, False, "d") == 252769
    # each G grow block adds 153978 weights plus its own LayerNorm (2*H*W*2) and drops the old one
    assert count(1, False, "g") == 226602 - 12000 + 153978 + 48000
    assert count(2, False, "g") == 226602 - 12000 + 2 * 153978 + 192000
    # each D grow block adds 100128; the straight twin drops the previous from-audio conv (384)
    assert count(1, False, "d") == 252769 + 100128 - 384
    assert count(1, True, "d") == 252769 + 100128 + 1
    assert count(1, True, "g") == count(1, False, "g") + 1


def test_losses_known_answers():
    fake, real = T([[0.0], [0.0]]), T([[1.0], [3.0]])
    assert abs(float(R.discriminator_loss(fake, real)) - (-1.9998)) < 1e-12
    assert abs(float(R.generator_loss(fake, real)) - 0.0002) < 1e-12
    same = T([[0.5]] * 4)
    assert abs(float(R.discriminator_loss(same, same)) - (1e-4 * 0.5 - 9.99e-5 * 0.5)) < 1e-15
    assert abs(float(R.generator_loss(same, same)) - 1e-4 * 0.5) < 1e-15


def test_soft_rectifier_known_answers():
    x = T([0.0, 50.0, -50.0])
    y = R.soft_rectifier(x, 20000.0)
    assert float(y[0]) == 0.0
    assert abs(float(y[1]) - (1.0 + 50.0 / 20000.0)) < 1e-12 and abs(float(y[2]) + (1.0 + 50.0 / 20000.0)) < 1e-12
    x0 = torch.zeros(1, dtype=torch.float64, requires_grad=True)
    g, = torch.autograd.grad(R.soft_rectifier(x0, 7.0).sum(), x0)
    assert abs(float(g) - (1.0 + 1.0 / 7.0)) < 1e-12


def test_same_padding_of_4x4_on_height_4():
    """4x4 'same' on H=4 reads rows -1..+2 around each output row (pad 1 before, 2 after)."""
    x = np.zeros((1, 4, 6, 1))
    x[0, :, 3, 0] = [1.0, 10.0, 100.0, 1000.0]
    k = np.zeros((4, 4, 1, 1))
    k[:, 1, 0, 0] = [1.0, 2.0, 3.0, 4.0]          # taps along H at kernel column 1 (= centre for pad 1)
    y = R.conv2d(T(x), T(k), T([0.0])).numpy()[0, :, 3, 0]
    # y[i] = sum_a k[a] * x[i + a - 1]
    want = [2 * 1 + 3 * 10 + 4 * 100, 1 * 1 + 2 * 10 + 3 * 100 + 4 * 1000, 1 * 10 + 2 * 100 + 3 * 1000, 1 * 100 + 2 * 1000]
    assert np.allclose(y, want)
    assert np.allclose(N.conv2d(x, k, np.zeros(1))[0, :, 3, 0], want)


def test_upsample_then_2x2_stride2_is_a_1x1_with_summed_taps():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 5, 2))
    k = rng.standard_normal((2, 2, 2, 7))
    b = rng.standard_normal(7)
    y = R.conv2d(R.upsample2(T(x)), T(k), T(b), strides=(2, 2), padding="valid").numpy()
    want = x @ k.sum(axis=(0, 1)) + b
    assert np.allclose(y, want)


def test_weighted_sum_endpoints_and_layernorm_and_mbstd():
    rng = np.random.default_rng(1)
    a, b = rng.standard_normal((2, 3, 4, 2)), rng.standard_normal((2, 3, 4, 2))
    assert np.array_equal(R.weighted_sum(T(a), T(b), 0.0).numpy(), a)
    assert np.array_equal(R.weighted_sum(T(a), T(b), 1.0).numpy(), b)
    x = rng.standard_normal((3, 4, 5, 2)) * 3.0 + 1.0
    y = R.layer_norm(T(x), T(np.ones((4, 5, 2))), T(np.zeros((4, 5, 2)))).numpy()
    for i in range(3):
        var = x[i].var()
        assert abs(y[i].mean()) < 1e-12 and abs(y[i].var() - var / (var + 1e-3)) < 1e-12
    const = R.layer_norm(T(np.full((1, 4, 5, 2), 3.0)), T(np.ones((4, 5, 2))), T(np.zeros((4, 5, 2))))
    assert float(const.abs().max()) == 0.0
    one = rng.standard_normal((1, 4, 6, 3))
    same = np.concatenate([one] * 4, axis=0)
    out = R.minibatch_stddev(T(same)).numpy()
    assert out.shape == (4, 5, 6, 3) and np.allclose(out[:, 4], 1e-4)
    assert np.array_equal(out[:, :4], same)


LAYER_CASES = [
    ("conv same 3x3", lambda x, k, b: (R.conv2d(T(x), T(k), T(b)), N.conv2d(x, k, b)), (2, 5, 7, 3), (3, 3, 3, 4)),
    ("conv same 4x4", lambda x, k, b: (R.conv2d(T(x), T(k), T(b)), N.conv2d(x, k, b)), (2, 4, 7, 3), (4, 4, 3, 4)),
    ("conv same 2x2", lambda x, k, b: (R.conv2d(T(x), T(k), T(b)), N.conv2d(x, k, b)), (2, 4, 7, 3), (2, 2, 3, 4)),
    ("conv valid s2", lambda x, k, b: (R.conv2d(T(x), T(k), T(b), (2, 2), "valid"), N.conv2d(x, k, b, (2, 2), "valid")),
     (2, 4, 8, 3), (2, 2, 3, 4)),
    ("conv valid (1,2)", lambda x, k, b: (R.conv2d(T(x), T(k), T(b), (1, 2), "valid"), N.conv2d(x, k, b, (1, 2), "valid")),
     (2, 1, 10, 2), (1, 2, 2, 4)),
]


@pytest.mark.parametrize("name,fn,xs,ks", LAYER_CASES)
def test_conv_against_naive(name, fn, xs, ks):
    rng = np.random.default_rng(2)
    x, k, b = rng.standard_normal(xs), rng.standard_normal(ks), rng.standard_normal(ks[-1])
    a, n = fn(x, k, b)
    assert np.allclose(a.numpy(), n, atol=1e-12), name


@pytest.mark.parametrize("ks", [(2, 1), (1, 2), (1, 15)])
def test_conv_transpose_against_naive(ks):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 3, 4, 5))
    k = rng.standard_normal(ks + (6, 5))
    b = rng.standard_normal(6)
    assert np.allclose(R.conv2d_transpose(T(x), T(k), T(b), ks).numpy(), N.conv2d_transpose(x, k, b, ks), atol=1e-12)


def test_small_layers_against_naive():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((8, 4, 6, 2))
    assert np.allclose(R.upsample2(T(x)).numpy(), N.upsample2(x))
    assert np.allclose(R.avgpool2(T(x)).numpy(), N.avgpool2(x))
    g, bt = rng.standard_normal((4, 6, 2)), rng.standard_normal((4, 6, 2))
    assert np.allclose(R.layer_norm(T(x), T(g), T(bt)).numpy(), N.layer_norm(x, g, bt), atol=1e-12)
    for n in (1, 2, 4, 8):
        assert np.allclose(R.minibatch_stddev(T(x[:n])).numpy(), N.minibatch_stddev(x[:n]), atol=1e-12)
    assert np.allclose(R.soft_rectifier(T(x), 3.0).numpy(), N.soft_rectifier(x, 3.0))
    assert np.allclose(R.weighted_sum(T(x), T(2 * x), 0.3).numpy(), N.weighted_sum(x, 2 * x, 0.3))


def test_adam_first_step_is_sign_like():
    p, g, v = T([1.0, 1.0, 1.0]), T([0.3, -2.0, 0.0]), T([0.0, 0.0, 0.0])
    p2, v2 = R.adam_update(p, g, v, 1)
    lr_t = 0.0005 * math.sqrt(1 - 0.99)
    want = 1.0 - lr_t * np.array([0.3, -2.0, 0.0]) / (np.sqrt(0.01 * np.array([0.09, 4.0, 0.0])) + 1e-7)
    assert np.allclose(p2.numpy(), want, atol=1e-15)
    assert abs(float(p2[0]) - (1 - 0.0005)) < 1e-8 and float(p2[2]) == 1.0


def test_activation_alpha_table():
    assert [R.activation_alphas(s) for s in range(7)] == [
        (40000.0, 20000.0), (40000.0, 20000.0), (60000.0, 40000.0), (90000.0, 80000.0), (120000.0, 120000.0),
        (150000.0, 150000.0), (200000.0, 11000.0)]


def test_oracle_reproduces_golden_stage0():
    """The committed fixture is what the oracle produces today (guards against oracle drift)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "step_s0_n4.npz")
    with np.load(path) as f:
        gold = {k.replace("~", "/"): f[k] for k in f.files}
    W = R.synthetic_weights(1, seed=100)
    real, z4 = R.synthetic_inputs(0, 4, seed=0)
    res = R.train_step(W, 1, 0, False, real, z4)
    got = np.array([res.metrics[k] for k in ("delta_0", "cu_0", "ci_0", "delta_1", "cu_1", "ci_1")])
    assert np.allclose(got, gold["metrics"], atol=1e-10)
    assert np.allclose([res.d_loss, res.g_loss], gold["losses"], atol=1e-10)
    assert np.allclose(res.fake_logits_d_step.reshape(-1), gold["fake_logits"], atol=1e-10)
