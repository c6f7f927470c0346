"""GPU parity of every kernel behind the C-ABI, layer by layer, against the CPU oracle
(oracle/ref_step.py, float64 + autograd) on the same seeded inputs.  Tolerance: float32 device
arithmetic vs float64 truth, relative L2 error <= 2e-5 for outputs and gradients (stated per test).
"""
import numpy as np
import pytest
import torch

import music_generator_amd as M
from music_generator_amd import layers as L
from music_generator_amd.graph import Input, Model
from oracle import ref_step as R

from helpers import assert_close, randomize, rel_l2, run_single

pytestmark = pytest.mark.gpu

TOL = 2e-5


def _oracle(fn, x, dy, weights):
    """Evaluate fn(x, **weights) in float64 with autograd; returns (y, dx, {name: grad})."""
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in weights.items()}
    y = fn(xt, wt)
    loss = (y * torch.tensor(dy, dtype=torch.float64)).sum()
    gs = torch.autograd.grad(loss, [xt] + list(wt.values()), allow_unused=True)
    grads = {k: (g.numpy() if g is not None else None) for k, g in zip(wt.keys(), gs[1:])}
    return y.detach().numpy(), gs[0].numpy(), grads


def _check(layer_fn, oracle_fn, in_shape, batch, seed=0, tol=TOL, scale=None):
    inp = Input(shape=in_shape)
    model = Model(inp, layer_fn(inp))
    w = randomize(model, seed, scale)
    rng = np.random.default_rng(seed + 17)
    x = rng.uniform(-1, 1, (batch,) + tuple(in_shape)).astype(np.float32)
    out_shape = (batch,) + tuple(model.output_shape[1:])
    dy = rng.standard_normal(out_shape).astype(np.float32)
    y, dx, grads = run_single(model, x, dy)
    short = {k.rsplit("/", 1)[1]: v for k, v in w.items()}
    yo, dxo, go = _oracle(oracle_fn, x, dy, short)
    assert_close(y, yo, tol, "output")
    assert_close(dx, dxo, tol, "input gradient")
    for k, v in grads.items():
        assert_close(v, go[k.rsplit("/", 1)[1]], tol, "gradient of " + k)


CONV_CASES = [
    # (H, W, Cin, Cout, ksize, strides, padding, batch)
    (4, 37, 128, 32, (3, 3), (1, 1), "same", 3),
    (4, 75, 32, 32, (4, 4), (1, 1), "same", 2),       # 4x4 'same' on H=4: pad 1 before / 2 after
    (4, 750, 32, 32, (4, 4), (1, 1), "same", 3),      # 4x4 tile kernel: 12 column tiles of 64, ragged last one
    (8, 130, 32, 32, (4, 4), (1, 1), "same", 2),      # 4x4 tile kernel over two tile rows of 4
    (8, 50, 32, 32, (3, 3), (1, 1), "same", 2),
    (4, 130, 32, 32, (3, 3), (1, 1), "same", 2),      # 4-row tiles over an image of 4 rows: padding tap rows skipped
    (12, 70, 32, 32, (3, 3), (1, 1), "same", 2),      # 4-row tiles over three tile rows: nothing to skip inside
    (16, 70, 128, 32, (3, 3), (1, 1), "same", 2),     # LDS-tiled path, 4 input-channel chunks
    (8, 33, 32, 32, (3, 3), (1, 1), "same", 3),       # LDS-tiled path, ragged right edge
    (12, 45, 64, 32, (3, 3), (1, 1), "same", 2),      # Winograd kernel: 4 channel steps, partial second tile row
    (24, 40, 96, 32, (3, 3), (1, 1), "same", 1),      # Winograd kernel: 6 channel steps (rolled step loop)
    (20, 33, 32, 32, (3, 3), (1, 1), "same", 2),      # Winograd kernel: 4 valid rows in the last tile row
    (16, 640, 32, 32, (3, 3), (1, 1), "same", 14),    # Winograd kernel: 280 tiles, workgroups walk two tiles
    (24, 95, 32, 32, (2, 2), (1, 1), "same", 1),
    (8, 50, 32, 32, (2, 2), (1, 1), "same", 2),       # 2x2 'same': pad 0 before / 1 after
    (8, 52, 32, 32, (2, 2), (2, 2), "valid", 3),
    (16, 50, 32, 32, (2, 2), (2, 2), "valid", 3),     # tile kernel: stride-2 gather forward, depth-to-space data-gradient
    (32, 132, 32, 32, (2, 2), (2, 2), "valid", 2),    # the same over two tile rows and three tile columns
    (6, 41, 2, 128, (1, 1), (1, 1), "same", 2),
    (6, 41, 32, 128, (1, 1), (1, 1), "same", 2),
    (16, 70, 32, 128, (1, 1), (1, 1), "same", 3),     # tile kernels: weight gradient over four output chunks
    (4, 45, 32, 128, (1, 1), (1, 1), "same", 2),      # the same on 4-row tiles
    (1, 100, 2, 128, (1, 2), (1, 2), "valid", 5),     # generator stage-0 input conv
    (8, 60, 2, 128, (2, 2), (2, 2), "valid", 2),      # generator grow-block input conv
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("act", [0.0, 7.0])
def test_conv2d(case, act):
    h, w, ci, co, ks, st, pad, batch = case

    def build(t):
        t = L.Conv2D(co, ks, strides=st, padding=pad, kernel_initializer="he_normal")(t)
        return M.SoftRectifier(start_alpha=act)(t) if act else t

    def oracle(x, wt):
        y = R.conv2d(x, wt["kernel"], wt["bias"], strides=st, padding=pad)
        return R.soft_rectifier(y, act) if act else y

    _check(build, oracle, (h, w, ci), batch)


@pytest.mark.parametrize("ci", [32, 128])
def test_winograd_equals_direct_kernel_at_full_size(ci, monkeypatch):
    """Size-independent check at the benchmark's shapes (batch 64, 32x6000, too large for the float64
    oracle): the Winograd kernel and the direct tile kernel are two float32 evaluations of the same
    convolution + tanh(x)+x/alpha, output and saved derivative.  Tolerance: relative L2 <= 2e-6 (each
    kernel is within ~4e-7 of float64 at the oracle's sizes), max abs <= 2e-5 * max|y|."""
    n, h, w = 64, 32, 6000
    inp = Input(shape=(h, w, ci))
    t = L.Conv2D(32, (3, 3), padding="same", kernel_initializer="he_normal")(inp)
    model = Model(inp, M.SoftRectifier(start_alpha=20000.0)(t))
    randomize(model, 3)
    x = torch.rand((n, h, w, ci), device="cuda") * 2 - 1
    outs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("MGX_WINOGRAD", flag)
        y, tape = model.forward(x, save=True)
        saved = [v for entry in tape if isinstance(entry, tuple) for v in entry if isinstance(v, torch.Tensor)]
        outs.append((y.clone(), [v.clone() for v in saved if v.shape == y.shape and v.data_ptr() not in (y.data_ptr(), x.data_ptr())]))
        del y, tape, saved
    (ya, sa), (yb, sb) = outs
    pairs = [(ya, yb)] + list(zip(sa, sb))
    assert len(sa) == len(sb) == 1          # the saved derivative of the fused activation
    for a, b in pairs:
        diff = (a - b).double()
        rel = float(diff.norm() / b.double().norm())
        assert rel <= 2e-6, rel
        assert float(diff.abs().max()) <= 2e-5 * float(b.abs().max())


CONVT_CASES = [
    (4, 50, 128, 16, (2, 1), 3), (8, 50, 32, 64, (1, 2), 2), (4, 50, 32, 64, (1, 15), 2),
    (1, 50, 128, 4, (2, 1), 3), (2, 50, 8, 16, (2, 1), 3),
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv2d_transpose(case):
    h, w, ci, co, ks, batch = case
    _check(lambda t: L.Conv2DTranspose(co, ks, strides=ks, padding="valid", kernel_initializer="he_normal")(t),
           lambda x, wt: R.conv2d_transpose(x, wt["kernel"], wt["bias"], ks), (h, w, ci), batch)


@pytest.mark.parametrize("ci,co", [(16, 32), (64, 128), (128, 256), (256, 64), (64, 8), (8, 2), (4, 8)])
def test_dense(ci, co):
    _check(lambda t: L.Dense(co)(t), lambda x, wt: R.dense(x, wt["kernel"], wt["bias"]), (3, 45, ci), 2)


@pytest.mark.parametrize("ci,co", [(64, 128), (128, 64), (256, 128)])
def test_dense_many_rows(ci, co):
    """67 200 rows = 525 row tiles of 128: every workgroup of the LDS-weight GEMM kernel walks two or
    three tiles (the A operands of the next tile are loaded during the current one), forward and
    data-gradient."""
    _check(lambda t: L.Dense(co)(t), lambda x, wt: R.dense(x, wt["kernel"], wt["bias"]), (4, 4200, ci), 4)


def test_dense1_with_static_opt_tanh():
    def build(t):
        return M.StaticOptTanh(alpha=5.0)(L.Dense(1)(L.Flatten()(t)))

    def oracle(x, wt):
        return R.soft_rectifier(R.dense(x.reshape(x.shape[0], -1), wt["kernel"], wt["bias"]), 5.0)

    _check(build, oracle, (5, 30, 32), 6, scale=0.01)


def test_upsample_and_avgpool():
    _check(lambda t: L.UpSampling2D()(t), lambda x, wt: R.upsample2(x), (3, 37, 2), 3)
    _check(lambda t: L.AveragePooling2D()(t), lambda x, wt: R.avgpool2(x), (6, 38, 2), 3)


def test_layer_normalization():
    _check(lambda t: L.LayerNormalization(axis=[1, 2, 3])(t),
           lambda x, wt: R.layer_norm(x, wt["gamma"], wt["beta"]), (4, 750, 2), 5)
    _check(lambda t: L.LayerNormalization(axis=[1, 2, 3])(t),
           lambda x, wt: R.layer_norm(x, wt["gamma"], wt["beta"]), (16, 3000, 2), 2)
    # 2^20 elements per sample: the second pass walks the batch per element position (gamma / beta read once)
    _check(lambda t: L.LayerNormalization(axis=[1, 2, 3])(t),
           lambda x, wt: R.layer_norm(x, wt["gamma"], wt["beta"]), (64, 8192, 2), 3)


@pytest.mark.parametrize("batch", [1, 2, 4, 8, 12])
def test_minibatch_stddev(batch):
    _check(lambda t: M.MinibatchStdDev()(t), lambda x, wt: R.minibatch_stddev(x), (4, 75, 32), batch)


def test_soft_rectifier_standalone():
    _check(lambda t: M.SoftRectifier(start_alpha=3.0)(t), lambda x, wt: R.soft_rectifier(x, 3.0), (3, 10, 4), 2)


def test_add_and_fanout():
    """Six parallel Dense branches from one tensor, summed: exercises Add and gradient fan-in."""
    def build(t):
        return L.Add()([L.Dense(8)(t) for _ in range(6)])

    inp = Input(shape=(3, 21, 16))
    model = Model(inp, build(inp))
    w = randomize(model, 3)
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (2, 3, 21, 16)).astype(np.float32)
    dy = rng.standard_normal((2, 3, 21, 8)).astype(np.float32)
    y, dx, grads = run_single(model, x, dy)
    names = sorted({k.rsplit("/", 1)[0] for k in w})
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    yo = sum(R.dense(xt, wt[n + "/kernel"], wt[n + "/bias"]) for n in names)
    gs = torch.autograd.grad((yo * torch.tensor(dy, dtype=torch.float64)).sum(), [xt] + list(wt.values()))
    assert_close(y, yo.detach().numpy(), TOL, "sum of branches")
    assert_close(dx, gs[0].numpy(), TOL, "fan-in input gradient")
    for k, g in zip(wt.keys(), gs[1:]):
        assert_close(grads[k], g.numpy(), TOL, k)


@pytest.mark.parametrize("alpha", [0.0, 0.37, 1.0])
def test_weighted_sum(alpha):
    inp = Input(shape=(4, 30, 2))
    a = L.Dense(2)(inp)
    b = L.Dense(2)(inp)
    ws = M.WeightedSum()
    model = Model(inp, ws([a, b]))
    w = randomize(model, 1)
    ws.alpha.assign(np.asarray(alpha, dtype=np.float32))
    rng = np.random.default_rng(2)
    x = rng.uniform(-1, 1, (3, 4, 30, 2)).astype(np.float32)
    dy = rng.standard_normal((3, 4, 30, 2)).astype(np.float32)
    y, dx, grads = run_single(model, x, dy)
    names = sorted({k.rsplit("/", 1)[0] for k in w if "dense" in k})
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items() if "dense" in k}
    al = torch.tensor(float(alpha), dtype=torch.float64, requires_grad=True)
    ya = R.dense(xt, wt[names[0] + "/kernel"], wt[names[0] + "/bias"])
    yb = R.dense(xt, wt[names[1] + "/kernel"], wt[names[1] + "/bias"])
    yo = R.weighted_sum(ya, yb, al)
    gs = torch.autograd.grad((yo * torch.tensor(dy, dtype=torch.float64)).sum(), [xt, al] + list(wt.values()))
    assert_close(y, yo.detach().numpy(), TOL, "blend")
    # at the ends of the fade the blend is a bit-exact pass-through of one input: (1-alpha)*a + alpha*b with a
    # factor of exactly 0 and 1 (the reference's fade-in starts at alpha = 0 and ends at 1)
    if alpha in (0.0, 1.0):
        branch = Model(inp, a if alpha == 0.0 else b)
        passed = branch(x).cpu().numpy()
        assert np.array_equal(y, passed), "alpha = %g is not a pass-through" % alpha
    assert_close(dx, gs[0].numpy(), TOL, "input gradient")
    assert_close(grads[ws.alpha.name], gs[1].numpy(), TOL, "alpha gradient")
    for k, g in zip(wt.keys(), gs[2:]):
        assert_close(grads[k], g.numpy(), TOL, k)


def test_losses_known_answers():
    """discriminator_loss(fake=[0,0], real=[1,3]) = -1.9998 and generator_loss = 0.0002; equal
    means give 1e-4*ci - 9.99e-5*cu and 1e-4*ci (trainer/models.py:468-484)."""
    fake = np.zeros((2, 1), np.float32)
    real = np.array([[1.0], [3.0]], np.float32)
    assert abs(float(M.discriminator_loss(fake, real)) - (-1.9998)) < 1e-6
    assert abs(float(M.generator_loss(fake, real)) - 0.0002) < 1e-6
    same = np.full((4, 1), 0.5, np.float32)
    assert abs(float(M.discriminator_loss(same, same)) - (1e-4 * 0.5 - 9.99e-5 * 0.5)) < 1e-9
    assert abs(float(M.generator_loss(same, same)) - 1e-4 * 0.5) < 1e-9


def test_adam_first_two_steps():
    inp = Input(shape=(2, 9, 4))
    model = Model(inp, L.Dense(5)(inp))
    w = randomize(model, 4)
    opt = M.Adam(lr=0.0005, beta_1=0, beta_2=0.99, epsilon=10e-8)
    rng = np.random.default_rng(9)
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in w.items()}
    vs = {k: torch.zeros_like(t) for k, t in p.items()}
    for t in (1, 2):
        flat = model.bind_grads()
        g = {v.name: rng.standard_normal(v.shape).astype(np.float32) for v in model.trainable_variables}
        for v in model.trainable_variables:
            v.grad.copy_(torch.from_numpy(g[v.name].reshape(-1)))
        opt.apply_model_gradients(model)
        for k in p:
            p[k], vs[k] = R.adam_update(p[k], torch.tensor(g[k], dtype=torch.float64), vs[k], t)
    for v in model.trainable_variables:
        assert_close(v.numpy(), p[v.name].numpy(), 1e-6, v.name)
    for k, val in opt.second_moments().items():
        assert_close(val, vs[k].numpy(), 1e-6, "v of " + k)


@pytest.mark.parametrize("stage,batch", [(0, 3), (1, 2)])
def test_evaluator_forward_matches_oracle(stage, batch):
    """Evaluator inference (trainer/evaluacion.py:27-110; 128 -> 128 convolutions 1x1 / 3x3 / 4x4 /
    2x2 'same' / 2x2 stride 2, Flatten, Dense(9), softmax) against the float64 restatement.  Weights
    are scaled so that activations stay O(1) through the 10 / 19 linear layers; bounds: logits
    rel-L2 1e-4, class probabilities abs 1e-5, Inception score rel 1e-5."""
    import music_generator_amd as M
    from music_generator_amd import evaluacion as E
    from oracle import ref_evaluator as RE
    M.clear_session()
    model = E.define_evaluator(stage + 1)[stage]
    rng = np.random.default_rng(40 + stage)
    kernels, biases = [], []
    convs = [l for l in model.layers if isinstance(l, M.layers.Conv2D)]
    for l in convs:
        fan_in = l.kernel_size[0] * l.kernel_size[1] * l.kernel.shape[2]
        k = (rng.standard_normal(l.kernel.shape) / np.sqrt(fan_in)).astype(np.float32)
        b = (0.05 * rng.standard_normal(l.bias.shape)).astype(np.float32)
        l.kernel.assign(k), l.bias.assign(b)
        kernels.append(k), biases.append(b)
    head = model.layers[-1]
    dk = (rng.standard_normal(head.kernel.shape) / np.sqrt(head.kernel.shape[0])).astype(np.float32)
    db = (0.1 * rng.standard_normal(head.bias.shape)).astype(np.float32)
    head.kernel.assign(dk), head.bias.assign(db)
    x = rng.uniform(-1, 1, (batch, 4 << stage, 750 << stage, 2)).astype(np.float32)
    probs = model(torch.from_numpy(x).cuda(), training=False).cpu().numpy()
    logits_ref, probs_ref = RE.forward(stage, kernels, biases, dk, db, x)
    assert probs.shape == (batch, 9) and np.allclose(probs.sum(axis=1), 1.0, atol=1e-6)
    assert np.abs(probs - probs_ref).max() < 1e-5, np.abs(probs - probs_ref).max()
    # logits recovered up to the row constant softmax removes
    lg = np.log(probs.astype(np.float64))
    lg += (logits_ref - lg).mean(axis=1, keepdims=True)
    assert rel_l2(lg, logits_ref) < 1e-4, rel_l2(lg, logits_ref)
    assert E.calculate_inception_score(probs) == pytest.approx(RE.inception_score(probs_ref), rel=1e-5)


def test_softmax_rows_wide_and_saturated():
    from music_generator_amd import ops
    rng = np.random.default_rng(9)
    x = rng.standard_normal((37, 200)).astype(np.float32) * 30.0       # wider than a wavefront, saturating
    xd = torch.from_numpy(x).cuda()
    y = torch.empty_like(xd)
    ops.softmax_rows(xd, y)
    want = torch.softmax(torch.from_numpy(x).double(), dim=1).numpy()
    assert np.abs(y.cpu().numpy() - want).max() < 2e-6
    with pytest.raises(ops.MgxError):
        ops.softmax_rows(xd, torch.empty((37, 199), device="cuda"))


def test_evaluator_fit_matches_oracle():
    """Two fitting steps of the stage-0 evaluator (trainer/evaluacion.py:158-161: Adam with a first
    moment, categorical cross-entropy on the softmax head) against float64 autograd: losses rel
    1e-5, gradients of the second batch rel-L2 1e-4, parameters after the second update abs 2e-7
    = lr/50 for 99.99 % of every tensor's entries and 2.1 lr for all (the first Adam updates are
    ~lr * sign(g): where a gradient of either step is ~0, float32 noise decides the direction).
    lr = 1e-5 here: with keras' default 1e-3 one sign-like step moves this 10-layer linear network
    far enough for float32 and float64 trajectories to part."""
    import music_generator_amd as M
    from music_generator_amd import evaluacion as E, ops
    from music_generator_amd import backend as K
    from oracle import ref_evaluator as RE
    M.clear_session()
    model = E.define_evaluator(1)[0]
    rng = np.random.default_rng(77)
    convs = [l for l in model.layers if isinstance(l, M.layers.Conv2D)]
    kernels, biases = [], []
    for l in convs:
        fan_in = l.kernel_size[0] * l.kernel_size[1] * l.kernel.shape[2]
        k = (rng.standard_normal(l.kernel.shape) / np.sqrt(fan_in)).astype(np.float32)
        b = (0.05 * rng.standard_normal(l.bias.shape)).astype(np.float32)
        l.kernel.assign(k), l.bias.assign(b)
        kernels.append(k), biases.append(b)
    head = model.layers[-1]
    dk = (rng.standard_normal(head.kernel.shape) / np.sqrt(head.kernel.shape[0])).astype(np.float32)
    db = (0.1 * rng.standard_normal(head.bias.shape)).astype(np.float32)
    head.kernel.assign(dk), head.bias.assign(db)
    batches = []
    for _ in range(2):
        x = rng.uniform(-1, 1, (4, 4, 750, 2)).astype(np.float32)
        y = np.eye(9, dtype=np.float32)[rng.integers(0, 9, 4)]
        batches.append((x, y))
    opt = M.Adam(lr=1e-5)
    losses = []
    for x, y in batches:
        xb, yb = model._to_device(x), E.labels_to_device(y)
        flat = model.bind_grads(0)
        flat.zero_()
        logits, tape = model.forward(xb, save=True)
        prob, rows, dlogits = K.empty(logits.shape), K.empty((4,)), K.empty(logits.shape)
        ops.softmax_xent(logits.contiguous(), yb, prob, rows, dlogits)
        model.backward(tape, dlogits, need_input_grad=False)
        grads = [v.grad.cpu().numpy().reshape(v.shape).copy() for l in convs + [head] for v in (l.kernel, l.bias)]
        opt.apply_model_gradients(model)
        losses.append(float(rows.mean()))
    ref_losses, ref_grads, ref_params = RE.fit_steps(0, kernels, biases, dk, db, batches, lr=1e-5)
    assert losses == pytest.approx(ref_losses, rel=1e-5)
    params = [v.numpy() for l in convs + [head] for v in (l.kernel, l.bias)]
    for g, rg, p, rp in zip(grads, ref_grads, params, ref_params):
        assert rel_l2(g, rg) < 1e-4, rel_l2(g, rg)
        diff = np.abs(p - rp)
        assert np.quantile(diff, 0.9999) < 2e-7 and diff.max() < 2.1e-5, (np.quantile(diff, 0.9999), diff.max())
    # the fitting loop itself: history shapes, validation on the training set, loss going down
    x = np.concatenate([b[0] for b in batches]); y = np.concatenate([b[1] for b in batches])
    hist = E.fit_evaluator(model, x, y, epochs=3, batch_size=4, optimizer=M.Adam(lr=1e-5), validation_data=(x, y))
    assert [len(hist[k]) for k in ("loss", "accuracy", "val_loss", "val_accuracy")] == [3, 3, 3, 3]
    assert np.isfinite(hist["loss"]).all() and hist["val_loss"][-1] < hist["val_loss"][0]


@pytest.mark.parametrize("g,m,per,shape", [(4, 4, 1, (4, 8, 2)), (4, 16, 4, (8, 100, 2)), (4, 6, 4, (4, 750, 2)),
                                           (2, 3, 2, (1, 100, 2)), (1, 5, 2, (4, 2))])
def test_gather_groups_takes_whole_minibatch_stddev_groups(g, m, per, shape):
    """mgx_gather_groups against an index gather: micro-batch k holds samples a * m + m0 + b (a < g, b < cnt),
    a-major, including the short last micro-batch (m not a multiple of the groups per micro-batch)."""
    from music_generator_amd import ops
    rng = np.random.default_rng(g * 100 + m)
    x = torch.from_numpy(rng.standard_normal((g * m,) + shape).astype(np.float32)).cuda()
    for m0 in range(0, m, per):
        cnt = min(per, m - m0)
        idx = (np.arange(g)[:, None] * m + m0 + np.arange(cnt)[None, :]).reshape(-1)
        got = ops.gather_groups(x, g, m, m0, cnt)
        assert torch.equal(got, x[torch.as_tensor(idx, device="cuda")])
    with pytest.raises(ops.MgxError):
        ops.gather_groups(x, g, m, m - 1, 2)          # the run would leave its group block


@pytest.mark.parametrize("n,h,w", [(2, 8, 100), (3, 16, 70), (2, 4, 45), (2, 12, 33)])
@pytest.mark.parametrize("alpha", [0.0, 0.37, 1.0])
def test_conv1x1_blend_equals_two_launches(n, h, w, alpha):
    """mgx_conv1x1_blend_f32 (1x1 convolution 32 -> 128 + activation with the fade-in blend in its epilogue) against
    mgx_conv_gemm_f32 followed by mgx_weighted_sum_fwd: bit for bit, on 8-row and 4-row tiles and ragged widths."""
    from music_generator_amd import ops
    M.clear_session()
    conv = L.Conv2D(128, (1, 1), padding="same", kernel_initializer="he_normal")
    conv(Input(shape=(h, w, 32)))
    rng = np.random.default_rng(n * 1000 + h + w)
    conv.bias.assign(rng.uniform(-0.5, 0.5, (128,)).astype(np.float32))
    x = torch.from_numpy(rng.uniform(-1, 1, (n, h, w, 32)).astype(np.float32)).cuda()
    old = torch.from_numpy(rng.uniform(-2, 2, (n, h, w, 128)).astype(np.float32)).cuda()
    al = torch.tensor([alpha], device="cuda")
    d = conv._desc("f", n, h, w)
    assert ops.conv1x1_blend_supported(d)
    mid = torch.empty((n, h, w, 128), device="cuda")
    want = torch.empty_like(mid)
    ops.conv_gemm(d, x, conv.kernel.value(), conv.bias.value(), mid, act_alpha=7.0)
    ops.weighted_sum_fwd(old, mid, al, want)
    got = torch.full_like(want, float("nan"))
    ops.conv1x1_blend(d, x, conv.kernel.value(), conv.bias.value(), old, al, got, 7.0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got).all()), "positions left unwritten"
    assert torch.equal(got, want), float((got - want).abs().max())


def test_discriminator_fade_in_forward_with_and_without_blend_epilogue():
    """A tape-free float32 forward of a fade-in discriminator with the blend in the 1x1 convolution's epilogue
    (Model.use_conv_chains, the default) and layer by layer: the same logits bit for bit; with a tape the
    WeightedSum layer runs."""
    from music_generator_amd import ops
    M.clear_session()
    d = M.define_discriminator(3)
    model = d[2][1]
    randomize(model, seed=3)
    for layer in model.layers:
        if isinstance(layer, M.WeightedSum):
            M.backend.set_value(layer.alpha, 0.37)
    x = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (4, 16, 3000, 2)).astype(np.float32)).cuda()
    blends = []
    orig = ops.conv1x1_blend
    ops.conv1x1_blend = lambda *a, **k: (blends.append(1), orig(*a, **k))[1]
    flag = Model.use_conv_chains
    try:
        Model.use_conv_chains = True
        y1 = model(x).clone()
        assert len(blends) == 1
        model.forward(x, save=True)
        assert len(blends) == 1
        Model.use_conv_chains = False
        y0 = model(x).clone()
        assert len(blends) == 1
    finally:
        Model.use_conv_chains = flag
        ops.conv1x1_blend = orig
    assert torch.equal(y0, y1)
