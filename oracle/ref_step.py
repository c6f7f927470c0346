"""CPU ORACLE (test infrastructure, never shipped, never on the product path).

A PyTorch-CPU restatement of the progressive-GAN training step of the reference
(`trainer/models.py`, driven by `trainer/task.py`).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.

PARITY UNPINNED: the reference holds no tests, fixtures or golden vectors for this
path and its arithmetic lives in TensorFlow/Keras (requirements.txt:3-4), which is not
installed here, so nothing produced by the reference itself anchors these functions.
They are pinned only by hand-derived known-answer tests (tests/test_oracle_kat.py) and
by an independent naive numpy restatement of every layer (oracle/naive.py).

Conventions (trainer/models.py, Keras defaults): NHWC tensors, float32 in the
reference; this oracle runs in float64 by default (dtype argument) so that it can act
as the "truth" a float32 device result is compared against.

Every function cites the reference lines it follows.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# constants of the reference
# --------------------------------------------------------------------------------------
LATENT_DIM = (1, 100, 2)          # trainer/task.py:78
BASE_SHAPE = (4, 750, 2)          # trainer/models.py:309
LN_EPS = 1e-3                     # Keras LayerNormalization default epsilon (models.py:400,405,451)
ADAM = dict(lr=0.0005, beta_1=0.0, beta_2=0.99, epsilon=10e-8)   # models.py:502-503


def stage_hw(stage: int, base_shape: Tuple[int, ...] = BASE_SHAPE) -> Tuple[int, int]:
    """Resolution of progressive stage `stage` (dataset.py:85-93; doubling per block).
    `base_shape` is the `input_shape` argument of define_discriminator (models.py:309): the
    discriminator family can be grown from any base resolution, the generator's is fixed."""
    return base_shape[0] * (1 << stage), base_shape[1] * (1 << stage)


def activation_alphas(stage: int, base_h: int = BASE_SHAPE[0]) -> Tuple[float, float]:
    """(StaticOptTanh alpha, SoftRectifier alpha) chosen by add_discriminator_block for the
    block that creates `stage` (>=1); keyed in the reference on the OLD model's height
    (models.py:232-251), which is `base_h` doubled stage - 1 times.  Stage 0 uses
    40000 / 20000 (models.py:341, 200)."""
    if stage == 0:
        return 40000.0, 20000.0
    old_h = base_h * (1 << (stage - 1))
    alpha, soft = 40000.0, 20000.0
    if old_h == 8:
        alpha, soft = 60000.0, 40000.0
    if old_h == 16:
        alpha, soft = 90000.0, 80000.0
    if old_h == 32:
        alpha, soft = 120000.0, 120000.0
    if old_h == 64:
        alpha, soft = 150000.0, 150000.0
    if old_h == 128:
        alpha, soft = 200000.0, 11000.0
    if old_h == 256:
        alpha, soft = 350000.0, 200000.0
    return alpha, soft


# --------------------------------------------------------------------------------------
# layer names: Keras-style auto names in the creation order of task.py:82-86
# (define_discriminator(n) first, then define_generator(n))
# --------------------------------------------------------------------------------------
class _Namer:
    def __init__(self):
        self.count: Dict[str, int] = {}

    def __call__(self, base: str) -> str:
        i = self.count.get(base, 0)
        self.count[base] = i + 1
        return base if i == 0 else "%s_%d" % (base, i)


def layer_names(n_blocks: int) -> Dict[str, object]:
    """Names of every weighted layer, replaying the construction order of
    define_discriminator (models.py:309-354 / 229-305) then define_generator
    (models.py:412-463 / 358-408)."""
    nm = _Namer()
    d: List[dict] = []
    # stage-0 discriminator, models.py:312-341
    d.append(dict(from_audio=nm("conv2d"), convs=[nm("conv2d") for _ in range(9)], dense=nm("dense")))
    for s in range(1, n_blocks):
        # add_discriminator_block, models.py:254-297
        blk = dict(from_audio=nm("conv2d"), convs=[nm("conv2d") for _ in range(9)], to128=nm("conv2d"))
        blk["ws"] = nm("weighted_sum")
        d.append(blk)
    g: List[dict] = []
    # stage-0 generator, models.py:415-451
    g0 = dict(featured=nm("conv2d"), branches=[])
    for _ in range(4):
        ct1 = nm("conv2d_transpose"); de1 = nm("dense")
        ct2 = nm("conv2d_transpose"); de2 = nm("dense")
        ct3 = nm("conv2d_transpose"); de3 = nm("dense")
        g0["branches"].append([ct1, de1, ct2, de2, ct3, de3])
    g0["tail"] = [nm("dense") for _ in range(4)]
    g0["ln"] = nm("layer_normalization")
    g.append(g0)
    for s in range(1, n_blocks):
        # add_generator_block, models.py:362-405
        gb = dict(featured=nm("conv2d"), branches=[])
        for _ in range(6):
            ct1 = nm("conv2d_transpose"); de1 = nm("dense")
            ct2 = nm("conv2d_transpose"); de2 = nm("dense")
            gb["branches"].append([ct1, de1, ct2, de2])
        gb["tail"] = [nm("dense") for _ in range(4)]
        gb["ln"] = nm("layer_normalization")
        gb["ws"] = nm("weighted_sum")
        gb["ln_fade"] = nm("layer_normalization")
        g.append(gb)
    return dict(d=d, g=g)


# --------------------------------------------------------------------------------------
# layer arithmetic (TensorFlow/Keras semantics restated)
# --------------------------------------------------------------------------------------
def conv2d(x, kernel, bias, strides=(1, 1), padding="same"):
    """Keras Conv2D, NHWC, kernel (kh,kw,Cin,Cout), cross-correlation.  'same' pads
    (k-1)//2 before and the rest after (TF convention for stride 1)."""
    kh, kw = kernel.shape[0], kernel.shape[1]
    xn = x.permute(0, 3, 1, 2)
    if padding == "same":
        assert strides == (1, 1)
        pt, pl = (kh - 1) // 2, (kw - 1) // 2
        xn = F.pad(xn, (pl, kw - 1 - pl, pt, kh - 1 - pt))
    y = F.conv2d(xn, kernel.permute(3, 2, 0, 1), bias, stride=strides)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose(x, kernel, bias, strides):
    """Keras Conv2DTranspose with kernel == stride, 'valid': kernel (kh,kw,Cout,Cin);
    out[n, i*kh+a, j*kw+b, co] = sum_ci x[n,i,j,ci] * K[a,b,co,ci] + bias[co]."""
    kh, kw, co, ci = kernel.shape
    assert (kh, kw) == tuple(strides)
    n, h, w, _ = x.shape
    y = torch.einsum("nijc,aboc->niajbo", x, kernel)
    return y.reshape(n, h * kh, w * kw, co) + bias


def dense(x, kernel, bias):
    """Keras Dense on the last axis of a rank-4 tensor (tensordot), no activation."""
    return x @ kernel + bias


def upsample2(x):
    """Keras UpSampling2D() nearest x2 (models.py:362)."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def avgpool2(x):
    """Keras AveragePooling2D() 2x2 stride 2 (models.py:292)."""
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).mean(dim=(2, 4))


def layer_norm(x, gamma, beta):
    """Keras LayerNormalization(axis=[1,2,3]): per-sample moments over H*W*C, biased
    variance, epsilon 1e-3, gamma/beta of shape (H,W,C) (models.py:400,405,451)."""
    mean = x.mean(dim=(1, 2, 3), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2, 3), keepdim=True)
    return (x - mean) * torch.rsqrt(var + LN_EPS) * gamma + beta


def weighted_sum(a, b, alpha):
    """models.py:30-35: ((1-a) * input1) + (a * input2)."""
    return (1.0 - alpha) * a + alpha * b


def minibatch_stddev(x):
    """models.py:167-192.  The comments there assume NCHW but the data is NHWC, so the
    statistic is tiled to (N,1,W,C) and concatenated on axis 1 (a ROW is appended)."""
    n = x.shape[0]
    g = min(4, n)
    s = x.shape
    y = x.reshape(g, -1, s[1], s[2], s[3])
    y = y - y.mean(dim=0, keepdim=True)
    y = (y ** 2).mean(dim=0)
    y = torch.sqrt(y + 1e-8)
    y = y.mean(dim=(1, 2, 3), keepdim=True)          # [M,1,1,1]
    y = y.repeat(g, 1, s[2], s[3])                   # tf.tile(y,[G,1,W,C]) -> [N,1,W,C]
    return torch.cat([x, y], dim=1)


def soft_rectifier(x, alpha):
    """models.py:205-206 / 219-220: tanh(x) + x/alpha (divide_no_nan; alpha != 0 here)."""
    return torch.tanh(x) + x / alpha


# --------------------------------------------------------------------------------------
# model graphs
# --------------------------------------------------------------------------------------
def _p(W, name, kind):
    return W["%s/%s" % (name, kind)]


def _conv_sr(W, name, x, alpha, **kw):
    return soft_rectifier(conv2d(x, _p(W, name, "kernel"), _p(W, name, "bias"), **kw), alpha)


def _d_base_body(W, names, x):
    """Stage-0 discriminator after its from-audio conv: nine 'same' convs 3x3/4x4/3x3,
    MinibatchStdDev, Flatten, Dense(1) (models.py:317-340).  Returns the pre-activation
    logit; the final StaticOptTanh belongs to the model being evaluated."""
    d0 = names["d"][0]
    for cname in d0["convs"]:
        x = _conv_sr(W, cname, x, 20000.0)
    x = minibatch_stddev(x)
    x = x.reshape(x.shape[0], -1)
    return dense(x, _p(W, d0["dense"], "kernel"), _p(W, d0["dense"], "bias"))


def _d_block(W, names, s, x):
    """New convolutions of add_discriminator_block for stage s after its from-audio conv
    (models.py:259-281): 3x3,3x3,2x2 | 3x3,3x3,2x2 | 3x3,3x3,2x2/s2 valid | 1x1->128."""
    blk = names["d"][s]
    _, soft = activation_alphas(s, x.shape[1] >> s)
    for i, cname in enumerate(blk["convs"]):
        if i == 8:
            x = _conv_sr(W, cname, x, soft, strides=(2, 2), padding="valid")
        else:
            x = _conv_sr(W, cname, x, soft)
    return _conv_sr(W, blk["to128"], x, soft)


def _from_audio(W, names, s, x):
    _, soft = activation_alphas(s, x.shape[1] >> s)
    return _conv_sr(W, names["d"][s]["from_audio"], x, soft)


def discriminator(W, names, stage, fade_in, x, alpha=None):
    """define_discriminator()[stage][1 if fade_in else 0] applied to x (models.py:309-354,
    229-305).  `alpha` is the WeightedSum value of the fade-in twin.  The base resolution
    (define_discriminator's `input_shape`) is read off x: its height is the base height
    doubled `stage` times."""
    base_h = x.shape[1] >> stage
    if stage == 0:
        h = _from_audio(W, names, 0, x)
    else:
        h = _d_block(W, names, stage, _from_audio(W, names, stage, x))
        if fade_in:
            old = _from_audio(W, names, stage - 1, avgpool2(x))       # models.py:292-295
            h = weighted_sum(old, h, alpha)                            # models.py:297
        for s in range(stage - 1, 0, -1):                              # old_model.layers[3:]
            h = _d_block(W, names, s, h)
    logit = _d_base_body(W, names, h)
    sot_alpha, _ = activation_alphas(stage, base_h)
    return soft_rectifier(logit, sot_alpha)                            # StaticOptTanh


def _g_tail(W, tail, x):
    for dname in tail:
        x = dense(x, _p(W, dname, "kernel"), _p(W, dname, "bias"))
    return x


def _g_block_end(W, names, stage, z):
    """Output of the last Dense(2) of stage `stage` (the tensor old_model.layers[-2].output
    that add_generator_block taps, models.py:360), before any LayerNormalization."""
    g = names["g"][stage]
    if stage == 0:
        f = conv2d(z, _p(W, g["featured"], "kernel"), _p(W, g["featured"], "bias"),
                   strides=(1, 2), padding="valid")                   # models.py:416
        acc = None
        for br in g["branches"]:                                       # models.py:418-444
            h = conv2d_transpose(f, _p(W, br[0], "kernel"), _p(W, br[0], "bias"), (2, 1))
            h = dense(h, _p(W, br[1], "kernel"), _p(W, br[1], "bias"))
            h = conv2d_transpose(h, _p(W, br[2], "kernel"), _p(W, br[2], "bias"), (2, 1))
            h = dense(h, _p(W, br[3], "kernel"), _p(W, br[3], "bias"))
            h = conv2d_transpose(h, _p(W, br[4], "kernel"), _p(W, br[4], "bias"), (1, 15))
            h = dense(h, _p(W, br[5], "kernel"), _p(W, br[5], "bias"))
            acc = h if acc is None else acc + h
        return _g_tail(W, g["tail"], acc), None
    prev, _ = _g_block_end(W, names, stage - 1, z)
    up = upsample2(prev)                                               # models.py:362
    f = conv2d(up, _p(W, g["featured"], "kernel"), _p(W, g["featured"], "bias"),
               strides=(2, 2), padding="valid")                        # models.py:363
    acc = None
    for br in g["branches"]:                                           # models.py:365-393
        h = conv2d_transpose(f, _p(W, br[0], "kernel"), _p(W, br[0], "bias"), (2, 1))
        h = dense(h, _p(W, br[1], "kernel"), _p(W, br[1], "bias"))
        h = conv2d_transpose(h, _p(W, br[2], "kernel"), _p(W, br[2], "bias"), (1, 2))
        h = dense(h, _p(W, br[3], "kernel"), _p(W, br[3], "bias"))
        acc = h if acc is None else acc + h
    return _g_tail(W, g["tail"], acc), up


def generator(W, names, stage, fade_in, z, alpha=None):
    """define_generator()[stage][1 if fade_in else 0] applied to z (models.py:412-463,
    358-408)."""
    g = names["g"][stage]
    end, up = _g_block_end(W, names, stage, z)
    if stage > 0 and fade_in:
        merged = weighted_sum(up, end, alpha)                          # models.py:404
        return layer_norm(merged, _p(W, g["ln_fade"], "gamma"), _p(W, g["ln_fade"], "beta"))
    return layer_norm(end, _p(W, g["ln"], "gamma"), _p(W, g["ln"], "beta"))


def trainable_names(names, stage, fade_in, which: str) -> List[str]:
    """Names ('layer/kind') of model.trainable_variables of the stage/twin, as a set-like
    list (order is irrelevant to Adam).  Fade-in twins include the old from-audio conv
    (discriminator) / the fade-in LayerNormalization (generator) and, because
    backend.variable is trainable by default (models.py:27), the WeightedSum alpha."""
    out: List[str] = []

    def kb(n):
        out.extend([n + "/kernel", n + "/bias"])

    if which == "d":
        d = names["d"]
        kb(d[stage]["from_audio"])
        for s in range(stage, 0, -1):
            for c in d[s]["convs"]:
                kb(c)
            kb(d[s]["to128"])
        for c in d[0]["convs"]:
            kb(c)
        kb(d[0]["dense"])
        if stage > 0 and fade_in:
            kb(d[stage - 1]["from_audio"])
            out.append(d[stage]["ws"] + "/alpha")
    else:
        g = names["g"]
        for s in range(0, stage + 1):
            kb(g[s]["featured"])
            for br in g[s]["branches"]:
                for n in br:
                    kb(n)
            for n in g[s]["tail"]:
                kb(n)
        ln = g[stage]["ln_fade"] if (stage > 0 and fade_in) else g[stage]["ln"]
        out.extend([ln + "/gamma", ln + "/beta"])
        if stage > 0 and fade_in:
            out.append(g[stage]["ws"] + "/alpha")
    return out


# --------------------------------------------------------------------------------------
# losses (models.py:468-484)
# --------------------------------------------------------------------------------------
def _sign(lamb, delta):
    # tf.math.divide_no_nan(lamb, delta): 0 when delta == 0.  Its analytic derivative is
    # zero wherever it is defined, so it is treated as a constant of the step.
    if float(delta.detach()) == 0.0:
        return torch.zeros_like(lamb)
    return (lamb / delta).detach()


def discriminator_loss(fake_logits, real_logits):
    ci = real_logits.mean()
    cu = fake_logits.mean()
    lamb = cu - ci
    delta = lamb.abs()
    sgn = _sign(lamb, delta)
    sign = sgn + 0.0001
    sign_2 = (sgn + 0.0000999) * -1.0
    return sign * ci + sign_2 * cu


def generator_loss(fake_logits, real_logits):
    ci = real_logits.mean()
    cu = fake_logits.mean()
    lamb = cu - ci
    delta = lamb.abs()
    sign = _sign(lamb, delta) + 0.0001
    return sign * ci + delta


# --------------------------------------------------------------------------------------
# Adam (tf.keras OptimizerV2 Adam, non-amsgrad) with the hyper-parameters of models.py:502
# --------------------------------------------------------------------------------------
def adam_update(p, g, v, t, lr=ADAM["lr"], beta_2=ADAM["beta_2"], eps=ADAM["epsilon"]):
    """One Adam step with beta_1 = 0 (so m == g).  t is the 1-based iteration.  Returns
    (new_p, new_v)."""
    lr_t = lr * math.sqrt(1.0 - beta_2 ** t)          # / (1 - beta_1**t) == 1
    v = beta_2 * v + (1.0 - beta_2) * g * g
    return p - lr_t * g / (torch.sqrt(v) + eps), v


# --------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------
def weight_shapes(n_blocks: int, base_shape: Tuple[int, ...] = BASE_SHAPE) -> Dict[str, Tuple[int, ...]]:
    """Shape of every variable of define_discriminator(n, input_shape=base_shape) +
    define_generator(n).  Only Dense(1) of the discriminator depends on the base shape
    (Flatten of (H+1, W, 32), models.py:338-340); the generator's resolution is fixed by its
    architecture (models.py:415-451)."""
    nm = layer_names(n_blocks)
    sh: Dict[str, Tuple[int, ...]] = {}

    def conv(name, kh, kw, ci, co):
        sh[name + "/kernel"] = (kh, kw, ci, co); sh[name + "/bias"] = (co,)

    def convt(name, kh, kw, co, ci):
        sh[name + "/kernel"] = (kh, kw, co, ci); sh[name + "/bias"] = (co,)

    def dn(name, i, o):
        sh[name + "/kernel"] = (i, o); sh[name + "/bias"] = (o,)

    d0 = nm["d"][0]
    conv(d0["from_audio"], 1, 1, 2, 128)
    ks0 = [3, 4, 3] * 3
    for i, c in enumerate(d0["convs"]):
        conv(c, ks0[i], ks0[i], 128 if i == 0 else 32, 32)
    dn(d0["dense"], (base_shape[0] + 1) * base_shape[1] * 32, 1)
    for s in range(1, n_blocks):
        b = nm["d"][s]
        conv(b["from_audio"], 1, 1, 2, 128)
        ks = [3, 3, 2] * 3
        for i, c in enumerate(b["convs"]):
            conv(c, ks[i], ks[i], 128 if i == 0 else 32, 32)
        conv(b["to128"], 1, 1, 32, 128)
        sh[b["ws"] + "/alpha"] = ()
    g0 = nm["g"][0]
    conv(g0["featured"], 1, 2, 2, 128)
    for br in g0["branches"]:
        convt(br[0], 2, 1, 4, 128); dn(br[1], 4, 8)
        convt(br[2], 2, 1, 16, 8); dn(br[3], 16, 32)
        convt(br[4], 1, 15, 64, 32); dn(br[5], 64, 128)
    for n_, (i, o) in zip(g0["tail"], [(128, 256), (256, 64), (64, 8), (8, 2)]):
        dn(n_, i, o)
    sh[g0["ln"] + "/gamma"] = (4, 750, 2); sh[g0["ln"] + "/beta"] = (4, 750, 2)
    for s in range(1, n_blocks):
        gb = nm["g"][s]
        h, w = stage_hw(s)
        conv(gb["featured"], 2, 2, 2, 128)
        for br in gb["branches"]:
            convt(br[0], 2, 1, 16, 128); dn(br[1], 16, 32)
            convt(br[2], 1, 2, 64, 32); dn(br[3], 64, 128)
        for n_, (i, o) in zip(gb["tail"], [(128, 256), (256, 64), (64, 8), (8, 2)]):
            dn(n_, i, o)
        for ln in (gb["ln"], gb["ln_fade"]):
            sh[ln + "/gamma"] = (h, w, 2); sh[ln + "/beta"] = (h, w, 2)
        sh[gb["ws"] + "/alpha"] = ()
    return sh


def synthetic_weights(n_blocks: int, seed: int = 100, perturb: bool = True,
                      base_shape: Tuple[int, ...] = BASE_SHAPE) -> Dict[str, np.ndarray]:
    """Deterministic float32 weights for parity runs.  TF's initialisers cannot be
    reproduced outside TF, so scales only approximate them: kernels ~ N(0, 2/fan_in)
    (he_normal-like, models.py:256 etc.) or U(+-sqrt(6/(fi+fo))) for Dense (glorot).
    With perturb=True biases, gamma and beta get small non-trivial values so that every
    parameter influences the result (Keras would start them at 0 / 1 / 0)."""
    out: Dict[str, np.ndarray] = {}
    for i, (name, shape) in enumerate(sorted(weight_shapes(n_blocks, base_shape).items())):
        rng = np.random.default_rng(seed * 100003 + i)
        kind = name.rsplit("/", 1)[1]
        if kind == "kernel":
            if len(shape) == 4:
                if "transpose" in name:
                    fan_in = shape[0] * shape[1] * shape[2]   # Keras: fan_in from dim -2 (= Cout here)
                else:
                    fan_in = shape[0] * shape[1] * shape[2]
                w = rng.standard_normal(shape) * math.sqrt(2.0 / fan_in)
            else:
                lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                w = rng.uniform(-lim, lim, shape)
                if shape[1] == 1:
                    w = w * 0.25      # keep the logit in the unsaturated range of tanh
        elif kind == "bias":
            w = rng.uniform(-0.1, 0.1, shape) if perturb else np.zeros(shape)
        elif kind == "gamma":
            w = 1.0 + (rng.uniform(-0.2, 0.2, shape) if perturb else 0.0) * np.ones(shape)
        elif kind == "beta":
            w = rng.uniform(-0.2, 0.2, shape) if perturb else np.zeros(shape)
        elif kind == "alpha":
            w = np.zeros(shape)
        else:
            raise KeyError(name)
        out[name] = np.asarray(w, dtype=np.float32)
    return out


# --------------------------------------------------------------------------------------
# the training step (models.py:72-158)
# --------------------------------------------------------------------------------------
class StepResult:
    """Everything the parity tests compare."""

    def __init__(self):
        self.metrics: Dict[str, float] = {}
        self.d_loss = None
        self.g_loss = None
        self.d_grads: Dict[str, np.ndarray] = {}
        self.g_grads: Dict[str, np.ndarray] = {}
        self.fake_d_step: Optional[np.ndarray] = None   # generated audio of the D step
        self.fake_logits_d_step = None
        self.real_logits_d_step = None
        self.weights: Dict[str, np.ndarray] = {}        # all weights after the step
        self.v_d: Dict[str, np.ndarray] = {}
        self.v_g: Dict[str, np.ndarray] = {}


def train_step(weights: Dict[str, np.ndarray], n_blocks: int, stage: int, fade_in: bool,
               real: np.ndarray, z4: List[np.ndarray], alpha: Optional[float] = None,
               v_d: Optional[Dict[str, np.ndarray]] = None,
               v_g: Optional[Dict[str, np.ndarray]] = None,
               t: int = 1, trainable_alpha: bool = True, dtype=torch.float64,
               keep_grads: bool = True) -> StepResult:
    """One WGAN.train_step (models.py:72-158) with injected weights, the four latent draws
    z4 (in the order D step, metric 0, G step, metric 1) and the fade-in alpha of this
    step (models.py:74-82; alpha = actual_step / (train_steps - 1) under eager semantics).

    v_d / v_g are the Adam second-moment states of the two optimisers of this WGAN (None =
    fresh, as compiled by define_composite, models.py:501-521) and t the 1-based Adam
    iteration.  trainable_alpha=True reproduces the reference, where the WeightedSum alpha
    is a trainable variable of both twins and receives an Adam update inside the step.
    """
    names = layer_names(n_blocks)
    W = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
    fade = bool(fade_in and stage > 0)
    if fade:
        assert alpha is not None
        W[names["d"][stage]["ws"] + "/alpha"] = torch.tensor(float(alpha), dtype=dtype)
        W[names["g"][stage]["ws"] + "/alpha"] = torch.tensor(float(alpha), dtype=dtype)
    real_t = torch.tensor(real, dtype=dtype)
    z = [torch.tensor(zz, dtype=dtype) for zz in z4]
    res = StepResult()

    def d_alpha():
        return W[names["d"][stage]["ws"] + "/alpha"] if fade else None

    def g_alpha():
        return W[names["g"][stage]["ws"] + "/alpha"] if fade else None

    def G(zz):
        return generator(W, names, stage, fade, zz, g_alpha())

    def D(x):
        return discriminator(W, names, stage, fade, x, d_alpha())

    def apply(which, grads, vstate):
        new_v = {}
        for name, g in grads.items():
            v = torch.zeros_like(W[name]) if (vstate is None or name not in vstate) \
                else torch.tensor(vstate[name], dtype=dtype)
            p, v = adam_update(W[name].detach(), g, v, t)
            W[name] = p
            new_v[name] = v.numpy().copy()
        return new_v

    # ---- discriminator step (models.py:104-122; d_steps == 1, models.py:499,514) ----
    d_names = trainable_names(names, stage, fade, "d")
    if not trainable_alpha:
        d_names = [n for n in d_names if not n.endswith("/alpha")]
    for n in d_names:
        W[n] = W[n].detach().requires_grad_(True)
    fake = G(z[0])
    fake_logits = D(fake)
    real_logits = D(real_t)
    d_loss = discriminator_loss(fake_logits, real_logits)
    grads = torch.autograd.grad(d_loss, [W[n] for n in d_names])
    res.d_loss = float(d_loss.detach())
    res.fake_d_step = fake.detach().numpy().copy()
    res.fake_logits_d_step = fake_logits.detach().numpy().copy()
    res.real_logits_d_step = real_logits.detach().numpy().copy()
    gd = {n: g.detach() for n, g in zip(d_names, grads)}
    if keep_grads:
        res.d_grads = {n: g.numpy().copy() for n, g in gd.items()}
    for n in d_names:
        W[n] = W[n].detach()
    res.v_d = apply("d", gd, v_d)

    # ---- metric block 0 (models.py:124-130) ----
    with torch.no_grad():
        fl = D(G(z[1])); rl = D(real_t)
        ci0, cu0 = rl.mean(), fl.mean()
        res.metrics.update(delta_0=float((cu0 - ci0).abs()), cu_0=float(cu0), ci_0=float(ci0))

    # ---- generator step (models.py:133-148) ----
    g_names = trainable_names(names, stage, fade, "g")
    if not trainable_alpha:
        g_names = [n for n in g_names if not n.endswith("/alpha")]
    for n in g_names:
        W[n] = W[n].detach().requires_grad_(True)
    gen_logits = D(G(z[2]))
    with torch.no_grad():
        real_logits = D(real_t)
    g_loss = generator_loss(gen_logits, real_logits)
    grads = torch.autograd.grad(g_loss, [W[n] for n in g_names])
    res.g_loss = float(g_loss.detach())
    gg = {n: g.detach() for n, g in zip(g_names, grads)}
    if keep_grads:
        res.g_grads = {n: g.numpy().copy() for n, g in gg.items()}
    for n in g_names:
        W[n] = W[n].detach()
    res.v_g = apply("g", gg, v_g)

    # ---- metric block 1 (models.py:150-156) ----
    with torch.no_grad():
        fl = D(G(z[3])); rl = D(real_t)
        ci1, cu1 = rl.mean(), fl.mean()
        res.metrics.update(delta_1=float((cu1 - ci1).abs()), cu_1=float(cu1), ci_1=float(ci1))

    res.weights = {k: v.detach().numpy().copy() for k, v in W.items()}
    return res


def generator_step_grads(weights: Dict[str, np.ndarray], n_blocks: int, stage: int, fade_in: bool,
                         real: np.ndarray, z: np.ndarray, trainable_alpha: bool = True, dtype=torch.float64):
    """Gradients of generator_loss w.r.t. the generator's trainable variables for the given
    weights (models.py:133-144), e.g. with the discriminator weights a device run produced after
    its D update: isolates the generator-step arithmetic from the sign-like first Adam step."""
    names = layer_names(n_blocks)
    W = {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in weights.items()}
    fade = bool(fade_in and stage > 0)
    g_names = trainable_names(names, stage, fade, "g")
    if not trainable_alpha:
        g_names = [n for n in g_names if not n.endswith("/alpha")]
    for n in g_names:
        W[n] = W[n].detach().requires_grad_(True)
    ga = W[names["g"][stage]["ws"] + "/alpha"] if fade else None
    da = W[names["d"][stage]["ws"] + "/alpha"] if fade else None
    gen = generator(W, names, stage, fade, torch.tensor(z, dtype=dtype), ga)
    gen_logits = discriminator(W, names, stage, fade, gen, da)
    with torch.no_grad():
        real_logits = discriminator(W, names, stage, fade, torch.tensor(real, dtype=dtype), da)
    loss = generator_loss(gen_logits, real_logits)
    grads = torch.autograd.grad(loss, [W[n] for n in g_names])
    return {n: g.detach().numpy().copy() for n, g in zip(g_names, grads)}


def synthetic_inputs(stage: int, batch: int, seed: int = 0, base_shape: Tuple[int, ...] = BASE_SHAPE):
    """real ~ U(-1,1) (N,H,W,2) and four latent draws ~ N(0,1) (N,1,100,2), float32, from
    numpy default_rng(seed) / default_rng(seed+1..seed+4)."""
    h, w = stage_hw(stage, base_shape)
    real = np.random.default_rng(seed).uniform(-1.0, 1.0, (batch, h, w, 2)).astype(np.float32)
    z4 = [np.random.default_rng(seed + 1 + i).standard_normal((batch,) + LATENT_DIM).astype(np.float32)
          for i in range(4)]
    return real, z4
