"""GPU parity of whole models and of the full WGAN.train_step against the CPU oracle and the
committed golden vectors (tests/golden/step_*.npz, produced by tests/golden/make_golden.py).

Stated tolerances (float32 device arithmetic vs float64 oracle):
  metrics / losses      abs 2e-5 (values are O(1))
  logits                abs 2e-5
  generated audio       max abs 1e-4 (post-LayerNorm, O(1))
  D-step gradients      rel-L2 5e-5 per tensor on the stored checksum vector (measured 2e-6..3e-5;
                        the worst is the scalar Dense(1) bias, a cancelling sum of the two seeds);
                        5e-4 for the scalar fade-in alpha (see d_grad_tol)
  G-step gradients      vs the golden fixture rel-L2 2e-2: they are taken AFTER the discriminator's
                        Adam update, whose first step is sign-like (beta_1 = 0), so float32-level
                        differences of near-zero D gradients move single D weights by up to ~1e-4;
                        the saturated output tanh turns the resulting logit shift dx into a factor
                        exp(-2*dx) on the generator gradients (measured 2e-5..5e-3).  The arithmetic
                        of the generator step itself is pinned separately at rel-L2 1e-4 by
                        test_generator_step_with_device_discriminator, which hands the device's
                        own post-update discriminator weights to the oracle.
                        (abs 1e-9 for tensors whose gradient is numerically zero)
  updated weights       abs 2e-5 on elements whose gradient is not within 1e-3 of the tensor's
                        rms (Adam with beta_1 = 0 is sign-like on its first step, so a 1-ulp sign
                        flip of a ~zero gradient moves the weight by 2*lr = 1e-3)
"""
import os

import numpy as np
import pytest
import torch

import music_generator_amd as M
from oracle import ref_step as R

from helpers import assert_close, rel_l2

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {
    "s0_n16": (1, 0, False, 16, None, True),
    "s0_n4": (1, 0, False, 4, None, True),
    "s1_straight_n4": (2, 1, False, 4, None, True),
    "s1_fade_a0_n4": (2, 1, True, 4, (0, 101), True),
    "s1_fade_a037_n4": (2, 1, True, 4, (37, 101), True),
    "s1_fade_a1_n4": (2, 1, True, 4, (100, 101), True),
    "s1_fade_a037_fixed_alpha_n4": (2, 1, True, 4, (37, 101), False),
    "s2_fade_a037_n4": (3, 2, True, 4, (37, 101), True),
    "s3_fade_a037_n4": (4, 3, True, 4, (37, 101), True),
}


def summarize(a, k=64):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, a.size - 1, num=min(k, a.size)).astype(np.int64)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[idx]])


def d_grad_tol(name):
    """rel-L2 bound of a D-step gradient tensor: 5e-5, and 5e-4 for the scalar fade-in alpha, whose
    gradient sum(dY * (new - old)) over the whole activation cancels to ~1e-3 of its terms."""
    return 5e-4 if name.endswith("/alpha") else 5e-5


def build(n_blocks, trainable_alpha=True, seed=100):
    M.clear_session()
    d = M.define_discriminator(n_blocks)
    g = M.define_generator(n_blocks)
    comp = M.define_composite(d, g, (1, 100, 2), trainable_alpha=trainable_alpha)
    W = R.synthetic_weights(n_blocks, seed=seed)
    seen = set()
    for lst in (d, g):
        for pair in lst:
            for model in pair:
                for layer in model.layers:
                    for v in layer.weights:
                        if id(v) not in seen:
                            seen.add(id(v))
                            v.assign(W[v.name])
    return d, g, comp, W


def device_step(name):
    n_blocks, stage, fade, batch, ks, tr_alpha = CASES[name]
    d, g, comp, W = build(n_blocks, tr_alpha)
    gan = comp[stage][1 if fade else 0]
    real, z4 = R.synthetic_inputs(stage, batch, seed=0)
    if ks:
        gan.set_train_steps(ks[1])
        gan.actual_step = ks[0]
        for layer in gan.generator.layers:
            if isinstance(layer, M.WeightedSum):
                M.backend.set_value(layer.alpha, ks[0] / float(ks[1] - 1))
    # the D step of the oracle records the generated audio: regenerate it here before the update
    fake = gan.generator(z4[0]).cpu().numpy()
    gan.inject_latents(z4)
    gan.keep_logits = True
    metrics = gan.train_step(real)
    torch.cuda.synchronize()
    return gan, metrics, fake


@pytest.mark.parametrize("name", list(CASES))
def test_train_step_matches_golden(name):
    gan, metrics, fake = device_step(name)
    with np.load(os.path.join(GOLDEN, "step_%s.npz" % name)) as f:
        gold = {k.replace("~", "/"): f[k] for k in f.files}
    got = np.array([float(metrics[k]) for k in ("delta_0", "cu_0", "ci_0", "delta_1", "cu_1", "ci_1")])
    assert np.abs(got - gold["metrics"]).max() < 2e-5, (got, gold["metrics"])
    losses = np.array([float(gan.last_losses["d_loss"]), float(gan.last_losses["g_loss"])])
    assert np.abs(losses - gold["losses"]).max() < 2e-5, (losses, gold["losses"])
    assert np.abs(summarize(fake, 256)[2:] - gold["fake_audio"][2:]).max() < 1e-4
    for key in ("fake", "real"):
        have = gan.last_logits[key].cpu().numpy()
        assert np.abs(have - gold[key + "_logits"]).max() < 2e-5, (key, have, gold[key + "_logits"])
    compare_step_with_fixture(gan, gold, 2e-2)


def compare_step_with_fixture(gan, gold, g_tol):
    for prefix, model in (("dgrad|", gan.discriminator), ("ggrad|", gan.generator)):
        for v in model.trainable_variables:
            want = gold[prefix + v.name]
            have = summarize(v.grad.cpu().numpy())
            if np.abs(want).max() < 1e-12:
                assert np.abs(have).max() < 1e-9, v.name
            else:
                tol = d_grad_tol(v.name) if prefix == "dgrad|" else g_tol
                err = rel_l2(have[2:], want[2:])
                assert err < tol, "%s%s: rel-L2 %.3e" % (prefix, v.name, err)
                assert abs(have[1] - want[1]) <= tol * want[1] + 1e-12, v.name
    for model in (gan.discriminator, gan.generator):
        for v in model.trainable_variables:
            want = gold["w|" + v.name]
            have = summarize(v.numpy())
            gsum = gold[("dgrad|" if model is gan.discriminator else "ggrad|") + v.name]
            rms = gsum[1] / np.sqrt(max(1, v.size))
            keep = np.abs(gsum[2:]) > 1e-3 * rms
            if keep.any():
                assert np.abs(have[2:] - want[2:])[keep].max() < 2e-5, v.name


def test_third_step_with_warmed_adam_matches_golden():
    """Adam at t = 3 with second moments warmed by two steps: the oracle advances the weights and both
    optimisers' states two steps on the host, the device takes them over and runs the third step, which
    is compared with the fixture of the oracle's third step.  With a warmed `v` the discriminator update is
    no longer sign-like, so the generator-step gradients, taken after it, are held to rel-L2 1e-4 (5e-4 for
    the cancelling bias / alpha / beta sums) instead of the 2e-2 of a first step."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_golden
    name, n_blocks, stage, batch, n_steps = make_golden.T3
    cur, vd, vg = make_golden.warmed_state(n_blocks, stage, batch, n_steps, 2)
    d, g, comp, _ = build(n_blocks)
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
    gan.keep_logits = True
    metrics = gan.train_step(real)
    with np.load(os.path.join(GOLDEN, "step_%s.npz" % name)) as f:
        gold = {k.replace("~", "/"): f[k] for k in f.files}
    got = np.array([float(metrics[k]) for k in ("delta_0", "cu_0", "ci_0", "delta_1", "cu_1", "ci_1")])
    assert np.abs(got - gold["metrics"]).max() < 2e-5, (got, gold["metrics"])
    for key in ("fake", "real"):
        assert np.abs(gan.last_logits[key].cpu().numpy() - gold[key + "_logits"]).max() < 2e-5, key
    assert gan.d_optimizer.iterations == 3 and gan.g_optimizer.iterations == 3
    for v in gan.generator.trainable_variables:
        want, have = gold["ggrad|" + v.name], summarize(v.grad.cpu().numpy())
        if np.abs(want).max() < 1e-12:
            continue
        tol = 5e-4 if v.name.endswith(("/bias", "/alpha", "/beta")) else 1e-4
        assert rel_l2(have[2:], want[2:]) < tol, "%s: %.3e" % (v.name, rel_l2(have[2:], want[2:]))
    for v in gan.discriminator.trainable_variables:
        want, have = gold["dgrad|" + v.name], summarize(v.grad.cpu().numpy())
        if np.abs(want).max() >= 1e-12:
            assert rel_l2(have[2:], want[2:]) < d_grad_tol(v.name), v.name
    # updated weights: no sign-like first step here, every element is comparable
    for model in (gan.discriminator, gan.generator):
        for v in model.trainable_variables:
            assert np.abs(summarize(v.numpy())[2:] - gold["w|" + v.name][2:]).max() < 2e-5, v.name


def test_train_step_matches_oracle_directly():
    """Same comparison without the fixture: the oracle runs on the host in this process (stage 1
    fade-in, batch 4, alpha = 37/100), every gradient tensor is compared in full."""
    name = "s1_fade_a037_n4"
    n_blocks, stage, fade, batch, ks, tr_alpha = CASES[name]
    gan, metrics, fake = device_step(name)
    W = R.synthetic_weights(n_blocks, seed=100)
    real, z4 = R.synthetic_inputs(stage, batch, seed=0)
    res = R.train_step(W, n_blocks, stage, fade, real, z4, alpha=ks[0] / float(ks[1] - 1))
    for k, v in res.metrics.items():
        assert abs(float(metrics[k]) - v) < 2e-5, k
    assert np.abs(fake - res.fake_d_step).max() < 1e-4
    for grads, model in ((res.d_grads, gan.discriminator), (res.g_grads, gan.generator)):
        for v in model.trainable_variables:
            want = grads[v.name]
            have = v.grad.cpu().numpy().reshape(v.shape)
            if np.abs(want).max() < 1e-12:
                assert np.abs(have).max() < 1e-9, v.name
            else:
                tol = d_grad_tol(v.name) if model is gan.discriminator else 2e-2
                assert rel_l2(have, want) < tol, "%s: %.3e" % (v.name, rel_l2(have, want))


@pytest.mark.parametrize("name", ["s0_n4", "s1_straight_n4", "s1_fade_a037_n4", "s2_fade_a037_n4"])
def test_generator_step_with_device_discriminator(name):
    """Generator-step gradients against the oracle evaluated with the discriminator weights the
    device produced in its own D update (the generator update does not touch them): rel-L2 1e-4,
    and 5e-4 for biases and the scalar alpha (per-channel / global sums over the whole activation
    in front of a LayerNormalization, which removes most of a constant shift: the sums cancel to a
    small fraction of their terms, measured up to 1.6e-4)."""
    n_blocks, stage, fade, batch, ks, tr_alpha = CASES[name]
    gan, metrics, fake = device_step(name)
    W = R.synthetic_weights(n_blocks, seed=100)
    for v in gan.discriminator.trainable_variables:
        W[v.name] = v.numpy()
    if ks:
        names = R.layer_names(n_blocks)
        W[names["g"][stage]["ws"] + "/alpha"] = np.asarray(ks[0] / float(ks[1] - 1), np.float32)
    real, z4 = R.synthetic_inputs(stage, batch, seed=0)
    want = R.generator_step_grads(W, n_blocks, stage, fade, real, z4[2], trainable_alpha=tr_alpha)
    for v in gan.generator.trainable_variables:
        have = v.grad.cpu().numpy().reshape(v.shape)
        if np.abs(want[v.name]).max() < 1e-12:
            assert np.abs(have).max() < 1e-9, v.name
        else:
            err = rel_l2(have, want[v.name])
            tol = 5e-4 if v.name.endswith(("/bias", "/alpha", "/beta")) else 1e-4
            assert err < tol, "%s: %.3e" % (v.name, err)


@pytest.mark.parametrize("stage,fade", [(0, False), (2, False), (2, True)])
def test_model_forward_parity(stage, fade):
    """generator(z) and discriminator(x) of a stage against the oracle graphs."""
    n_blocks = stage + 1
    d, g, comp, W = build(n_blocks)
    names = R.layer_names(n_blocks)
    Wt = {k: torch.tensor(v, dtype=torch.float64) for k, v in W.items()}
    alpha = 0.25
    tw = 1 if fade else 0
    if fade:
        for model in (d[stage][1], g[stage][1]):
            for layer in model.layers:
                if isinstance(layer, M.WeightedSum):
                    M.backend.set_value(layer.alpha, alpha)
        Wt[names["d"][stage]["ws"] + "/alpha"] = torch.tensor(alpha, dtype=torch.float64)
        Wt[names["g"][stage]["ws"] + "/alpha"] = torch.tensor(alpha, dtype=torch.float64)
    real, z4 = R.synthetic_inputs(stage, 4, seed=3)
    audio = g[stage][tw](z4[0], training=False).cpu().numpy()
    want = R.generator(Wt, names, stage, fade, torch.tensor(z4[0], dtype=torch.float64), alpha).numpy()
    assert np.abs(audio - want).max() < 1e-4
    logits = d[stage][tw](real).cpu().numpy()
    want = R.discriminator(Wt, names, stage, fade, torch.tensor(real, dtype=torch.float64), alpha).numpy()
    assert np.abs(logits - want).max() < 2e-5


def test_two_consecutive_steps_fade_ramp():
    """Two steps of a fade-in WGAN: alpha follows k/(n-1), Adam state carries over (t = 1, 2)."""
    n_blocks, stage, batch, n_steps = 2, 1, 4, 5
    d, g, comp, W = build(n_blocks)
    gan = comp[stage][1]
    gan.set_train_steps(n_steps)
    cur = dict(W)
    vd = vg = None
    for k in range(2):
        real, z4 = R.synthetic_inputs(stage, batch, seed=10 + k)
        gan.inject_latents(z4)
        metrics = gan.train_step(real)
        res = R.train_step(cur, n_blocks, stage, True, real, z4, alpha=k / float(n_steps - 1), v_d=vd, v_g=vg,
                           t=k + 1, keep_grads=False)
        cur, vd, vg = res.weights, res.v_d, res.v_g
        for key, val in res.metrics.items():
            assert abs(float(metrics[key]) - val) < (5e-5 if k == 0 else 5e-4), (k, key, float(metrics[key]), val)
    assert gan.actual_step == 2


def test_micro_batching_reproduces_whole_batch_step():
    """WGAN.micro_batch splits the batch by whole MinibatchStdDev groups; unit-seed gradients are
    summed per micro-batch and scaled once the batch-wide sign is known, so the step must equal the
    whole-batch step up to float32 summation order (D gradients rel-L2 5e-5, metrics abs 1e-5)."""
    n_blocks, stage, batch = 2, 1, 16
    real, z4 = R.synthetic_inputs(stage, batch, seed=5)
    results = []
    for mb in (None, 4, 8, (8, 2), (8, "reuse")):
        d, g, comp, W = build(n_blocks)
        gan = comp[stage][1]
        if isinstance(mb, tuple) and mb[1] == "reuse":      # one evaluation of D(real) after the D update
            mb, gan.reuse_real_logits = mb[0], True
        elif isinstance(mb, tuple):                         # (micro-batch, generator backward chunk)
            mb, gan.generator_chunk = mb
        gan.micro_batch = mb
        gan.set_train_steps(11)
        gan.actual_step = 3
        gan.inject_latents(z4)
        metrics = {k: float(v) for k, v in gan.train_step(real).items()}
        grads = {}
        for model in (gan.discriminator, gan.generator):
            for v in model.trainable_variables:
                grads[v.name] = v.grad.cpu().numpy().copy()
        results.append((metrics, grads))
    base_m, base_g = results[0]
    for metrics, grads in results[1:]:
        for k in base_m:
            assert abs(metrics[k] - base_m[k]) < 1e-5, k
        for k, v in base_g.items():
            if np.abs(v).max() < 1e-12:
                continue
            tol = 5e-5 if not k.startswith(("dense_", "conv2d_transpose", "layer_norm", "weighted_sum_1")) else 2e-3
            assert rel_l2(grads[k], v) < tol, "%s: %.3e" % (k, rel_l2(grads[k], v))


def test_bench_workload_micro_batched_equals_whole_batch():
    """The bench's own workload under pytest: stage 3 (32 x 6000) fade-in step at batch 64, whole batch
    against micro-batches of 16 (4 MinibatchStdDev groups each).  Metrics abs 1e-5, discriminator gradients
    rel-L2 5e-5; generator gradients (taken after the sign-like first Adam update of D, see the module
    docstring) 2e-2; all finite."""
    n_blocks, stage, batch = 4, 3, 64
    real, z4 = R.synthetic_inputs(stage, batch, seed=21)
    results = []
    for mb in (None, 16):
        d, g, comp, W = build(n_blocks)
        gan = comp[stage][1]
        gan.micro_batch = mb
        gan.set_train_steps(11)
        gan.actual_step = 4
        gan.inject_latents(z4)
        metrics = {k: float(v) for k, v in gan.train_step(real).items()}
        grads = {v.name: v.grad.cpu().numpy().copy() for model in (gan.discriminator, gan.generator)
                 for v in model.trainable_variables}
        d_names = {v.name for v in gan.discriminator.trainable_variables}
        results.append((metrics, grads))
        del gan, d, g, comp
        torch.cuda.empty_cache()
    (m0, g0), (m1, g1) = results
    for k in m0:
        assert np.isfinite(m0[k]) and abs(m0[k] - m1[k]) < 1e-5, (k, m0[k], m1[k])
    for k, v in g0.items():
        assert np.isfinite(g1[k]).all(), k
        if np.abs(v).max() < 1e-12:
            continue
        tol = (5e-4 if k.endswith("/alpha") else 5e-5) if k in d_names else 2e-2
        assert rel_l2(g1[k], v) < tol, "%s: %.3e" % (k, rel_l2(g1[k], v))


@pytest.mark.parametrize("stage,fade,batch,budget", [(0, False, 1, None), (0, False, 5, None), (0, False, 37, None),
                                                     (1, False, 1, None), (1, True, 4, None), (2, True, 3, None),
                                                     (2, True, 3, 48 << 20), (1, False, 1, 1 << 20),
                                                     (2, True, 3, "auto")])
def test_fused_generator_backward_matches_layerwise(stage, fade, batch, budget):
    """The fused grow-block kernels (saving forward, register-chained data gradient, tile-major
    weight gradients) against the layer-by-layer taped path on the same device: output abs 2e-5 x max|y|,
    every gradient tensor rel-L2 2e-5 (float32 summation order is the only difference).  Batch
    1 and 3 leave the last 32-position tile partly empty; with a side-buffer budget the backward
    re-computes the block's tensors over several position ranges.  Stage 0 isolates the fused head of the base
    generator (latent -> Dense(32) of the four branches in one launch, its 34 gradient tensors and d/dz in two);
    batch 37 gives its backward waves several columns each and a ragged last pass."""
    from music_generator_amd.graph import FusedGroup, Model
    from music_generator_amd import backend as K
    n_blocks = stage + 1
    d, g, comp, W = build(n_blocks)
    model = g[stage][1 if fade else 0]
    if fade:
        for layer in model.layers:
            if isinstance(layer, M.WeightedSum):
                M.backend.set_value(layer.alpha, 0.37)
    real, z4 = R.synthetic_inputs(stage, batch, seed=7)
    rng = np.random.default_rng(3)
    dy = torch.tensor(rng.standard_normal((batch,) + tuple(model.output.shape[1:])).astype(np.float32), device="cuda")
    results = []
    auto = budget == "auto"        # no bound given and (pretended) too little free memory: the bound is chosen
    avail_fn, auto_from = K.available_bytes, FusedGroup.auto_budget_from_bytes
    for fused in (False, True):
        Model.use_fused_backward = fused
        FusedGroup.side_budget_bytes = None if auto else budget
        if auto:
            FusedGroup.auto_budget_from_bytes = 0
            K.available_bytes = lambda: (1 << 20, 16 << 20)
        try:
            model.bind_grads().zero_()
            y, tape = model.forward(z4[0], save=True)
            dz = model.backward(tape, dy, need_input_grad=True)
            torch.cuda.synchronize()
        finally:
            Model.use_fused_backward = True
            FusedGroup.side_budget_bytes = None
            FusedGroup.auto_budget_from_bytes = auto_from
            K.available_bytes = avail_fn
        results.append((y.cpu().numpy(), dz.cpu().numpy(),
                        {v.name: v.grad.cpu().numpy().copy() for v in model.trainable_variables}))
    (y0, dz0, g0), (y1, dz1, g1) = results
    assert np.abs(y0 - y1).max() < 2e-5 * max(1.0, float(np.abs(y0).max()))
    assert rel_l2(dz1, dz0) < 2e-5
    for name, want in g0.items():
        if np.abs(want).max() < 1e-12:
            assert np.abs(g1[name]).max() < 1e-9, name
        else:
            assert rel_l2(g1[name], want) < 2e-5, "%s: %.3e" % (name, rel_l2(g1[name], want))


def test_two_rank_device_step_equals_single_process(tmp_path):
    """Two processes sharing the one GPU (gloo carries the collectives; RCCL needs one device per
    rank) run the device train step on their MinibatchStdDev-group shards of a batch of 16 with the
    overlapped gradient all-reduce; metrics (abs 1e-5) and the reduced D / G gradients (rel-L2 5e-5
    / 2e-3 as for micro-batching) must equal the single-process step on the whole batch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "dp_worker.py")
    single, multi = str(tmp_path / "single.npz"), str(tmp_path / "multi.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, str(script), root, single], env=env, capture_output=True, text=True,
                         timeout=100)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29631", str(script), root, multi]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=100)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    with np.load(single) as a, np.load(multi) as b:
        for k in a.files:
            if k.startswith("m|"):
                assert abs(float(a[k]) - float(b[k])) < 1e-5, k
            elif np.abs(a[k]).max() >= 1e-12:
                name = k[2:]
                tol = 5e-5 if not name.startswith(("dense_", "conv2d_transpose", "layer_norm", "weighted_sum_1")) else 2e-3
                assert rel_l2(b[k], a[k]) < tol, "%s: %.3e" % (k, rel_l2(b[k], a[k]))


def test_rccl_path_executes_in_a_group_of_one(tmp_path):
    """The `nccl` backend (= RCCL) needs one device per rank, so a one-GPU box cannot run two ranks; a process
    group of ONE rank with the collectives forced (DataParallel(force_collectives=True)) still sends every
    call of the data-parallel step through RCCL: the weight broadcast, the three asynchronous gradient
    all-reduces on RCCL's stream with their waits, the blocking 2-float logit-sum all-reduces.  Sums over one
    rank change nothing: the step (in micro-batches of 4) must equal the plain single-process step (metrics
    abs 1e-5; gradients as for micro-batching)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tests", "dp_worker.py")
    single, rccl = str(tmp_path / "single.npz"), str(tmp_path / "rccl.npz")
    env = dict(os.environ, OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(script), root, single], env=env, capture_output=True, text=True,
                         timeout=100)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    out = subprocess.run([sys.executable, str(script), root, rccl], env=dict(env, MGX_DP_RCCL_ONE="1", MGX_DP_TRACE="1"),
                         capture_output=True, text=True, timeout=150)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("all_reduce_async begin") == 3 and out.stdout.count("wait_for end") == 3, out.stdout[-3000:]
    with np.load(single) as a, np.load(rccl) as b:
        for k in a.files:
            if k.startswith("m|"):
                assert abs(float(a[k]) - float(b[k])) < 1e-5, k
            elif np.abs(a[k]).max() >= 1e-12:
                name = k[2:]
                tol = 5e-5 if not name.startswith(("dense_", "conv2d_transpose", "layer_norm", "weighted_sum_1")) else 2e-3
                assert rel_l2(b[k], a[k]) < tol, "%s: %.3e" % (k, rel_l2(b[k], a[k]))


def test_progressive_schedule_two_stages():
    """trainer/task.py:108-184 on the device with a tiny in-memory dataset: stage 0 fit, then the
    stage-1 fade-in twin (alpha ramps 0 -> 1 over its steps) and the straight-through twin, which
    share their layers: the fade-in phase must move the weights the straight phase then starts
    from, every metric stays finite, and each phase runs int(n/batch)*epochs steps."""
    from music_generator_amd import task
    M.clear_session()
    d, g, comp = task.build(2)
    for pair in comp:
        for gan in pair:
            gan.seed(7)
    w_before = g[1][0].trainable_variables[-1].numpy().copy()
    hist = task.train(g, d, comp, task.synthetic_provider(8, seed=1), batch_sizes=[4, 4], epochs_norm=[2, 1],
                      epochs_fade=[0, 3])
    assert [len(h["ci_1"]) for h in hist] == [2, 3, 1]
    for h in hist:
        for k in ("delta_0", "cu_0", "ci_0", "delta_1", "cu_1", "ci_1"):
            assert np.isfinite(h[k]).all(), k
    fade, straight = comp[1][1], comp[1][0]
    assert comp[0][0].actual_step == 2 * 2 and fade.actual_step == 2 * 3 and straight.actual_step == 2 * 1
    assert fade.train_steps == 6
    alphas = [float(l.alpha.numpy()) for l in fade.generator.layers if isinstance(l, M.WeightedSum)]
    # the last fade-in step ran with alpha = (n-1)/(n-1) = 1, then Adam moved it by at most lr
    assert all(abs(a - 1.0) <= 5e-4 + 1e-6 for a in alphas), alphas
    assert np.abs(g[1][0].trainable_variables[-1].numpy() - w_before).max() > 0
    assert g[1][0].layers[1] is g[1][1].layers[1]


def test_checkpoint_resume_continues_bit_for_bit(tmp_path):
    """A run stopped after epoch 2 of the stage-1 fade-in phase and resumed from its checkpoint in
    a freshly built family (weights, Adam second moments and iteration counts, step counters,
    latent generator state, schedule position) must end with exactly the weights and the metric
    history of the uninterrupted run: every kernel of the step reduces in a fixed order."""
    from music_generator_amd import checkpoint, task

    def family():
        d, g, comp = task.build(2)
        for pair in comp:
            for gan in pair:
                gan.seed(7)
        return d, g, comp

    kw = dict(batch_sizes=[4, 4], epochs_norm=[2, 2], epochs_fade=[0, 3])
    M.clear_session()
    d, g, comp = family()
    cbs = lambda stage, fade: [checkpoint.CheckpointCallback(str(tmp_path), d, g, comp, stage, fade, every=1)]
    hist = task.train(g, d, comp, task.synthetic_provider(8, seed=1), callbacks_for=cbs, **kw)
    final = [v.numpy() for v in checkpoint._unique_variables([d, g])]
    final_v = comp[1][0].g_optimizer._state["v"].cpu().numpy()

    # stage 1 is 8x1500; the fade-in and the straight phase both write epoch2: the later one wins,
    # so take the fade-in phase's file from a run that stops there
    M.clear_session()
    d1, g1, comp1 = family()
    stop = {}

    class StopAfter:
        model = None

        def on_epoch_end(self, epoch, logs=None):
            if stop.get("armed") and epoch + 1 == 2:
                stop["path"] = checkpoint.save_checkpoint(str(tmp_path / "stop"), d1, g1, comp1, 1, True, 2)
                raise KeyboardInterrupt

    def cbs1(stage, fade):
        stop["armed"] = stage == 1 and fade
        return [StopAfter()]

    with pytest.raises(KeyboardInterrupt):
        task.train(g1, d1, comp1, task.synthetic_provider(8, seed=1), callbacks_for=cbs1, **kw)

    M.clear_session()
    d2, g2, comp2 = family()
    pos = checkpoint.load_checkpoint(stop["path"], d2, g2, comp2)
    assert (pos["stage"], pos["fade_in"], pos["epochs_done"]) == (1, True, 2)
    hist2 = task.train(g2, d2, comp2, task.synthetic_provider(8, seed=1), resume=pos, **kw)
    assert [len(h["ci_1"]) for h in hist2] == [1, 2]
    assert hist2[0]["ci_1"] == hist[1]["ci_1"][2:] and hist2[1] == hist[2]
    for a, b in zip(final, checkpoint._unique_variables([d2, g2])):
        assert np.array_equal(a, b.numpy()), b.name
    assert np.array_equal(final_v, comp2[1][0].g_optimizer._state["v"].cpu().numpy())
    assert comp2[1][1].actual_step == comp[1][1].actual_step and comp2[1][0].d_optimizer.iterations == 4


def test_monitor_writes_audio_and_inception_score(tmp_path):
    """The epoch-end hook of trainer/models.py:528-551 on the device: generator inference in
    batches of 10, one peak-normalised WAV per batch on the saving epochs, evaluator predictions of
    every batch folded into one Inception score per epoch (1 <= IS <= 9 classes)."""
    from scipy.io import wavfile
    from music_generator_amd import audio_out, evaluacion, task
    M.clear_session()
    d, g, comp = task.build(1)
    comp[0][0].seed(3)
    evaluador = evaluacion.define_evaluator(1)[0]
    sink = audio_out.WavSink(out_dir=str(tmp_path), pick=None, seed=1)
    monitor = M.GANMonitor(evaluador=evaluador, num_examples=2, sink=sink, save_every=2)
    task.train(g, d, comp, task.synthetic_provider(4, seed=2), batch_sizes=[4], epochs_norm=[2], epochs_fade=[0],
               callbacks_for=lambda stage, fade: [monitor])
    assert sorted(monitor.inception_scores) == [1, 2]
    assert all(1.0 - 1e-6 <= v <= 9.0 + 1e-6 for v in monitor.inception_scores.values())
    assert [os.path.relpath(p, tmp_path) for p in sink.written] == ["4x750/epoch1_0.wav", "4x750/epoch1_1.wav"]
    rate, sig = wavfile.read(sink.written[0])
    assert rate == 750 and sig.shape == (3000, 2) and np.isfinite(sig).all()
    assert np.abs(sig).max(axis=0) == pytest.approx([1.0, 1.0])


@pytest.mark.parametrize("stage,fade", [(0, False), (1, True)])
def test_graph_replayed_step_equals_eager_step(stage, fade):
    """WGAN.use_graph: the step captured into one HIP graph (second call) and replayed (later calls)
    must walk exactly the trajectory of the eager step: same latent stream (the graph advances the
    seeded generator), fade-in alpha and Adam step sizes fed through device scalars.  Metrics and
    weights are compared bit for bit after 5 steps on changing batches."""
    runs = []
    for use_graph in (False, True):
        d, g, comp, _ = build(stage + 1)
        gan = comp[stage][1 if fade else 0]
        gan.seed(21)
        gan.use_graph = use_graph
        gan.set_train_steps(5)
        metrics = []
        for k in range(5):
            real, _ = R.synthetic_inputs(stage, 4, seed=30 + k)
            out = gan.train_step(real)
            metrics.append({key: float(v) for key, v in out.items()})
        weights = [v.numpy() for model in (gan.discriminator, gan.generator) for v in model.trainable_variables]
        runs.append((metrics, weights, gan.actual_step, gan.d_optimizer.iterations, gan.g_optimizer.iterations))
    (m_e, w_e, *c_e), (m_g, w_g, *c_g) = runs
    assert c_e == c_g == [5, 5, 5]
    assert m_e == m_g
    for a, b in zip(w_e, w_g):
        assert np.array_equal(a, b)


def test_graph_replays_without_host_reads_follow_the_eager_trajectory():
    """24 replays on device-resident batches with no host read in between: the host runs replays ahead of the
    GPU, so the per-step scalars (fade-in alpha, the two Adam step sizes, which change by a factor of 3 over the
    first ten updates) must reach each replay through a pinned slot of its own.  Weights after the run must
    equal the eager run's bit for bit; the ring of slots is walked three times."""
    stage, fade, steps = 1, True, 26
    batches = [torch.from_numpy(R.synthetic_inputs(stage, 4, seed=200 + k)[0]).cuda() for k in range(4)]
    runs = []
    for use_graph in (False, True):
        d, g, comp, _ = build(stage + 1)
        gan = comp[stage][1]
        gan.seed(9)
        gan.use_graph = use_graph
        gan.set_train_steps(steps)
        torch.cuda.synchronize()
        for k in range(steps):
            gan.train_step(batches[k % 4])
        torch.cuda.synchronize()
        runs.append([v.numpy() for model in (gan.discriminator, gan.generator) for v in model.trainable_variables])
        if use_graph:
            entry = next(e for e in gan._graphs.values() if e.get("graph") is not None)
            assert steps - 2 >= 3 * gan.SCALAR_SLOTS and all(ev is not None for ev in entry["host_events"])
    for a, b in zip(*runs):
        assert np.array_equal(a, b)


def test_graph_mode_leaves_optimizers_with_a_first_moment_alone():
    """The automatic graph mode steps aside for optimisers whose update cannot read its step size from device
    memory (beta_1 != 0): such a WGAN trains eagerly at any batch size, and forcing use_graph raises before
    anything is captured."""
    d, g, comp, _ = build(1)
    gan = M.WGAN(discriminator=d[0][0], generator=g[0][0], latent_dim=(1, 100, 2), discriminator_extra_steps=1)
    gan.compile(d_optimizer=M.Adam(lr=0.0005, beta_1=0.5, beta_2=0.99), g_optimizer=M.Adam(lr=0.0005, beta_1=0.5, beta_2=0.99))
    gan.seed(3)
    real, _ = R.synthetic_inputs(0, 4, seed=1)
    for _ in range(3):
        out = gan.train_step(real)
    assert all(np.isfinite(float(v)) for v in out.values())
    assert gan.d_optimizer.iterations == 3 and not any(e.get("graph") for e in gan._graphs.values())
    gan.use_graph = True
    with pytest.raises(NotImplementedError):
        gan.train_step(real)
    assert gan.d_optimizer.iterations == 3 and not gan._graphs


def test_no_lane_stays_forked_after_a_step():
    """Every stream a step forks (the real-batch lane, the generator-ahead lane) is joined again before the step
    returns, eagerly and inside a capture (a capture that ends with an unjoined fork fails)."""
    from music_generator_amd import backend as K
    d, g, comp, _ = build(2)
    gan = comp[1][1]
    gan.seed(1)
    gan.set_train_steps(4)
    real, _ = R.synthetic_inputs(1, 4, seed=2)
    for _ in range(3):                      # eager, capture, replay
        gan.train_step(real)
        assert K.open_lanes() == 0
    lane = K.fork(2)
    with lane:
        pass
    assert K.open_lanes() == 1
    K.join_all()
    assert K.open_lanes() == 0


@pytest.mark.parametrize("stage,fade,use_graph", [(0, False, False), (1, True, False), (1, True, True)])
def test_real_pass_on_second_stream_changes_nothing(stage, fade, use_graph):
    """WGAN.overlap_passes: every discriminator pass over the real batch on the second stream (own gradient
    buffer, own scratch lane) beside the pass over the generated batch.  Same kernels on the same data: metrics
    and weights after 4 steps must equal the one-stream step bit for bit, eagerly and as a replayed graph."""
    runs = []
    for overlap in (False, True):
        d, g, comp, _ = build(stage + 1)
        gan = comp[stage][1 if fade else 0]
        gan.seed(77)
        gan.use_graph = use_graph
        gan.overlap_passes = overlap
        gan.set_train_steps(4)
        metrics = []
        for k in range(4):
            real, _ = R.synthetic_inputs(stage, 8, seed=90 + k)
            out = gan.train_step(real)
            metrics.append({key: float(v) for key, v in out.items()})
        weights = [v.numpy() for model in (gan.discriminator, gan.generator) for v in model.trainable_variables]
        runs.append((metrics, weights))
    (m_a, w_a), (m_b, w_b) = runs
    assert m_a == m_b
    for a, b in zip(w_a, w_b):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("stage,fade,use_graph", [(0, False, False), (1, True, True), (2, True, False)])
def test_generator_forwards_ahead_on_third_stream_change_nothing(stage, fade, use_graph):
    """WGAN.early_generator (default in the small-batch step): the generated batches of metric block 0 and of the
    generator update are computed on a third stream beside the discriminator update.  The latent draws keep their
    order and the generator is not touched by the discriminator update, so metrics and weights after 4 steps must
    equal the in-order step bit for bit, eagerly and as a replayed graph."""
    runs = []
    for early in (False, True):
        d, g, comp, _ = build(stage + 1)
        gan = comp[stage][1 if fade else 0]
        gan.seed(41)
        gan.use_graph = use_graph
        gan.early_generator = early
        gan.set_train_steps(4)
        metrics = []
        for k in range(4):
            real, _ = R.synthetic_inputs(stage, 8, seed=60 + k)
            out = gan.train_step(real)
            metrics.append({key: float(v) for key, v in out.items()})
        weights = [v.numpy() for model in (gan.discriminator, gan.generator) for v in model.trainable_variables]
        runs.append((metrics, weights))
    (m_a, w_a), (m_b, w_b) = runs
    assert m_a == m_b
    for a, b in zip(w_a, w_b):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("stage,fade,n,use_graph", [(0, False, 16, False), (1, True, 8, False), (2, True, 4, True),
                                                    (1, False, 2, False)])
def test_merged_discriminator_passes_equal_separate_passes(stage, fade, n, use_graph):
    """WGAN.merge_passes (opt-in, for whole batches of 16 and fewer): the generated and the real batch go
    through the discriminator as ONE pass of 2n samples, MinibatchStdDev keeping the two apart, and the backward
    is seeded with the per-sample loss derivative instead of unit seeds that are combined afterwards.  The same
    passes over the same data in another float32 summation order: with the learning rates at 0 (a first Adam
    update is sign-like) metrics abs 1e-5, discriminator gradients rel-L2 5e-5 (alpha 5e-4), generator
    gradients 2e-3; and one real step each must leave finite, equal-to-1e-5 metrics."""
    runs = []
    for merge in (False, True):
        d, g, comp, _ = build(stage + 1)
        gan = comp[stage][1 if fade else 0]
        gan.seed(31)
        gan.use_graph = use_graph
        gan.merge_passes = merge
        gan.set_train_steps(4)
        assert gan._merge(n) == merge
        lrs = gan.d_optimizer.lr, gan.g_optimizer.lr
        gan.d_optimizer.lr = gan.g_optimizer.lr = 0.0
        real, _ = R.synthetic_inputs(stage, n, seed=12)
        for _ in range(3 if use_graph else 1):          # eager, capture, replay
            m = {k: float(v) for k, v in gan.train_step(real).items()}
        grads = {v.name: v.grad.cpu().numpy().copy() for model in (gan.discriminator, gan.generator)
                 for v in model.trainable_variables}
        d_names = {v.name for v in gan.discriminator.trainable_variables}
        gan.d_optimizer.lr, gan.g_optimizer.lr = lrs
        gan.use_graph = False
        m2 = {k: float(v) for k, v in gan.train_step(real).items()}
        runs.append((m, grads, m2))
    (m_a, g_a, m2_a), (m_b, g_b, m2_b) = runs
    for k in m_a:
        assert np.isfinite(m_b[k]) and abs(m_a[k] - m_b[k]) < 1e-5, (k, m_a[k], m_b[k])
        assert np.isfinite(m2_b[k]) and abs(m2_a[k] - m2_b[k]) < 5e-4, (k, m2_a[k], m2_b[k])
    for k, v in g_a.items():
        if np.abs(v).max() < 1e-12:
            continue
        tol = (5e-4 if k.endswith("/alpha") else 5e-5) if k in d_names else 2e-3
        assert rel_l2(g_b[k], v) < tol, "%s: %.3e" % (k, rel_l2(g_b[k], v))


def test_graph_survives_scratch_reallocation():
    """A captured graph carries the raw addresses of the library's scratch buffers.  Shape A is stepped twice
    (eager, then captured), a larger shape B follows and the workspace is grown beyond anything seen so far
    (the old buffer goes back to the allocator and is overwritten), then A is stepped again: the stale graph
    must be captured anew, and the whole trajectory must equal the eager one bit for bit."""
    from music_generator_amd import backend as K
    stage, fade = 1, True
    sizes = [4, 4, 8, 8, 4, 4, 8]
    runs = []
    for use_graph in (False, True):
        d, g, comp, _ = build(stage + 1)
        gan = comp[stage][1]
        gan.seed(5)
        gan.use_graph = use_graph
        gan.set_train_steps(len(sizes))
        metrics, captures = [], []
        for k, n in enumerate(sizes):
            if k == 4:
                gen0 = K.buffer_generation()
                grown = K.workspace(2 * K.workspace(1).numel())
                assert K.buffer_generation() > gen0
                junk = torch.full((grown.numel() // 8,), float("nan"), device="cuda")   # lands on freed scratch
            real, _ = R.synthetic_inputs(stage, n, seed=60 + k)
            out = gan.train_step(real)
            metrics.append({key: float(v) for key, v in out.items()})
            captures.append(sorted((key[0][0], e.get("generation")) for key, e in gan._graphs.items()))
        del junk
        weights = [v.numpy() for model in (gan.discriminator, gan.generator) for v in model.trainable_variables]
        runs.append((metrics, weights, captures))
    (m_e, w_e, _), (m_g, w_g, caps) = runs
    assert m_e == m_g
    for a, b in zip(w_e, w_g):
        assert np.array_equal(a, b)
    # step 4 (shape A again) captured a second time under the new generation; shape B followed at step 6
    gen_a_before, gen_a_after = dict(caps[3])[4], dict(caps[4])[4]
    assert gen_a_before is not None and gen_a_after > gen_a_before
    assert dict(caps[6])[8] == gen_a_after


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (torch.distributed.run as a
    child, before the parent touches the GPU) and relays rank 0's line.  RCCL refuses two ranks on one
    device, so the rehearsal on this one-GPU box carries the collectives over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MGX_BENCH_BACKEND="gloo", MGX_BENCH_C4="2,8,4")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--stage", "1", "--batch", "8",
                          "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["collective_backend"] == "gloo"
    assert out["config"]["global_batch"] == 16 and out["value"] > 0
    # the full-resolution data-parallel object of the multi-rank line, rehearsed at stage 2 (16 x 3000), batch 8 per rank
    c4 = out["c4"]
    assert c4["n_gpus"] == 2 and c4["global_batch"] == 16 and c4["rccl_ranks"] == 2 and c4["value"] > 0
    assert c4["collectives"]["allreduce_bytes_per_step"] > 0 and c4["dtype"] == "bf16"
    assert all(np.isfinite(v) for v in c4["last_metrics"].values())
