"""Chunked generator-only inference (music_generator_amd.inference.generate_clips: the call shape
g_model(z, training=False) of trainer/utils.py:80-81, 103-104) against the CPU oracle, and at the
full-resolution stage through size-independent properties.

Tolerances: float32 audio vs the float64 oracle max abs 1e-4 (post-LayerNormalization values are O(1));
bf16 generator blocks vs the float32 path rel-L2 5e-2 on the audio."""
import numpy as np
import pytest
import torch

import music_generator_amd as M
from music_generator_amd.inference import ChecksumSink, HostRingSink, draw_latents, generate_clips
from oracle import ref_step as R

from helpers import rel_l2
from test_gpu_step import build

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("chunk", [1, 2, 5, 8])
def test_generate_clips_matches_oracle_stage2(chunk):
    """5 clips at stage 2 (16 x 3000): every chunking (ragged last chunk, one chunk, chunk > n) hands the
    sink the same clips in order, equal to the oracle's generator on the same latent vectors."""
    stage, n = 2, 5
    d, g, comp, W = build(stage + 1)
    names = R.layer_names(stage + 1)
    z = np.random.default_rng(4).standard_normal((n, 1, 100, 2)).astype(np.float32)
    got = []
    done = generate_clips(g[stage][0], n, chunk, lambda first, audio: got.append((first, audio.cpu().numpy())), z=z)
    assert done == n
    assert [f for f, _ in got] == list(range(0, n, chunk))
    assert [a.shape[0] for _, a in got] == [min(chunk, n - f) for f in range(0, n, chunk)]
    audio = np.concatenate([a for _, a in got])
    Wt = {k: torch.tensor(v, dtype=torch.float64) for k, v in W.items()}
    want = R.generator(Wt, names, stage, False, torch.tensor(z, dtype=torch.float64), None).numpy()
    assert audio.shape == want.shape == (n, 16, 3000, 2)
    assert np.abs(audio - want).max() < 1e-4


def test_sinks_agree_and_latents_do_not_depend_on_chunk():
    stage, n = 1, 6
    d, g, comp, W = build(stage + 1)
    model = g[stage][0]
    za, zb = draw_latents(n, seed=9), draw_latents(n, seed=9)
    assert torch.equal(za, zb) and za.shape == (n, 1, 100, 2)
    full = model(za).cpu().numpy()
    chk = ChecksumSink()
    generate_clips(model, n, 4, chk, seed=9)
    res = chk.result()
    flat = full.reshape(n, -1).astype(np.float64)
    assert np.allclose(res[:, 0], flat.sum(1), rtol=0, atol=1e-6 * flat.shape[1] ** 0.5)
    assert np.allclose(res[:, 1], (flat * flat).sum(1), rtol=1e-6)
    assert np.array_equal(res[:, 2].astype(np.float32), np.abs(full.reshape(n, -1)).max(1))
    seen = {}
    ring = HostRingSink(full.shape[1:], 4, lambda first, a: seen.update({first: a.copy()}), slots=2)
    generate_clips(model, n, 4, ring, seed=9)
    ring.flush()
    assert sorted(seen) == [0, 4]
    assert np.array_equal(np.concatenate([seen[0], seen[4]]), full)
    assert generate_clips(model, 0, 4, chk) == 0
    with pytest.raises(ValueError):
        generate_clips(model, 4, 0)
    with pytest.raises(ValueError):
        generate_clips(model, 4, 2, z=np.zeros((3, 1, 100, 2), np.float32))


@pytest.mark.parametrize("precision", ["float32", "bfloat16"])
def test_full_resolution_inference_properties(precision):
    """Stage 6 (256 x 48000), 10 clips in chunks of 8 (one ragged chunk), freshly initialised generator:
    its last layer is a per-clip LayerNormalization over (H, W, C) with gamma = 1, beta = 0, so every clip
    has mean 0 and mean square var/(var + 1e-3) (just under 1) whatever came before; outputs are finite;
    a clip does not depend on which chunk it was generated in."""
    M.backend.set_precision(precision)
    try:
        M.clear_session()
        g = M.define_generator(7)[6][0]
        n, size = 10, 256 * 48000 * 2
        chk = ChecksumSink()
        first_clip = {}

        def sink(first, audio):
            chk(first, audio)
            assert audio.shape[1:] == (256, 48000, 2) and audio.dtype == torch.float32
            if first == 8:
                first_clip["late"] = audio[0, ::16, ::64].cpu().numpy()

        assert generate_clips(g, n, 8, sink, seed=3) == n
        res = chk.result()
        assert res.shape == (n, 3) and np.isfinite(res).all()
        assert np.abs(res[:, 0] / size).max() < 1e-4, res[:, 0] / size
        ms = res[:, 1] / size
        assert (ms > 0.0).all() and (ms < 1.0 + 1e-4).all(), ms
        # clip 8 again, now first of a chunk of its own
        alone = g(draw_latents(n, seed=3)[8:9])[0, ::16, ::64].cpu().numpy()
        if precision == "float32":
            assert np.array_equal(alone, first_clip["late"])
        else:
            assert rel_l2(alone, first_clip["late"]) < 1e-6
    finally:
        M.backend.set_precision("float32")


def test_bf16_inference_close_to_float32():
    stage, n = 3, 4
    d, g, comp, W = build(stage + 1)
    z = draw_latents(n, seed=5)
    out = {}
    for precision in ("float32", "bfloat16"):
        M.backend.set_precision(precision)
        try:
            got = []
            generate_clips(g[stage][1], n, 2, lambda first, a: got.append(a.cpu().numpy()), z=z)
            out[precision] = np.concatenate(got)
        finally:
            M.backend.set_precision("float32")
    err = rel_l2(out["bfloat16"], out["float32"])
    assert err < 5e-2, err


@pytest.mark.parametrize("stage,fade", [(0, False), (1, True), (2, False), (2, True)])
def test_collapsed_generator_equals_layer_path(stage, fade):
    """The opt-in collapsed inference path (one table lookup + normalisation per output value) against the
    layer kernels and against the float64 oracle on the same latents.  Both float32 evaluations are within 1e-4
    (max abs, post-normalisation values are O(1)) of the oracle; against each other 5e-5."""
    from music_generator_amd.inference import CollapsedGenerator
    d, g, comp, W = build(stage + 1)
    names = R.layer_names(stage + 1)
    model = g[stage][1 if fade else 0]
    alpha = 0.37 if fade else None
    Wt = {k: torch.tensor(v, dtype=torch.float64) for k, v in W.items()}
    if fade:
        for layer in model.layers:
            if isinstance(layer, M.WeightedSum):
                M.backend.set_value(layer.alpha, alpha)
        Wt[names["g"][stage]["ws"] + "/alpha"] = torch.tensor(alpha, dtype=torch.float64)
    n = 6
    z = np.random.default_rng(11).standard_normal((n, 1, 100, 2)).astype(np.float32)
    zt = torch.tensor(z, device="cuda")
    layers = model(zt, training=False).cpu().numpy()
    cg = CollapsedGenerator(model)
    got = cg(zt).cpu().numpy()
    want = R.generator(Wt, names, stage, fade, torch.tensor(z, dtype=torch.float64), alpha).numpy()
    assert got.shape == layers.shape == want.shape
    assert np.abs(got - want).max() < 1e-4
    assert np.abs(got - layers).max() < 5e-5
    # after a weight change the table is stale until refresh()
    cg.norm.beta.assign(cg.norm.beta.numpy() + 0.5)
    assert np.abs(cg(zt).cpu().numpy() - (got + 0.5)).max() < 1e-5      # beta is read at call time
    first = model.layers[1]
    first.kernel.assign(first.kernel.numpy() * 1.5)
    cg.refresh()
    assert np.abs(cg(zt).cpu().numpy() - model(zt, training=False).cpu().numpy()).max() < 5e-5


def test_collapsed_generator_table_is_float32_under_bf16_precision():
    """CollapsedGenerator.refresh() with bf16 precision set globally: the probes still run through the float32
    layer kernels, so the table equals the one built under float32 bit for bit and the precision setting is
    left as it was."""
    from music_generator_amd.inference import CollapsedGenerator
    d, g, comp, W = build(3)
    model = g[2][0]
    want = CollapsedGenerator(model)
    M.backend.set_precision("bfloat16")
    try:
        got = CollapsedGenerator(model)
        assert M.backend.precision() == "bfloat16"
    finally:
        M.backend.set_precision("float32")
    assert torch.equal(got.table, want.table) and torch.equal(got.gram, want.gram)


def test_generate_clips_collapsed_sink_order_and_full_resolution():
    """generate_clips(collapse=True): same chunking contract; at the full-resolution stage (256 x 48000) two
    clips equal the layer path (relative L2 below 1e-5) and are normalised (mean 0, variance 1 - eps effect below 1e-3,
    gamma = 1 and beta = 0 at initialisation)."""
    M.clear_session()
    g = M.define_generator(7)
    model = g[6][0]
    z = draw_latents(2, seed=5)
    got = []
    assert generate_clips(model, 2, 1, lambda first, a: got.append((first, a.clone())), z=z, collapse=True) == 2
    assert [f for f, _ in got] == [0, 1]
    audio = torch.cat([a for _, a in got])
    assert tuple(audio.shape) == (2, 256, 48000, 2)
    ref = model(z, training=False)
    # 7 blocks of random linear maps give heavy-tailed values (|x| up to ~130 here; the largest difference
    # measured is 2.8e-4): max abs <= 5e-6 max|x|
    assert float((audio - ref).abs().max()) <= 5e-6 * float(ref.abs().max())
    assert float((audio - ref).double().norm() / ref.double().norm()) < 1e-5
    flat = audio.reshape(2, -1).double()
    assert float(flat.mean(1).abs().max()) < 1e-4
    assert float((flat.var(1, unbiased=False) - 1.0).abs().max()) < 2e-3
