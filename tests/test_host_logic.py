"""CPU tests of the host side: the models.py-compatible construction API (layer order, shared
weights, shapes, names), the fade-in schedule arithmetic, the progressive driver's step counts,
the data-parallel sharding rule (incl. a 2-rank gloo run), and that libmgx.so exports every
symbol include/mgx.h declares.  No kernel is launched here.
"""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import music_generator_amd as M
from music_generator_amd import _cabi, parallel, task
from music_generator_amd.layers import Conv2D, Dense, LayerNormalization
from oracle import ref_step as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mgx.h")).read()
    declared = set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_cabi.PROTOTYPES), declared ^ set(_cabi.PROTOTYPES)
    lib = ctypes.CDLL(_cabi.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _cabi.load().mgx_version() >= 1


def test_missing_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    d = M.define_discriminator(1)
    with pytest.raises(RuntimeError):
        d[0][0](np.zeros((4, 4, 750, 2), np.float32))


def test_list_structure_and_shapes():
    d = M.define_discriminator(3)
    g = M.define_generator(3)
    assert len(d) == 3 and len(g) == 3
    assert d[0][0] is d[0][1] and g[0][0] is g[0][1]
    assert d[1][0] is not d[1][1]
    for s in range(3):
        h, w = 4 << s, 750 << s
        for tw in (0, 1):
            assert g[s][tw].output_shape == (None, h, w, 2)
            assert g[s][tw].input_shape == (None, 1, 100, 2)
            assert d[s][tw].input_shape == (None, h, w, 2)
            assert d[s][tw].output_shape == (None, 1)
    assert list(d[1][0].input[0].shape) == [8, 1500, 2]


def test_layers_are_shared_between_stages_and_twins():
    d = M.define_discriminator(3)
    g = M.define_generator(3)
    # the stage-0 body is the tail of every later discriminator
    tail0 = d[0][0].layers[3:-1]
    for s in (1, 2):
        for tw in (0, 1):
            assert all(any(l is m for m in d[s][tw].layers) for l in tail0)
    # fade-in twin re-uses the previous from-audio conv + activation (old_model.layers[1], [2])
    assert any(l is d[0][0].layers[1] for l in d[1][1].layers)
    assert not any(l is d[0][0].layers[1] for l in d[1][0].layers)
    # generator: old LayerNormalization dropped, new one per twin, everything else shared
    ln = [l for l in g[1][0].layers if isinstance(l, LayerNormalization)]
    lnf = [l for l in g[1][1].layers if isinstance(l, LayerNormalization)]
    assert len(ln) == 1 and len(lnf) == 1 and ln[0] is not lnf[0]
    assert isinstance(g[1][0].layers[-1], LayerNormalization) and isinstance(g[1][0].layers[-2], Dense)
    assert g[1][0].layers[-2].units == 2
    assert isinstance(d[1][0].layers[1], Conv2D) and isinstance(d[1][0].layers[2], M.SoftRectifier)


def test_variable_names_and_shapes_match_the_oracle_replay():
    d = M.define_discriminator(3)
    g = M.define_generator(3)
    names, shapes = R.layer_names(3), R.weight_shapes(3)
    for s in range(3):
        for tw, fade in ((0, False), (1, True)):
            for which, lst in (("d", d), ("g", g)):
                model = lst[s][tw]
                assert sorted(v.name for v in model.trainable_variables) == sorted(R.trainable_names(names, s, fade, which))
                for v in model.trainable_variables:
                    assert tuple(v.shape) == tuple(shapes[v.name])
    assert d[0][0].count_params() == 252769 and g[0][0].count_params() == 226602


def test_activation_constants_per_stage():
    d = M.define_discriminator(7)
    for s in range(7):
        sot = [l for l in d[s][0].layers if isinstance(l, M.StaticOptTanh)]
        assert len(sot) == 1 and sot[0].alpha == R.activation_alphas(s)[0]
        first_sr = d[s][0].layers[2]
        assert isinstance(first_sr, M.SoftRectifier) and first_sr.start_alpha == R.activation_alphas(s)[1]


def test_fusion_plan_pairs_every_conv_with_its_activation():
    d = M.define_discriminator(2)
    plan = d[1][1]._plan
    fused = [op for op in plan if op.act_node is not None]
    # 11 new convs + old from-audio + 9 base convs + Dense(1) -> 22 fused launches
    assert len(fused) == 22
    assert all(isinstance(op.node.layer, (Conv2D, Dense)) for op in fused)


def test_composite_and_optimizer_settings():
    d = M.define_discriminator(2)
    g = M.define_generator(2)
    comp = M.define_composite(d, g, (1, 100, 2))
    assert len(comp) == 2 and all(len(p) == 2 for p in comp)
    gan = comp[1][1]
    assert gan.fade_in and not comp[1][0].fade_in and gan.d_steps == 1
    assert gan.discriminator is d[1][1] and gan.generator is g[1][1]
    for opt in (gan.d_optimizer, gan.g_optimizer):
        assert (opt.lr, opt.beta_1, opt.beta_2, opt.epsilon) == (0.0005, 0.0, 0.99, 10e-8)
    assert comp[1][0].d_optimizer is not comp[1][1].d_optimizer
    ws = [l for l in d[1][1].layers if isinstance(l, M.WeightedSum)]
    assert len(ws) == 1 and ws[0].alpha.trainable
    comp2 = M.define_composite(d, g, (1, 100, 2), trainable_alpha=False)
    assert not ws[0].alpha.trainable and comp2[1][1].discriminator.count_params() == d[1][0].count_params() + 384


def test_fade_in_needs_two_steps():
    if not torch.cuda.is_available():
        d = M.define_discriminator(2)
        g = M.define_generator(2)
        gan = M.define_composite(d, g, (1, 100, 2))[1][1]
        gan.set_train_steps(1)
        with pytest.raises(ValueError):
            gan.train_step(np.zeros((4, 8, 1500, 2), np.float32))


def test_schedule_constants_and_step_counts():
    assert task.BATCH_SIZES == [16, 8, 4, 2, 2, 2, 2]
    assert task.EPOCHS_NORM == [200, 220, 250, 280, 300, 350, 420]
    assert task.EPOCHS_FADE == [10, 15, 32, 25, 30, 35, 40]
    assert task.phase_steps(1003, 16, 200) == 62 * 200
    assert task.phase_steps(37, 8, 15) == 4 * 15


def test_shard_indices_keep_minibatch_groups_whole():
    n, world = 32, 4
    m = n // 4
    seen = []
    for r in range(world):
        idx = parallel.shard_indices(n, r, world)
        assert len(idx) == n // world
        loc = idx.reshape(4, -1)                      # local [G, M/R]
        for col in range(loc.shape[1]):
            assert len({int(i) % m for i in loc[:, col]}) == 1      # one global group per local group
            assert sorted(int(i) // m for i in loc[:, col]) == [0, 1, 2, 3]
        seen.extend(idx.tolist())
    assert sorted(seen) == list(range(n))
    with pytest.raises(ValueError):
        parallel.shard_indices(12, 0, 2)              # 3 groups do not split over 2 ranks


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from music_generator_amd.parallel import DataParallel, shard_indices
from oracle import ref_step as R
dp = DataParallel(backend="gloo")
n_blocks, stage, batch = 1, 0, 8
names = R.layer_names(n_blocks)
W = {k: torch.tensor(v, dtype=torch.float64) for k, v in R.synthetic_weights(n_blocks, seed=100).items()}
real, z4 = R.synthetic_inputs(stage, batch, seed=0)
dn = R.trainable_names(names, stage, False, "d")

def d_grads(real_b, z_b, n_global, sums=None):
    P = {k: (v.clone().requires_grad_(True) if k in dn else v) for k, v in W.items()}
    fake = R.generator(P, names, stage, False, torch.tensor(z_b, dtype=torch.float64))
    fl = R.discriminator(P, names, stage, False, fake)
    rl = R.discriminator(P, names, stage, False, torch.tensor(real_b, dtype=torch.float64))
    local = torch.stack([fl.sum(), rl.sum()]).detach()
    if sums is None:
        return local
    cu, ci = sums[0] / n_global, sums[1] / n_global
    sgn = torch.sign(cu - ci)
    # d(loss)/d(logit) with the GLOBAL batch in the denominator (parallel.py docstring)
    loss = ((sgn + 1e-4) * rl.sum() - (sgn + 9.99e-5) * fl.sum()) / n_global
    return torch.autograd.grad(loss, [P[k] for k in dn])

idx = shard_indices(batch, dp.rank, dp.world_size)
sums = d_grads(real[idx], z4[0][idx], batch)
dp.all_reduce(sums)
flat = torch.cat([g.reshape(-1) for g in d_grads(real[idx], z4[0][idx], batch, sums)])
dp.all_reduce(flat)
if dp.rank == 0:
    full_sums = d_grads(real, z4[0], batch)
    ref = torch.cat([g.reshape(-1) for g in d_grads(real, z4[0], batch, full_sums)])
    err = float((flat - ref).norm() / ref.norm())
    assert torch.allclose(sums, full_sums, rtol=1e-12, atol=1e-12), (sums, full_sums)
    assert err < 1e-10, err
    print("DP-OK", err)
dist.destroy_process_group()
"""


def test_two_rank_gloo_data_parallel_equals_single_process(tmp_path):
    """World-size-2 rehearsal on CPU (gloo): sharding by MinibatchStdDev groups + all-reduce of the
    logit sums and of the gradients reproduces the single-process discriminator gradients.  The
    per-rank arithmetic is the oracle's; what is under test is parallel.py's rule and collectives."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29613", OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29613", str(script), ROOT]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DP-OK" in out.stdout


def test_cabi_rejects_bad_arguments_before_touching_the_device():
    """Argument validation of the entry points happens on the host, before any launch: null
    pointers, inconsistent descriptors, short workspaces and misaligned position ranges return
    the documented status codes (no GPU needed to observe them)."""
    lib = _cabi.load()
    d = _cabi.ConvDesc()
    for k, v in dict(nimg=1, hin=16, win=32, cin=32, hout=16, wout=32, kh=3, kw=3, sh=1, sw=1, pt=1, pl=1, flip=0,
                     ngemm=32, dkh=1, dkw=1, cout=32, wsk=32, wsn=1, wtap=1024, nbias=32, accumulate=0).items():
        setattr(d, k, v)
    d.act_alpha = 0.0
    assert lib.mgx_conv_gemm_f32(ctypes.byref(d), None, None, None, None, None, None, None) == _cabi.MGX_ERR_ARG
    bad = _cabi.ConvDesc.from_buffer_copy(d)
    bad.ngemm = 48                                       # != dkh*dkw*cout
    fake = ctypes.c_void_p(4096)                         # never dereferenced: validation fails first
    assert lib.mgx_conv_gemm_f32(ctypes.byref(bad), fake, None, fake, fake, fake, None, None) == _cabi.MGX_ERR_ARG
    assert lib.mgx_conv_wgrad_workspace(None) < 0
    need = lib.mgx_conv_wgrad_workspace(ctypes.byref(d))
    assert need > 0
    assert lib.mgx_conv_wgrad_f32(ctypes.byref(d), fake, fake, None, fake, fake, fake, need - 1, None) \
        == _cabi.MGX_ERR_WORKSPACE
    # forward and weight gradient: Winograd kernels (tile kernels when switched off)
    wino = 3 if os.environ.get("MGX_WINOGRAD", "1") != "0" else 1
    assert lib.mgx_conv_route(ctypes.byref(d), 0) == wino and lib.mgx_conv_route(ctypes.byref(d), 1) == wino
    assert lib.mgx_conv_route(ctypes.byref(d), 2) == wino          # with an input multiplier: 32 channels, linear
    d.hin = d.hout = 4                                   # 4-row tiles of the direct kernels
    assert lib.mgx_conv_route(ctypes.byref(d), 0) == 1 and lib.mgx_conv_route(ctypes.byref(d), 1) == 1
    d.kh = d.kw = 4                                      # 4x4 'same' 32 -> 32: tile kernels too (weights 16 x 1024)
    d.wtap = 1024
    assert lib.mgx_conv_route(ctypes.byref(d), 0) == 1 and lib.mgx_conv_route(ctypes.byref(d), 1) == 1
    d.kh = d.kw = 3
    d.hin = d.hout = 6                                   # not a multiple of 4: generic kernel
    assert lib.mgx_conv_route(ctypes.byref(d), 0) == 0
    # fused generator block: unset weight pointers, misaligned range, short workspace
    w = _cabi.GBlockWeights()
    assert lib.mgx_gblock_fwd_f32(ctypes.byref(w), fake, fake, fake, None, 1, 4, 750, 0, 0, None) == _cabi.MGX_ERR_ARG
    for name, _ in _cabi.GBlockWeights._fields_:
        cur = getattr(w, name)
        if isinstance(cur, ctypes.Array):
            for i in range(len(cur)):
                cur[i] = 4096
        else:
            setattr(w, name, 4096)
    assert lib.mgx_gblock_fwd_f32(ctypes.byref(w), fake, fake, fake, None, 1, 4, 750, 64, 0, None) == _cabi.MGX_ERR_ARG
    assert lib.mgx_gblock_dgrad_f32(ctypes.byref(w), fake, fake, fake, fake, 1, 4, 750, 3000, 0, None) \
        == _cabi.MGX_ERR_ARG
    assert lib.mgx_gblock_wgrad_f32(fake, fake, ctypes.byref(w), fake, 16, 3000, None) == _cabi.MGX_ERR_WORKSPACE
    assert lib.mgx_gblock_side_floats(1, 4, 750) == lib.mgx_gblock_side_floats_n(3000) == 24 * 4 * 4104 * 32


def test_wav_sink_round_trip(tmp_path):
    """(H, W, 2) clip -> (H*W, 2) float WAV at rate W (trainer/utils.py:84, 95, 108), read back
    with scipy."""
    from scipy.io import wavfile
    from music_generator_amd import audio_out
    rng = np.random.default_rng(3)
    audio = rng.uniform(-0.5, 0.5, (3, 4, 750, 2)).astype(np.float32)
    paths = audio_out.write_examples(audio, "gen", 7, out_dir=str(tmp_path))
    assert [os.path.relpath(p, tmp_path) for p in paths] == ["4x750/gen0.wav", "4x750/gen7.wav", "4x750/gen14.wav"]
    rate, sig = wavfile.read(paths[1])
    assert rate == 750 and sig.dtype == np.float32 and sig.shape == (3000, 2)
    assert np.array_equal(sig, audio[1].reshape(3000, 2))
    sink = audio_out.WavSink(out_dir=str(tmp_path), prefix="s_", pick=2)
    sink(5, 1, torch.from_numpy(audio))
    rate, sig = wavfile.read(sink.written[0])
    assert sink.written[0].endswith("4x750/s_epoch5_1.wav") and rate == 750
    want = audio[2].reshape(3000, 2) / np.max(np.abs(audio[2].reshape(3000, 2)), axis=0)
    assert np.allclose(sig, want, rtol=0, atol=1e-7) and np.max(np.abs(sig)) == pytest.approx(1.0)
    silent = audio_out.clip_to_signal(np.zeros((4, 750, 2), np.float32), normalise=True)
    assert np.all(silent == 0)
    with pytest.raises(ValueError):
        audio_out.clip_to_signal(np.zeros((4, 750, 3), np.float32))


def test_checkpoint_restores_weights_counters_and_position(tmp_path):
    """Host side of the checkpoint: every shared variable once, WGAN counters, schedule position;
    a family of another depth is refused."""
    from music_generator_amd import checkpoint
    M.backend.clear_session()
    d1, g1, c1 = task.build(2)
    rng = np.random.default_rng(11)
    variables = checkpoint._unique_variables([d1, g1])
    assert len({id(v) for v in variables}) == len(variables)
    for v in variables:
        v.assign(rng.standard_normal(v.shape).astype(np.float32))
    c1[1][1].actual_step, c1[1][1].total_steps, c1[1][1].train_steps = 17, 40, 60
    c1[1][1].d_optimizer.iterations = 17
    path = checkpoint.save_checkpoint(str(tmp_path / "ck"), d1, g1, c1, stage=1, fade_in=True, epochs_done=3,
                                      extra={"note": "x"})
    assert path.endswith("ck.npz") and not os.path.exists(path + ".tmp.npz")
    d2, g2, c2 = task.build(2)              # fresh names (the session counters moved on), fresh values
    pos = checkpoint.load_checkpoint(path, d2, g2, c2)
    assert pos == {"stage": 1, "fade_in": True, "epochs_done": 3, "extra": {"note": "x"}}
    for a, b in zip(variables, checkpoint._unique_variables([d2, g2])):
        assert a.shape == b.shape and np.array_equal(a.numpy(), b.numpy())
    assert (c2[1][1].actual_step, c2[1][1].total_steps, c2[1][1].train_steps) == (17, 40, 60)
    assert c2[1][1].d_optimizer.iterations == 17 and c2[1][0].d_optimizer.iterations == 0
    # sharing survives: the straight and fade-in twins still hold the same variable objects
    last = lambda m: [l for l in m.layers if l.weights][-1].weights[0]
    assert last(d2[1][0]) is last(d2[1][1])
    d3, g3, c3 = task.build(3)
    with pytest.raises(ValueError):
        checkpoint.load_checkpoint(path, d3, g3, c3)
    # the optimisers' hyper-parameters travel with the checkpoint: another learning rate is refused
    d4, g4, c4 = task.build(2)
    c4[0][0].g_optimizer.lr = 1e-3
    with pytest.raises(ValueError, match="lr"):
        checkpoint.load_checkpoint(path, d4, g4, c4)


def test_resume_skips_finished_phases():
    """task.train(resume=...) runs only what is left: the named phase from `epochs_done` on, then
    the later ones."""
    calls = []

    class FakeGan:
        def __init__(self, tag):
            self.tag = tag

        def set_train_steps(self, n):
            pass

        def fit(self, data, batch_size, epochs, callbacks, verbose, initial_epoch):
            calls.append((self.tag, epochs, initial_epoch))
            return type("H", (), {"history": {}})()

    class Shape:
        def __init__(self, h, w):
            self.output_shape = (None, h, w, 2)

    g_models = [[Shape(4, 750)] * 2, [Shape(8, 1500)] * 2, [Shape(16, 3000)] * 2]
    gans = [[FakeGan("s0"), FakeGan("s0f")], [FakeGan("s1"), FakeGan("s1f")], [FakeGan("s2"), FakeGan("s2f")]]
    provider = lambda h, w: np.zeros((4, 1, 1, 2), np.float32)
    task.train(g_models, None, gans, provider, batch_sizes=[2, 2, 2], epochs_norm=[5, 6, 7], epochs_fade=[0, 2, 3])
    assert calls == [("s0", 5, 0), ("s1f", 2, 0), ("s1", 6, 0), ("s2f", 3, 0), ("s2", 7, 0)]
    del calls[:]
    task.train(g_models, None, gans, provider, batch_sizes=[2, 2, 2], epochs_norm=[5, 6, 7], epochs_fade=[0, 2, 3],
               resume={"stage": 1, "fade_in": False, "epochs_done": 4})
    assert calls == [("s1", 6, 4), ("s2f", 3, 0), ("s2", 7, 0)]


def test_evaluator_family_structure_and_inception_score():
    """trainer/evaluacion.py:27-110: per stage the convolution list the oracle replays, old layers
    shared from index 3 on, (None, 9) outputs; trainer/utils.py:65-76 against explicit loops."""
    from music_generator_amd import evaluacion as E
    from oracle import ref_evaluator as RE
    M.backend.clear_session()
    ev = E.define_evaluator(3)
    for s, m in enumerate(ev):
        convs = [l for l in m.layers if isinstance(l, Conv2D)]
        assert [(l.kernel_size[0], l.kernel_size[1], l.strides[0], l.padding) for l in convs] == RE.conv_specs(s)
        assert all(l.filters == 128 for l in convs)
        assert m.input.shape == (None, 4 << s, 750 << s, 2) and m.output_shape == (None, 9)
        assert isinstance(m.layers[-1], Dense) and m.layers[-1].units == 9
    assert ev[1].layers[-1] is ev[0].layers[-1] and ev[2].layers[-3] is ev[0].layers[-3]
    assert ev[0].layers[2] not in ev[1].layers              # the old first 3x3 convolution is dropped
    assert ev[0].layers[3] in ev[1].layers
    rng = np.random.default_rng(0)
    p = rng.uniform(0.01, 1.0, (7, 9))
    p /= p.sum(axis=1, keepdims=True)
    assert E.calculate_inception_score(p) == pytest.approx(RE.inception_score(p), rel=1e-12)
    assert E.calculate_inception_score(np.full((5, 9), 1 / 9.0)) == pytest.approx(1.0)
    assert E.calculate_inception_score(np.eye(9)) == pytest.approx(9.0, rel=1e-9)


def test_directory_dataset_reader(tmp_path):
    """trainer/dataset.py:213-239: nine class folders per resolution, (channels, frames) samples
    reshaped as they are to (H, W, 2), one-hot labels, broken files skipped, five failures in a row
    end a folder."""
    from scipy.io import wavfile
    from music_generator_amd import dataset
    rng = np.random.default_rng(5)
    h, w = 4, 750
    base = tmp_path / "wav" / ("%d-%d" % (h, w))
    clips = {}
    for folder, count in ((1, 2), (3, 1)):
        (base / str(folder)).mkdir(parents=True)
        for i in range(count):
            sig = rng.uniform(-1, 1, (h * w, 2)).astype(np.float32)          # (frames, channels) on disk
            wavfile.write(str(base / str(folder) / ("%d.wav" % (i + 1))), w, sig)
            clips[(folder, i)] = sig
    (base / "1" / "3.wav").write_bytes(b"not a wav file")                    # skipped
    wavfile.write(str(base / "3" / "0_short.wav"), w, np.zeros((10, 2), np.float32))   # wrong length: skipped
    pcm = (rng.uniform(-1, 1, (h * w, 2)) * 32767).astype(np.int16)
    (base / "9").mkdir()
    wavfile.write(str(base / "9" / "1.wav"), w, pcm)
    (base / "5").mkdir()
    for i in range(6):
        (base / "5" / ("%d.wav" % i)).write_bytes(b"junk")
    wavfile.write(str(base / "5" / "z.wav"), w, clips[(1, 0)])               # after five failures: never reached
    x, y = dataset.read_dataset((h, w), "wav", root=str(tmp_path))
    assert x.shape == (4, h, w, 2) and x.dtype == np.float32 and y.shape == (4, 9)
    assert y.argmax(axis=1).tolist() == [0, 0, 2, 8] and np.all(y.sum(axis=1) == 1)
    assert np.array_equal(x[0], clips[(1, 0)].T.reshape(h, w, 2))            # audiofile layout, reshaped as is
    assert np.array_equal(x[2], clips[(3, 0)].T.reshape(h, w, 2))
    assert np.allclose(x[3], (pcm.astype(np.float32) / 32768.0).T.reshape(h, w, 2))
    xi, _ = dataset.read_dataset((h, w), "wav", root=str(tmp_path), channels_first=False)
    assert np.array_equal(xi[1], clips[(1, 1)].reshape(h, w, 2))
    assert dataset.directory_provider(str(tmp_path))(h, w).shape == (4, h, w, 2)
    assert dataset.read_dataset((8, 1500), "wav", root=str(tmp_path))[0].shape == (0, 8, 1500, 2)
    with pytest.raises(NotImplementedError):
        dataset.read_dataset((h, w), "mp3", root=str(tmp_path))


def test_generate_clips_validates_before_touching_the_device():
    """Argument errors are raised before any device work; without a GPU the call fails loudly (there is no
    host fallback), it does not return host-computed audio."""
    from music_generator_amd import inference
    with pytest.raises(ValueError):
        inference.generate_clips(None, 4, 0)
    with pytest.raises(ValueError):
        inference.generate_clips(None, -1, 2)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="ROCm GPU"):
            inference.generate_clips(lambda z, training=False: z, 2, 2)
    z = inference.draw_latents(3, seed=1, device="cpu")
    assert z.shape == (3, 1, 100, 2) and torch.equal(z, inference.draw_latents(3, seed=1, device="cpu"))


def test_fit_shards_every_global_batch_over_the_ranks():
    """WGAN.fit with a DataParallel attached: batch_size is the global batch, every rank draws the same
    permutation and steps on its own MinibatchStdDev groups of each global batch; the union over the ranks
    is the single-process batch, group membership is kept; a batch that does not split raises."""
    class Dist:
        def __init__(self, rank, world):
            self.rank, self.world_size = rank, world

    x = np.arange(32, dtype=np.float32).reshape(32, 1)
    seen = {}
    for rank in (None, 0, 1):
        M.clear_session()
        gan = M.WGAN(discriminator=_Stub(), generator=_Stub(), latent_dim=(1, 100, 2))
        gan.dist = None if rank is None else Dist(rank, 2)
        fed = []
        gan.train_step = lambda batch, fed=fed: fed.append(np.asarray(batch).reshape(-1).copy()) or {}
        gan.fit(x, batch_size=16, epochs=2, shuffle_seed=3)
        seen[rank] = fed
    assert len(seen[None]) == len(seen[0]) == len(seen[1]) == 4
    for whole, a, b in zip(seen[None], seen[0], seen[1]):
        assert a.size == b.size == 8
        assert np.array_equal(whole[parallel.shard_indices(16, 0, 2)], a)
        assert np.array_equal(whole[parallel.shard_indices(16, 1, 2)], b)
        assert sorted(np.concatenate([a, b]).tolist()) == sorted(whole.tolist())
    other = M.WGAN(discriminator=_Stub(), generator=_Stub(), latent_dim=(1, 100, 2))
    other.train_step = lambda batch: {}
    other.dist = Dist(0, 3)
    with pytest.raises(ValueError):
        other.fit(x, batch_size=16, epochs=1)
    a, b = [], []
    for seed, out in ((0, a), (1, b)):
        other.dist = None
        other.train_step = lambda batch, out=out: out.append(np.asarray(batch).reshape(-1).copy()) or {}
        other.fit(x, batch_size=32, epochs=1, shuffle_seed=seed)
    assert not np.array_equal(a[0], b[0])


class _Stub:
    layers = ()
