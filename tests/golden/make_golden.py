"""Generates tests/golden/step_*.npz from the CPU oracle (oracle/ref_step.py, float64).

The reference itself cannot run here (TensorFlow/Keras are absent), so these vectors pin the
ORACLE, not the reference: they detect drift of the oracle and let the GPU tests run without
re-evaluating it.  Inputs are regenerated from seeds on both sides; only outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_step as R  # noqa: E402

# name: (n_blocks, stage, fade_in, batch, fade step k of n, trainable_alpha)
CASES = {
    "s0_n16": (1, 0, False, 16, None, True),
    "s0_n4": (1, 0, False, 4, None, True),
    "s1_straight_n4": (2, 1, False, 4, None, True),
    "s1_fade_a0_n4": (2, 1, True, 4, (0, 101), True),
    "s1_fade_a037_n4": (2, 1, True, 4, (37, 101), True),
    "s1_fade_a1_n4": (2, 1, True, 4, (100, 101), True),
    "s1_fade_a037_fixed_alpha_n4": (2, 1, True, 4, (37, 101), False),
    "s2_fade_a037_n4": (3, 2, True, 4, (37, 101), True),
    "s3_fade_a037_n4": (4, 3, True, 4, (37, 101), True),      # the bench workload's graph (32 x 6000) at batch 4
}
# third consecutive step of a fade-in WGAN (Adam t = 3, second moments warmed by two steps): stage 1,
# batch 4, 5 steps in the stage (alpha = k/4), inputs of step k drawn with seed 10 + k
T3 = ("s1_fade_t3_n4", 2, 1, 4, 5)


def summarize(a, k=64):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    idx = np.linspace(0, a.size - 1, num=min(k, a.size)).astype(np.int64)
    return np.concatenate([[a.sum(), np.sqrt((a * a).sum())], a[idx]])


def warmed_state(n_blocks, stage, batch, n_steps, k_done):
    """Weights and Adam second moments after `k_done` oracle steps of the fade-in WGAN."""
    cur, vd, vg = R.synthetic_weights(n_blocks, seed=100), None, None
    for k in range(k_done):
        real, z4 = R.synthetic_inputs(stage, batch, seed=10 + k)
        res = R.train_step(cur, n_blocks, stage, True, real, z4, alpha=k / float(n_steps - 1), v_d=vd, v_g=vg,
                           t=k + 1, keep_grads=False)
        cur, vd, vg = res.weights, res.v_d, res.v_g
    return cur, vd, vg


def run_t3():
    name, n_blocks, stage, batch, n_steps = T3
    cur, vd, vg = warmed_state(n_blocks, stage, batch, n_steps, 2)
    real, z4 = R.synthetic_inputs(stage, batch, seed=12)
    res = R.train_step(cur, n_blocks, stage, True, real, z4, alpha=2 / float(n_steps - 1), v_d=vd, v_g=vg, t=3)
    return pack(res)


def run_case(name):
    n_blocks, stage, fade, batch, ks, tr_alpha = CASES[name]
    W = R.synthetic_weights(n_blocks, seed=100)
    real, z4 = R.synthetic_inputs(stage, batch, seed=0)
    alpha = ks[0] / float(ks[1] - 1) if ks else None
    return pack(R.train_step(W, n_blocks, stage, fade, real, z4, alpha=alpha, trainable_alpha=tr_alpha))


def pack(res):
    out = {"metrics": np.array([res.metrics[k] for k in ("delta_0", "cu_0", "ci_0", "delta_1", "cu_1", "ci_1")]),
           "losses": np.array([res.d_loss, res.g_loss]),
           "fake_logits": res.fake_logits_d_step.reshape(-1), "real_logits": res.real_logits_d_step.reshape(-1),
           "fake_audio": summarize(res.fake_d_step, 256)}
    for k, v in res.d_grads.items():
        out["dgrad|" + k] = summarize(v)
    for k, v in res.g_grads.items():
        out["ggrad|" + k] = summarize(v)
    for k in list(res.d_grads) + list(res.g_grads):
        out["w|" + k] = summarize(res.weights[k])
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name in (sys.argv[1:] or list(CASES) + [T3[0]]):
        out = run_t3() if name == T3[0] else run_case(name)
        np.savez_compressed(os.path.join(here, "step_%s.npz" % name), **{k.replace("/", "~"): v for k, v in out.items()})
        print(name, out["metrics"], out["losses"])
