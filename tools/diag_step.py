"""Diagnostic: per-tensor gradient errors of one train step vs the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import music_generator_amd as M
from oracle import ref_step as R
from helpers import rel_l2
import test_gpu_step as T

for name in sys.argv[1:] or ["s0_n4", "s1_straight_n4", "s1_fade_a037_n4"]:
    n_blocks, stage, fade, batch, ks, tr_alpha = T.CASES[name]
    gan, metrics, fake = T.device_step(name)
    W = R.synthetic_weights(n_blocks, seed=100)
    real, z4 = R.synthetic_inputs(stage, batch, seed=0)
    res = R.train_step(W, n_blocks, stage, fade, real, z4, alpha=(ks[0] / float(ks[1] - 1)) if ks else None, trainable_alpha=tr_alpha)
    print(name, {k: (float(metrics[k]) - v) for k, v in res.metrics.items()})
    for tag, grads, model in (("D", res.d_grads, gan.discriminator), ("G", res.g_grads, gan.generator)):
        errs = []
        for v in model.trainable_variables:
            want = grads[v.name]; have = v.grad.cpu().numpy().reshape(v.shape)
            errs.append((rel_l2(have, want), v.name, float(np.abs(want).max())))
        errs.sort(reverse=True)
        print(tag, "worst:", [(("%.2e" % e), n, "%.1e" % m) for e, n, m in errs[:6]], "median %.2e" % errs[len(errs)//2][0])
        werr = []
        for v in model.trainable_variables:
            werr.append((float(np.abs(v.numpy() - res.weights[v.name]).max()), v.name))
        werr.sort(reverse=True)
        print(tag, "weights worst abs:", [("%.2e" % e, n) for e, n in werr[:4]])
