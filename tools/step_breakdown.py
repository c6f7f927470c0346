#!/usr/bin/env python3
"""Per-shape timing of the contraction launches of one train step (HIP events on the launch
stream): python tools/step_breakdown.py [--stage 3] [--batch 64]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import music_generator_amd as M
from music_generator_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
args = ap.parse_args()
orig_g, orig_w = ops._gemm_key, ops._wgrad_key


def shape(d):
    return "%dx%d s%d %d->%d rows=%d %s" % (d.kh, d.kw, d.sh, d.cin, d.ngemm, d.nimg * d.hout * d.wout,
                                           "scatter" if d.dkh * d.dkw > 1 else "")


ops._gemm_key = lambda d, mul=False: orig_g(d, mul) + " | " + shape(d)
ops._wgrad_key = lambda d: orig_w(d) + " | " + shape(d)
M.clear_session()
d = M.define_discriminator(args.stage + 1)
g = M.define_generator(args.stage + 1)
gan = M.define_composite(d, g, (1, 100, 2))[args.stage][1]
gan.set_train_steps(100)
h, w = 4 << args.stage, 750 << args.stage
real = torch.rand((args.batch, h, w, 2), device="cuda") * 2 - 1
gan.train_step(real)
torch.cuda.synchronize()
prof = ops.LaunchProfiler()
ops.PROFILER = prof
gan.train_step(real)
torch.cuda.synchronize()
ops.PROFILER = None
rows = sorted(prof.summary().items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
for key, (n, ms, fl, by) in rows[:45]:
    print("%7.2f ms %5.1f%% n=%3d %6.1f TF/s %6.2f TB/s  %s" % (ms, 100 * ms / tot, n, fl / ms / 1e9, by / ms / 1e9, key))
print("total %.1f ms" % tot)
