#!/usr/bin/env python3
"""Time the fused from-audio + 3x3 kernel against the two launches it replaces:
    python tools/bench_baudio.py [--h 32 --w 6000 --n 64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import music_generator_amd as M
from music_generator_amd import layers as L, ops
from music_generator_amd.graph import Input

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=32); ap.add_argument("--w", type=int, default=6000)
ap.add_argument("--n", type=int, default=64); ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
M.clear_session()
la = L.Conv2D(128, (1, 1), padding="same", kernel_initializer="he_normal")
lb = L.Conv2D(32, (3, 3), padding="same", kernel_initializer="he_normal")
la(Input(shape=(a.h, a.w, 2))); lb(Input(shape=(a.h, a.w, 128)))
x = torch.rand((a.n, a.h, a.w, 2), device="cuda") * 2 - 1
px = a.n * a.h * a.w
mid = torch.empty((a.n, a.h, a.w, 128), dtype=torch.bfloat16, device="cuda")
y0 = torch.empty((a.n, a.h, a.w, 32), dtype=torch.bfloat16, device="cuda")
y1 = torch.empty_like(y0)


def timed(f):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / a.reps


def separate():
    ops.pointwise2_bf16_fwd(x, la.kernel.value(), la.bias.value(), mid, None, px, 80000.0)
    ops.conv_gemm_bf16(lb._desc("f", a.n, a.h, a.w), mid, lb.kernel.value(), lb.bias.value(), y0, act_alpha=80000.0)


def fused():
    ops.audio_conv3_bf16(x, la.kernel.value(), la.bias.value(), 80000.0, lb.kernel.value(), lb.bias.value(), 80000.0, y1)


t0, t1 = timed(separate), timed(fused)
flops = 2.0 * 9 * 128 * 32 * px
print("two launches %.3f ms   fused %.3f ms (%4.0f TFLOP/s of the 3x3)   x%.2f" % (t0, t1, flops / t1 / 1e9, t0 / t1))
assert torch.equal(y0, y1)
