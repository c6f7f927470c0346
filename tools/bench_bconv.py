#!/usr/bin/env python3
"""Time one bf16 convolution launch of the discriminator blocks on the GPU:
    python tools/bench_bconv.py [--h 32 --w 6000 --n 64 --cin 32 --cout 32 --k 3]
forward with activation (with and without the derivative store) and the data-gradient with multiplier."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import music_generator_amd as M
from music_generator_amd import layers as L, ops
from music_generator_amd.graph import Input

ap = argparse.ArgumentParser()
ap.add_argument("--h", type=int, default=32); ap.add_argument("--w", type=int, default=6000)
ap.add_argument("--n", type=int, default=64); ap.add_argument("--cin", type=int, default=32)
ap.add_argument("--cout", type=int, default=32); ap.add_argument("--k", type=int, default=3)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
M.clear_session()
layer = L.Conv2D(a.cout, (a.k, a.k), padding="same", kernel_initializer="he_normal")
layer(Input(shape=(a.h, a.w, a.cin)))
x = torch.randn((a.n, a.h, a.w, a.cin), device="cuda").to(torch.bfloat16)
y = torch.empty((a.n, a.h, a.w, a.cout), dtype=torch.bfloat16, device="cuda")
dact = torch.empty_like(y)
mul = torch.rand((a.n, a.h, a.w, a.cin), device="cuda").to(torch.bfloat16)
dx = torch.empty_like(x)
kd, bd = layer.kernel.value(), layer.bias.value()
px = a.n * a.h * a.w
flops = 2.0 * a.k * a.k * a.cin * a.cout * px


def timed(f):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / a.reps


fd = layer._desc("f", a.n, a.h, a.w)
dd = layer._desc("d", a.n, a.h, a.w)
for name, f, nbytes in (
        ("fwd+act+dact", lambda: ops.conv_gemm_bf16(fd, x, kd, bd, y, dact=dact, act_alpha=7.0), px * 2 * (a.cin + 2 * a.cout)),
        ("fwd+act", lambda: ops.conv_gemm_bf16(fd, x, kd, bd, y, act_alpha=7.0), px * 2 * (a.cin + a.cout)),
        ("fwd+bias", lambda: ops.conv_gemm_bf16(fd, x, kd, bd, y, act_alpha=0.0), px * 2 * (a.cin + a.cout)),
        ("dgrad", lambda: ops.conv_gemm_bf16(dd, y, kd, None, dx), px * 2 * (a.cout + a.cin)),
        ("dgrad*mul", lambda: ops.conv_gemm_bf16(dd, y, kd, None, dx, ymul=mul), px * 2 * (a.cout + 2 * a.cin))):
    ms = timed(f)
    print("%-14s %.3f ms  %6.0f GB/s (algorithmic)  %6.0f TFLOP/s" % (name, ms, nbytes / ms / 1e6, flops / ms / 1e9), flush=True)
