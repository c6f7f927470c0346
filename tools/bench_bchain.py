#!/usr/bin/env python3
"""Time the row-streaming chain kernel against the layer-by-layer launches it replaces:
    python tools/bench_bchain.py [--h 32 --w 6000 --n 64]"""
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


def timed(f):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.reps):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / a.reps


x = torch.randn((a.n, a.h, a.w, 32), device="cuda").to(torch.bfloat16)
px = a.n * a.h * a.w
for ks in ((3, 3, 2), (3, 3), (3, 2)):
    layers = [L.Conv2D(32, (k, k), padding="same", kernel_initializer="he_normal") for k in ks]
    for l in layers:
        l(Input(shape=(a.h, a.w, 32)))
    bufs = [torch.empty_like(x) for _ in ks]
    alphas = [80000.0] * len(ks)

    def separate():
        cur = x
        for l, b in zip(layers, bufs):
            ops.conv_gemm_bf16(l._desc("f", a.n, a.h, a.w), cur, l.kernel.value(), l.bias.value(), b, act_alpha=80000.0)
            cur = b

    y = torch.empty_like(x)

    def chain():
        ops.conv_chain_bf16(x, [l.kernel.value() for l in layers], [l.bias.value() for l in layers], alphas, ks, y)

    t0, t1 = timed(separate), timed(chain)
    flops = 2.0 * sum(k * k for k in ks) * 1024 * px
    print("%-10s layer by layer %.3f ms (%4.0f TFLOP/s)   chain %.3f ms (%4.0f TFLOP/s, %4.0f GB/s in+out)   x%.2f"
          % (ks, t0, flops / t0 / 1e9, t1, flops / t1 / 1e9, px * 128 / t1 / 1e6, t0 / t1), flush=True)
    assert torch.equal(y, bufs[-1])
