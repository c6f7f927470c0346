"""Micro-benchmark of single layers through the Python layer objects.
usage: python tools/bench_kernels.py conv3|conv3_128|conv2|gblock [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import music_generator_amd as M
from music_generator_amd import layers as L
from music_generator_amd.graph import Input, Model

which = sys.argv[1] if len(sys.argv) > 1 else "conv3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
N, H, W = 64, 32, 6000
cfg = {"conv3": (32, 32, (3, 3)), "conv3_128": (128, 32, (3, 3)), "conv2": (32, 32, (2, 2))}
if which in cfg:
    ci, co, ks = cfg[which]
    inp = Input(shape=(H, W, ci))
    t = L.Conv2D(co, ks, padding="same", kernel_initializer="he_normal")(inp)
    model = Model(inp, M.SoftRectifier(start_alpha=20000.0)(t))
    x = torch.rand((N, H, W, ci), device="cuda") * 2 - 1
    flops = 2.0 * N * H * W * ks[0] * ks[1] * ci * co
    y, tape = model.forward(x, save=True)
    dy = torch.randn_like(y)
    model.bind_grads()
    def fwd(): model.forward(x, save=True)
    def bwd():
        yy, tp = model.forward(x, save=True)
        model.backward(tp, dy, need_input_grad=True)
    for name, fn, mult in (("fwd", fwd, 1.0), ("fwd+dgrad+wgrad", bwd, 3.0)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%s %s: %.3f ms  %.1f TFLOP/s" % (which, name, dt * 1e3, flops * mult / dt / 1e12))
elif which == "gblock":
    g = M.define_generator(4)
    z = torch.randn((N, 1, 100, 2), device="cuda")
    m = g[3][0]
    m(z); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): m(z)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("G fwd stage 3 batch %d: %.3f ms  %.1f TFLOP/s (60.6 GFLOP/sample)" % (N, dt * 1e3, 60.6e9 * N / dt / 1e12))
