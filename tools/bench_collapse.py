"""The collapsed generator (inference.CollapsedGenerator) at the full-resolution stage: difference to the layer
path on two clips, then clips/s and write rate for chunks of 8, 32 and 64.  PYTHONPATH=. python tools/bench_collapse.py"""
import torch, music_generator_amd as M
from music_generator_amd.inference import CollapsedGenerator, draw_latents
import time
M.clear_session()
g = M.define_generator(7)
model = g[6][0]
z = draw_latents(2, seed=5)
ref = model(z, training=False)
cg = CollapsedGenerator(model)
out = cg(z)
d = (out - ref).double()
print("max|ref|", float(ref.abs().max()), "max abs diff", float(d.abs().max()), "rel l2", float(d.norm() / ref.double().norm()))
i = int(d.abs().argmax()); print("at value", float(ref.reshape(-1)[i]))
for m in (8, 32, 64):
    zz = draw_latents(m, seed=1)
    o = cg(zz); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        o = cg(zz)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print("chunk", m, "ms", dt * 1e3, "clips/s", m / dt, "GB/s written", m * 98.304e6 / dt / 1e9)
    del o
