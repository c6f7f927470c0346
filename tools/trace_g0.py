#!/usr/bin/env python3
"""Per-launch times of one stage-0 generator forward + backward (batch from argv), in launch order."""
import sys
import torch
sys.path.insert(0, ".")
import music_generator_amd as M
from music_generator_amd import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = M.define_generator(1)[0][0]
z = torch.randn((n, 1, 100, 2), device="cuda")
dy = torch.randn((n, 4, 750, 2), device="cuda")
for rep in range(3):
    ops.PROFILER = ops.LaunchProfiler() if rep == 2 else None
    g.bind_grads().zero_()
    y, tape = g.forward(z, save=True)
    g.backward(tape, dy)
    torch.cuda.synchronize()
for key, flops, e0, e1, nbytes in ops.PROFILER.records:
    print("%-40s %8.1f us  %8.2f MFLOP %8.2f MB" % (key, e0.elapsed_time(e1) * 1e3, flops / 1e6, nbytes / 1e6))
