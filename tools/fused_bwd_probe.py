#!/usr/bin/env python3
"""A few launches of the fused bf16 convolution backward and of the two separate kernels on one full-resolution
micro-batch image (256 x 48000 x 32), for counter passes:
    rocprofv3 --pmc <counters> -- python3 tools/fused_bwd_probe.py [3|2]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import music_generator_amd as M
from music_generator_amd import layers as L, ops
from music_generator_amd.graph import Input
k = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, h, w = 1, 256, 48000
M.clear_session()
layer = L.Conv2D(32, (k, k), strides=(1, 1), padding="same", kernel_initializer="he_normal"); layer(Input(shape=(h, w, 32)))
x = (torch.rand((n, h, w, 32), device="cuda") * 2 - 1).bfloat16()
g = torch.randn((n, h, w, 32), device="cuda").bfloat16()
kd = layer.kernel.value()
dw = torch.zeros(layer.kernel.shape, device="cuda").view(-1); db = torch.zeros((32,), device="cuda")
dx = torch.empty_like(x)
fd, dd = layer._desc("f", n, h, w), layer._desc("d", n, h, w)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
for rep in range(3):
    ev[0].record(); ops.conv_bwd_bf16(fd, x, g, kd, dw, db, dx, 7.0)
    ev[1].record(); ops.conv_gemm_bf16(dd, g, kd, None, dx, ymul=x, act_alpha=7.0)
    ev[2].record(); ops.conv_wgrad_bf16(fd, x, g, dw, db)
    ev[3].record(); torch.cuda.synchronize()
print("k=%d: fused %.3f ms, data gradient %.3f ms, weight gradient %.3f ms" % (k, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), ev[2].elapsed_time(ev[3])))
