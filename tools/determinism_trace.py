#!/usr/bin/env python3
"""Run-to-run determinism of a taped generator forward + backward, op by op: every function of
music_generator_amd.ops is wrapped to record checksums of its tensor arguments after the call; four runs on the same
inputs are compared and the first op whose record differs from the second run's is printed (the first run also
builds descriptors, so it is left out).
    python tools/determinism_trace.py <stage> <batch>        (LEAN=0 in the environment: the full generator tape)
A missing wait state between a conversion and an inline-assembly MFMA in the taped generator-block forward was found
with it: the outputs differed by a few flipped roundings between runs."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import music_generator_amd as M
from music_generator_amd import ops
stage, n = int(sys.argv[1]), int(sys.argv[2])
M.backend.set_precision("bfloat16")
M.clear_session()
g = M.define_generator(stage + 1)
model = g[stage][1]
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
z = torch.randn((n, 1, 100, 2), device="cuda", generator=gen)
dy = torch.randn((n, 4 << stage, 750 << stage, 2), device="cuda", generator=gen)
rec = []
SKIP = {"gblock_fwd_bf16", "gblock_dgrad_bf16"}        # side buffers hold unwritten rows: checksum only their fp32 args
def cs(t):
    if t.dtype in (torch.float32,):
        return (tuple(t.shape), t.double().sum().item(), t.double().abs().sum().item())
    if t.dtype == torch.bfloat16:
        return (tuple(t.shape), "bf16", t.float().double().abs().sum().item())
    return None
def wrap(name, fn):
    def inner(*a, **kw):
        out = fn(*a, **kw)
        tens = [x for x in list(a) + list(kw.values()) if isinstance(x, torch.Tensor)]
        if isinstance(out, torch.Tensor): tens.append(out)
        elif isinstance(out, (tuple, list)): tens += [x for x in out if isinstance(x, torch.Tensor)]
        sums = []
        for t in tens:
            if name.startswith("gblock") and t.dtype != torch.float32: continue
            if t.numel() > 0: sums.append(cs(t))
        if 'desc' not in name: rec[-1].append((name, sums))
        return out
    return inner
for name in dir(ops):
    fn = getattr(ops, name)
    if isinstance(fn, types.FunctionType) and not name.startswith("_") and fn.__module__ == ops.__name__ \
            and name not in ("check", "gblock_side_elems_bf16", "gblock_side_elems_bf16_lean"):
        setattr(ops, name, wrap(name, fn))
M.backend.GBLOCK_LEAN_TAPE = os.environ.get("LEAN", "1") == "1"
for it in range(4):
    rec.append([])
    y, tape = model.forward(z, save=True)
    model.bind_grads().zero_()
    nfwd = len(rec[-1])
    model.backward(tape, dy, need_input_grad=False)
    torch.cuda.synchronize()
    del tape
    if it > 1:
        for i, (a, b) in enumerate(zip(rec[1], rec[it])):
            if a != b:
                print("run", it, "first divergence at record", i, "(forward records:", nfwd, ")")
                for j in range(max(0, i - 3), min(len(rec[1]), i + 2)):
                    print("   ", j, rec[1][j][0], "SAME" if rec[1][j] == rec[it][j] else "DIFF")
                    if rec[1][j] != rec[it][j]:
                        for x, yv in zip(rec[1][j][1], rec[it][j][1]):
                            print("        ", x, yv if x != yv else "")
                break
        else:
            print("run", it, "identical records", len(rec[1]))
