#!/usr/bin/env python3
"""Per-kernel, per-size timing of one full-resolution bf16 train step (HIP events on the launch stream; launches
of one kernel are told apart by their algorithmic FLOPs / bytes, i.e. by block and channel count):
    python tools/c3_breakdown.py [--stage 6] [--batch 8] [--micro-batch 4] [--budget-gib 8]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import music_generator_amd as M
from music_generator_amd import ops
from music_generator_amd.graph import FusedGroup

ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=6)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--micro-batch", type=int, default=4)
ap.add_argument("--budget-gib", type=float, default=8.0)
ap.add_argument("--precision", default="bfloat16")
ap.add_argument("--top", type=int, default=60)
args = ap.parse_args()

M.backend.set_precision(args.precision)
M.clear_session()
d = M.define_discriminator(args.stage + 1)
g = M.define_generator(args.stage + 1)
gan = M.define_composite(d, g, (1, 100, 2))[args.stage][1]
gan.set_train_steps(100)
gan.micro_batch = args.micro_batch or None
FusedGroup.side_budget_bytes = int(args.budget_gib * 2 ** 30) if args.budget_gib else None
h, w = 4 << args.stage, 750 << args.stage
real = torch.rand((args.batch, h, w, 2), device="cuda") * 2 - 1
gan.train_step(real)
torch.cuda.synchronize()


class ByShape(ops.LaunchProfiler):
    def stop(self, key, flops, e0, nbytes=0.0):
        super().stop("%s | %.3g GF %.3g MB" % (key, flops / 1e9, nbytes / 1e6), flops, e0, nbytes)


prof = ByShape()
ops.PROFILER = prof
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
gan.train_step(real)
t1.record()
torch.cuda.synchronize()
ops.PROFILER = None
rows = sorted(prof.summary().items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for _, v in rows)
print("step %.1f ms, timed launches %.1f ms, %d micro-batches" % (t0.elapsed_time(t1), tot, args.batch // (args.micro_batch or args.batch)))
for key, (n, ms, fl, by) in rows[:args.top]:
    print("%8.2f ms %5.1f%% n=%4d %7.1f TF/s %6.2f TB/s  %s" % (ms, 100 * ms / tot, n, fl / ms / 1e9, by / ms / 1e9, key))
