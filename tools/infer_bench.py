#!/usr/bin/env python3
"""Generator-only inference throughput (clips/s) at a progressive stage on one MI355X: batches of
latent noise -> audio tensors through the fused grow-block forward, outputs left in HBM and
overwritten batch after batch (a 4096-clip run at the final stage is 400 GB of audio).

    python tools/infer_bench.py [--stage 6] [--batch 8] [--clips 64]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import music_generator_amd as M

G_GFLOP_PER_CLIP = [0.29, 1.18, 4.74, 18.97, 75.9, 303.6, 1214.4]   # generator forward, stages 0..6 (60.6 at stage 3 / 3.2)

ap = argparse.ArgumentParser()
ap.add_argument("--stage", type=int, default=6)
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--clips", type=int, default=64)
args = ap.parse_args()
M.clear_session()
g = M.define_generator(args.stage + 1)[args.stage][0]
z = torch.randn((args.batch, 1, 100, 2), device="cuda")
y = g(z)
torch.cuda.synchronize()
n_batches = max(1, args.clips // args.batch)
t0 = time.perf_counter()
for _ in range(n_batches):
    z.normal_()
    y = g(z)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
clips = n_batches * args.batch
h, w = 4 << args.stage, 750 << args.stage
print(json.dumps({"metric": "generator inference clips/s", "stage": args.stage, "hw": [h, w], "batch": args.batch,
                  "clips": clips, "clips_per_s": round(clips / dt, 3), "ms_per_clip": round(dt / clips * 1e3, 3),
                  "seconds_for_4096_clips": round(4096 * dt / clips, 1),
                  "finite": bool(torch.isfinite(y).all().item()),
                  "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
