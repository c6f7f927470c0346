#!/usr/bin/env python3
"""Per-stage G+D train-step throughput on one MI355X (fp32): the "samples/sec per progressive
stage" table.  For each stage it runs `steps` timed steps of the straight-through twin (fade-in
with --fade) on synthetic audio resident in HBM and prints one JSON line per stage.

    python tools/stage_sweep.py [--stages 0,1,2,3,4,5] [--steps 2] [--fade]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import music_generator_amd as M

STEP_GFLOP = [13.6, 59.9, 245.0, 985.0, 3947.0, 15794.0, 63182.0]
# (batch, micro_batch) per stage: large batches at low resolution, memory-bounded micro-batches above
PLAN = {0: (256, None), 1: (256, None), 2: (128, None), 3: (64, None), 4: (32, 16), 5: (16, 4), 6: (4, 4)}
# generator backward chunk (samples) and side-buffer budget (bytes) of the fused generator blocks:
# only where a micro-batch of generator activations does not fit
GEN_CHUNK = {}
SIDE_BUDGET = {6: 8 << 30}

ap = argparse.ArgumentParser()
ap.add_argument("--stages", default="0,1,2,3,4,5")
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--fade", action="store_true")
ap.add_argument("--precision", default="float32")
ap.add_argument("--batch", type=int, default=0, help="override the batch of every listed stage")
ap.add_argument("--micro-batch", type=int, default=0)
args = ap.parse_args()
M.backend.set_precision(args.precision)
for stage in [int(s) for s in args.stages.split(",")]:
    batch, mb = PLAN[stage]
    if args.batch:
        batch = args.batch
    if args.micro_batch:
        mb = args.micro_batch
    M.clear_session()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats()
    d = M.define_discriminator(stage + 1)
    g = M.define_generator(stage + 1)
    fade = args.fade and stage > 0
    gan = M.define_composite(d, g, (1, 100, 2))[stage][1 if fade else 0]
    gan.micro_batch = mb
    gan.generator_chunk = GEN_CHUNK.get(stage)
    from music_generator_amd.graph import FusedGroup
    FusedGroup.side_budget_bytes = SIDE_BUDGET.get(stage)
    gan.set_train_steps(100)
    h, w = 4 << stage, 750 << stage
    real = torch.rand((batch, h, w, 2), device="cuda") * 2 - 1
    gan.train_step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        m = gan.train_step(real)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"stage": stage, "hw": [h, w], "batch": batch, "micro_batch": mb, "generator_chunk": GEN_CHUNK.get(stage),
                      "side_budget_gib": (SIDE_BUDGET[stage] >> 30) if stage in SIDE_BUDGET else None,
                      "fade_in": fade, "precision": args.precision,
                      "ms_per_step": round(dt * 1e3, 2), "samples_per_s": round(batch / dt, 3),
                      "algorithmic_tflops": round(STEP_GFLOP[stage] * batch / dt / 1e3, 2),
                      "frac_f32_mfma_peak": round(STEP_GFLOP[stage] * batch / dt / 1e3 / 157.3, 4),
                      "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                      "peak_reserved_gib": round(torch.cuda.max_memory_reserved() / 2**30, 1),
                      "alloc_retries": torch.cuda.memory_stats().get("num_alloc_retries", 0),
                      "delta_1": float(m["delta_1"])}), flush=True)
    del gan, d, g, real
