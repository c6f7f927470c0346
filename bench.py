#!/usr/bin/env python3
"""Benchmark of the G+D training step (trainer/models.py:72-158) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1: run directly;
    N > 1: launched by torch.distributed.run, one rank per GPU, data parallel over RCCL)

A "step" is one WGAN.train_step over one batch of synthetic audio that is already resident in
HBM.  Default workload (N = 1): BASELINE.json configs[1] — mid-stage fade-in step, stage 3
(32 x 6000), batch 64 per GPU, fp32.  With N GPUs the global batch is 64*N (weak scaling).

Rank 0 prints ONE JSON line with, beside the contract fields,
  roofline      the dominant kernel of the timed region: algorithmic FLOPs / launch duration,
                measured with HIP events on the launch stream, against the dense f32 MFMA peak;
  cpu_baseline  the CPU oracle's train step (float32, PyTorch-CPU) on a bounded sample of the
                same workload, on this host's cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3      # dense f32 matrix peak of MI355X
# train-step GFLOP per sample (6 G forwards-equivalents + 13 D), stages 0..6
STEP_GFLOP_PER_SAMPLE = [13.6, 59.9, 245.0, 985.0, 3947.0, 15794.0, 63182.0]


def cpu_baseline(stage, fade, sample_batch, n_blocks):
    """Time the oracle's train step on the host (the reference's TensorFlow CPU path cannot run
    here: TensorFlow/Keras are not installed)."""
    import numpy as np
    import torch
    from oracle import ref_step as R
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    W = R.synthetic_weights(n_blocks, seed=100)
    real, z4 = R.synthetic_inputs(stage, sample_batch, seed=0)
    t0 = time.time()
    R.train_step(W, n_blocks, stage, fade, real, z4, alpha=0.5 if fade else None, dtype=torch.float32,
                 keep_grads=False)
    dt = time.time() - t0
    return {"value": sample_batch / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "sample": "one oracle train step (float32, PyTorch-CPU) of the same stage-%d %s workload at batch %d, %.1f s"
                      % (stage, "fade-in" if fade else "straight", sample_batch, dt)}


def measured_traffic(kernel, stage, fade, batch):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc runs of this same command), or None
    when no pass for this workload and kernel is on file."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    wl = table.get("stage%d_%s_b%d" % (stage, "fade" if fade else "straight", batch), {})
    entry = wl.get(kernel)
    if entry:
        return entry["hbm_bytes_per_launch"]
    # a key without template arguments stands for all instantiations of the kernel: launch-weighted mean
    inst = [e for k, e in wl.items() if isinstance(e, dict) and k.startswith(kernel + "<")]
    n = sum(e["launches_per_step"] for e in inst)
    return round(sum(e["hbm_bytes_per_launch"] * e["launches_per_step"] for e in inst) / n) if n else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--straight", action="store_true", help="straight-through twin instead of fade-in")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--reuse-real-logits", action="store_true",
                    help="evaluate discriminator(real) once after the D update instead of three times (same values)")
    ap.add_argument("--cpu-sample-batch", type=int, default=2)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one HIP graph (WGAN.use_graph): for small batches, where the host's "
                         "launch rate bounds the step; no per-kernel roofline in this mode")
    ap.add_argument("--precision", default="float32", choices=["float32", "bfloat16"],
                    help="bfloat16: 32/128-channel discriminator activations stored as bf16, bf16 MFMA, f32 accumulation")
    ap.add_argument("--micro-batch", type=int, default=0, help="samples per micro-batch (multiple of 4; 0 = whole batch)")
    ap.add_argument("--kernels", action="store_true", help="also print the per-kernel time table of the timed steps")
    ap.add_argument("--host-input", action="store_true",
                    help="hand every step a host (numpy) batch, so the host-to-device copy is inside the timed "
                         "region: the PCIe-inclusive rate, not the headline value")
    args = ap.parse_args()

    import numpy as np
    import torch
    import music_generator_amd as M
    from music_generator_amd import ops
    from music_generator_amd.parallel import DataParallel

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
    # one rank per GPU; MGX_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs
    # than ranks (RCCL refuses two ranks on one device)
    backend = os.environ.get("MGX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank)
    dist = DataParallel(backend=backend) if world > 1 else None

    M.backend.set_precision(args.precision)
    stage, fade = args.stage, (not args.straight and args.stage > 0)
    n_blocks = stage + 1
    M.clear_session()
    d = M.define_discriminator(n_blocks)
    g = M.define_generator(n_blocks)
    comp = M.define_composite(d, g, (1, 100, 2))
    gan = comp[stage][1 if fade else 0]
    gan.dist = dist
    gan.reuse_real_logits = args.reuse_real_logits
    if args.micro_batch:
        gan.micro_batch = args.micro_batch
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is a single-GPU mode")
        gan.use_graph = True
        args.warmup = max(args.warmup, 2)        # eager step, then the capturing step
    if dist is not None:
        dist.broadcast_weights(gan.discriminator, gan.generator)
    gan.set_train_steps(100)
    gan.seed(1234 + rank)
    h, w = 4 << stage, 750 << stage
    gen = torch.Generator(device="cuda")
    gen.manual_seed(99 + rank)
    real = torch.rand((args.batch, h, w, 2), device="cuda", generator=gen) * 2.0 - 1.0   # synthetic audio in HBM

    if args.host_input:
        real = real.cpu().numpy()

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        gan.train_step(real)
    sync()
    prof = None if args.graph else ops.LaunchProfiler()
    ops.PROFILER = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = gan.train_step(real)
    sync()
    elapsed = time.perf_counter() - t0
    ops.PROFILER = None
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    last = {k: float(v) for k, v in metrics.items()}
    if dist is not None:
        torch.distributed.destroy_process_group()
    if not all(np.isfinite(list(last.values()))):
        raise SystemExit("non-finite metrics: %r" % (last,))

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    global_batch = args.batch * world
    value = global_batch * args.steps / elapsed
    if prof is None:
        print(json.dumps({"metric": "G+D train-step samples/sec", "value": round(value, 3), "unit": "samples/s",
                          "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "stage-%d (%dx%d) %s G+D train step, batch %d, fp32, HIP graph replay"
                                                 % (stage, h, w, "fade-in" if fade else "straight-through", args.batch)},
                          "step_tflops_algorithmic": round(STEP_GFLOP_PER_SAMPLE[stage] * args.batch / ms_per_step, 2),
                          "last_metrics": last}))
        return
    summ = prof.summary()
    if args.kernels:
        for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1]):
            print("%-40s n=%5d  %9.2f ms  %7.1f TFLOP/s  %7.1f GB/s" % (k, v[0], v[1], v[2] / max(v[1], 1e-9) / 1e9,
                                                                      v[3] / max(v[1], 1e-9) / 1e6), file=sys.stderr)
    # bandwidth-bound kernels (zero FLOPs recorded): achieved GB/s against the 8 TB/s HBM peak
    hbm_ops = {k: {"launches": v[0], "ms": round(v[1], 3), "GB/s": round(v[3] / v[1] / 1e6, 1),
                   "frac_of_8TBps": round(v[3] / v[1] / 1e6 / 8000.0, 3)}
               for k, v in summ.items() if v[2] == 0.0 and v[1] > 0}
    summ = {k: v for k, v in summ.items() if v[2] > 0.0}
    key, (launches, ms, flops, nbytes) = max(summ.items(), key=lambda kv: kv[1][1])
    achieved = flops / (ms * 1e-3) / 1e12
    contraction_ms = sum(v[1] for v in summ.values())
    roofline = {"bound": "mfma", "kernel": key, "achieved": round(achieved, 2), "peak": F32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                "traffic": measured_traffic(key, stage, fade, args.batch),
                "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 2),
                "algorithmic_bytes_per_launch": round(nbytes / launches),
                "launches": launches, "avg_launch_ms": round(ms / launches, 4),
                "share_of_step_time": round(ms / args.steps / ms_per_step, 3),
                "all_contractions_share_of_step_time": round(contraction_ms / args.steps / ms_per_step, 3),
                "step_tflops_algorithmic": round(STEP_GFLOP_PER_SAMPLE[stage] * args.batch / ms_per_step, 2)}
    if key.startswith("wino3_"):
        # the kernel's algorithm is Winograd F(2x2,3x3): `achieved` counts the multiply-adds it executes
        # (16 per 2x2 patch and channel pair); the same launches as a rate of direct-convolution FLOPs:
        roofline["direct_convolution_equivalent_tflops"] = round(achieved / ops.WINOGRAD_FLOP_FRACTION, 2)
    out = {"metric": "G+D train-step samples/sec", "value": round(value, 3), "unit": "samples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic (host-resident batch, copied to HBM inside every step)" if args.host_input else "synthetic",
           "config": {"workload": "stage-%d (%dx%d) %s G+D train step, batch %d per GPU, fp32"
                                  % (stage, h, w, "fade-in" if fade else "straight-through", args.batch),
                      "global_batch": global_batch, "parallelism": "dp%d" % world,
                      "discriminator_real_passes_per_step": 2 if args.reuse_real_logits else 4},
           "roofline": roofline, "hbm_ops": hbm_ops, "last_metrics": last}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(stage, fade, args.cpu_sample_batch, n_blocks)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
