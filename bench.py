#!/usr/bin/env python3
"""Benchmark of the G+D training step (trainer/models.py:72-158) on MI355X.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without WORLD_SIZE in the environment starts its own N ranks
(`python -m torch.distributed.run`, one per GPU, RCCL) before anything touches the GPU and relays
rank 0's line; under an external torch.distributed.run it is a rank itself.

A "step" is one WGAN.train_step over one batch of synthetic audio that is already resident in HBM.
Timed workload (`value`): BASELINE.json configs[1] — mid-stage fade-in step, stage 3 (32 x 6000),
batch 64 per GPU, fp32; with N GPUs the global batch is 64*N (weak scaling, batch sharded by
MinibatchStdDev groups, gradient all-reduce over RCCL).

Rank 0 prints ONE JSON line with, beside the contract fields,
  roofline         the dominant kernel of the timed region: algorithmic FLOPs / launch duration, measured
                   with HIP events on the launch stream, against the dense MFMA peak of its dtype;
  cpu_baseline     the CPU oracle's train step (float32, PyTorch-CPU) on a bounded sample of the same
                   workload, on this host's cores (N = 1 only); cpu_baseline_c1: BASELINE configs[0]
                   (stage 0, batch 16) on the CPU, whole batch;
  per_stage        (N = 1) samples/s of the fade-in step at every progressive stage 0-6, fp32 and bf16;
  c3               (N = 1) BASELINE configs[2]: the full-resolution stage (256 x 48000), batch 256 in
                   micro-batches of one MinibatchStdDev group, bf16: one timed step after a warm-up on 8
                   samples, with the roofline of its dominant kernel against the bf16 dense MFMA peak;
  c5               (N = 1) BASELINE configs[4]: generator-only inference at the final stage through
                   music_generator_amd.inference.generate_clips, clips/s in fp32 and bf16.

    python bench.py --infer [--clips 4096] [--chunk 8] [--precision bfloat16]
prints the inference line alone (metric "generator inference clips/s"; a step is one chunk).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3      # dense f32 matrix peak of MI355X
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 matrix peak (no sparsity)
# train-step GFLOP per sample (6 G forwards-equivalents + 13 D), stages 0..6
STEP_GFLOP_PER_SAMPLE = [13.6, 59.9, 245.0, 985.0, 3947.0, 15794.0, 63182.0]
# (batch, micro-batch) of the per-stage table: large batches at low resolution, memory-bounded above
STAGE_PLAN = {0: (256, None), 1: (256, None), 2: (128, None), 3: (64, None), 4: (32, 16), 5: (16, 4), 6: (4, 4)}
# bf16 tapes are half the size: the same batches run whole (measured 5 % / 11 % faster than in micro-batches)
STAGE_MICRO_BATCH_BF16 = {4: None, 5: None}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--straight", action="store_true", help="straight-through twin instead of fade-in")
    ap.add_argument("--precision", default="float32", choices=["float32", "bfloat16"],
                    help="bfloat16: 32/128-channel discriminator activations and the generator blocks on bf16 MFMAs "
                         "with f32 accumulation")
    ap.add_argument("--micro-batch", type=int, default=0, help="samples per micro-batch (multiple of 4; 0 = whole batch)")
    ap.add_argument("--side-budget-gib", type=float, default=0.0,
                    help="bound of the generator block's tape in GiB (re-computes above it; 0 = unbounded)")
    ap.add_argument("--kernels", action="store_true", help="also print the per-kernel time table of the timed steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip per_stage and c3 (N = 1 extras)")
    ap.add_argument("--reuse-real-logits", action="store_true",
                    help="evaluate discriminator(real) once after the D update instead of three times (same values)")
    ap.add_argument("--cpu-sample-batch", type=int, default=2)
    ap.add_argument("--c3-batch", type=int, default=256)
    ap.add_argument("--c3-steps", type=int, default=1, help="timed full-resolution steps of the c3 / c4 objects")
    ap.add_argument("--c5-clips", type=int, default=4096, help="clips of the bf16 leg of the c5 object (the float32 leg "
                                                               "is a 64-clip sample)")
    ap.add_argument("--infer", action="store_true", help="generator-only inference line (BASELINE configs[4])")
    ap.add_argument("--clips", type=int, default=256, help="--infer: clips to generate in the timed region")
    ap.add_argument("--chunk", type=int, default=8, help="--infer: clips per generator call")
    ap.add_argument("--collapsed", action="store_true",
                    help="--infer: also time the opt-in collapsed generator (labelled, float32, chunks of 64)")
    ap.add_argument("--infer-stage", type=int, default=6)
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one HIP graph (WGAN.use_graph): for small batches, where the host's "
                         "launch rate bounds the step; no per-kernel roofline in this mode")
    ap.add_argument("--host-input", action="store_true",
                    help="hand every step a host (numpy) batch, so the host-to-device copy is inside the timed "
                         "region: the PCIe-inclusive rate, not the headline value")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: start N ranks as a child
    torch.distributed.run (this process has not touched the GPU) and exit with its status."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def cpu_baseline(stage, fade, sample_batch, n_blocks, timed=3):
    """Median of `timed` oracle train steps after one warm-up, on the host (the reference's TensorFlow CPU
    path cannot run here: TensorFlow/Keras are not installed)."""
    import torch
    from oracle import ref_step as R
    # the cores this process may use: its affinity mask, or the thread count the box's operator set
    # (os.cpu_count() is the whole machine, of which a one-GPU lease is granted a share)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if os.environ.get("OMP_NUM_THREADS", "").isdigit():
        cores = min(cores, int(os.environ["OMP_NUM_THREADS"]))
    torch.set_num_threads(cores)
    W = R.synthetic_weights(n_blocks, seed=100)
    real, z4 = R.synthetic_inputs(stage, sample_batch, seed=0)
    times = []
    for i in range(timed + 1):
        t0 = time.time()
        R.train_step(W, n_blocks, stage, fade, real, z4, alpha=0.5 if fade else None, dtype=torch.float32,
                     keep_grads=False)
        if i:
            times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(sample_batch / dt, 4), "unit": "samples/s", "cores": cores,
            "host_cpu_count": os.cpu_count(), "kind": "port",
            "sample": "oracle train step (float32, PyTorch-CPU) of the stage-%d %s workload at batch %d: 1 warm-up, "
                      "median of %d steps, %.2f s each" % (stage, "fade-in" if fade else "straight", sample_batch,
                                                          timed, dt)}


def _norm(name):
    return name.replace(", ", ",")


def _is_instance(table_key, kernel):
    """`kernel` is a profiler key (`wino3_kernel`, `bconv_kernel<3,3,1>`); the tables hold full template
    instances (`wino3_kernel<1,2,false,32>`, `bconv_kernel<3,3,1,16,1>`)."""
    tk = _norm(table_key)
    return tk == kernel or tk.startswith(kernel + "<") or (kernel.endswith(">") and tk.startswith(kernel[:-1] + ","))


# A launch-profiler key that stands for one library call issuing several kernels: the kernels it covers
KERNEL_FAMILIES = {
    "gblock_wgrad_bf16_kernel<*>": ("gblock_lean_branch_kernel", "gblock_lean_tail_kernel", "gblock_wgrad_bf16_kernel",
                                    "gblock_unpack_kernel"),
}


def _members(names, kernel):
    fam = KERNEL_FAMILIES.get(kernel)
    if fam is None:
        return [k for k in names if _is_instance(k, kernel)]
    return [k for k in names if any(_is_instance(k, m) for m in fam)]


def measured_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/pmc_traffic.json:
    2 x FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc runs of the same workload), or None
    when no pass for this workload and kernel is on file."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            table = json.load(f)
    except (OSError, ValueError):
        return None
    wl = table.get(workload, {})
    entry = wl.get(kernel)
    if entry:
        return entry["hbm_bytes_per_launch"]
    inst = [wl[k] for k in _members([k for k, e in wl.items() if isinstance(e, dict)], kernel)]
    # a family: bytes of all its kernels per library call (every member is launched once or more per call)
    n = max((e["launches_per_step"] for e in inst), default=0) if kernel in KERNEL_FAMILIES else \
        sum(e["launches_per_step"] for e in inst)
    return round(sum(e["hbm_bytes_per_launch"] * e["launches_per_step"] for e in inst) / n) if n else None


def measured_mfma_busy(kernel, digest):
    """Matrix-unit occupancy of `kernel` from the committed SQ counter pass of the same workload
    (profiles/<digest>_digest.json: SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs),
    launch-weighted over the kernel's template instances), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", digest + "_digest.json")) as f:
            rows = json.load(f)["kernels"]
    except (OSError, ValueError, KeyError):
        return None
    names = set(_members([r["kernel"] for r in rows], kernel))
    inst = [r for r in rows if "mfma_busy_frac" in r and r["kernel"] in names]
    t = sum(r["calls"] * r["avg_us"] for r in inst)
    return round(sum(r["mfma_busy_frac"] * r["calls"] * r["avg_us"] for r in inst) / t, 4) if t else None


def build_gan(M, stage, fade, precision, micro_batch=None, side_budget_gib=0.0):
    from music_generator_amd.graph import FusedGroup
    M.backend.set_precision(precision)
    M.clear_session()
    d = M.define_discriminator(stage + 1)
    g = M.define_generator(stage + 1)
    gan = M.define_composite(d, g, (1, 100, 2))[stage][1 if fade else 0]
    gan.micro_batch = micro_batch or None
    FusedGroup.side_budget_bytes = int(side_budget_gib * 2 ** 30) if side_budget_gib else None
    gan.set_train_steps(100)
    return gan


HBM_PEAK_GBPS = 8000.0


def roofline_of(summ, steps, ms_per_step, peak, traffic=None):
    """The kernel family with the largest share of the timed region, priced against the roof that bounds it: the
    dense MFMA peak of its dtype, or, when its algorithmic FLOPs per byte lie below the ridge (peak / 8 TB/s), the
    HBM peak; the other roof's figures ride along."""
    contr = {k: v for k, v in summ.items() if v[2] > 0.0}
    key, (launches, ms, flops, nbytes) = max(contr.items(), key=lambda kv: kv[1][1])
    achieved = flops / (ms * 1e-3) / 1e12
    gbps = nbytes / ms / 1e6
    ridge = peak * 1e12 / (HBM_PEAK_GBPS * 1e9)
    intensity = flops / nbytes if nbytes else float("inf")
    if intensity < ridge:
        head = {"bound": "hbm", "kernel": key, "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(gbps / HBM_PEAK_GBPS, 4), "traffic": traffic,
                "algorithmic_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                "mfma_TFLOPs": round(achieved, 2), "mfma_peak_TFLOPs": peak, "mfma_frac": round(achieved / peak, 4)}
    else:
        head = {"bound": "mfma", "kernel": key, "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "algorithmic_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1)}
    head.update({
                 "algorithmic_gflop_per_launch": round(flops / launches / 1e9, 2),
                 "algorithmic_bytes_per_launch": round(nbytes / launches),
                 "algorithmic_GBps": round(nbytes / ms / 1e6, 1),
                 "launches": launches, "avg_launch_ms": round(ms / launches, 4),
                 "share_of_step_time": round(ms / steps / ms_per_step, 3),
                 "all_contractions_share_of_step_time": round(sum(v[1] for v in contr.values()) / steps / ms_per_step, 3)})
    return key, head


def progress(msg):
    """One line on stderr: long phases stay visibly alive for whoever watches the run."""
    print("[bench %6.1f s] %s" % (time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.perf_counter()


C3_DIGEST = next((d for d in ("r3e_c3", "r3_c3", "r2_c3") if os.path.exists(os.path.join(ROOT, "profiles", d + "_digest.json"))), "r2_c3")
C2_DIGEST = "r3_c2" if os.path.exists(os.path.join(ROOT, "profiles", "r3_c2_digest.json")) else "r2_c2"
C3_MICRO_BATCH = 4          # samples per micro-batch of the full-resolution step (whole MinibatchStdDev groups)


def per_stage_table(torch, M):
    """2 timed fade-in steps (after one warm-up) per stage and precision on one GPU."""
    out = []
    for stage in range(7):
        batch, mb = STAGE_PLAN[stage]
        row = {"stage": stage, "hw": [4 << stage, 750 << stage], "batch": batch, "micro_batch": mb}
        for precision in ("float32", "bfloat16"):
            torch.cuda.empty_cache()
            if precision == "bfloat16" and stage in STAGE_MICRO_BATCH_BF16:
                mb = STAGE_MICRO_BATCH_BF16[stage]
                row["micro_batch_bf16"] = mb
            # generator tape bound: the full-resolution stage, and the whole bf16 batch of stage 5 (its discriminator
            # tape leaves no room for a kept generator tape)
            bound = stage == 6 or (stage == 5 and precision == "bfloat16")
            gan = build_gan(M, stage, stage > 0, precision, mb, 8.0 if bound else 0.0)
            real = torch.rand((batch, 4 << stage, 750 << stage, 2), device="cuda") * 2 - 1
            gan.train_step(real)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(2):
                gan.train_step(real)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 2
            tag = "f32" if precision == "float32" else "bf16"
            row["samples_per_s_" + tag] = round(batch / dt, 2)
            row["tflops_algorithmic_" + tag] = round(STEP_GFLOP_PER_SAMPLE[stage] * batch / dt / 1e3, 1)
            progress("stage %d %s: %.1f samples/s" % (stage, tag, batch / dt))
            del gan, real
        out.append(row)
    return out


def c3_run(torch, M, ops, batch, micro_batch=C3_MICRO_BATCH, steps=1, dist=None, stage=6, rank=0):
    """BASELINE configs[2]: stage 6 (256 x 48000), batch `batch`, bf16, micro-batches of one MinibatchStdDev
    group (4 samples), generator tape kept or bounded as the executor decides from the free memory (at this size it
    keeps the lean tape: no re-computing forward).  Warm-up: one step on 8
    samples (same micro-batch shapes); then `steps` timed steps on the whole batch.
    With `dist` (a DataParallel over more than one rank) this is BASELINE configs[3]: every rank steps on
    `batch` samples of its own with the gradient all-reduces of the data-parallel step, and the time is the
    maximum over ranks."""
    torch.cuda.empty_cache()
    tag = "c4" if dist is not None else "c3"
    gan = build_gan(M, stage, True, "bfloat16", micro_batch, 0.0)
    h, w = 4 << stage, 750 << stage
    if dist is not None:
        gan.dist = dist
        dist.broadcast_weights(gan.discriminator, gan.generator)
        dist.record_waits = True
    gan.seed(4321 + rank)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7 + rank)
    real = torch.rand((batch, h, w, 2), device="cuda", generator=gen).mul_(2.0).sub_(1.0)
    warm = min(batch, max(8, micro_batch))

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        progress("%s: batch of %d clips resident, warm-up step on %d samples" % (tag, batch, warm))
    gan.train_step(real[:warm])
    sync()
    if dist is not None:
        dist.take_wait_stall_ms()
    torch.cuda.reset_peak_memory_stats()
    if rank == 0:
        progress("%s: %d timed step(s) on %d samples (about 40 s each)" % (tag, steps, batch))
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    t0 = time.perf_counter()
    for _ in range(steps):
        metrics = gan.train_step(real)
    sync()
    dt = (time.perf_counter() - t0) / steps
    ops.PROFILER = None
    comm = None
    if dist is not None:
        t = torch.tensor([dt], device="cuda" if torch.distributed.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        comm = collective_report(torch, gan, dist, steps)
    if rank == 0:
        progress("%s: %.1f s per step, %.2f samples/s per GPU" % (tag, dt, batch / dt))
    summ = prof.summary()
    key, roof = roofline_of(summ, steps, dt * 1e3, BF16_MFMA_PEAK_TFLOPS)
    # counter passes of the same step at batch 16 (the launches are those of the batch-256 step)
    roof["traffic"] = measured_traffic(key, "stage6_fade_mb4_bf16")
    roof["mfma_busy_frac_sq_counters"] = measured_mfma_busy(key, C3_DIGEST)
    top = sorted(summ.items(), key=lambda kv: -kv[1][1])[:8]
    world = dist.world_size if dist is not None else 1
    out = {"workload": "stage-%d (%dx%d) fade-in G+D train step, batch %d per GPU (micro-batches of %d), bf16 storage / "
                       "bf16 MFMA / f32 accumulation, f32 master weights" % (stage, h, w, batch, micro_batch),
           "value": round(world * batch / dt, 3), "unit": "samples/s", "steps": steps,
           "warmup": "1 step on %d samples" % warm,
           "ms_per_step": round(dt * 1e3, 1), "dtype": "bf16",
           "step_tflops_algorithmic": round(world * STEP_GFLOP_PER_SAMPLE[stage] * batch / dt / 1e3, 1),
           "frac_of_bf16_mfma_peak": round(STEP_GFLOP_PER_SAMPLE[stage] * batch / dt / 1e3 / BF16_MFMA_PEAK_TFLOPS, 4),
           "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
           "roofline": roof,
           "top_kernels": {k: {"ms": round(v[1] / steps, 1), "TFLOP/s": round(v[2] / v[1] / 1e9, 1),
                               "GB/s": round(v[3] / v[1] / 1e6, 1)} for k, v in top},
           "last_metrics": {k: float(v) for k, v in metrics.items()}}
    if dist is not None:
        out.update({"n_gpus": world, "global_batch": world * batch, "rccl_ranks": torch.distributed.get_world_size(),
                    "collective_backend": torch.distributed.get_backend(), "collectives": comm})
    del gan, real
    torch.cuda.empty_cache()
    M.backend.set_precision("float32")
    return out


def collective_report(torch, gan, dist, steps):
    """What the gradient all-reduces of the timed data-parallel steps cost: bytes per step, the time the compute
    stream stood still in front of mgx_combine_grads / Adam waiting for them (events around every wait), and the
    same three buffers reduced once more on an otherwise idle GPU for comparison; the share of the stand-alone
    time that did NOT show up as a stall is the overlap achieved."""
    stall_ms = dist.take_wait_stall_ms() / steps
    bufs = [gan.discriminator.bind_grads(0), gan.discriminator.bind_grads(1), gan.generator.bind_grads(0)]
    gan.discriminator.bind_grads(0)
    nbytes = sum(b.numel() * 4 for b in bufs)
    alone_ms = None
    if torch.distributed.get_backend() == "nccl":
        scratch = [torch.zeros_like(b) for b in bufs]
        for t in scratch:
            torch.distributed.all_reduce(t)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in scratch:
            torch.distributed.all_reduce(t)
        e1.record()
        torch.cuda.synchronize()
        alone_ms = e0.elapsed_time(e1)
    return {"allreduce_bytes_per_step": nbytes, "buffers": {"discriminator_fake_pass": bufs[0].numel() * 4,
                                                          "discriminator_real_pass": bufs[1].numel() * 4,
                                                          "generator": bufs[2].numel() * 4},
            "logit_sum_allreduces_per_step": 4,
            "compute_stream_stall_ms_per_step": round(stall_ms, 3),
            "same_allreduces_alone_ms": None if alone_ms is None else round(alone_ms, 3),
            "overlap_achieved": None if not alone_ms else round(max(0.0, 1.0 - stall_ms / alone_ms), 3)}


def c1_run(torch, M, steps=100):
    """BASELINE configs[0] on the GPU: stage 0 (4 x 750) straight-through step at the reference's batch of 16,
    float32, the way train_step runs it by default at this size (one HIP graph replayed per step)."""
    torch.cuda.empty_cache()
    gan = build_gan(M, 0, False, "float32")
    gan.seed(5)
    real = torch.rand((16, 4, 750, 2), device="cuda") * 2.0 - 1.0
    for _ in range(4):
        gan.train_step(real)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        metrics = gan.train_step(real)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    replayed = any(e.get("graph") is not None for e in gan._graphs.values())
    return {"workload": "stage-0 (4x750) straight-through G+D train step, batch 16, fp32%s"
                        % (", HIP graph replay" if replayed else ""),
            "value": round(16 / dt, 1), "unit": "samples/s", "steps": steps, "ms_per_step": round(dt * 1e3, 3),
            "dtype": "f32", "step_tflops_algorithmic": round(STEP_GFLOP_PER_SAMPLE[0] * 16 / dt / 1e3, 2),
            "last_metrics": {k: float(v) for k, v in metrics.items()}}


def infer_run(torch, M, ops, precision, stage, clips, chunk):
    """BASELINE configs[4]: `clips` latent vectors -> audio through generate_clips in chunks, every chunk
    reduced on the device to per-clip checksums (a 4096-clip run at the final stage is 403 GB of float32
    audio: it is consumed chunk by chunk, never resident).  One warm-up chunk, then the timed run."""
    from music_generator_amd.inference import ChecksumSink, generate_clips
    M.backend.set_precision(precision)
    M.clear_session()
    torch.cuda.empty_cache()
    g = M.define_generator(stage + 1)[stage][0]
    generate_clips(g, chunk, chunk, ChecksumSink(), seed=1)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    sink = ChecksumSink()
    prof = ops.LaunchProfiler()
    ops.PROFILER = prof
    t0 = time.perf_counter()
    n = generate_clips(g, clips, chunk, sink, seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.PROFILER = None
    chk = sink.result()
    h, w = 4 << stage, 750 << stage
    summ = prof.summary()
    peak = BF16_MFMA_PEAK_TFLOPS if precision == "bfloat16" else F32_MFMA_PEAK_TFLOPS
    nchunks = (clips + chunk - 1) // chunk
    _, roof = roofline_of(summ, nchunks, dt * 1e3 / nchunks, peak)
    out = {"metric": "generator inference clips/s", "value": round(n / dt, 3), "unit": "clips/s", "n_gpus": 1,
           "steps": nchunks, "warmup": 1, "ms_per_step": round(dt / nchunks * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32" if precision == "float32" else "bf16",
           "data": "synthetic (standard-normal latent vectors)",
           "config": {"workload": "generator-only inference, stage %d (%dx%d), %d clips in chunks of %d, audio reduced "
                                  "to per-clip checksums on the device" % (stage, h, w, clips, chunk)},
           "clips": n, "seconds": round(dt, 2),
           "seconds_for_4096_clips": round(4096 * dt / n, 1),
           "seconds_for_4096_clips_is": "measured" if n == 4096 else "extrapolated from %d clips" % n,
           "gflop_per_clip": round(sum(v[2] for v in summ.values()) / n / 1e9, 1),
           "tflops_algorithmic": round(sum(v[2] for v in summ.values()) / dt / 1e12, 1),
           "finite": bool(np_isfinite(chk)), "clip_mean_abs_max": float(abs(chk[:, 0]).max() / (h * w * 2)),
           "clip_mean_square_range": [float(chk[:, 1].min() / (h * w * 2)), float(chk[:, 1].max() / (h * w * 2))],
           "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
           "roofline": roof,
           "hbm_ops": {k: {"launches": v[0], "ms": round(v[1], 3), "bytes_per_launch": round(v[3] / v[0]),
                           "GB/s": round(v[3] / v[1] / 1e6, 1), "frac_of_8TBps": round(v[3] / v[1] / 1e6 / 8000.0, 3)}
                       for k, v in summ.items() if v[2] == 0.0 and v[1] > 0}}
    del g
    torch.cuda.empty_cache()
    M.backend.set_precision("float32")
    return out


def collapsed_infer_run(torch, M, ops, stage, clips, chunk):
    """The labelled inference optimisation next to BASELINE configs[4] (not the reference's computation order and
    no part of any MFMA figure): the generator's layers in front of its LayerNormalization are all linear, so
    inference.CollapsedGenerator measures them once with five probe latents and then writes every clip directly.
    Same sink and latents as infer_run; the table build is timed separately."""
    from music_generator_amd.inference import ChecksumSink, CollapsedGenerator, generate_clips
    M.backend.set_precision("float32")
    M.clear_session()
    torch.cuda.empty_cache()
    g = M.define_generator(stage + 1)[stage][0]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cg = CollapsedGenerator(g)
    torch.cuda.synchronize()
    build_ms = (time.perf_counter() - t0) * 1e3
    generate_clips(cg, chunk, chunk, ChecksumSink(), seed=1)
    torch.cuda.synchronize()
    sink = ChecksumSink()
    t0 = time.perf_counter()
    n = generate_clips(cg, clips, chunk, sink, seed=2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    chk = sink.result()
    h, w = 4 << stage, 750 << stage
    clip_bytes = h * w * 2 * 4
    return {"value": round(n / dt, 1), "unit": "clips/s", "clips": n, "chunk": chunk,
            "seconds_for_4096_clips": round(4096 * dt / n, 3), "table_build_ms": round(build_ms, 1),
            "hbm_GBps_written_and_checksummed": round(2 * n * clip_bytes / dt / 1e9, 1),
            "frac_of_8TBps": round(2 * n * clip_bytes / dt / 8e12, 3), "finite": bool(np_isfinite(chk)),
            "clip_mean_square_range": [float(chk[:, 1].min() / (h * w * 2)), float(chk[:, 1].max() / (h * w * 2))],
            "note": "opt-in algebraic collapse of the all-linear generator (inference.CollapsedGenerator, float32): "
                    "each clip is written once from a 5-vector table and read once by the checksum sink; equals the "
                    "layer path to float32 rounding (tests/test_gpu_inference.py); not the reference's computation "
                    "order, no MFMA work, not the c5 headline"}


def np_isfinite(a):
    import numpy as np
    return np.isfinite(a).all()


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch
    import music_generator_amd as M
    from music_generator_amd import ops
    from music_generator_amd.parallel import DataParallel

    if args.infer:
        if world > 1:
            raise SystemExit("--infer is a single-GPU line (clips are independent: run one process per GPU)")
        torch.cuda.set_device(0)
        r = infer_run(torch, M, ops, args.precision, args.infer_stage, args.clips, args.chunk)
        if args.collapsed:
            r["collapsed_f32"] = collapsed_infer_run(torch, M, ops, args.infer_stage, max(args.clips, 64), 64)
        print(json.dumps(r))
        return

    # one rank per GPU; MGX_BENCH_BACKEND=gloo rehearses the multi-rank flow on a box with fewer GPUs
    # than ranks (RCCL refuses two ranks on one device)
    backend = os.environ.get("MGX_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank)
    dist = DataParallel(backend=backend) if world > 1 else None

    stage, fade = args.stage, (not args.straight and args.stage > 0)
    n_blocks = stage + 1
    gan = build_gan(M, stage, fade, args.precision, args.micro_batch, args.side_budget_gib)
    gan.dist = dist
    gan.reuse_real_logits = args.reuse_real_logits
    if args.graph:
        if world > 1:
            raise SystemExit("--graph is a single-GPU mode")
        gan.use_graph = True
        args.warmup = max(args.warmup, 2)        # eager step, then the capturing step
    trace = (lambda msg: print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)) \
        if os.environ.get("MGX_BENCH_TRACE") else (lambda msg: None)
    trace("models built")
    if dist is not None:
        dist.broadcast_weights(gan.discriminator, gan.generator)
    trace("weights broadcast")
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
        trace("warm-up step enqueued")
    sync()
    trace("warm-up done")
    prof = None if args.graph else ops.LaunchProfiler()
    ops.PROFILER = prof
    t0 = time.perf_counter()
    for _ in range(args.steps):
        metrics = gan.train_step(real)
    sync()
    elapsed = time.perf_counter() - t0
    ops.PROFILER = None
    backend_name = torch.distributed.get_backend() if dist is not None else None
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    last = {k: float(v) for k, v in metrics.items()}
    ranks = torch.distributed.get_world_size() if dist is not None else 1
    c4 = None
    if dist is not None and not args.no_extras and not args.graph:
        # BASELINE configs[3]: the full-resolution bf16 step, batch 256 per GPU, data-parallel over all ranks
        # (MGX_BENCH_C4="stage,batch,micro-batch" shrinks it for rehearsals on a box with fewer GPUs than ranks)
        c4_stage, c4_batch, c4_mb = (int(v) for v in os.environ.get("MGX_BENCH_C4", "6,%d,%d" % (args.c3_batch, C3_MICRO_BATCH)).split(","))
        del gan, real
        c4 = c3_run(torch, M, ops, c4_batch, c4_mb, steps=args.c3_steps, dist=dist, stage=c4_stage, rank=rank)
        gan = real = None
    if dist is not None:
        torch.distributed.destroy_process_group()
    if not all(np.isfinite(list(last.values()))):
        raise SystemExit("non-finite metrics: %r" % (last,))
    if rank != 0:
        return

    ms_per_step = elapsed / args.steps * 1e3
    progress("timed region: %.1f ms per step" % ms_per_step)
    global_batch = args.batch * world
    value = global_batch * args.steps / elapsed
    dtype = "f32" if args.precision == "float32" else "bf16"
    pname = "fp32" if args.precision == "float32" else "bf16 storage + bf16 MFMA, f32 accumulation"
    out = {"metric": "G+D train-step samples/sec", "value": round(value, 3), "unit": "samples/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype,
           "data": "synthetic (host-resident batch, copied to HBM inside every step)" if args.host_input else "synthetic",
           "config": {"workload": "stage-%d (%dx%d) %s G+D train step, batch %d per GPU, %s%s"
                                  % (stage, h, w, "fade-in" if fade else "straight-through", args.batch, pname,
                                     ", HIP graph replay" if args.graph else ""),
                      "global_batch": global_batch, "parallelism": "dp%d" % world,
                      "discriminator_real_passes_per_step": 2 if args.reuse_real_logits else 4},
           "step_tflops_algorithmic": round(STEP_GFLOP_PER_SAMPLE[stage] * args.batch / ms_per_step, 2),
           "last_metrics": last}
    if world > 1:
        out["rccl_ranks"] = ranks
        out["collective_backend"] = backend_name
        if c4 is not None:
            out["c4"] = c4
    if prof is not None:
        summ = prof.summary()
        if args.kernels:
            for k, v in sorted(summ.items(), key=lambda kv: -kv[1][1]):
                print("%-40s n=%5d  %9.2f ms  %7.1f TFLOP/s  %7.1f GB/s" % (k, v[0], v[1], v[2] / max(v[1], 1e-9) / 1e9,
                                                                          v[3] / max(v[1], 1e-9) / 1e6), file=sys.stderr)
        # bandwidth-bound kernels (zero FLOPs recorded): achieved GB/s against the 8 TB/s HBM peak
        out["hbm_ops"] = {k: {"launches": v[0], "ms": round(v[1], 3), "GB/s": round(v[3] / v[1] / 1e6, 1),
                              "frac_of_8TBps": round(v[3] / v[1] / 1e6 / 8000.0, 3)}
                          for k, v in summ.items() if v[2] == 0.0 and v[1] > 0}
        key, roof = roofline_of(summ, args.steps, ms_per_step,
                                BF16_MFMA_PEAK_TFLOPS if args.precision == "bfloat16" else F32_MFMA_PEAK_TFLOPS)
        wl = "stage%d_%s_b%d%s" % (stage, "fade" if fade else "straight", args.batch,
                                   "_bf16" if args.precision == "bfloat16" else "")
        roof["traffic"] = measured_traffic(key, wl)
        if (stage, fade, args.batch) == (3, True, 64):
            roof["mfma_busy_frac_sq_counters"] = measured_mfma_busy(key, "r3d_c2bf16" if args.precision == "bfloat16"
                                                                    else C2_DIGEST)
        if key.startswith("wino3_") and roof["bound"] == "mfma":
            # the kernel's algorithm is Winograd F(2x2,3x3): `achieved` counts the multiply-adds it executes
            # (16 per 2x2 patch and channel pair); the same launches as a rate of direct-convolution FLOPs:
            roof["direct_convolution_equivalent_tflops"] = round(roof["achieved"] / ops.WINOGRAD_FLOP_FRACTION, 2)
        out["roofline"] = roof
    if world == 1 and not args.graph:
        del gan, real
        gan = real = None
        if not args.no_extras:
            out["per_stage"] = per_stage_table(torch, M)
            out["c1"] = c1_run(torch, M)
            progress("c1: %.2f ms per step" % out["c1"]["ms_per_step"])
            out["c3"] = c3_run(torch, M, ops, args.c3_batch, steps=args.c3_steps)
            out["c5"] = {}
            for precision in ("float32", "bfloat16"):
                # bf16: the whole 4096-clip run of BASELINE configs[4]; float32: a 64-clip sample of it
                r = infer_run(torch, M, ops, precision, 6, 64 if precision == "float32" else args.c5_clips, 8)
                out["c5"]["f32" if precision == "float32" else "bf16"] = {
                    k: r[k] for k in ("value", "unit", "clips", "seconds", "seconds_for_4096_clips",
                                      "seconds_for_4096_clips_is", "gflop_per_clip", "tflops_algorithmic", "roofline",
                                      "hbm_ops", "peak_hbm_gib", "finite")}
                out["c5"]["workload"] = r["config"]["workload"]
                progress("c5 %s: %.1f clips/s" % (precision, r["value"]))
            out["c5"]["collapsed_f32"] = collapsed_infer_run(torch, M, ops, 6, 1024, 64)
            progress("c5 collapsed: %.0f clips/s" % out["c5"]["collapsed_f32"]["value"])
        if not args.no_cpu_baseline:
            progress("CPU baseline")
            out["cpu_baseline"] = cpu_baseline(stage, fade, args.cpu_sample_batch, n_blocks)
            out["cpu_baseline_c1"] = cpu_baseline(0, False, 16, 1, timed=5)
            progress("done")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
