"""Worker of test_two_rank_device_step_equals_single_process and test_rccl_path_executes_in_a_group_of_one
(not collected by pytest):
    python tests/dp_worker.py <repo root> <out.npz>   (1 process, or 2 under torch.distributed.run)
MGX_DP_RCCL_ONE=1: a process group of ONE rank on the `nccl` backend (= RCCL) with the collectives forced, so
that broadcast, the asynchronous gradient all-reduces and the logit-sum all-reduces all run through RCCL."""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch
import music_generator_amd as M
from music_generator_amd.parallel import DataParallel
from oracle import ref_step as R
from test_gpu_step import build

out = sys.argv[2]
world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
rccl_one = bool(os.environ.get("MGX_DP_RCCL_ONE"))
if rccl_one:
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dp = DataParallel(backend="nccl", force_collectives=True)
    assert dp.active and torch.distributed.get_backend() == "nccl"
    dp.record_waits = True
else:
    dp = DataParallel(backend="gloo") if world > 1 else None
n_blocks, stage, batch = 2, 1, 16
d, g, comp, W = build(n_blocks)
gan = comp[stage][1]
gan.dist = dp
if rccl_one:
    dp.broadcast_weights(gan.discriminator, gan.generator)
gan.micro_batch = 4 if (world > 1 or rccl_one) else None
gan.set_train_steps(11)
gan.actual_step = 3
real, z4 = R.synthetic_inputs(stage, batch, seed=5)
if dp is not None and not rccl_one:
    real = dp.shard(real)
    z4 = [dp.shard(z) for z in z4]
if os.environ.get("MGX_DP_TRACE") and dp is not None:
    import functools
    for name in ("all_reduce", "all_reduce_async", "wait_for"):
        fn = getattr(dp, name)
        def wrap(*a, _fn=fn, _name=name, **k):
            print("rank", dp.rank, _name, "begin", flush=True)
            r = _fn(*a, **k)
            print("rank", dp.rank, _name, "end", flush=True)
            return r
        setattr(dp, name, wrap)
gan.inject_latents(z4)
print("rank", 0 if dp is None else dp.rank, "step begin", flush=True)
metrics = {k: float(v) for k, v in gan.train_step(real).items()}
torch.cuda.synchronize()
print("rank", 0 if dp is None else dp.rank, "step end", flush=True)
if rccl_one:
    stall = dp.take_wait_stall_ms()
    assert stall >= 0.0 and not dp._wait_events
    print("stall ms in front of the gradient combines:", stall, flush=True)
if dp is None or dp.rank == 0:
    arrays = {"m|" + k: np.float64(v) for k, v in metrics.items()}
    for model in (gan.discriminator, gan.generator):
        for v in model.trainable_variables:
            arrays["g|" + v.name.replace("/", "~")] = v.grad.cpu().numpy()
    np.savez(out, **arrays)
if dp is not None:
    torch.distributed.destroy_process_group()
