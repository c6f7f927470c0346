"""Single-node data parallelism for the training step: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" for CPU rehearsal).

The reference is single-device (its MirroredStrategy is commented out, trainer/task.py:73-80).
Per-sample independence holds everywhere except MinibatchStdDev (groups of 4 samples taken with
stride M = N/4, trainer/models.py:168-176) and the batch-mean logits inside both losses
(trainer/models.py:469-470), so a global batch of N on R ranks is reproduced exactly when
  * rank r owns the groups m in [r*M/R, (r+1)*M/R), i.e. samples {g*M + m}, kept g-major so the
    local reshape [G, M/R, ...] forms the same groups;
  * the pair (sum fake logits, sum real logits) is all-reduced before the loss kernel, which
    divides by the GLOBAL batch; local gradients are then already scaled by 1/N and
  * the flat gradient buffers are all-reduced (sum) before each Adam step.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist


def shard_indices(n_global: int, rank: int, world_size: int) -> np.ndarray:
    """Indices (into the global batch) of the samples rank `rank` processes, in local order."""
    g = min(4, n_global)
    if n_global % g:
        raise ValueError("global batch %d is not divisible by the MinibatchStdDev group size %d" % (n_global, g))
    m = n_global // g
    if m % world_size:
        raise ValueError("the %d MinibatchStdDev groups of a batch of %d do not split over %d ranks"
                         % (m, n_global, world_size))
    per = m // world_size
    local_groups = np.arange(rank * per, (rank + 1) * per)
    return (np.arange(g)[:, None] * m + local_groups[None, :]).reshape(-1)


class DataParallel:
    """Process-group handle attached to a WGAN (`wgan.dist = DataParallel()`)."""

    def __init__(self, backend: Optional[str] = None, init: bool = True, force_collectives: bool = False):
        if init and not dist.is_initialized():
            backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if backend == "nccl":
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group(backend=backend)
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        # collectives are skipped in a group of one unless forced (a one-rank group still runs every call
        # through the backend: the way to execute the RCCL path on a box with a single GPU)
        self.active = dist.is_initialized() and (self.world_size > 1 or force_collectives)
        # True: every wait_for() is bracketed by two events on the waiting stream, so that the time the stream
        # stood still for a collective can be read afterwards (take_wait_stall_ms)
        self.record_waits = False
        self._wait_events = []

    def all_reduce(self, t: torch.Tensor):
        """In-place sum over ranks, enqueued on the current stream (RCCL) / blocking (gloo)."""
        if self.active:
            if t.is_cuda and dist.get_backend() != "nccl":
                # gloo rehearsal with device tensors: reduce a host copy
                host = t.detach().cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                t.copy_(host)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t

    def all_reduce_async(self, t: torch.Tensor):
        """Start an in-place sum over ranks that runs beside the kernels enqueued after this call
        (RCCL uses its own stream, ordered after the work already on the current one).  Returns a
        handle for wait_for(); None when there is nothing to do."""
        if self.active:
            if dist.get_backend() != "nccl":      # gloo rehearsal: reduce in place now
                self.all_reduce(t)
                return None
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)
        return None

    def wait_for(self, handle):
        """Order the current stream behind an all_reduce_async (blocks the host only for gloo)."""
        if handle is not None:
            if self.record_waits and torch.cuda.is_available():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                handle.wait()
                e1.record()
                self._wait_events.append((e0, e1))
            else:
                handle.wait()

    def take_wait_stall_ms(self) -> float:
        """Milliseconds the waiting stream spent between 'reached the wait' and 'the collective had finished',
        summed over the waits recorded since the last call (synchronises the device)."""
        if not self._wait_events:
            return 0.0
        torch.cuda.synchronize()
        total = sum(e0.elapsed_time(e1) for e0, e1 in self._wait_events)
        self._wait_events = []
        return float(total)

    def shard(self, global_batch):
        idx = shard_indices(global_batch.shape[0], self.rank, self.world_size)
        if isinstance(global_batch, torch.Tensor):
            return global_batch[torch.as_tensor(idx, device=global_batch.device)]
        return global_batch[idx]

    def broadcast_weights(self, *models):
        """Make every replica start from rank 0's weights."""
        if not self.active:
            return
        seen = set()
        for model in models:
            for layer in model.layers:
                for v in layer.weights:
                    if id(v) not in seen:
                        seen.add(id(v))
                        t = v.value()
                        if t.is_cuda and dist.get_backend() != "nccl":
                            # gloo rehearsal with device tensors: broadcast a host copy (as all_reduce does)
                            host = t.detach().cpu()
                            dist.broadcast(host, src=0)
                            t.copy_(host)
                        else:
                            dist.broadcast(t, src=0)

    def barrier(self):
        if self.active:
            dist.barrier()
