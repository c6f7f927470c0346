"""Generator-only inference in chunks: `g_model(z, training=False)` of trainer/utils.py:80-81 and
103-104 for more clips than fit in HBM at once.

A final-stage clip is 256 x 48000 x 2 float32 = 98.3 MB, so the 4096 clips of a sampling run are
403 GB: the latent vectors are drawn for the whole run (their stream does not depend on the chunk
size), the generator runs `chunk` clips at a time, and every finished chunk is handed to `sink`
before the next one is computed.  The chunk tensor lives on the device and is only valid during the
sink call of its chunk; a sink that keeps data copies it (HostRingSink) or reduces it
(ChecksumSink).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import backend as K
from . import ops


def draw_latents(n: int, latent_dim: Sequence[int] = (1, 100, 2), seed: Optional[int] = None, device=None):
    """`tf.random.normal((n,) + latent_dim)` of utils.py:80,103: standard-normal latent vectors on the
    device (800 B a clip).  With a seed the stream is reproducible and independent of the chunk size."""
    dev = device if device is not None else K.device()
    gen = None
    if seed is not None:
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(seed))
    return torch.randn((int(n),) + tuple(int(v) for v in latent_dim), device=dev, dtype=torch.float32, generator=gen)


class CollapsedGenerator:
    """An opt-in inference path that is NOT the reference's computation order: the generator of
    trainer/models.py:358-399, 412-450 has no non-linearity in front of its final LayerNormalization and every
    layer there acts on one position or on disjoint windows, so the tensor entering the normalisation is an affine
    function of the 4 latent numbers of a column pair, the same for every pair.  `refresh()` measures that function
    with the layer kernels themselves (five probe latents through the model up to the normalisation's input) and
    keeps it as a table; a call then writes every clip once, normalised, without running the layers
    (mgx_collapse_apply).  Results differ from `g_model(z)` by float32 rounding only (the test states the bound).
    Call `refresh()` again after the generator's weights (or a fade-in alpha) changed."""

    def __init__(self, g_model):
        from .graph import Model
        from .layers import LayerNormalization
        self.g_model = g_model
        self.norm = g_model.layers[-1]
        if not isinstance(self.norm, LayerNormalization):
            raise ValueError("the generator is expected to end in LayerNormalization(axis=[1, 2, 3])")
        self.pre_model = Model(g_model.input, g_model.layers[-2].output)
        lat = tuple(int(v) for v in g_model.input_shape[1:])
        if len(lat) != 3 or lat[0] != 1 or lat[1] % 2 or lat[2] != 2:
            raise ValueError("latent shape (1, even, 2) expected, got %s" % (lat,))
        self.latent_dim = lat
        self.ncols = lat[1] // 2
        self.refresh()

    def refresh(self):
        """Measure the table with the float32 layer kernels, whatever precision is set globally: probes run
        through the bf16 generator kernels would carry their rounding (~1e-3) into every clip written later."""
        probe_z = torch.zeros((5, self.ncols, 4), dtype=torch.float32, device=K.device())
        for k in range(4):
            probe_z[1 + k, :, k] = 1.0
        precision = K.precision()
        K.set_precision("float32")
        try:
            probe = self.pre_model(probe_z.reshape((5,) + self.latent_dim), training=False).contiguous()
        finally:
            K.set_precision(precision)
        if probe.shape[2] % self.ncols:
            raise ValueError("output width %d is not a multiple of the %d latent column pairs" % (probe.shape[2], self.ncols))
        self.h, self.wp = int(probe.shape[1]), int(probe.shape[2]) // self.ncols
        self.table, self.gram = ops.collapse_table(probe, self.ncols)

    def __call__(self, z, training: bool = False):
        if not torch.is_tensor(z):
            z = torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32))
        z = z.to(device=K.device(), dtype=torch.float32)
        if tuple(z.shape[1:]) != self.latent_dim:
            raise ValueError("latent tensor of shape %s, the generator takes %s" % (tuple(z.shape), self.latent_dim))
        z = z.reshape(z.shape[0], self.ncols * 4).contiguous()
        return ops.collapse_apply(self.table, self.gram, z, self.norm.gamma.value(), self.norm.beta.value(), self.h,
                                  self.wp, self.ncols, self.norm.epsilon)


def generate_clips(g_model, n: int, chunk: int = 8, sink: Optional[Callable] = None,
                   latent_dim: Sequence[int] = (1, 100, 2), seed: Optional[int] = None, z=None,
                   collapse: bool = False) -> int:
    """Run the generator on `n` latent vectors, `chunk` at a time (the last chunk may be shorter),
    calling `sink(first_index, audio)` with each (m, H, W, 2) float32 device tensor in order.
    `z` supplies the latent vectors ((n,) + latent_dim, array or tensor) instead of drawing them.
    `collapse=True` runs the chunks through CollapsedGenerator (built here from the present weights).
    Returns the number of clips generated.  Nothing is synchronised with the host here: a sink that
    only enqueues device work lets consecutive chunks overlap its copies."""
    if collapse:
        g_model = CollapsedGenerator(g_model)
    n, chunk = int(n), int(chunk)
    if n < 0 or chunk < 1:
        raise ValueError("generate_clips needs n >= 0 and chunk >= 1 (got n=%d, chunk=%d)" % (n, chunk))
    if z is None:
        z = draw_latents(n, latent_dim, seed)
    else:
        z = torch.as_tensor(np.asarray(z) if not torch.is_tensor(z) else z, dtype=torch.float32).to(K.device())
        if tuple(z.shape) != (n,) + tuple(int(v) for v in latent_dim):
            raise ValueError("latent tensor of shape %s for n=%d, latent_dim=%s" % (tuple(z.shape), n, tuple(latent_dim)))
    done = 0
    while done < n:
        m = min(chunk, n - done)
        audio = g_model(z[done:done + m], training=False)
        if sink is not None:
            sink(done, audio)
        del audio
        done += m
    return done


class ChecksumSink:
    """Reduces every chunk on the device to per-clip sum, sum of squares and absolute peak (float64
    accumulators, one pass over the chunk, no host synchronisation until `result()`)."""

    def __init__(self):
        self._rows: List[torch.Tensor] = []
        self.clips = 0

    def __call__(self, first: int, audio: torch.Tensor):
        m = audio.shape[0]
        row = torch.empty((m, 3), dtype=torch.float64, device=audio.device)
        ops.clip_checksum(audio.contiguous(), row)     # one read of the chunk (mgx_clip_checksum)
        self._rows.append(row)
        self.clips += m

    def result(self) -> np.ndarray:
        """(clips, 3) float64: sum, sum of squares, peak of every clip in generation order."""
        if not self._rows:
            return np.zeros((0, 3))
        return torch.cat(self._rows).cpu().numpy()


class HostRingSink:
    """Streams chunks to a ring of pinned host buffers on a copy stream of its own, so the copy of chunk
    k overlaps the computation of chunk k+1; `consume(first_index, array)` (a host callable, e.g. a WAV
    writer) runs on a slot's content just before the slot is reused and in `flush()`."""

    def __init__(self, clip_shape: Sequence[int], chunk: int, consume: Callable, slots: int = 2):
        self.consume = consume
        self.buffers = [torch.empty((int(chunk),) + tuple(int(v) for v in clip_shape), dtype=torch.float32).pin_memory()
                        for _ in range(int(slots))]
        self.device_stage = [None] * int(slots)
        self.pending = [None] * int(slots)            # (event, first index, clip count)
        self.stream = torch.cuda.Stream()
        self.k = 0

    def _drain(self, slot: int):
        if self.pending[slot] is not None:
            ev, first, m = self.pending[slot]
            ev.synchronize()
            self.consume(first, self.buffers[slot][:m].numpy())
            self.pending[slot] = None

    def __call__(self, first: int, audio: torch.Tensor):
        slot = self.k % len(self.buffers)
        self.k += 1
        self._drain(slot)
        m = audio.shape[0]
        # the generator reuses the chunk's memory for the next chunk: the copy stream must finish reading it
        # first, so the copy is ordered after the producer and the tensor is tied to the copy stream
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self.buffers[slot][:m].copy_(audio, non_blocking=True)
            audio.record_stream(self.stream)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pending[slot] = (ev, first, m)

    def flush(self):
        order = sorted((p[1], s) for s, p in enumerate(self.pending) if p is not None)
        for _, slot in order:
            self._drain(slot)
