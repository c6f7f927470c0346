"""Host-side plumbing shared by the layer objects: Keras-style unique layer names, variables that
live in HBM, the device handle and a reusable workspace.  PyTorch is used here only to allocate
device memory, obtain the HIP stream and copy between host and device.
"""
from __future__ import annotations

import re
from typing import Dict, Optional, Tuple

import numpy as np
import os

import torch

_name_counts: Dict[str, int] = {}


# Counts every event that moves or drops a scratch buffer whose raw address kernels are handed
# (workspace / side-buffer growth, clear_session).  A captured HIP graph has those addresses baked
# into its launches: it records the count at capture time and is re-captured when the count moved.
_buffer_generation = [0]


def buffer_generation() -> int:
    return _buffer_generation[0]


def live_scratch():
    """The scratch tensors currently handed out (a graph keeps them referenced while it may replay)."""
    return list(_workspace.values()) + [t for pair in _side.values() for t in pair]


def clear_session():
    """Reset the auto-naming counters (keras.backend.clear_session analogue)."""
    _name_counts.clear()
    _side.clear()
    _buffer_generation[0] += 1


def unique_name(cls_name: str) -> str:
    """Keras auto-name: snake_case class name, then _1, _2, ... per class."""
    s = re.sub(r"(.)([A-Z][a-z0-9]+)", r"\1_\2", cls_name)
    base = re.sub(r"([a-z])([A-Z])", r"\1_\2", s).lower()
    i = _name_counts.get(base, 0)
    _name_counts[base] = i + 1
    return base if i == 0 else "%s_%d" % (base, i)


def device() -> torch.device:
    """The MI355X this process drives.  Raises when no GPU is visible: the layer kernels exist
    only as gfx950 code objects and there is no host fallback."""
    if not torch.cuda.is_available():
        raise RuntimeError("music_generator_amd needs a ROCm GPU (MI355X / gfx950): no device visible")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class Variable:
    """A trainable tensor.  Created on the host (numpy) and moved to HBM the first time a model
    that owns it runs; `grad` is a view into the flat gradient buffer of the model currently
    being differentiated (bound by Model.bind_grads)."""

    def __init__(self, name: str, value: np.ndarray, trainable: bool = True):
        self.name = name
        self.shape: Tuple[int, ...] = tuple(np.shape(value))
        self.trainable = trainable
        self._host: Optional[np.ndarray] = np.ascontiguousarray(value, dtype=np.float32)
        self._dev: Optional[torch.Tensor] = None
        self.grad: Optional[torch.Tensor] = None

    @property
    def size(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n

    def value(self) -> torch.Tensor:
        if self._dev is None:
            host = self._host.reshape(-1) if self.shape else self._host.reshape(1)
            self._dev = torch.from_numpy(host).to(device())
            self._host = None
        return self._dev

    def numpy(self) -> np.ndarray:
        if self._dev is not None:
            return self._dev.detach().cpu().numpy().reshape(self.shape).copy()
        return self._host.reshape(self.shape).copy()

    def assign(self, value):
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(value, dtype=np.float32), self.shape))
        if self._dev is not None:
            self._dev.copy_(torch.from_numpy(arr.reshape(-1) if self.shape else arr.reshape(1)))
        else:
            self._host = arr.copy()

    def fill(self, scalar: float):
        """Set every element to a host scalar without a host<->device synchronisation."""
        if self._dev is not None:
            self._dev.fill_(float(scalar))
        else:
            self._host = np.full(self.shape, scalar, dtype=np.float32)


def variable(value, name: str = "variable") -> Variable:
    """keras.backend.variable analogue (trainable by default, as in models.py:27)."""
    return Variable(name, np.asarray(value, dtype=np.float32))


def set_value(var: Variable, value):
    """keras.backend.set_value analogue (models.py:82)."""
    if np.ndim(value) == 0:
        var.fill(float(value))
    else:
        var.assign(value)


def get_value(var: Variable):
    return var.numpy()


# Arithmetic of the discriminator's grow blocks: "float32" (v_mfma_f32_32x32x2_f32 everywhere) or
# "bfloat16" (32 / 128-channel activations at 8 rows and more stored as bf16, bf16 MFMA with float32
# accumulation; weights, gradients of variables, 2-channel audio tensors and the 4 x 750 base block stay
# float32).
_precision = ["float32"]
BF16_MIN_ROWS = 8
# What a taped bf16 convolution + SoftRectifier keeps for its backward pass beside its input.  False (default):
# nothing; the derivative 1 - tanh(v)^2 + 1/alpha is recovered from the layer's stored bf16 OUTPUT (which the next
# layer's tape holds anyway as its input) as max(1 + 1/alpha - y^2, 1/alpha): half the tape and a third less HBM
# traffic in every taped forward.  True: the derivative is written as a bf16 tensor of its own by the forward
# epilogue (one more rounding-exact value per element, twice the tape).
BF16_SAVE_DERIVATIVE = False
# bf16 32 -> 32 'same' convolutions whose backward needs both gradients: one kernel that reads the layer input and
# the incoming gradient once (mgx_conv_bwd_bf16) instead of a data-gradient and a weight-gradient launch.
BF16_FUSED_CONV_BACKWARD = os.environ.get("MGX_FUSED_CONV_BWD", "1") != "0"
# bf16 generator grow blocks: leave the outputs of the Conv2DTranspose(64) / Dense(256) layers and the gradients
# arriving at them off the tape (5120 of its 8138 rows per 32 positions); the weight-gradient kernels recompute
# them from the neighbouring rows.  False: every tensor on the tape, every weight gradient read from it.
GBLOCK_LEAN_TAPE = os.environ.get("MGX_GBLOCK_LEAN", "1") != "0"


def set_precision(name: str):
    if name in ("bf16", "bfloat16"):
        _precision[0] = "bfloat16"
    elif name in ("fp32", "f32", "float32"):
        _precision[0] = "float32"
    else:
        raise ValueError("precision must be 'float32' or 'bfloat16'")


def precision() -> str:
    return _precision[0]


def bf16_enabled() -> bool:
    return _precision[0] == "bfloat16"


_workspace: Dict[Tuple[int, int], torch.Tensor] = {}

# Lane 0 is the stream the caller made current; lane 1 is the device's second stream, on which WGAN's step
# runs the pass over the real batch beside the pass over the generated one when a batch is too small to fill
# the chip; lane 2 a third one for generator forwards that can run ahead.  Scratch buffers are per lane: the kernels of one lane are ordered by its stream, the two lanes
# are not ordered against each other between fork() and join().
_lane = [0]
_lane_streams: Dict[Tuple[int, int], "torch.cuda.Stream"] = {}


class _SideLane:
    """`with fork() as lane:` enqueues on the second stream, after everything enqueued so far on the current
    one; `lane.join()` (called outside the block) makes the current stream wait for that work.  Also valid
    under HIP graph capture, where it becomes a fork / join of the captured graph."""

    def __init__(self, lane: int = 1):
        if lane < 1:
            raise ValueError("side lanes are numbered from 1")
        dev = device()
        key = (dev.index or 0, lane)
        st = _lane_streams.get(key)
        if st is None:
            st = torch.cuda.Stream(device=dev)
            _lane_streams[key] = st
        self.stream = st
        self.lane = lane
        self._ctx = None

    def __enter__(self):
        if _lane[0] != 0:
            raise RuntimeError("the second lane cannot be forked from itself")
        self.origin = torch.cuda.current_stream()
        self.stream.wait_stream(self.origin)
        self._ctx = torch.cuda.stream(self.stream)
        self._ctx.__enter__()
        _lane[0] = self.lane
        _open_lanes[id(self.stream)] = self
        return self

    def __exit__(self, *exc):
        _lane[0] = 0
        ctx, self._ctx = self._ctx, None
        return ctx.__exit__(*exc)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.stream)
        _open_lanes.pop(id(self.stream), None)


# Lanes that were forked and not yet joined, by stream.  Under HIP graph capture a forked stream that is not
# joined back to the stream the capture began on before the capture ends is an error of the capture (an
# unjoined fork; this ROCm release can crash inside hipStreamEndCapture instead of reporting it), also for a
# lane that only ran leaf work nothing waits for: join_all() closes whatever is open.
_open_lanes: Dict[int, "_SideLane"] = {}


def join_all():
    """Make the current stream wait for every lane forked and not joined so far."""
    for lane in list(_open_lanes.values()):
        lane.join()


def open_lanes() -> int:
    return len(_open_lanes)


def current_lane() -> int:
    """0 on the caller's stream, 1 / 2 inside `with fork(lane):`."""
    return _lane[0]


def fork(lane: int = 1) -> _SideLane:
    """Lane 1 carries the passes over the real batch; lane 2 the generator forwards of the later parts of a step
    that only need the generator's present weights."""
    return _SideLane(lane)


def workspace(nbytes: int) -> torch.Tensor:
    """A reusable scratch buffer of at least `nbytes` on the current device, one per lane: the kernels
    of a lane run on one stream, so its consecutive users are ordered."""
    dev = device()
    key = (dev.index or 0, _lane[0])
    cur = _workspace.get(key)
    if cur is None or cur.numel() < nbytes:
        cur = None
        _workspace.pop(key, None)
        cur = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
        _workspace[key] = cur
        _buffer_generation[0] += 1
    return cur


_side: Dict[Tuple[int, int], list] = {}


def side_buffers(nfloats: int):
    """Two persistent float32 scratch tensors of at least `nfloats` each (grow-only, per device):
    the bounded side buffers of a re-computing fused backward.  Kept out of the caching allocator
    so that their memory is not carved up between two uses."""
    dev = device()
    key = (dev.index or 0, _lane[0])
    cur = _side.get(key)
    if cur is None or cur[0].numel() < nfloats:
        _side.pop(key, None)
        cur = None
        cur = [torch.empty((int(nfloats),), dtype=torch.float32, device=dev) for _ in range(2)]
        _side[key] = cur
        _buffer_generation[0] += 1
    return cur


def available_bytes():
    """(bytes a new allocation can still get: free device memory plus what the caching allocator holds unused,
    total device memory)."""
    free, total = torch.cuda.mem_get_info()
    return free + torch.cuda.memory_reserved() - torch.cuda.memory_allocated(), total


def empty(shape, dev=None, dtype=torch.float32) -> torch.Tensor:
    return torch.empty(tuple(int(s) for s in shape), dtype=dtype, device=dev or device())


def zeros(shape, dev=None) -> torch.Tensor:
    return torch.zeros(tuple(int(s) for s in shape), dtype=torch.float32, device=dev or device())
