"""Thin typed wrappers over the C-ABI: torch tensors in, device pointers out.  Every wrapper checks
on the host that operand sizes match what the kernel and its grid assume before launching.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _cabi
from . import backend as K
from ._cabi import ConvDesc, MgxError, check


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
        raise MgxError("operand must be a contiguous float32 device tensor")
    return t.data_ptr()


def _lib():
    return _cabi.load()


class LaunchProfiler:
    """Optional per-launch timing of the contraction kernels with HIP events recorded on the
    launch stream (set `ops.PROFILER = LaunchProfiler()`; None = off).  Keys mirror the template
    instantiation the C side dispatches to, so they can be matched with a rocprofv3 kernel trace."""

    def __init__(self):
        self.records = []          # (key, flops, start_event, end_event)

    def start(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def stop(self, key, flops, e0, nbytes=0.0):
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.records.append((key, flops, e0, e1, nbytes))

    def summary(self):
        """{key: (launches, total_ms, total_flops, total_bytes)} — call after torch.cuda.synchronize()."""
        out = {}
        for key, flops, e0, e1, nbytes in self.records:
            n, ms, fl, by = out.get(key, (0, 0.0, 0.0, 0.0))
            out[key] = (n + 1, ms + e0.elapsed_time(e1), fl + flops, by + nbytes)
        return out


PROFILER: Optional[LaunchProfiler] = None


def _hbm_op(name):
    """Decorator for the wrappers of bandwidth-bound kernels: when a LaunchProfiler is active the call
    is timed and the bytes of its tensor arguments (each read or written once, the algorithmic
    traffic) are recorded under `name` with zero FLOPs."""
    def deco(fn):
        def wrapped(*args, **kw):
            if PROFILER is None:
                return fn(*args, **kw)
            nbytes = 0.0
            for a in list(args) + list(kw.values()):
                if isinstance(a, torch.Tensor):
                    nbytes += a.numel() * a.element_size()
                elif isinstance(a, (list, tuple)):
                    nbytes += sum(t.numel() * t.element_size() for t in a if isinstance(t, torch.Tensor))
            e0 = PROFILER.start()
            out = fn(*args, **kw)
            PROFILER.stop(name, 0.0, e0, nbytes)
            return out
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return wrapped
    return deco


def _gemm_key(d: ConvDesc, mul: bool = False) -> str:
    route = _lib().mgx_conv_route(C.byref(d), 2 if mul else 0)
    if route == 2:
        return "pointwise_to2_kernel" if d.ngemm == 2 else "pointwise2_kernel"
    if route == 3:
        return "wino3_kernel"
    if route == 4:
        return "pgemm_kernel<%d,%d,%s>" % (d.cin // 8, d.ngemm // 32, "true" if d.wsk == 1 else "false")
    if route == 5:
        return "tapsum_kernel"
    if route == 1:
        kh, kw = (1, 1) if d.sh == 2 else (d.kh, d.kw)      # the strided form runs as a 1x1 kernel over gathered windows
        return "tconv3_kernel<%d,%d,%s,%s>" % (kh, kw, "true" if d.wsk == 1 else "false",
                                               "true" if mul else "false")
    nt = (d.ngemm + 31) // 32
    pn = "4,1" if nt == 1 else ("2,2" if nt == 2 else "2,4")
    return "conv_gemm_kernel<%s,%d,%s>" % (pn, 0 if d.wsn == 1 else 1, "true" if d.cin % 8 == 0 else "false")


def _wgrad_key(d: ConvDesc) -> str:
    route = _lib().mgx_conv_route(C.byref(d), 1)
    if route == 3:
        return "wino3_wgrad_kernel"
    if route == 2:
        return "pointwise2_wgrad_kernel"
    if route == 1:
        return "twgrad_kernel<%d,%d>" % (d.kh, d.kw)
    taps, kt, nt = d.kh * d.kw, (d.cin + 31) // 32, (d.ngemm + 31) // 32
    if taps % 9 == 0:
        return "wgrad_kernel<9,1,1>"
    if taps % 8 == 0:
        return "wgrad_kernel<8,1,1>"
    if taps % 4 == 0:
        return "wgrad_kernel<4,1,1>"
    return "wgrad_kernel<1,%d,%d>" % (2 if kt >= 2 else 1, 4 if nt >= 4 else (2 if nt >= 2 else 1))


def _conv_flops(d: ConvDesc) -> float:
    """Algorithmic FLOPs of the contraction: 2 * rows * taps * cin * ngemm."""
    return 2.0 * d.nimg * d.hout * d.wout * d.kh * d.kw * d.cin * d.ngemm


WINOGRAD_FLOP_FRACTION = 16.0 / 36.0


def _gemm_flops(d: ConvDesc, key: str) -> float:
    """FLOPs of the algorithm the launch runs: the direct contraction, or for the Winograd
    F(2x2,3x3) kernel its 16 multiply-adds per 2x2 output patch and channel pair in place of 36."""
    return _conv_flops(d) * (WINOGRAD_FLOP_FRACTION if key.startswith("wino3_") else 1.0)


def _conv_bytes(d: ConvDesc, n_in: int = 1, n_out: int = 1) -> float:
    """Algorithmic HBM bytes of one contraction launch: every input-shaped operand read once,
    every output-shaped operand written once, the weights once."""
    xe, ye, we = _conv_sizes(d)
    return 4.0 * (n_in * xe + n_out * ye + we + d.nbias)


def make_conv_desc(**kw) -> ConvDesc:
    d = ConvDesc()
    defaults = dict(kh=1, kw=1, sh=1, sw=1, pt=0, pl=0, flip=0, dkh=1, dkw=1, wtap=0, nbias=0,
                    accumulate=0, act_alpha=0.0)
    defaults.update(kw)
    for k, v in defaults.items():
        setattr(d, k, v)
    return d


def _conv_sizes(d: ConvDesc):
    x_elems = d.nimg * d.hin * d.win * d.cin
    y_elems = d.nimg * d.hout * d.dkh * d.wout * d.dkw * d.cout
    w_last = (d.kh * d.kw - 1) * d.wtap + (d.cin - 1) * d.wsk + (d.ngemm - 1) * d.wsn
    return x_elems, y_elems, w_last + 1


def conv_gemm(d: ConvDesc, x, w, bias, y, xmul=None, dact=None, accumulate=False, act_alpha=0.0):
    xe, ye, we = _conv_sizes(d)
    if x.numel() != xe or y.numel() != ye or w.numel() < we:
        raise MgxError("conv_gemm: operand sizes do not match the descriptor (%d/%d, %d/%d, %d/%d)"
                       % (x.numel(), xe, y.numel(), ye, w.numel(), we))
    if xmul is not None and xmul.numel() != xe:
        raise MgxError("conv_gemm: xmul size mismatch")
    if dact is not None and dact.numel() != ye:
        raise MgxError("conv_gemm: dact size mismatch")
    if d.nbias and (bias is None or bias.numel() != d.nbias):
        raise MgxError("conv_gemm: bias size mismatch")
    d.accumulate = 1 if accumulate else 0
    d.act_alpha = float(act_alpha)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_conv_gemm_f32(C.byref(d), _p(x), _p(xmul), _p(w), _p(bias) if d.nbias else None,
                                   _p(y), _p(dact), K.stream_ptr()), "mgx_conv_gemm_f32")
    if e0 is not None:
        key = _gemm_key(d, xmul is not None)
        PROFILER.stop(key, _gemm_flops(d, key), e0,
                      _conv_bytes(d, 2 if xmul is not None else 1,
                                  (2 if dact is not None else 1) + (1 if accumulate else 0)))


def conv_wgrad(d: ConvDesc, x, dy, dw, dbias, dymul=None):
    xe, ye, we = _conv_sizes(d)
    if x.numel() != xe or dy.numel() != ye or dw.numel() < we:
        raise MgxError("conv_wgrad: operand sizes do not match the descriptor")
    if dymul is not None and dymul.numel() != ye:
        raise MgxError("conv_wgrad: dymul size mismatch")
    if dbias is not None and dbias.numel() != d.nbias:
        raise MgxError("conv_wgrad: bias gradient size mismatch")
    lib = _lib()
    nbytes = lib.mgx_conv_wgrad_workspace(C.byref(d))
    if nbytes <= 0:
        raise MgxError("conv_wgrad: bad descriptor")
    ws = K.workspace(nbytes)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(lib.mgx_conv_wgrad_f32(C.byref(d), _p(x), _p(dy), _p(dymul), _p(dw), _p(dbias),
                                 ws.data_ptr(), ws.numel(), K.stream_ptr()), "mgx_conv_wgrad_f32")
    if e0 is not None:
        key = _wgrad_key(d)
        PROFILER.stop(key, _gemm_flops(d, key), e0, _conv_bytes(d, 1, 2 if dymul is not None else 1))


def gblock_pack_floats() -> int:
    return int(_lib().mgx_gblock_pack_floats())


def gblock_side_floats(npos) -> int:
    """Floats of a side buffer (activations or gradients) of a fused grow block for npos positions."""
    return int(_lib().mgx_gblock_side_floats_n(npos))


def gblock_packt_floats() -> int:
    return int(_lib().mgx_gblock_packt_floats())


# 2 * multiply-adds of one grow block per low-resolution position (4 output positions)
GBLOCK_FLOP_PER_POSITION = 2.0 * (4 * 2 * 128 + 6 * (2 * 16 * 128 + 2 * 16 * 32 + 4 * 64 * 32 + 4 * 64 * 128)
                                  + 4 * (128 * 256 + 256 * 64 + 64 * 8 + 8 * 2))


def _gblock_range(nimg, hl, wl, pos0, npos):
    total = nimg * hl * wl
    if pos0 < 0 or pos0 >= total or pos0 % 128:
        raise MgxError("gblock: the first position of a range must be a multiple of 128 inside the grid")
    return min(npos, total - pos0) if npos and npos > 0 else total - pos0


def clip_checksum(audio, out):
    """audio (m, ...) contiguous float32 -> out (m, 3) float64: per-clip sum, sum of squares, absolute peak."""
    m = audio.shape[0]
    elems = audio.numel() // max(m, 1)
    if audio.dtype != torch.float32 or not audio.is_contiguous() or m < 1 or elems % 4:
        raise MgxError("clip_checksum: contiguous float32 clips with a multiple of 4 elements expected")
    if out.dtype != torch.float64 or not out.is_contiguous() or tuple(out.shape) != (m, 3):
        raise MgxError("clip_checksum: out must be a contiguous float64 (m, 3)")
    nbytes = int(_lib().mgx_clip_checksum_workspace(m))
    ws = K.workspace(nbytes)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_clip_checksum(_p(audio), m, elems, out.data_ptr(), ws.data_ptr(), nbytes, K.stream_ptr()),
          "mgx_clip_checksum")
    if e0 is not None:
        PROFILER.stop("clip_checksum", 0.0, e0, 4.0 * audio.numel())


GHEAD_FLOP_PER_SAMPLE = 2.0 * 50 * (128 * 4 + 4 * (8 * 128 + 16 * 4 + 64 * 8 + 128 * 16))


def _ghead_tensors(ts, nimg):
    from ._cabi import GHeadTensors
    if len(ts) != 4:
        raise MgxError("ghead: four branch tensors expected")
    t = GHeadTensors()
    for b, y in enumerate(ts):
        if y.dtype != torch.float32 or not y.is_contiguous() or y.numel() != nimg * 4 * 50 * 32:
            raise MgxError("ghead: branch tensor %d is not a contiguous float32 (n, 4, 50, 32)" % b)
        t.y[b] = y.data_ptr()
    return t


def ghead_fwd(weights, z, ys):
    """weights: _cabi.GHeadWeights; z (n,1,100,2) -> ys[b] (n,4,50,32), b = 0..3 (the head of the stage-0
    generator in one launch)."""
    nimg = z.shape[0]
    if z.dtype != torch.float32 or not z.is_contiguous() or z.numel() != nimg * 200:
        raise MgxError("ghead_fwd: z must be a contiguous float32 (n, 1, 100, 2)")
    t = _ghead_tensors(ys, nimg)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_ghead_fwd_f32(C.byref(weights), _p(z), nimg, C.byref(t), K.stream_ptr()), "mgx_ghead_fwd_f32")
    if e0 is not None:
        PROFILER.stop("ghead_fwd_kernel", GHEAD_FLOP_PER_SAMPLE * nimg, e0, 4.0 * nimg * (200 + 4 * 6400))


def ghead_bwd(weights, z, dys, grads=None, dz=None):
    """Given dys[b] = d(loss)/d(ys[b]): adds the weight / bias gradients of the head's 34 tensors into `grads`
    (_cabi.GHeadWeights of gradient pointers; None: skipped) and writes d(loss)/dz into `dz` (None: skipped)."""
    nimg = z.shape[0]
    if z.dtype != torch.float32 or not z.is_contiguous() or z.numel() != nimg * 200:
        raise MgxError("ghead_bwd: z must be a contiguous float32 (n, 1, 100, 2)")
    if grads is None and dz is None:
        raise MgxError("ghead_bwd: nothing to compute")
    if dz is not None and (dz.dtype != torch.float32 or not dz.is_contiguous() or dz.numel() != z.numel()):
        raise MgxError("ghead_bwd: dz must match z")
    t = _ghead_tensors(dys, nimg)
    nbytes, ws = 0, None
    if grads is not None:
        nbytes = int(_lib().mgx_ghead_bwd_workspace(nimg))
        if nbytes <= 0:
            raise MgxError("ghead_bwd: bad batch size")
        ws = K.workspace(nbytes)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_ghead_bwd_f32(C.byref(weights), _p(z), nimg, C.byref(t), C.byref(grads) if grads is not None else None,
                                   _p(dz), ws.data_ptr() if ws is not None else None, nbytes, K.stream_ptr()), "mgx_ghead_bwd_f32")
    if e0 is not None:
        PROFILER.stop("ghead_bwd_kernel", 3.0 * GHEAD_FLOP_PER_SAMPLE * nimg, e0, 4.0 * nimg * (200 + 4 * 6400))


def gblock_fwd(weights, pack, x, y, nimg, hl, wl, act=None, pos0=0, npos=0):
    """weights: _cabi.GBlockWeights; x (nimg,hl,wl,2) -> y (nimg,2hl,2wl,2) for the low-resolution
    positions [pos0, pos0+npos) (default: all).  `act` (optional) receives the intermediate tensors
    of the range for the backward entry points; y may then be None."""
    n = _gblock_range(nimg, hl, wl, pos0, npos)
    if x.numel() != nimg * hl * wl * 2 or pack.numel() < gblock_pack_floats():
        raise MgxError("gblock_fwd: size mismatch")
    if y is None and act is None:
        raise MgxError("gblock_fwd: nothing to write")
    if y is not None and y.numel() != 4 * x.numel():
        raise MgxError("gblock_fwd: output size mismatch")
    if act is not None and act.numel() < gblock_side_floats(n):
        raise MgxError("gblock_fwd: activation buffer too small")
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_gblock_fwd_f32(C.byref(weights), _p(pack), _p(x), _p(y), _p(act), nimg, hl, wl, pos0, n,
                                    K.stream_ptr()), "mgx_gblock_fwd_f32")
    if e0 is not None:
        PROFILER.stop("gblock_fwd_kernel<%s>" % ("true" if act is not None else "false"),
                      GBLOCK_FLOP_PER_POSITION * n, e0,
                      4.0 * (10 * n + (gblock_side_floats(n) if act is not None else 0)))


def gblock_dgrad(weights, packt, dy, delta, dx, nimg, hl, wl, pos0=0, npos=0):
    """dy (nimg,2hl,2wl,2) -> dx (nimg,hl,wl,2) for the positions of the range; `delta` receives the
    intermediate gradients of the range."""
    n = _gblock_range(nimg, hl, wl, pos0, npos)
    if dx.numel() != nimg * hl * wl * 2 or dy.numel() != 4 * dx.numel() or packt.numel() < gblock_packt_floats():
        raise MgxError("gblock_dgrad: size mismatch")
    if delta.numel() < gblock_side_floats(n):
        raise MgxError("gblock_dgrad: gradient buffer too small")
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_gblock_dgrad_f32(C.byref(weights), _p(packt), _p(dy), _p(delta), _p(dx), nimg, hl, wl, pos0, n,
                                      K.stream_ptr()), "mgx_gblock_dgrad_f32")
    if e0 is not None:
        PROFILER.stop("gblock_dgrad_kernel", GBLOCK_FLOP_PER_POSITION * n, e0,
                      4.0 * (10 * n + gblock_side_floats(n)))


def gblock_wgrad(act, delta, grads, npos):
    """Adds the weight / bias gradients of the block, computed from the side buffers of `npos`
    positions, into the tensors `grads` points to (_cabi.GBlockWeights holding gradient pointers)."""
    need = gblock_side_floats(npos)
    if act.numel() < need or delta.numel() < need:
        raise MgxError("gblock_wgrad: side buffer too small")
    nbytes = int(_lib().mgx_gblock_wgrad_workspace_n(npos))
    ws = K.workspace(nbytes)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_gblock_wgrad_f32(_p(act), _p(delta), C.byref(grads), ws.data_ptr(), ws.numel(), npos,
                                      K.stream_ptr()), "mgx_gblock_wgrad_f32")
    if e0 is not None:
        PROFILER.stop("gblock_wgrad_kernel<*>", GBLOCK_FLOP_PER_POSITION * npos, e0, 8.0 * need)


@_hbm_op("upsample_fwd")
def upsample2x_fwd(x, y, nimg, h, w, c):
    if x.numel() != nimg * h * w * c or y.numel() != 4 * x.numel():
        raise MgxError("upsample2x_fwd: size mismatch")
    check(_lib().mgx_upsample2x_fwd(_p(x), _p(y), nimg, h, w, c, K.stream_ptr()), "mgx_upsample2x_fwd")


def upsample2x_bwd(dy, dx, nimg, h, w, c, accumulate):
    if dx.numel() != nimg * h * w * c or dy.numel() != 4 * dx.numel():
        raise MgxError("upsample2x_bwd: size mismatch")
    check(_lib().mgx_upsample2x_bwd(_p(dy), _p(dx), nimg, h, w, c, int(accumulate), K.stream_ptr()),
          "mgx_upsample2x_bwd")


@_hbm_op("avgpool_fwd")
def avgpool2x_fwd(x, y, nimg, h, w, c):
    if x.numel() != nimg * h * w * c or 4 * y.numel() != x.numel():
        raise MgxError("avgpool2x_fwd: size mismatch")
    check(_lib().mgx_avgpool2x_fwd(_p(x), _p(y), nimg, h, w, c, K.stream_ptr()), "mgx_avgpool2x_fwd")


def avgpool2x_bwd(dy, dx, nimg, h, w, c, accumulate):
    if dx.numel() != nimg * h * w * c or 4 * dy.numel() != dx.numel():
        raise MgxError("avgpool2x_bwd: size mismatch")
    check(_lib().mgx_avgpool2x_bwd(_p(dy), _p(dx), nimg, h, w, c, int(accumulate), K.stream_ptr()),
          "mgx_avgpool2x_bwd")


@_hbm_op("add_n")
def add_n(xs: Sequence[torch.Tensor], y):
    n = y.numel()
    if not 1 <= len(xs) <= 8 or any(t.numel() != n for t in xs):
        raise MgxError("add_n: size mismatch")
    arr = (C.c_void_p * len(xs))(*[_p(t) for t in xs])
    check(_lib().mgx_add_n(arr, len(xs), _p(y), n, K.stream_ptr()), "mgx_add_n")


def gather_groups(src, g, m, m0, cnt):
    """mgx_gather_groups: samples a * m + m0 + b (a < g, b < cnt) of a float32 batch of g * m samples, a-major."""
    if src.dtype != torch.float32 or src.shape[0] != g * m:
        raise MgxError("gather_groups: a float32 batch of g * m samples is expected")
    row = src.numel() // src.shape[0]
    if row % 4:
        raise MgxError("gather_groups: samples must be a multiple of 4 floats")
    dst = K.empty((g * cnt,) + tuple(src.shape[1:]))
    check(_lib().mgx_gather_groups(_p(src), _p(dst), row, int(g), int(m), int(m0), int(cnt), K.stream_ptr()),
          "mgx_gather_groups")
    return dst


def collapse_table(probe, ncols):
    """mgx_collapse_table: (table, gram) of the collapsed generator from the probe tensor (5, H, ncols * wp, 2)."""
    if probe.dtype != torch.float32 or probe.dim() != 4 or probe.shape[0] != 5 or probe.shape[3] != 2 \
            or probe.shape[2] % ncols or not probe.is_contiguous():
        raise MgxError("collapse_table: a contiguous float32 probe of shape (5, H, ncols * wp, 2) is expected")
    h, wp = int(probe.shape[1]), int(probe.shape[2]) // int(ncols)
    table = K.empty((h * wp, 2, 5))
    gram = K.empty((20,), dtype=torch.float64)
    check(_lib().mgx_collapse_table(_p(probe), _p(table), gram.data_ptr(), h, wp, int(ncols), K.stream_ptr()),
          "mgx_collapse_table")
    return table, gram


def collapse_apply(table, gram, z, gamma, beta, h, wp, ncols, eps):
    """mgx_collapse_apply: normalised clips (n, h, ncols * wp, 2) of the latents z (n, ncols * 4 numbers)."""
    n = int(z.shape[0])
    if z.dtype != torch.float32 or not z.is_contiguous() or z.numel() != n * ncols * 4:
        raise MgxError("collapse_apply: latents must be contiguous float32, ncols * 4 numbers a clip")
    if table.numel() != h * wp * 10 or gram.numel() != 20 or gram.dtype != torch.float64 or gamma.numel() != h * wp * ncols * 2 \
            or beta.numel() != gamma.numel():
        raise MgxError("collapse_apply: table / gamma / beta do not match h, wp, ncols")
    out = K.empty((n, h, ncols * wp, 2))
    stats = K.empty((n, 2))
    check(_lib().mgx_collapse_apply(_p(table), gram.data_ptr(), _p(z), _p(gamma), _p(beta), _p(out), _p(stats), n, int(h),
                                    int(wp), int(ncols), float(eps), K.stream_ptr()), "mgx_collapse_apply")
    return out


def add_inplace(dst, src):
    if dst.numel() != src.numel():
        raise MgxError("add_inplace: size mismatch")
    check(_lib().mgx_add_inplace(_p(dst), _p(src), dst.numel(), K.stream_ptr()), "mgx_add_inplace")


@_hbm_op("wsum_fwd")
def weighted_sum_fwd(a, b, alpha, y):
    if a.numel() != b.numel() or a.numel() != y.numel() or alpha.numel() != 1:
        raise MgxError("weighted_sum_fwd: size mismatch")
    check(_lib().mgx_weighted_sum_fwd(_p(a), _p(b), _p(alpha), _p(y), y.numel(), K.stream_ptr()),
          "mgx_weighted_sum_fwd")


@_hbm_op("wsum_bwd")
def weighted_sum_bwd(dy, a, b, alpha, da, acc_a, db, acc_b, dalpha):
    n = dy.numel()
    for t in (a, b, da, db):
        if t is not None and t.numel() != n:
            raise MgxError("weighted_sum_bwd: size mismatch")
    lib = _lib()
    ws = K.workspace(lib.mgx_reduce_workspace())
    check(lib.mgx_weighted_sum_bwd(_p(dy), _p(a), _p(b), _p(alpha), _p(da), int(acc_a), _p(db), int(acc_b),
                                   _p(dalpha), ws.data_ptr(), ws.numel(), n, K.stream_ptr()),
          "mgx_weighted_sum_bwd")


@_hbm_op("layernorm_fwd")
def layernorm_fwd(x, gamma, beta, y, stats, nimg, e, eps):
    if x.numel() != nimg * e or y.numel() != nimg * e or gamma.numel() != e or beta.numel() != e \
            or stats.numel() != 2 * nimg:
        raise MgxError("layernorm_fwd: size mismatch")
    lib = _lib()
    ws = K.workspace(lib.mgx_layernorm_workspace(nimg, e))
    check(lib.mgx_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), _p(stats), nimg, e, float(eps),
                                ws.data_ptr(), ws.numel(), K.stream_ptr()), "mgx_layernorm_fwd")


@_hbm_op("layernorm_bwd")
def layernorm_bwd(dy, x, gamma, stats, dx, acc_x, dgamma, dbeta, nimg, e):
    if dy.numel() != nimg * e or x.numel() != nimg * e or gamma.numel() != e or stats.numel() != 2 * nimg:
        raise MgxError("layernorm_bwd: size mismatch")
    for t, n in ((dx, nimg * e), (dgamma, e), (dbeta, e)):
        if t is not None and t.numel() != n:
            raise MgxError("layernorm_bwd: size mismatch")
    lib = _lib()
    ws = K.workspace(lib.mgx_layernorm_workspace(nimg, e))
    check(lib.mgx_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(stats), _p(dx), int(acc_x), _p(dgamma), _p(dbeta),
                                nimg, e, ws.data_ptr(), ws.numel(), K.stream_ptr()), "mgx_layernorm_bwd")


@_hbm_op("mbstd_fwd")
def mbstd_fwd(x, y, stat, n, h, w, c):
    e = h * w * c
    g = min(4, n)
    if x.numel() != n * e or y.numel() != n * (e + w * c) or n % g or stat.numel() != n // g:
        raise MgxError("mbstd_fwd: size mismatch (batch must be a multiple of min(4, batch))")
    lib = _lib()
    ws = K.workspace(lib.mgx_mbstd_workspace(n, e))
    check(lib.mgx_mbstd_fwd(_p(x), _p(y), _p(stat), n, h, w, c, ws.data_ptr(), ws.numel(), K.stream_ptr()),
          "mgx_mbstd_fwd")


@_hbm_op("mbstd_bwd")
def mbstd_bwd(dy, x, dx, n, h, w, c):
    e = h * w * c
    if x.numel() != n * e or dx.numel() != n * e or dy.numel() != n * (e + w * c):
        raise MgxError("mbstd_bwd: size mismatch")
    lib = _lib()
    ws = K.workspace(lib.mgx_mbstd_workspace(n, e))
    check(lib.mgx_mbstd_bwd(_p(dy), _p(x), _p(dx), n, h, w, c, ws.data_ptr(), ws.numel(), K.stream_ptr()),
          "mgx_mbstd_bwd")


def softmax_xent(logits, onehot, prob, loss_rows, dlogits=None):
    if logits.dim() != 2 or onehot.shape != logits.shape or prob.shape != logits.shape \
            or loss_rows.numel() != logits.shape[0] or (dlogits is not None and dlogits.shape != logits.shape):
        raise MgxError("softmax_xent: shape mismatch")
    check(_lib().mgx_softmax_xent(_p(logits), _p(onehot), _p(prob), _p(loss_rows), _p(dlogits), logits.shape[0],
                                  logits.shape[1], K.stream_ptr()), "mgx_softmax_xent")


def adam_multi_m(p_table, n_table, off_table, g_flat, m_flat, v_flat, blk_tensor, blk_start, nblocks, lr_t, beta1,
                 beta2, eps):
    check(_lib().mgx_adam_multi_m(p_table.data_ptr(), n_table.data_ptr(), off_table.data_ptr(), _p(g_flat), _p(m_flat),
                                  _p(v_flat), blk_tensor.data_ptr(), blk_start.data_ptr(), nblocks, float(lr_t),
                                  float(beta1), float(beta2), float(eps), K.stream_ptr()), "mgx_adam_multi_m")


def softmax_rows(x, y):
    if x.dim() != 2 or x.shape != y.shape:
        raise MgxError("softmax_rows: expected two (rows, cols) tensors of one shape")
    check(_lib().mgx_softmax_rows(_p(x), _p(y), x.shape[0], x.shape[1], K.stream_ptr()), "mgx_softmax_rows")


def softrect_fwd(x, y, alpha):
    if x.numel() != y.numel():
        raise MgxError("softrect_fwd: size mismatch")
    check(_lib().mgx_softrect_fwd(_p(x), _p(y), x.numel(), float(alpha), K.stream_ptr()), "mgx_softrect_fwd")


def softrect_bwd(dy, x, dx, accumulate, alpha):
    if x.numel() != dy.numel() or x.numel() != dx.numel():
        raise MgxError("softrect_bwd: size mismatch")
    check(_lib().mgx_softrect_bwd(_p(dy), _p(x), _p(dx), int(accumulate), x.numel(), float(alpha), K.stream_ptr()),
          "mgx_softrect_bwd")


def dense1_fwd(x, w, b, y, n, e):
    if x.numel() != n * e or w.numel() != e or y.numel() != n or (b is not None and b.numel() != 1):
        raise MgxError("dense1_fwd: size mismatch")
    check(_lib().mgx_dense1_fwd(_p(x), _p(w), _p(b), _p(y), n, e, K.stream_ptr()), "mgx_dense1_fwd")


def dense1_bwd(dy, x, w, dx, dw, db, n, e):
    if dy.numel() != n or x.numel() != n * e or w.numel() != e:
        raise MgxError("dense1_bwd: size mismatch")
    for t, m in ((dx, n * e), (dw, e), (db, 1)):
        if t is not None and t.numel() != m:
            raise MgxError("dense1_bwd: size mismatch")
    check(_lib().mgx_dense1_bwd(_p(dy), _p(x), _p(w), _p(dx), _p(dw), _p(db), n, e, K.stream_ptr()),
          "mgx_dense1_bwd")


def logit_sums(fake, real, sums, accumulate=False):
    n = fake.numel() if fake is not None else real.numel()
    if sums.numel() != 2 or (fake is not None and real is not None and fake.numel() != real.numel()):
        raise MgxError("logit_sums: size mismatch")
    check(_lib().mgx_logit_sums(_p(fake), _p(real), n, _p(sums), int(accumulate), K.stream_ptr()), "mgx_logit_sums")


def combine_grads(out, a, b, ca, cb):
    n = out.numel()
    if a.numel() != n or (b is not None and b.numel() != n) or ca.numel() != 1 or (cb is not None and cb.numel() != 1):
        raise MgxError("combine_grads: size mismatch")
    check(_lib().mgx_combine_grads(_p(out), _p(a), _p(b), _p(ca), _p(cb), n, K.stream_ptr()), "mgx_combine_grads")


def wgan_loss(sums, n_global, mode, out4, dfake, dreal, n_local):
    if sums.numel() != 2 or (out4 is not None and out4.numel() != 4):
        raise MgxError("wgan_loss: size mismatch")
    for t in (dfake, dreal):
        if t is not None and t.numel() != n_local:
            raise MgxError("wgan_loss: size mismatch")
    check(_lib().mgx_wgan_loss(_p(sums), n_global, mode, _p(out4), _p(dfake), _p(dreal), n_local, K.stream_ptr()),
          "mgx_wgan_loss")


def adam_multi(p_table, n_table, off_table, g_flat, v_flat, blk_tensor, blk_start, nblocks, lr_t, beta2, eps):
    """lr_t: a host float, or a 1-element device tensor read when the kernel runs (captured steps)."""
    if isinstance(lr_t, torch.Tensor):
        check(_lib().mgx_adam_multi_devlr(p_table.data_ptr(), n_table.data_ptr(), off_table.data_ptr(), _p(g_flat),
                                          _p(v_flat), blk_tensor.data_ptr(), blk_start.data_ptr(), nblocks, _p(lr_t),
                                          float(beta2), float(eps), K.stream_ptr()), "mgx_adam_multi_devlr")
        return
    check(_lib().mgx_adam_multi(p_table.data_ptr(), n_table.data_ptr(), off_table.data_ptr(), _p(g_flat), _p(v_flat),
                                blk_tensor.data_ptr(), blk_start.data_ptr(), nblocks, float(lr_t), float(beta2),
                                float(eps), K.stream_ptr()), "mgx_adam_multi")


# --------------------------------------------------------------------------------------------
# bf16 path: 32 / 128-channel discriminator activations in bfloat16, float32 accumulation
# --------------------------------------------------------------------------------------------
def _pb(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if t.dtype != torch.bfloat16 or not t.is_cuda or not t.is_contiguous():
        raise MgxError("operand must be a contiguous bfloat16 device tensor")
    return t.data_ptr()


def conv_bf16_supported(d: ConvDesc) -> bool:
    return bool(_lib().mgx_conv_bf16_supported(C.byref(d)))


def _conv_bytes_bf16(d: ConvDesc, n_in: int = 1, n_out: int = 1) -> float:
    xe, ye, we = _conv_sizes(d)
    return 2.0 * (n_in * xe + n_out * ye) + 4.0 * (we + d.nbias)


def conv_gemm_bf16(d: ConvDesc, x, w, bias, y, dact=None, ymul=None, act_alpha=0.0):
    """mgx_conv_gemm_bf16: x, y (and dact, ymul) bfloat16; w, bias float32.  ymul with act_alpha > 0: ymul is the
    stored output of the activation whose derivative multiplies the result (no derivative tensor kept)."""
    xe, ye, we = _conv_sizes(d)
    if x.numel() != xe or y.numel() != ye or w.numel() < we:
        raise MgxError("conv_gemm_bf16: operand sizes do not match the descriptor")
    for t in (dact, ymul):
        if t is not None and t.numel() != ye:
            raise MgxError("conv_gemm_bf16: dact / ymul size mismatch")
    if d.nbias and (bias is None or bias.numel() != d.nbias):
        raise MgxError("conv_gemm_bf16: bias size mismatch")
    d.accumulate = 0
    d.act_alpha = float(act_alpha)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_conv_gemm_bf16(C.byref(d), _pb(x), _p(w), _p(bias) if d.nbias else None, _pb(y), _pb(dact),
                                    _pb(ymul), K.stream_ptr()), "mgx_conv_gemm_bf16")
    if e0 is not None:
        kk = 1 if (d.sh == 2 or d.dkh == 2) else d.kh
        key = "bconv_kernel<%d,%d,%d>" % (kk, kk, (4 if act_alpha > 0 else 2) if ymul is not None
                                          else (1 if (d.nbias or act_alpha > 0) else 0))
        PROFILER.stop(key, _conv_flops(d), e0,
                      _conv_bytes_bf16(d, 1, 1 + (1 if dact is not None else 0) + (1 if ymul is not None else 0)))


def conv1x1_blend_supported(d: ConvDesc) -> bool:
    return bool(_lib().mgx_conv1x1_blend_supported(C.byref(d)))


def conv1x1_blend(d: ConvDesc, x, w, bias, old, alpha, y, act_alpha):
    """mgx_conv1x1_blend_f32: y = (1 - alpha) old + alpha act(conv1x1(x)), float32 tensors, alpha on the device."""
    xe, ye, we = _conv_sizes(d)
    if x.numel() != xe or y.numel() != ye or old.numel() != ye or w.numel() < we or alpha.numel() != 1:
        raise MgxError("conv1x1_blend: operand sizes do not match the descriptor")
    if d.nbias and (bias is None or bias.numel() != d.nbias):
        raise MgxError("conv1x1_blend: bias size mismatch")
    d.accumulate = 0
    d.act_alpha = float(act_alpha)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_conv1x1_blend_f32(C.byref(d), _p(x), _p(w), _p(bias) if d.nbias else None, _p(old), _p(alpha),
                                       _p(y), K.stream_ptr()), "mgx_conv1x1_blend_f32")
    if e0 is not None:
        key = _gemm_key(d, False)
        PROFILER.stop(key, _gemm_flops(d, key), e0, _conv_bytes(d, 1, 2))


def conv1x1_blend_bf16(d: ConvDesc, x, w, bias, old, alpha, y, act_alpha):
    """mgx_conv1x1_blend_bf16: y = (1 - alpha) old + alpha act(conv1x1(x)); x, old, y bfloat16, alpha on the device."""
    xe, ye, we = _conv_sizes(d)
    if x.numel() != xe or y.numel() != ye or old.numel() != ye or w.numel() < we or alpha.numel() != 1:
        raise MgxError("conv1x1_blend_bf16: operand sizes do not match the descriptor")
    if d.nbias and (bias is None or bias.numel() != d.nbias):
        raise MgxError("conv1x1_blend_bf16: bias size mismatch")
    d.accumulate = 0
    d.act_alpha = float(act_alpha)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_conv1x1_blend_bf16(C.byref(d), _pb(x), _p(w), _p(bias) if d.nbias else None, _pb(old), _p(alpha),
                                        _pb(y), K.stream_ptr()), "mgx_conv1x1_blend_bf16")
    if e0 is not None:
        PROFILER.stop("bconv_kernel<1,1,3>", _conv_flops(d), e0, _conv_bytes_bf16(d, 1, 2))


def conv_dgrad_to_audio_bf16(d: ConvDesc, g, w, y128, act_alpha, audio_w, dx, accumulate):
    """mgx_conv_dgrad_to_audio_bf16: data gradient of a 3x3 convolution 128 -> 32 (d: its data-gradient descriptor)
    times the derivative recovered from y128 (the from-audio layer's stored output, act_alpha its alpha), contracted
    with the from-audio kernel audio_w (2, 128) into dx (n, h, w, 2) float32."""
    xe, ye, we = _conv_sizes(d)
    if g.numel() != xe or y128.numel() != ye or w.numel() < we or audio_w.numel() != 256 or dx.numel() * 64 != ye:
        raise MgxError("conv_dgrad_to_audio_bf16: operand sizes do not match the descriptor")
    if dx.dtype != torch.float32 or g.dtype != torch.bfloat16 or y128.dtype != torch.bfloat16:
        raise MgxError("conv_dgrad_to_audio_bf16: g, y128 bfloat16; dx float32")
    d.accumulate = 0
    d.act_alpha = float(act_alpha)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_conv_dgrad_to_audio_bf16(C.byref(d), _pb(g), _p(w), _pb(y128), _p(audio_w), _p(dx),
                                              1 if accumulate else 0, K.stream_ptr()), "mgx_conv_dgrad_to_audio_bf16")
    if e0 is not None:
        PROFILER.stop("bconv_kernel<3,3,6>", _conv_flops(d), e0, 2.0 * (xe + ye) + 4.0 * we + dx.numel() * 4)


def conv1x1_blend_audio_bf16(d: ConvDesc, x, w, bias, audio, audio_w, audio_b, audio_alpha, alpha, y, act_alpha):
    """mgx_conv1x1_blend_audio_bf16: y = (1 - alpha) old + alpha act(conv1x1(x)) with old = the from-audio layer
    (kernel audio_w (2,128), bias audio_b, SoftRectifier audio_alpha) of the float32 audio (n, h, w, 2), re-computed in
    the epilogue instead of read; x, y bfloat16, alpha on the device."""
    xe, ye, we = _conv_sizes(d)
    if x.numel() != xe or y.numel() != ye or w.numel() < we or alpha.numel() != 1 or d.ngemm != 128:
        raise MgxError("conv1x1_blend_audio_bf16: operand sizes do not match the descriptor")
    if audio.dtype != torch.float32 or audio.numel() * 64 != ye or audio_w.numel() != 256 or audio_b.numel() != 128:
        raise MgxError("conv1x1_blend_audio_bf16: audio must be float32 (n, h, w, 2), its kernel (2,128), its bias (128,)")
    if d.nbias and (bias is None or bias.numel() != d.nbias):
        raise MgxError("conv1x1_blend_audio_bf16: bias size mismatch")
    d.accumulate = 0
    d.act_alpha = float(act_alpha)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_conv1x1_blend_audio_bf16(C.byref(d), _pb(x), _p(w), _p(bias) if d.nbias else None, _p(audio),
                                              _p(audio_w), _p(audio_b), float(audio_alpha), _p(alpha), _pb(y),
                                              K.stream_ptr()), "mgx_conv1x1_blend_audio_bf16")
    if e0 is not None:
        PROFILER.stop("bconv_kernel<1,1,5>", _conv_flops(d), e0, _conv_bytes_bf16(d, 1, 1) + audio.numel() * 4)


def conv_pair_bf16_supported(n, h, w, kb) -> bool:
    return bool(_lib().mgx_conv_pair_bf16_supported(int(n), int(h), int(w), int(kb)))


def conv_pair_bf16(x, wa, ba, alpha_a, wb, bb, alpha_b, kb, y):
    """mgx_conv_pair_bf16: Conv2D(32, 3x3, same) + SoftRectifier then Conv2D(32, kb x kb, same) + SoftRectifier on
    a bf16 (n, h, w, 32) tensor, no tape."""
    n, h, w, c = x.shape
    if c != 32 or tuple(y.shape) != (n, h, w, 32) or x.dtype != torch.bfloat16 or y.dtype != torch.bfloat16:
        raise MgxError("conv_pair_bf16: x and y must be bf16 (n, h, w, 32) tensors of one shape")
    if wa.numel() != 9 * 1024 or wb.numel() != kb * kb * 1024 or ba.numel() != 32 or bb.numel() != 32:
        raise MgxError("conv_pair_bf16: kernels must be (3,3,32,32) and (kb,kb,32,32), biases (32,)")
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_conv_pair_bf16(n, h, w, _pb(x), _p(wa), _p(ba), float(alpha_a), _p(wb), _p(bb), float(alpha_b),
                                    int(kb), _pb(y), K.stream_ptr()), "mgx_conv_pair_bf16")
    if e0 is not None:
        px = n * h * w
        PROFILER.stop("bpair_kernel<%d>" % kb, 2.0 * (9 + kb * kb) * 1024 * px, e0, px * 128)


def audio_conv3_bf16_supported(n, h, w) -> bool:
    return bool(_lib().mgx_audio_conv3_bf16_supported(int(n), int(h), int(w)))


def audio_conv3_bf16(x, wa, ba, alpha_a, wb, bb, alpha_b, y):
    """mgx_audio_conv3_bf16: Conv2D(128, 1x1) + SoftRectifier on float32 audio (n, h, w, 2), then Conv2D(32, 3x3, same)
    + SoftRectifier -> bf16 (n, h, w, 32), the 128-channel tensor between them never stored; no tape."""
    n, h, w, c = x.shape
    if c != 2 or x.dtype != torch.float32 or tuple(y.shape) != (n, h, w, 32) or y.dtype != torch.bfloat16:
        raise MgxError("audio_conv3_bf16: x must be float32 (n, h, w, 2), y bf16 (n, h, w, 32)")
    if wa.numel() != 256 or ba.numel() != 128 or wb.numel() != 9 * 128 * 32 or bb.numel() != 32:
        raise MgxError("audio_conv3_bf16: kernels must be (1,1,2,128) and (3,3,128,32), biases (128,) and (32,)")
    e0 = PROFILER.start() if PROFILER is not None else None
    check(_lib().mgx_audio_conv3_bf16(n, h, w, _p(x), _p(wa), _p(ba), float(alpha_a), _p(wb), _p(bb), float(alpha_b),
                                      _pb(y), K.stream_ptr()), "mgx_audio_conv3_bf16")
    if e0 is not None:
        px = n * h * w
        PROFILER.stop("baudio_kernel", 2.0 * (9 * 128 * 32 + 2 * 128) * px, e0, px * (8 + 64))


def conv_chain_bf16_supported(n, h, w, ksizes) -> bool:
    ks = (C.c_int * len(ksizes))(*[int(k) for k in ksizes])
    return bool(_lib().mgx_conv_chain_bf16_supported(int(n), int(h), int(w), len(ksizes), ks))


def conv_chain_bf16(x, weights, biases, alphas, ksizes, y):
    """mgx_conv_chain_bf16: two or three Conv2D(32, k x k, same) + SoftRectifier layers on a bf16 (n, h, w, 32)
    tensor in one launch.  y is the last layer's output, or a list with one tensor per layer (mgx_conv_chain_bf16_taped:
    every layer's output is written, for forward passes that keep a tape)."""
    n, h, w, c = x.shape
    nl = len(ksizes)
    taped = isinstance(y, (list, tuple))
    outs = list(y) if taped else [y]
    if taped and len(outs) != nl:
        raise MgxError("conv_chain_bf16: one output tensor per layer")
    if c != 32 or x.dtype != torch.bfloat16 or any(tuple(t.shape) != (n, h, w, 32) or t.dtype != torch.bfloat16 for t in outs):
        raise MgxError("conv_chain_bf16: x and y must be bf16 (n, h, w, 32) tensors of one shape")
    if not (len(weights) == len(biases) == len(alphas) == nl):
        raise MgxError("conv_chain_bf16: one kernel, bias and alpha per layer")
    for k, wt, b in zip(ksizes, weights, biases):
        if wt.numel() != k * k * 1024 or b.numel() != 32:
            raise MgxError("conv_chain_bf16: kernels must be (k,k,32,32), biases (32,)")
    ks = (C.c_int * nl)(*[int(k) for k in ksizes])
    wp = (C.c_void_p * nl)(*[_p(t) for t in weights])
    bp = (C.c_void_p * nl)(*[_p(t) for t in biases])
    al = (C.c_float * nl)(*[float(a) for a in alphas])
    e0 = PROFILER.start() if PROFILER is not None else None
    if taped:
        yp = (C.c_void_p * nl)(*[_pb(t) for t in outs])
        check(_lib().mgx_conv_chain_bf16_taped(n, h, w, nl, ks, _pb(x), wp, bp, al, yp, K.stream_ptr()),
              "mgx_conv_chain_bf16_taped")
    else:
        check(_lib().mgx_conv_chain_bf16(n, h, w, nl, ks, _pb(x), wp, bp, al, _pb(y), K.stream_ptr()),
              "mgx_conv_chain_bf16")
    if e0 is not None:
        px = n * h * w
        PROFILER.stop("bchain_kernel<%s%s>" % (",".join(str(int(k)) for k in ksizes), ",taped" if taped else ""),
                      2.0 * sum(k * k for k in ksizes) * 1024 * px, e0, px * 64 * (1 + (nl if taped else 1)))


def conv_wgrad_bf16(d: ConvDesc, x, g, dw, dbias):
    xe, ye, we = _conv_sizes(d)
    if x.numel() != xe or g.numel() != ye or dw.numel() < we:
        raise MgxError("conv_wgrad_bf16: operand sizes do not match the descriptor")
    if dbias is not None and dbias.numel() != d.nbias:
        raise MgxError("conv_wgrad_bf16: bias gradient size mismatch")
    lib = _lib()
    nbytes = lib.mgx_conv_wgrad_bf16_workspace(C.byref(d))
    if nbytes <= 0:
        raise MgxError("conv_wgrad_bf16: unsupported descriptor")
    ws = K.workspace(nbytes)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(lib.mgx_conv_wgrad_bf16(C.byref(d), _pb(x), _pb(g), _p(dw), _p(dbias), ws.data_ptr(), ws.numel(),
                                  K.stream_ptr()), "mgx_conv_wgrad_bf16")
    if e0 is not None:
        PROFILER.stop("bwgrad_kernel<%d,%d>" % (d.kh, d.kw), _conv_flops(d), e0, _conv_bytes_bf16(d, 1, 1))


def conv_bwd_bf16_supported(d: ConvDesc) -> bool:
    return bool(_lib().mgx_conv_bwd_bf16_supported(C.byref(d)))


def conv_bwd_bf16(d: ConvDesc, x, g, w, dw, dbias, dx, act_alpha=0.0):
    """Input gradient and weight / bias gradient of a bf16 32 -> 32 'same' convolution in one pass over x (the layer
    input) and g (the gradient with respect to its pre-activation); d is the FORWARD descriptor.  dx is multiplied by
    the activation derivative recovered from x when act_alpha > 0; dw / dbias are added to."""
    xe, ye, we = _conv_sizes(d)
    if x.numel() != xe or g.numel() != ye or dw.numel() < we or w.numel() < we or dx.numel() != xe:
        raise MgxError("conv_bwd_bf16: operand sizes do not match the descriptor")
    if dbias is not None and dbias.numel() != d.nbias:
        raise MgxError("conv_bwd_bf16: bias gradient size mismatch")
    lib = _lib()
    nbytes = lib.mgx_conv_bwd_bf16_workspace(C.byref(d))
    if nbytes <= 0:
        raise MgxError("conv_bwd_bf16: unsupported descriptor")
    ws = K.workspace(nbytes)
    e0 = PROFILER.start() if PROFILER is not None else None
    check(lib.mgx_conv_bwd_bf16(C.byref(d), _pb(x), _pb(g), _p(w), _p(dw), _p(dbias), _pb(dx), float(act_alpha),
                                ws.data_ptr(), ws.numel(), K.stream_ptr()), "mgx_conv_bwd_bf16")
    if e0 is not None:
        PROFILER.stop("bbwd_kernel<%d,%d>" % (d.kh, d.kw), 2 * _conv_flops(d), e0, _conv_bytes_bf16(d, 2, 1))


@_hbm_op("pointwise2_bf16_fwd")
def pointwise2_bf16_fwd(x, w, bias, y, dact, npos, act_alpha):
    if x.numel() != 2 * npos or y.numel() != 128 * npos or w.numel() != 256 or (bias is not None and bias.numel() != 128) \
            or (dact is not None and dact.numel() != y.numel()):
        raise MgxError("pointwise2_bf16_fwd: size mismatch")
    check(_lib().mgx_pointwise2_bf16_fwd(_p(x), _p(w), _p(bias), _pb(y), _pb(dact), npos, float(act_alpha),
                                         K.stream_ptr()), "mgx_pointwise2_bf16_fwd")


@_hbm_op("pointwise2_bf16_wgrad")
def pointwise2_bf16_wgrad(x, g, dw, db, npos):
    if x.numel() != 2 * npos or g.numel() != 128 * npos or dw.numel() != 256 or (db is not None and db.numel() != 128):
        raise MgxError("pointwise2_bf16_wgrad: size mismatch")
    lib = _lib()
    ws = K.workspace(lib.mgx_pointwise2_bf16_wgrad_workspace(npos))
    check(lib.mgx_pointwise2_bf16_wgrad(_p(x), _pb(g), _p(dw), _p(db), ws.data_ptr(), ws.numel(), npos,
                                        K.stream_ptr()), "mgx_pointwise2_bf16_wgrad")


@_hbm_op("pointwise_to2_bf16")
def pointwise_to2_bf16(g, w, dx, npos, accumulate):
    if g.numel() != 128 * npos or dx.numel() != 2 * npos or w.numel() != 256:
        raise MgxError("pointwise_to2_bf16: size mismatch")
    check(_lib().mgx_pointwise_to2_bf16(_pb(g), _p(w), _p(dx), npos, int(accumulate), K.stream_ptr()),
          "mgx_pointwise_to2_bf16")


@_hbm_op("wsum_bf16_fwd")
def weighted_sum_bf16_fwd(a, b, alpha, y):
    if a.numel() != b.numel() or a.numel() != y.numel() or alpha.numel() != 1:
        raise MgxError("weighted_sum_bf16_fwd: size mismatch")
    check(_lib().mgx_weighted_sum_bf16_fwd(_pb(a), _pb(b), _p(alpha), _pb(y), y.numel(), K.stream_ptr()),
          "mgx_weighted_sum_bf16_fwd")


@_hbm_op("wsum_bf16_bwd")
def weighted_sum_bf16_bwd(dy, a, b, alpha, da, db, dalpha, ma=None, mb=None, act_alpha_a=0.0, act_alpha_b=0.0):
    """ma / mb: saved activation derivatives of the producers of a / b; da / db leave multiplied by them.
    act_alpha_a / act_alpha_b (instead of ma / mb): a / b are the stored outputs of the activations that produced
    them and the derivatives are recovered from them."""
    n = dy.numel()
    for t in (a, b, da, db, ma, mb):
        if t is not None and t.numel() != n:
            raise MgxError("weighted_sum_bf16_bwd: size mismatch")
    lib = _lib()
    ws = K.workspace(lib.mgx_reduce_workspace())
    if act_alpha_a > 0.0 or act_alpha_b > 0.0:
        if ma is not None or mb is not None:
            raise MgxError("weighted_sum_bf16_bwd: stored derivatives and recovered ones exclude each other")
        check(lib.mgx_weighted_sum_bf16_bwd_act(_pb(dy), _pb(a), _pb(b), _p(alpha), _pb(da), _pb(db), _p(dalpha),
                                                float(act_alpha_a), float(act_alpha_b), ws.data_ptr(), ws.numel(), n,
                                                K.stream_ptr()), "mgx_weighted_sum_bf16_bwd_act")
        return
    check(lib.mgx_weighted_sum_bf16_bwd_mul(_pb(dy), _pb(a), _pb(b), _p(alpha), _pb(da), _pb(db), _p(dalpha), _pb(ma),
                                            _pb(mb), ws.data_ptr(), ws.numel(), n, K.stream_ptr()),
          "mgx_weighted_sum_bf16_bwd_mul")


def gblock_pack_bf16_bytes() -> int:
    return int(_lib().mgx_gblock_pack_bf16_bytes())


def gblock_packt_bf16_bytes() -> int:
    return int(_lib().mgx_gblock_packt_bf16_bytes())


def gblock_side_elems_bf16(npos) -> int:
    return int(_lib().mgx_gblock_side_elems_bf16(npos))


def gblock_side_elems_bf16_lean(npos) -> int:
    """bf16 elements of a lean side buffer (no ConvT(64) / Dense(256) outputs and gradients): 1544 of 4104 rows."""
    return int(_lib().mgx_gblock_side_elems_bf16_lean(npos))



def gblock_fwd_bf16(weights, packb, x, y, nimg, hl, wl, act=None, pos0=0, npos=0, lean=False):
    """Forward of a generator grow block on bf16 MFMAs: x (nimg,hl,wl,2) -> y (nimg,2hl,2wl,2), float32; `act`
    (bf16 side buffer) receives the tape of the range; y may then be None.  lean: the tape without the tensors
    gblock_wgrad_bf16(..., weights=) recomputes."""
    n = _gblock_range(nimg, hl, wl, pos0, npos)
    if x.numel() != nimg * hl * wl * 2 or packb.numel() * packb.element_size() < gblock_pack_bf16_bytes():
        raise MgxError("gblock_fwd_bf16: size mismatch")
    if y is None and act is None:
        raise MgxError("gblock_fwd_bf16: nothing to write")
    if y is not None and y.numel() != 4 * x.numel():
        raise MgxError("gblock_fwd_bf16: output size mismatch")
    lean = bool(lean and act is not None)
    side = gblock_side_elems_bf16_lean(n) if lean else gblock_side_elems_bf16(n)
    if act is not None and act.numel() < side:
        raise MgxError("gblock_fwd_bf16: activation buffer too small")
    e0 = PROFILER.start() if PROFILER is not None else None
    fn = _lib().mgx_gblock_fwd_bf16_lean if lean else _lib().mgx_gblock_fwd_bf16
    check(fn(C.byref(weights), packb.data_ptr(), _p(x), _p(y), _pb(act), nimg, hl, wl, pos0, n, K.stream_ptr()),
          "mgx_gblock_fwd_bf16")
    if e0 is not None:
        PROFILER.stop("gblock_fwd_bf16_kernel<%s>" % ("true" if act is not None else "false"),
                      GBLOCK_FLOP_PER_POSITION * n, e0, 4.0 * 10 * n + (2.0 * side if act is not None else 0))


def gblock_dgrad_bf16(weights, packt, dy, delta, dx, nimg, hl, wl, pos0=0, npos=0, lean=False):
    n = _gblock_range(nimg, hl, wl, pos0, npos)
    if dx.numel() != nimg * hl * wl * 2 or dy.numel() != 4 * dx.numel() \
            or packt.numel() * packt.element_size() < gblock_packt_bf16_bytes():
        raise MgxError("gblock_dgrad_bf16: size mismatch")
    side = gblock_side_elems_bf16_lean(n) if lean else gblock_side_elems_bf16(n)
    if delta.numel() < side:
        raise MgxError("gblock_dgrad_bf16: gradient buffer too small")
    e0 = PROFILER.start() if PROFILER is not None else None
    fn = _lib().mgx_gblock_dgrad_bf16_lean if lean else _lib().mgx_gblock_dgrad_bf16
    check(fn(C.byref(weights), packt.data_ptr(), _p(dy), _pb(delta), _p(dx), nimg, hl, wl, pos0, n, K.stream_ptr()),
          "mgx_gblock_dgrad_bf16")
    if e0 is not None:
        PROFILER.stop("gblock_dgrad_bf16_kernel", GBLOCK_FLOP_PER_POSITION * n, e0, 4.0 * 10 * n + 2.0 * side)


def gblock_wgrad_bf16(act, delta, grads, npos, weights=None):
    """Adds the block's 58 weight and bias gradients from the two side buffers.  With `weights` (the block's
    variables, as the forward saw them) the buffers are lean ones and the operands they leave out are recomputed."""
    need = gblock_side_elems_bf16_lean(npos) if weights is not None else gblock_side_elems_bf16(npos)
    if act.numel() < need or delta.numel() < need:
        raise MgxError("gblock_wgrad_bf16: side buffer too small")
    nbytes = int(_lib().mgx_gblock_wgrad_workspace_n(npos))
    ws = K.workspace(nbytes)
    e0 = PROFILER.start() if PROFILER is not None else None
    if weights is not None:
        check(_lib().mgx_gblock_wgrad_bf16_lean(C.byref(weights), _pb(act), _pb(delta), C.byref(grads), ws.data_ptr(),
                                                ws.numel(), npos, K.stream_ptr()), "mgx_gblock_wgrad_bf16_lean")
    else:
        check(_lib().mgx_gblock_wgrad_bf16(_pb(act), _pb(delta), C.byref(grads), ws.data_ptr(), ws.numel(), npos,
                                           K.stream_ptr()), "mgx_gblock_wgrad_bf16")
    if e0 is not None:
        PROFILER.stop("gblock_wgrad_bf16_kernel<*>", GBLOCK_FLOP_PER_POSITION * npos, e0, 4.0 * need)


@_hbm_op("mul_bf16")
def mul_bf16(a, b):
    """bf16(a * b): a bfloat16 or float32, b bfloat16, same shape."""
    if a.numel() != b.numel():
        raise MgxError("mul_bf16: size mismatch")
    out = torch.empty(b.shape, dtype=torch.bfloat16, device=b.device)
    f32 = a.dtype == torch.float32
    check(_lib().mgx_mul_bf16(_p(a) if f32 else _pb(a), int(f32), _pb(b), _pb(out), b.numel(), K.stream_ptr()),
          "mgx_mul_bf16")
    return out


@_hbm_op("mul_bf16")
def mul_dact_bf16(a, y, act_alpha):
    """bf16(a * f'(y)): a bfloat16 or float32; y the stored bfloat16 output of tanh(v) + v/alpha, from which the
    derivative is recovered."""
    if a.numel() != y.numel():
        raise MgxError("mul_dact_bf16: size mismatch")
    out = torch.empty(y.shape, dtype=torch.bfloat16, device=y.device)
    f32 = a.dtype == torch.float32
    check(_lib().mgx_mul_dact_bf16(_p(a) if f32 else _pb(a), int(f32), _pb(y), float(act_alpha), _pb(out), y.numel(),
                                   K.stream_ptr()), "mgx_mul_dact_bf16")
    return out


def cast_to_bf16(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(_lib().mgx_cast_f32_to_bf16(_p(x), _pb(y), x.numel(), K.stream_ptr()), "mgx_cast_f32_to_bf16")
    return y


def cast_to_f32(x: torch.Tensor) -> torch.Tensor:
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_lib().mgx_cast_bf16_to_f32(_pb(x), _p(y), x.numel(), K.stream_ptr()), "mgx_cast_bf16_to_f32")
    return y
