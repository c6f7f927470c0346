"""Keras-equivalent layer objects used by trainer/models.py, backed by the libmgx kernels.

Layouts follow Keras: NHWC activations; Conv2D kernel (kh,kw,Cin,Cout); Conv2DTranspose kernel
(kh,kw,Cout,Cin); Dense kernel (in,out) applied to the last axis; LayerNormalization gamma/beta
over the normalised axes.  Every contraction is expressed as one mgx_conv_desc (see
include/mgx.h): forward, data-gradient and weight-gradient differ only in the descriptor.
"""
from __future__ import annotations

import math
import zlib
from typing import List, Optional

import numpy as np
import torch

from . import backend as K
from . import ops
from .graph import Layer

_init_seed = [12345]


def set_init_seed(seed: int):
    """Seed of the (numpy) weight initialisers; TF's own RNG streams cannot be reproduced."""
    _init_seed[0] = int(seed)


def _rng(name: str):
    return np.random.default_rng((_init_seed[0] * 1000003 + zlib.crc32(name.encode())) % (2 ** 63))


def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rfs = int(np.prod(shape[:-2]))
    return shape[-2] * rfs, shape[-1] * rfs


def initial_value(initializer: str, shape, name: str) -> np.ndarray:
    """he_normal (truncated normal, stddev sqrt(2/fan_in)/0.8796) and glorot_uniform as Keras
    defines them (models.py:256 etc. pass 'he_normal'; Dense uses the glorot default)."""
    fan_in, fan_out = _fans(shape)
    rng = _rng(name)
    if initializer == "he_normal":
        std = math.sqrt(2.0 / max(1.0, fan_in)) / 0.87962566103423978
        v = rng.standard_normal(shape)
        bad = np.abs(v) > 2.0
        while bad.any():
            v[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(v) > 2.0
        return (v * std).astype(np.float32)
    if initializer == "glorot_uniform":
        lim = math.sqrt(6.0 / max(1.0, fan_in + fan_out))
        return rng.uniform(-lim, lim, shape).astype(np.float32)
    if initializer == "zeros":
        return np.zeros(shape, dtype=np.float32)
    if initializer == "ones":
        return np.ones(shape, dtype=np.float32)
    raise ValueError("unknown initializer %r" % (initializer,))


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


class _Contraction(Layer):
    """Shared machinery of Conv2D / Conv2DTranspose / Dense: descriptor caches, the fused
    tanh(x)+x/alpha epilogue and the two gradient calls."""
    fusable_activation = True

    def __init__(self, **kw):
        super().__init__(**kw)
        self._descs = {}

    def _fwd_desc(self, n, h, w):
        raise NotImplementedError

    def _dgrad_desc(self, n, h, w):
        raise NotImplementedError

    def _out_hw(self, h, w):
        raise NotImplementedError

    def _desc(self, kind, n, h, w):
        key = (kind, n, h, w)
        d = self._descs.get(key)
        if d is None:
            d = self._fwd_desc(n, h, w) if kind == "f" else self._dgrad_desc(n, h, w)
            self._descs[key] = d
        return d

    def forward(self, xs, save, act_alpha: float = 0.0):
        x = xs[0]
        n, h, w, _ = x.shape
        mode = self._bf16_mode(x, n, h, w)
        if mode is not None:
            return self._forward_bf16(mode, x, save, act_alpha)
        was_bf16 = x.dtype == torch.bfloat16
        if was_bf16:                       # this layer has no bf16 kernel: leave the bf16 region here
            x = ops.cast_to_f32(x)
        ho, wo = self._out_hw(h, w)
        y = K.empty((n, ho, wo, self.filters))
        dact = K.empty(y.shape) if (save and act_alpha > 0.0) else None
        ops.conv_gemm(self._desc("f", n, h, w), x, self.kernel.value(), self.bias.value(), y,
                      dact=dact, act_alpha=act_alpha)
        return y, ((_CastSaved((x, dact)) if was_bf16 else (x, dact)) if save else None)

    def backward(self, saved, dy, need, existing, wgrad, outmul=None):
        if isinstance(saved, _BSaved):
            return self._backward_bf16(saved, dy, need, existing, wgrad, outmul)
        x, dact = saved
        was_bf16 = isinstance(saved, _CastSaved)
        n, h, w, _ = x.shape
        if wgrad and self.trainable:
            if self.kernel.grad is None or self.bias.grad is None:
                raise RuntimeError("gradient buffers of %s are not bound" % self.name)
            ops.conv_wgrad(self._desc("f", n, h, w), x, dy, self.kernel.grad, self.bias.grad, dymul=dact)
        if not need[0]:
            return [None]
        if was_bf16:
            if existing[0] is not None:
                raise NotImplementedError("gradient fan-in into a bf16 activation")
            dx = K.empty(x.shape)
            ops.conv_gemm(self._desc("d", n, h, w), dy, self.kernel.value(), None, dx, xmul=dact)
            return [(ops.cast_to_bf16(dx), True)]
        dx = existing[0] if existing[0] is not None else K.empty(x.shape)
        ops.conv_gemm(self._desc("d", n, h, w), dy, self.kernel.value(), None, dx, xmul=dact,
                      accumulate=existing[0] is not None)
        return [(dx, True)]

    # ---- bf16 path (discriminator grow blocks) ----
    def _bf16_mode(self, x, n, h, w):
        """None, "conv" (bf16 tile kernels) or "audio" (2 float32 channels -> 128 bf16 channels)."""
        if not K.bf16_enabled() or x.dim() != 4 or h < K.BF16_MIN_ROWS or not isinstance(self, Conv2D):
            return None
        if x.dtype == torch.bfloat16:
            return "conv" if ops.conv_bf16_supported(self._desc("f", n, h, w)) else None
        if self.cin == 2 and self.filters == 128 and self.kernel_size == (1, 1):
            return "audio"
        return None

    def _forward_bf16(self, mode, x, save, act_alpha):
        n, h, w, _ = x.shape
        ho, wo = self._out_hw(h, w)
        y = K.empty((n, ho, wo, self.filters), dtype=torch.bfloat16)
        keep = save and act_alpha > 0.0
        dact = K.empty(y.shape, dtype=torch.bfloat16) if (keep and K.BF16_SAVE_DERIVATIVE) else None
        if mode == "audio":
            ops.pointwise2_bf16_fwd(x, self.kernel.value(), self.bias.value(), y, dact, n * h * w, act_alpha)
        else:
            ops.conv_gemm_bf16(self._desc("f", n, h, w), x, self.kernel.value(), self.bias.value(), y, dact=dact,
                               act_alpha=act_alpha)
        return y, (_BSaved(mode, x, dact, y if (keep and dact is None) else None, act_alpha) if save else None)

    def _backward_bf16(self, saved, dy, need, existing, wgrad, outmul):
        x = saved.x
        n, h, w, _ = x.shape
        if saved.has_derivative() and not saved.premultiplied:
            # dY (.) f': the kernel upstream could not apply it
            g = ops.mul_bf16(dy, saved.dact) if saved.dact is not None else ops.mul_dact_bf16(dy, saved.yact, saved.alpha)
        else:
            g = dy if dy.dtype == torch.bfloat16 else ops.cast_to_bf16(dy)
        do_w = wgrad and self.trainable
        if do_w and (self.kernel.grad is None or self.bias.grad is None):
            raise RuntimeError("gradient buffers of %s are not bound" % self.name)
        if saved.mode == "audio":
            if do_w:
                ops.pointwise2_bf16_wgrad(x, g, self.kernel.grad, self.bias.grad, n * h * w)
            if not need[0]:
                return [None]
            dx = existing[0] if existing[0] is not None else K.empty(x.shape)
            ops.pointwise_to2_bf16(g, self.kernel.value(), dx, n * h * w, existing[0] is not None)
            return [(dx, True)]
        fdesc = self._desc("f", n, h, w)
        fuse = do_w and need[0] and existing[0] is None and K.BF16_FUSED_CONV_BACKWARD \
            and (outmul is None or outmul.dact is None) and ops.conv_bwd_bf16_supported(fdesc)
        if do_w and not fuse:
            ops.conv_wgrad_bf16(fdesc, x, g, self.kernel.grad, self.bias.grad)
        if not need[0]:
            return [None]
        if existing[0] is not None:
            raise NotImplementedError("gradient fan-in into a bf16 activation")
        dx = K.empty(x.shape, dtype=torch.bfloat16)
        ymul, mul_alpha = None, 0.0
        if outmul is not None and outmul.has_derivative() and not outmul.premultiplied:
            # the stored derivative, or the producer's stored output (= this layer's input) it is recovered from
            ymul, mul_alpha = (outmul.dact, 0.0) if outmul.dact is not None else (outmul.yact, outmul.alpha)
            outmul.premultiplied = True      # the producer of x receives dX (.) f' and must not multiply again
        if fuse:
            # both gradients read x and g: one pass (x is the producer's stored output the derivative comes from)
            assert ymul is None or ymul is x
            ops.conv_bwd_bf16(fdesc, x, g, self.kernel.value(), self.kernel.grad, self.bias.grad, dx, mul_alpha)
        else:
            ops.conv_gemm_bf16(self._desc("d", n, h, w), g, self.kernel.value(), None, dx, ymul=ymul, act_alpha=mul_alpha)
        return [(dx, True)]


def audio_dgrad_eligible(conv, saved, audio_conv, audio_saved) -> bool:
    """`conv` (the block's first 3x3 convolution, 128 -> 32, taped on the bf16 kernels) reads the output of `audio_conv`
    (the from-audio layer, taped on the bf16 path with its output kept for the derivative): their two data gradients
    can run as one launch (mgx_conv_dgrad_to_audio_bf16) when no weight gradient is wanted."""
    if not (isinstance(saved, _BSaved) and isinstance(audio_saved, _BSaved) and isinstance(conv, Conv2D)):
        return False
    if saved.mode != "conv" or audio_saved.mode != "audio" or audio_saved.premultiplied:
        return False
    if audio_saved.dact is not None or audio_saved.yact is None or audio_saved.yact is not saved.x or not audio_saved.alpha > 0:
        return False
    if conv.cin != 128 or conv.filters != 32 or conv.kernel_size != (3, 3) or conv.strides != (1, 1) or conv.padding != "same":
        return False
    n, h, w, _ = saved.x.shape
    return ops.conv_bf16_supported(conv._desc("d", n, h, w))


def audio_dgrad_backward(conv, saved, audio_conv, audio_saved, dy, existing):
    """Gradient of the audio (n, h, w, 2) float32 from dy, the gradient at `conv`'s output: new tensor, or `existing`
    added to."""
    x = saved.x
    n, h, w, _ = x.shape
    if saved.has_derivative() and not saved.premultiplied:
        g = ops.mul_bf16(dy, saved.dact) if saved.dact is not None else ops.mul_dact_bf16(dy, saved.yact, saved.alpha)
    else:
        g = dy if dy.dtype == torch.bfloat16 else ops.cast_to_bf16(dy)
    dxa = existing if existing is not None else K.empty(tuple(audio_saved.x.shape))
    ops.conv_dgrad_to_audio_bf16(conv._desc("d", n, h, w), g, conv.kernel.value(), x, audio_saved.alpha,
                                 audio_conv.kernel.value(), dxa, existing is not None)
    return dxa


def conv_pair_eligible(a, b, x) -> bool:
    """Layers a, b (each followed by a SoftRectifier) as one launch of the pair kernel: a is Conv2D(32, 3x3, same),
    b Conv2D(32, 3x3 or 2x2, same), both on 32 channels, bf16 input of at least 16 rows, bf16 precision on."""
    if not (K.bf16_enabled() and isinstance(a, Conv2D) and isinstance(b, Conv2D)):
        return False
    if x.dim() != 4 or x.dtype != torch.bfloat16 or x.shape[-1] != 32:
        return False
    for l in (a, b):
        if l.padding != "same" or l.strides != (1, 1) or l.filters != 32 or l.cin != 32:
            return False
    if a.kernel_size != (3, 3) or b.kernel_size not in ((3, 3), (2, 2)):
        return False
    return ops.conv_pair_bf16_supported(x.shape[0], x.shape[1], x.shape[2], b.kernel_size[0])


def conv_pair_forward(a, alpha_a, b, alpha_b, x):
    y = K.empty(tuple(x.shape), dtype=torch.bfloat16)
    ops.conv_pair_bf16(x, a.kernel.value(), a.bias.value(), alpha_a, b.kernel.value(), b.bias.value(), alpha_b,
                       b.kernel_size[0], y)
    return y


def audio_conv_eligible(a, b, x) -> bool:
    """Layers a, b (each followed by a SoftRectifier) as one launch of the from-audio + 3x3 kernel: a is the
    Conv2D(128, 1x1) on the 2-channel float32 audio, b the Conv2D(32, 3x3, same) behind it; bf16 precision on."""
    if not (K.bf16_enabled() and isinstance(a, Conv2D) and isinstance(b, Conv2D)):
        return False
    if x.dim() != 4 or x.dtype != torch.float32 or x.shape[-1] != 2 or x.shape[1] < K.BF16_MIN_ROWS:
        return False
    if a.cin != 2 or a.filters != 128 or a.kernel_size != (1, 1) or a.strides != (1, 1):
        return False
    if b.cin != 128 or b.filters != 32 or b.kernel_size != (3, 3) or b.padding != "same" or b.strides != (1, 1):
        return False
    return ops.audio_conv3_bf16_supported(x.shape[0], x.shape[1], x.shape[2])


def audio_conv_forward(a, alpha_a, b, alpha_b, x):
    y = K.empty(tuple(x.shape[:3]) + (32,), dtype=torch.bfloat16)
    ops.audio_conv3_bf16(x, a.kernel.value(), a.bias.value(), alpha_a, b.kernel.value(), b.bias.value(), alpha_b, y)
    return y


def conv_blend_eligible(conv, x, old) -> bool:
    """`conv` (a 1x1 Conv2D followed by a SoftRectifier) and the WeightedSum that blends its output with `old` as one
    launch: both operands bf16 with the convolution on the bf16 tile kernel, or both float32 with the convolution on
    the float32 tile kernel; `old` of the output's shape."""
    if not isinstance(conv, Conv2D) or x.dim() != 4 or conv.kernel_size != (1, 1) or conv.strides != (1, 1):
        return False
    n, h, w, _ = x.shape
    if x.dtype != old.dtype or tuple(old.shape) != (n, h, w, conv.filters) or not old.is_contiguous():
        return False
    if x.dtype == torch.bfloat16:
        return K.bf16_enabled() and conv._bf16_mode(x, n, h, w) == "conv"
    if x.dtype == torch.float32:
        return conv._bf16_mode(x, n, h, w) is None and ops.conv1x1_blend_supported(conv._desc("f", n, h, w))
    return False


def conv_blend_forward(conv, act_alpha, x, old, alpha):
    n, h, w, _ = x.shape
    y = K.empty((n, h, w, conv.filters), dtype=x.dtype)
    blend = ops.conv1x1_blend_bf16 if x.dtype == torch.bfloat16 else ops.conv1x1_blend
    blend(conv._desc("f", n, h, w), x, conv.kernel.value(), conv.bias.value(), old, alpha, y, act_alpha)
    return y


def blend_audio_source(conv, x) -> bool:
    """`conv` (followed by a SoftRectifier) is a from-audio layer the bf16 path runs on mgx_pointwise2_bf16_fwd: its
    output can be left to the epilogue of the blend that reads it."""
    if not isinstance(conv, Conv2D) or x.dim() != 4 or x.dtype != torch.float32 or not x.is_contiguous():
        return False
    n, h, w, _ = x.shape
    return conv._bf16_mode(x, n, h, w) == "audio"


def conv_blend_audio_eligible(conv, x, audio) -> bool:
    """`conv` (the 1x1 Conv2D that ends a block, followed by a SoftRectifier) on the bf16 tile kernel with 128
    filters, and `audio` the float32 (n, h, w, 2) input of the from-audio layer whose output the blend would read."""
    if not isinstance(conv, Conv2D) or x.dim() != 4 or conv.kernel_size != (1, 1) or conv.strides != (1, 1):
        return False
    n, h, w, _ = x.shape
    if x.dtype != torch.bfloat16 or conv.filters != 128 or tuple(audio.shape) != (n, h, w, 2):
        return False
    return K.bf16_enabled() and conv._bf16_mode(x, n, h, w) == "conv"


def conv_blend_audio_forward(conv, act_alpha, x, audio_conv, audio_alpha, audio, alpha):
    n, h, w, _ = x.shape
    y = K.empty((n, h, w, conv.filters), dtype=torch.bfloat16)
    ops.conv1x1_blend_audio_bf16(conv._desc("f", n, h, w), x, conv.kernel.value(), conv.bias.value(), audio,
                                 audio_conv.kernel.value(), audio_conv.bias.value(), audio_alpha, alpha, y, act_alpha)
    return y


def conv_chain_eligible(layers, x) -> bool:
    """Two or three layers (each followed by a SoftRectifier) as one launch of the chain kernel: Conv2D(32, k x k,
    same) on 32 channels with kernel sizes {3,3,2}, {3,3} or {3,2}, bf16 input, bf16 precision on."""
    if not K.bf16_enabled() or x.dim() != 4 or x.dtype != torch.bfloat16 or x.shape[-1] != 32:
        return False
    if x.shape[1] < K.BF16_MIN_ROWS:
        return False
    ks = []
    for l in layers:
        if not isinstance(l, Conv2D) or l.padding != "same" or l.strides != (1, 1) or l.filters != 32 or l.cin != 32:
            return False
        if l.kernel_size[0] != l.kernel_size[1]:
            return False
        ks.append(l.kernel_size[0])
    if tuple(ks) not in ((3, 3, 2), (3, 3), (3, 2)):
        return False
    return ops.conv_chain_bf16_supported(x.shape[0], x.shape[1], x.shape[2], ks)


def conv_chain_forward(layers, alphas, x):
    y = K.empty(tuple(x.shape), dtype=torch.bfloat16)
    ops.conv_chain_bf16(x, [l.kernel.value() for l in layers], [l.bias.value() for l in layers], alphas,
                        [l.kernel_size[0] for l in layers], y)
    return y


def conv_chain_forward_taped(layers, alphas, x):
    """The chain as one launch that also writes the inner layers' outputs: returns the last output and the tape
    entries the layers' own forward calls would have produced (input, stored output, alpha of each)."""
    ys = [K.empty(tuple(x.shape), dtype=torch.bfloat16) for _ in layers]
    ops.conv_chain_bf16(x, [l.kernel.value() for l in layers], [l.bias.value() for l in layers], alphas,
                        [l.kernel_size[0] for l in layers], ys)
    saved = [_BSaved("conv", xin, None, yout, a) for xin, yout, a in zip([x] + ys[:-1], ys, alphas)]
    return ys[-1], saved


class _CastSaved(tuple):
    """(x, dact) of a float32 contraction whose input arrived as bf16 and was cast: the data-gradient goes
    back as bf16."""


class _BSaved:
    """Tape entry of a contraction that ran on the bf16 kernels: its input; for a fused activation either the
    saved derivative (`dact`) or the layer's stored output and the activation's alpha the derivative is recovered
    from (`yact`, `alpha`; the output is the next layer's taped input, so it costs no memory); and whether the
    gradient that will arrive has already been multiplied by that derivative (the data-gradient kernel of the
    consuming layer does it in its epilogue)."""
    __slots__ = ("mode", "x", "dact", "yact", "alpha", "premultiplied")

    def __init__(self, mode, x, dact, yact=None, alpha=0.0):
        self.mode, self.x, self.dact, self.yact, self.alpha, self.premultiplied = mode, x, dact, yact, float(alpha), False

    def has_derivative(self) -> bool:
        return self.dact is not None or self.yact is not None


class Conv2D(_Contraction):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", kernel_initializer="glorot_uniform",
                 **kw):
        super().__init__(**kw)
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.padding = padding
        self.kernel_initializer = kernel_initializer
        if padding == "same" and self.strides != (1, 1):
            raise NotImplementedError("'same' padding is implemented for stride 1 only (all the reference uses)")
        if padding == "valid" and self.strides != self.kernel_size and self.kernel_size != (1, 1):
            raise NotImplementedError("'valid' convolutions must have kernel == stride")

    def build(self, input_shape):
        cin = input_shape[-1]
        kh, kw = self.kernel_size
        self.cin = cin
        self.kernel = self.add_weight("kernel", initial_value(self.kernel_initializer, (kh, kw, cin, self.filters),
                                                              self.name + "/kernel"))
        self.bias = self.add_weight("bias", np.zeros((self.filters,), dtype=np.float32))

    def _out_hw(self, h, w):
        if self.padding == "same":
            return h, w
        kh, kw = self.kernel_size
        sh, sw = self.strides
        return (h - kh) // sh + 1, (w - kw) // sw + 1

    def compute_output_shape(self, s):
        ho, wo = self._out_hw(s[1], s[2])
        return (s[0], ho, wo, self.filters)

    def _fwd_desc(self, n, h, w):
        kh, kw = self.kernel_size
        sh, sw = self.strides
        ho, wo = self._out_hw(h, w)
        pt, pl = ((kh - 1) // 2, (kw - 1) // 2) if self.padding == "same" else (0, 0)
        return ops.make_conv_desc(nimg=n, hin=h, win=w, cin=self.cin, hout=ho, wout=wo, kh=kh, kw=kw, sh=sh, sw=sw,
                                  pt=pt, pl=pl, ngemm=self.filters, cout=self.filters, wsk=self.filters, wsn=1,
                                  wtap=self.cin * self.filters, nbias=self.filters)

    def _dgrad_desc(self, n, h, w):
        kh, kw = self.kernel_size
        sh, sw = self.strides
        ho, wo = self._out_hw(h, w)
        if (sh, sw) == (1, 1):
            # dX[p] = sum_tap dY[p - tap + pad] W[tap]^T : a 'same' gather with flipped taps
            pt, pl = ((kh - 1) // 2, (kw - 1) // 2) if self.padding == "same" else (0, 0)
            return ops.make_conv_desc(nimg=n, hin=ho, win=wo, cin=self.filters, hout=h, wout=w, kh=kh, kw=kw,
                                      pt=kh - 1 - pt, pl=kw - 1 - pl, flip=1, ngemm=self.cin, cout=self.cin,
                                      wsk=1, wsn=self.filters, wtap=self.cin * self.filters)
        # kernel == stride: every input pixel belongs to exactly one window -> 1x1 GEMM over
        # output pixels with a depth-to-space store of the (a,b,ci) columns
        if h != ho * kh or w != wo * kw:
            raise NotImplementedError("strided conv input must be a multiple of the kernel")
        return ops.make_conv_desc(nimg=n, hin=ho, win=wo, cin=self.filters, hout=ho, wout=wo,
                                  ngemm=kh * kw * self.cin, dkh=kh, dkw=kw, cout=self.cin, wsk=1, wsn=self.filters)


class Conv2DTranspose(_Contraction):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", kernel_initializer="glorot_uniform",
                 **kw):
        super().__init__(**kw)
        self.filters = int(filters)
        self.kernel_size = _pair(kernel_size)
        self.strides = _pair(strides)
        self.kernel_initializer = kernel_initializer
        if self.kernel_size != self.strides or padding != "valid":
            raise NotImplementedError("Conv2DTranspose needs kernel == stride and 'valid' (all the reference uses)")

    def build(self, input_shape):
        cin = input_shape[-1]
        kh, kw = self.kernel_size
        self.cin = cin
        self.kernel = self.add_weight("kernel", initial_value(self.kernel_initializer, (kh, kw, self.filters, cin),
                                                              self.name + "/kernel"))
        self.bias = self.add_weight("bias", np.zeros((self.filters,), dtype=np.float32))

    def _out_hw(self, h, w):
        return h * self.kernel_size[0], w * self.kernel_size[1]

    def compute_output_shape(self, s):
        return (s[0], s[1] * self.kernel_size[0], s[2] * self.kernel_size[1], self.filters)

    def _fwd_desc(self, n, h, w):
        kh, kw = self.kernel_size
        return ops.make_conv_desc(nimg=n, hin=h, win=w, cin=self.cin, hout=h, wout=w,
                                  ngemm=kh * kw * self.filters, dkh=kh, dkw=kw, cout=self.filters,
                                  wsk=1, wsn=self.cin, nbias=self.filters)

    def _dgrad_desc(self, n, h, w):
        kh, kw = self.kernel_size
        return ops.make_conv_desc(nimg=n, hin=h * kh, win=w * kw, cin=self.filters, hout=h, wout=w, kh=kh, kw=kw,
                                  sh=kh, sw=kw, ngemm=self.cin, cout=self.cin, wsk=self.cin, wsn=1,
                                  wtap=self.filters * self.cin)


class Dense(_Contraction):
    """Dense on the last axis.  Rank-4 inputs run as a 1x1 contraction; the rank-2 input with one
    unit (Flatten -> Dense(1), models.py:339-340) runs as a wavefront-reduction GEMV."""

    def __init__(self, units, kernel_initializer="glorot_uniform", **kw):
        super().__init__(**kw)
        self.filters = int(units)
        self.units = int(units)
        self.kernel_initializer = kernel_initializer

    def build(self, input_shape):
        cin = input_shape[-1]
        self.cin = cin
        self.kernel = self.add_weight("kernel", initial_value(self.kernel_initializer, (cin, self.units),
                                                              self.name + "/kernel"))
        self.bias = self.add_weight("bias", np.zeros((self.units,), dtype=np.float32))

    def _out_hw(self, h, w):
        return h, w

    def compute_output_shape(self, s):
        return tuple(s[:-1]) + (self.units,)

    def _fwd_desc(self, n, h, w):
        return ops.make_conv_desc(nimg=n, hin=h, win=w, cin=self.cin, hout=h, wout=w, ngemm=self.units,
                                  cout=self.units, wsk=self.units, wsn=1, nbias=self.units)

    def _dgrad_desc(self, n, h, w):
        return ops.make_conv_desc(nimg=n, hin=h, win=w, cin=self.units, hout=h, wout=w, ngemm=self.cin,
                                  cout=self.cin, wsk=1, wsn=self.units)

    def forward(self, xs, save, act_alpha: float = 0.0):
        x = xs[0]
        if x.dim() == 4:
            return super().forward(xs, save, act_alpha)
        n, e = x.shape
        if self.units == 1:
            y = K.empty((n, 1))
            ops.dense1_fwd(x, self.kernel.value(), self.bias.value(), y, n, e)
            if act_alpha > 0.0:
                pre = y
                y = K.empty((n, 1))
                ops.softrect_fwd(pre, y, act_alpha)
                return y, ((x, pre, act_alpha) if save else None)
            return y, ((x, None, 0.0) if save else None)
        y4, saved = super().forward([x.view(n, 1, 1, e)], save, act_alpha)
        return y4.view(n, self.units), saved

    def backward(self, saved, dy, need, existing, wgrad, outmul=None):
        if len(saved) == 2:
            x, _ = saved
            if x.dim() == 4 and dy.dim() == 4:
                return super().backward(saved, dy, need, existing, wgrad)
            n = x.shape[0]
            res = super().backward(saved, dy.view(n, 1, 1, self.units), need,
                                   [e.view(x.shape) if e is not None else None for e in existing], wgrad)
            return [None if r is None else (r[0].view(n, -1), r[1]) for r in res]
        x, pre, act_alpha = saved
        n, e = x.shape
        if pre is not None:
            dpre = K.empty((n, 1))
            ops.softrect_bwd(dy, pre, dpre, False, act_alpha)
            dy = dpre
        do_w = wgrad and self.trainable
        dx = None
        if need[0]:
            if existing[0] is not None:
                raise NotImplementedError("Dense(1) input gradient cannot accumulate")
            dx = K.empty((n, e))
        if do_w or dx is not None:
            ops.dense1_bwd(dy, x, self.kernel.value(), dx, self.kernel.grad if do_w else None,
                           self.bias.grad if do_w else None, n, e)
        return [(dx, True) if dx is not None else None]


class UpSampling2D(Layer):
    def __init__(self, size=(2, 2), **kw):
        super().__init__(**kw)
        if _pair(size) != (2, 2):
            raise NotImplementedError("only the default size (2,2) is used by the reference")

    def compute_output_shape(self, s):
        return (s[0], s[1] * 2, s[2] * 2, s[3])

    def forward(self, xs, save):
        x = xs[0]
        n, h, w, c = x.shape
        y = K.empty((n, 2 * h, 2 * w, c))
        ops.upsample2x_fwd(x, y, n, h, w, c)
        return y, (tuple(x.shape) if save else None)

    def backward(self, saved, dy, need, existing, wgrad):
        if not need[0]:
            return [None]
        n, h, w, c = saved
        dx = existing[0] if existing[0] is not None else K.empty(saved)
        ops.upsample2x_bwd(dy, dx, n, h, w, c, existing[0] is not None)
        return [(dx, True)]


class AveragePooling2D(Layer):
    def __init__(self, pool_size=(2, 2), **kw):
        super().__init__(**kw)
        if _pair(pool_size) != (2, 2):
            raise NotImplementedError("only the default pool size (2,2) is used by the reference")

    def compute_output_shape(self, s):
        return (s[0], s[1] // 2, s[2] // 2, s[3])

    def forward(self, xs, save):
        x = xs[0]
        n, h, w, c = x.shape
        y = K.empty((n, h // 2, w // 2, c))
        ops.avgpool2x_fwd(x, y, n, h, w, c)
        return y, (tuple(x.shape) if save else None)

    def backward(self, saved, dy, need, existing, wgrad):
        if not need[0]:
            return [None]
        n, h, w, c = saved
        dx = existing[0] if existing[0] is not None else K.empty(saved)
        ops.avgpool2x_bwd(dy, dx, n, h, w, c, existing[0] is not None)
        return [(dx, True)]


class Add(Layer):
    def compute_output_shape(self, shapes):
        return shapes[0]

    def _merge_function(self, inputs):
        y = K.empty(inputs[0].shape)
        ops.add_n(inputs, y)
        return y

    def forward(self, xs, save):
        return self._merge_function(xs), (len(xs) if save else None)

    def backward(self, saved, dy, need, existing, wgrad):
        out = []
        for nd, ex in zip(need, existing):
            if not nd:
                out.append(None)
            elif ex is not None:
                ops.add_inplace(ex, dy)
                out.append((ex, True))
            else:
                out.append((dy, False))      # alias: the executor never accumulates into it
        return out


class Flatten(Layer):
    def compute_output_shape(self, s):
        n = 1
        for d in s[1:]:
            n *= d
        return (s[0], n)

    def forward(self, xs, save):
        x = xs[0]
        return x.view(x.shape[0], -1), (tuple(x.shape) if save else None)

    def backward(self, saved, dy, need, existing, wgrad):
        if not need[0]:
            return [None]
        g = dy.view(saved)
        if existing[0] is not None:
            ops.add_inplace(existing[0], g)
            return [(existing[0], True)]
        return [(g, True)]


class LayerNormalization(Layer):
    """LayerNormalization(axis=[1,2,3]) with Keras defaults (epsilon 1e-3, gamma=1, beta=0)."""

    def __init__(self, axis=-1, epsilon=1e-3, **kw):
        super().__init__(**kw)
        ax = list(axis) if isinstance(axis, (list, tuple)) else [axis]
        if sorted(ax) != [1, 2, 3]:
            raise NotImplementedError("only axis=[1,2,3] is used by the reference")
        self.epsilon = float(epsilon)

    def build(self, input_shape):
        shp = tuple(input_shape[1:])
        self.gamma = self.add_weight("gamma", np.ones(shp, dtype=np.float32))
        self.beta = self.add_weight("beta", np.zeros(shp, dtype=np.float32))

    def forward(self, xs, save):
        x = xs[0]
        n = x.shape[0]
        e = x.numel() // n
        y = K.empty(x.shape)
        stats = K.empty((n, 2))
        ops.layernorm_fwd(x, self.gamma.value(), self.beta.value(), y, stats, n, e, self.epsilon)
        return y, ((x, stats) if save else None)

    def backward(self, saved, dy, need, existing, wgrad):
        x, stats = saved
        n = x.shape[0]
        e = x.numel() // n
        do_w = wgrad and self.trainable
        dx = None
        if need[0]:
            dx = existing[0] if existing[0] is not None else K.empty(x.shape)
        if dx is not None or do_w:
            ops.layernorm_bwd(dy, x, self.gamma.value(), stats, dx, existing[0] is not None,
                              self.gamma.grad if do_w else None, self.beta.grad if do_w else None, n, e)
        return [(dx, True) if dx is not None else None]
