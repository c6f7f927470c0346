"""Test infrastructure (never imported by the product path): the CPU oracle of oracle/ref_step.py with
bfloat16 STORAGE emulated at the points where the device's bf16 path stores bf16, and exact (float64)
arithmetic in between.

The device path (music_generator_amd with backend.set_precision("bfloat16")) keeps the 32 / 128-channel
activations of the discriminator's grow blocks (trainer/models.py:256-281) at 8 rows and more in bf16:
  * the output y of every Conv2D + SoftRectifier there is rounded to bf16; the backward pass takes the
    activation's derivative from that STORED output, max(1 + 1/alpha - y^2, 1/alpha) (SAVE_DERIVATIVE = False,
    the device's default: no derivative tensor is kept), or, with SAVE_DERIVATIVE = True
    (backend.BF16_SAVE_DERIVATIVE on the device), from a second tensor holding 1 - tanh^2 + 1/alpha rounded
    to bf16;
  * the weights of those convolutions are rounded to bf16 as they enter the matrix unit (the from-audio
    convolution with its 2 float32 input channels runs on float32 weights);
  * the gradient with respect to a pre-activation, dY (.) f', is rounded to bf16;
  * WeightedSum on such tensors rounds its output and both input gradients.
Everything else (2-channel audio tensors, the 4 x 750 base block, weight gradients, Adam) is float32 on
the device and exact here.  Comparing the device with THIS oracle isolates the kernels' arithmetic
from the cost of bf16 storage itself, which test_gpu_bf16.py reports separately against the plain oracle.
"""
import contextlib

import torch

from . import ref_step as R

MIN_ROWS = 8
SAVE_DERIVATIVE = False


def _round(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundValue(torch.autograd.Function):
    """Rounds the value; the gradient passes unchanged (float32 master weights)."""

    @staticmethod
    def forward(ctx, x):
        return _round(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBoth(torch.autograd.Function):
    """Rounds the value going forward and the gradient going back."""

    @staticmethod
    def forward(ctx, x):
        return _round(x)

    @staticmethod
    def backward(ctx, g):
        return _round(g)


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _round(g)


class _ActStored(torch.autograd.Function):
    """tanh(x) + x/alpha with the output stored as bf16 and the incoming gradient multiplied by the derivative the
    device uses (recovered from the stored output, or a derivative stored as bf16), the product stored as bf16."""

    @staticmethod
    def forward(ctx, pre, alpha):
        t = torch.tanh(pre)
        y = _round(t + pre / alpha)
        if SAVE_DERIVATIVE:
            d = _round(1.0 - t * t + 1.0 / alpha)
        else:
            d = torch.clamp(1.0 + 1.0 / alpha - y * y, min=1.0 / alpha)
        ctx.save_for_backward(d)
        return y

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return _round(g * d), None


def _conv_sr(W, name, x, alpha, **kw):
    if x.shape[1] < MIN_ROWS:
        kshape = R._p(W, name, "kernel").shape
        if tuple(kshape) == (1, 1, 32, 128):
            # the first float32 layer after the bf16 region (the 1x1 convolution of the lowest grow block on
            # 4 rows): its input gradient goes back into the region as bf16
            x = _RoundGrad.apply(x)
        return R.soft_rectifier(R.conv2d(x, R._p(W, name, "kernel"), R._p(W, name, "bias"), **kw), alpha)
    kernel = R._p(W, name, "kernel")
    if x.shape[-1] != 2:
        kernel = _RoundValue.apply(kernel)
    return _ActStored.apply(R.conv2d(x, kernel, R._p(W, name, "bias"), **kw), alpha)


def _weighted_sum(a, b, alpha):
    if a.shape[1] < MIN_ROWS or a.shape[-1] == 2:
        return _plain_weighted_sum(a, b, alpha)
    return _RoundBoth.apply(_plain_weighted_sum(_RoundGrad.apply(a), _RoundGrad.apply(b), alpha))


_plain_weighted_sum = R.weighted_sum


@contextlib.contextmanager
def bf16_storage():
    """Within the block, oracle.ref_step evaluates the discriminator with bf16 storage emulated."""
    saved = (R._conv_sr, R.weighted_sum)
    R._conv_sr, R.weighted_sum = _conv_sr, _weighted_sum
    try:
        yield
    finally:
        R._conv_sr, R.weighted_sum = saved
