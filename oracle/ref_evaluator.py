"""CPU restatement of the evaluator forward pass (trainer/evaluacion.py:27-110) and of
calculate_inception_score (trainer/utils.py:65-76).

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path.  Parity unpinned: the
reference holds no fixture for the evaluator and TensorFlow/Keras are not installed, so this
follows the source plus the documented Keras semantics (NHWC cross-correlation, 'same' padding
split (k-1)//2 before / rest after, Flatten row-major (h, w, c), softmax over the last axis).

Layer list per stage, as the reference wires it (indices are positions in `old_model.layers`,
evaluacion.py:59-63 re-uses the old model from index 3, i.e. WITHOUT its input layer, its 1x1
feature convolution and its first 3x3 convolution):
  stage 0: 1x1, then 3x[3x3, 4x4, 3x3], Flatten, Dense(9, softmax)
  stage s: 1x1, 2x[3x3, 3x3, 2x2 'same'], 3x3, 3x3, 2x2 stride 2 'valid', 1x1, then the tail of
           stage s-1 from its layer index 3.
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Spec = Tuple[int, int, int, str]          # (kh, kw, stride, padding)

BASE: List[Spec] = [(1, 1, 1, "same")] + [(3, 3, 1, "same"), (4, 4, 1, "same"), (3, 3, 1, "same")] * 3
BLOCK: List[Spec] = ([(1, 1, 1, "same")] + [(3, 3, 1, "same"), (3, 3, 1, "same"), (2, 2, 1, "same")] * 2
                     + [(3, 3, 1, "same"), (3, 3, 1, "same"), (2, 2, 2, "valid"), (1, 1, 1, "same")])


def conv_specs(stage: int) -> List[Spec]:
    """Convolutions of the stage-`stage` evaluator in execution order."""
    specs = list(BASE)
    for _ in range(stage):
        specs = list(BLOCK) + specs[2:]   # old layers[3:] = drop (input,) 1x1 and the first 3x3
    return specs


def conv_nhwc(x, k, b, stride, padding):
    """Keras Conv2D on NHWC float64 tensors, kernel (kh, kw, cin, cout)."""
    kh, kw = k.shape[0], k.shape[1]
    xt = x.permute(0, 3, 1, 2)
    if padding == "same":
        xt = F.pad(xt, ((kw - 1) // 2, kw - 1 - (kw - 1) // 2, (kh - 1) // 2, kh - 1 - (kh - 1) // 2))
    y = F.conv2d(xt, k.permute(3, 2, 0, 1), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def forward(stage: int, kernels, biases, dense_k, dense_b, x) -> Tuple[np.ndarray, np.ndarray]:
    """-> (logits, probabilities), float64.  kernels/biases follow conv_specs(stage)."""
    specs = conv_specs(stage)
    assert len(specs) == len(kernels) == len(biases)
    t = torch.as_tensor(np.asarray(x), dtype=torch.float64)
    for (kh, kw, s, pad), k, b in zip(specs, kernels, biases):
        k = torch.as_tensor(np.asarray(k), dtype=torch.float64)
        assert tuple(k.shape[:2]) == (kh, kw)
        t = conv_nhwc(t, k, torch.as_tensor(np.asarray(b), dtype=torch.float64), s, pad)
    flat = t.reshape(t.shape[0], -1)
    logits = flat @ torch.as_tensor(np.asarray(dense_k), dtype=torch.float64) + torch.as_tensor(
        np.asarray(dense_b), dtype=torch.float64)
    return logits.numpy(), torch.softmax(logits, dim=1).numpy()


def inception_score(p_yx, eps=1e-16) -> float:
    """trainer/utils.py:65-76, written with explicit loops."""
    p = np.asarray(p_yx, dtype=np.float64)
    n, c = p.shape
    p_y = [sum(p[i, j] for i in range(n)) / n for j in range(c)]
    total = 0.0
    for i in range(n):
        total += sum(p[i, j] * (np.log(p[i, j] + eps) - np.log(p_y[j] + eps)) for j in range(c))
    return float(np.exp(total / n))


def fit_steps(stage: int, kernels, biases, dense_k, dense_b, batches, lr=1e-3, beta_1=0.9, beta_2=0.999, eps=1e-7):
    """Batches of (x, one-hot y) through compile(optimizer="Adam", loss='categorical_crossentropy')
    + fit (trainer/evaluacion.py:160-161) in float64 with autograd: mean over the batch of
    -sum_c y log softmax(logits)_c (tf.keras takes the loss of a Softmax output from its logits,
    without clipping), keras' Adam update with bias-corrected step size
    lr*sqrt(1-beta_2^t)/(1-beta_1^t).  -> (losses, gradients of the LAST batch, parameters after
    the last update), parameter order: kernel0, bias0, kernel1, ..., dense kernel, dense bias."""
    specs = conv_specs(stage)
    params = []
    for k, b in zip(kernels, biases):
        params += [torch.tensor(np.asarray(k), dtype=torch.float64, requires_grad=True),
                   torch.tensor(np.asarray(b), dtype=torch.float64, requires_grad=True)]
    params += [torch.tensor(np.asarray(dense_k), dtype=torch.float64, requires_grad=True),
               torch.tensor(np.asarray(dense_b), dtype=torch.float64, requires_grad=True)]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    losses, grads = [], None
    for t, (x, y) in enumerate(batches, start=1):
        h = torch.as_tensor(np.asarray(x), dtype=torch.float64)
        for i, (kh, kw, s, pad) in enumerate(specs):
            h = conv_nhwc(h, params[2 * i], params[2 * i + 1], s, pad)
        logits = h.reshape(h.shape[0], -1) @ params[-2] + params[-1]
        loss = -(torch.as_tensor(np.asarray(y), dtype=torch.float64) * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()
        grads = torch.autograd.grad(loss, params)
        losses.append(float(loss.detach()))
        lr_t = lr * np.sqrt(1 - beta_2 ** t) / (1 - beta_1 ** t)
        with torch.no_grad():
            for q, g, mq, vq in zip(params, grads, m, v):
                mq.mul_(beta_1).add_((1 - beta_1) * g)
                vq.mul_(beta_2).add_((1 - beta_2) * g * g)
                q.sub_(lr_t * mq / (vq.sqrt() + eps))
    return losses, [g.numpy() for g in grads], [q.detach().numpy() for q in params]
