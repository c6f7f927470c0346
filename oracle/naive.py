"""CPU ORACLE helper (test infrastructure only): naive numpy/loop restatement of each
layer of trainer/models.py, written independently of oracle/ref_step.py (no torch, no
library convolution) so that the two can be checked against each other on small cases.

PARITY UNPINNED: see the header of oracle/ref_step.py.
"""
from __future__ import annotations

import numpy as np


def conv2d(x, k, b, strides=(1, 1), padding="same"):
    """Keras Conv2D NHWC, kernel (kh,kw,Cin,Cout); 'same' (stride 1) pads (k-1)//2 before
    and k-1-(k-1)//2 after; 'valid' pads nothing."""
    n, h, w, ci = x.shape
    kh, kw, _, co = k.shape
    sh, sw = strides
    if padding == "same":
        assert strides == (1, 1)
        pt, pl = (kh - 1) // 2, (kw - 1) // 2
        ho, wo = h, w
    else:
        pt = pl = 0
        ho, wo = (h - kh) // sh + 1, (w - kw) // sw + 1
    y = np.zeros((n, ho, wo, co), dtype=np.float64)
    for a in range(kh):
        for c in range(kw):
            for i in range(ho):
                hi = i * sh + a - pt
                if hi < 0 or hi >= h:
                    continue
                for j in range(wo):
                    wi = j * sw + c - pl
                    if wi < 0 or wi >= w:
                        continue
                    y[:, i, j, :] += x[:, hi, wi, :].astype(np.float64) @ k[a, c].astype(np.float64)
    return y + b


def conv2d_transpose(x, k, b, strides):
    """Keras Conv2DTranspose, kernel (kh,kw,Cout,Cin) == stride, 'valid'."""
    n, h, w, ci = x.shape
    kh, kw, co, _ = k.shape
    assert (kh, kw) == tuple(strides)
    y = np.zeros((n, h * kh, w * kw, co), dtype=np.float64)
    for i in range(h):
        for j in range(w):
            for a in range(kh):
                for c in range(kw):
                    y[:, i * kh + a, j * kw + c, :] = x[:, i, j, :].astype(np.float64) @ k[a, c].astype(np.float64).T
    return y + b


def dense(x, k, b):
    return x.astype(np.float64) @ k.astype(np.float64) + b


def upsample2(x):
    n, h, w, c = x.shape
    y = np.zeros((n, 2 * h, 2 * w, c), dtype=x.dtype)
    for i in range(2 * h):
        for j in range(2 * w):
            y[:, i, j] = x[:, i // 2, j // 2]
    return y


def avgpool2(x):
    n, h, w, c = x.shape
    y = np.zeros((n, h // 2, w // 2, c), dtype=np.float64)
    for i in range(h // 2):
        for j in range(w // 2):
            y[:, i, j] = (x[:, 2 * i, 2 * j].astype(np.float64) + x[:, 2 * i, 2 * j + 1]
                          + x[:, 2 * i + 1, 2 * j] + x[:, 2 * i + 1, 2 * j + 1]) / 4.0
    return y


def layer_norm(x, gamma, beta, eps=1e-3):
    y = np.zeros(x.shape, dtype=np.float64)
    for i in range(x.shape[0]):
        s = x[i].astype(np.float64)
        m = s.sum() / s.size
        v = ((s - m) ** 2).sum() / s.size
        y[i] = (s - m) / np.sqrt(v + eps) * gamma + beta
    return y


def weighted_sum(a, b, alpha):
    return (1.0 - alpha) * a.astype(np.float64) + alpha * b.astype(np.float64)


def minibatch_stddev(x):
    """models.py:167-192 with explicit group bookkeeping: sample n belongs to group
    m = n % M (M = N / G), and the statistic of group m fills row H of every member."""
    n, h, w, c = x.shape
    g = min(4, n)
    assert n % g == 0
    m_groups = n // g
    y = np.zeros((n, h + 1, w, c), dtype=np.float64)
    y[:, :h] = x
    for m in range(m_groups):
        members = [x[gi * m_groups + m].astype(np.float64) for gi in range(g)]
        mean = sum(members) / g
        var = sum((t - mean) ** 2 for t in members) / g
        stat = np.sqrt(var + 1e-8).sum() / (h * w * c)
        for gi in range(g):
            y[gi * m_groups + m, h] = stat
    return y


def soft_rectifier(x, alpha):
    return np.tanh(x.astype(np.float64)) + x.astype(np.float64) / alpha
