"""Shared helpers of the parity tests: comparison metric and a one-layer model runner."""
import numpy as np
import torch


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))


def assert_close(a, b, rtol=1e-5, what=""):
    """fp32 device result `a` against fp64 oracle `b`: relative L2 error <= rtol and max abs
    error <= 20*rtol*max|b| (+1e-7)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, "%s: shape %s vs %s" % (what, a.shape, b.shape)
    assert np.isfinite(a).all(), "%s: non-finite values" % what
    err = rel_l2(a, b)
    assert err <= rtol, "%s: rel-L2 error %.3e > %.1e" % (what, err, rtol)
    mx = float(np.abs(a - b).max()) if a.size else 0.0
    lim = 20 * rtol * float(np.abs(b).max() if b.size else 0.0) + 1e-7
    assert mx <= lim, "%s: max abs error %.3e > %.3e" % (what, mx, lim)


def run_single(model, x, dy, need_input_grad=True):
    """forward + backward of a Model on the device; returns (y, dx, {var name: grad})."""
    y, tape = model.forward(x, save=True)
    flat = model.bind_grads()
    flat.zero_()
    dyt = torch.from_numpy(np.ascontiguousarray(dy, dtype=np.float32)).to(y.device)
    dx = model.backward(tape, dyt, need_input_grad=need_input_grad)
    torch.cuda.synchronize()
    grads = {v.name: v.grad.detach().cpu().numpy().reshape(v.shape).copy() for v in model.trainable_variables}
    return y.cpu().numpy(), (dx.cpu().numpy() if dx is not None else None), grads


def randomize(model, seed=0, scale=None):
    """Give every variable of the model a reproducible non-trivial value; returns the dict."""
    out = {}
    for i, v in enumerate(sorted(model.trainable_variables, key=lambda t: t.name)):
        rng = np.random.default_rng(seed * 7919 + i)
        kind = v.name.rsplit("/", 1)[1]
        if kind == "kernel":
            fan = max(1, int(np.prod(v.shape[:-1])))
            val = rng.standard_normal(v.shape) * (scale if scale is not None else 1.0 / np.sqrt(fan))
        elif kind == "gamma":
            val = 1.0 + rng.uniform(-0.3, 0.3, v.shape)
        elif kind == "alpha":
            val = np.asarray(0.37)
        else:
            val = rng.uniform(-0.3, 0.3, v.shape)
        v.assign(val.astype(np.float32))
        out[v.name] = v.numpy()
    return out
