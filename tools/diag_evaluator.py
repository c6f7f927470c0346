"""Diagnostic: per-parameter gradient error of one evaluator fitting step against the float64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import music_generator_amd as M
from music_generator_amd import evaluacion as E, ops, backend as K
from oracle import ref_evaluator as RE
from helpers import rel_l2

M.clear_session()
model = E.define_evaluator(1)[0]
rng = np.random.default_rng(77)
convs = [l for l in model.layers if isinstance(l, M.layers.Conv2D)]
kernels, biases = [], []
for l in convs:
    fan_in = l.kernel_size[0] * l.kernel_size[1] * l.kernel.shape[2]
    k = (rng.standard_normal(l.kernel.shape) / np.sqrt(fan_in)).astype(np.float32)
    b = (0.05 * rng.standard_normal(l.bias.shape)).astype(np.float32)
    l.kernel.assign(k), l.bias.assign(b)
    kernels.append(k), biases.append(b)
head = model.layers[-1]
dk = (rng.standard_normal(head.kernel.shape) / np.sqrt(head.kernel.shape[0])).astype(np.float32)
db = (0.1 * rng.standard_normal(head.bias.shape)).astype(np.float32)
head.kernel.assign(dk), head.bias.assign(db)
x = rng.uniform(-1, 1, (4, 4, 750, 2)).astype(np.float32)
y = np.eye(9, dtype=np.float32)[rng.integers(0, 9, 4)]
xb, yb = model._to_device(x), E.labels_to_device(y)
flat = model.bind_grads(0); flat.zero_()
logits, tape = model.forward(xb, save=True)
prob, rows, dlogits = K.empty(logits.shape), K.empty((4,)), K.empty(logits.shape)
ops.softmax_xent(logits.contiguous(), yb, prob, rows, dlogits)
model.backward(tape, dlogits, need_input_grad=False)
grads = [v.grad.cpu().numpy().reshape(v.shape).copy() for l in convs + [head] for v in (l.kernel, l.bias)]
_, ref_grads, _ = RE.fit_steps(0, kernels, biases, dk, db, [(x, y)])
names = [n for l in convs + [head] for n in (l.name + "/kernel", l.name + "/bias")]
for n, g, rg in zip(names, grads, ref_grads):
    print("%-28s %-20s rel_l2 %.3e  |ref| %.3e" % (n, g.shape, rel_l2(g, rg), np.abs(rg).max()))
