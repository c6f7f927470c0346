"""Construction API and training step of the progressive GAN, mirroring trainer/models.py so that
trainer/task.py-style drivers can switch to this package unchanged:

    define_discriminator, add_discriminator_block, define_generator, add_generator_block,
    define_composite, WGAN, WeightedSum, MinibatchStdDev, SoftRectifier, StaticOptTanh,
    discriminator_loss, generator_loss, Adam, GANMonitor

Every numerical operation runs in a hand-written gfx950 kernel behind the C-ABI of include/mgx.h;
this file only wires layer objects and sequences kernel launches.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import backend
from . import backend as K
from . import ops
from . import _cabi
from .graph import FusedGroup, Input, Layer, Model
from .layers import (Add, AveragePooling2D, Conv2D, Conv2DTranspose, Dense, Flatten, LayerNormalization,
                     UpSampling2D)

# --------------------------------------------------------------------------------------------
# custom layers of trainer/models.py
# --------------------------------------------------------------------------------------------


class WeightedSum(Add):
    """Fade-in blend ((1-a)*x0 + a*x1), trainer/models.py:23-35.  As in the reference the blend
    factor is a variable created with backend.variable, hence trainable unless
    `alpha.trainable` is cleared (WGAN(trainable_alpha=False))."""

    def __init__(self, alpha=0.0, **kwargs):
        super().__init__(**kwargs)
        self.alpha = backend.Variable(self.name + "/alpha", np.asarray(alpha, dtype=np.float32))
        self.weights.append(self.alpha)

    def _merge_function(self, inputs):
        assert len(inputs) == 2                                   # models.py:32
        a, b = inputs
        if a.dtype != b.dtype:            # one side left the bf16 region: blend in float32
            a = a if a.dtype == torch.float32 else ops.cast_to_f32(a)
            b = b if b.dtype == torch.float32 else ops.cast_to_f32(b)
        y = K.empty(a.shape, dtype=a.dtype)
        if a.dtype == torch.bfloat16:
            ops.weighted_sum_bf16_fwd(a, b, self.alpha.value(), y)
        else:
            ops.weighted_sum_fwd(a, b, self.alpha.value(), y)
        return y

    def forward(self, xs, save):
        return self._merge_function(xs), ((xs[0], xs[1]) if save else None)

    # the executor hands backward() the tape entries of the layers that produced the two inputs (outmuls)
    takes_outmuls = True
    # in passes that keep no tape the executor may let the convolution that produces the second input blend in its
    # epilogue (mgx_conv1x1_blend_bf16) instead of running this layer
    is_weighted_sum = True

    def backward(self, saved, dy, need, existing, wgrad, outmuls=(None, None)):
        a, b = saved
        dalpha = self.alpha.grad if (wgrad and self.trainable and self.alpha.trainable) else None
        if dalpha is None and wgrad and self.trainable and self.alpha.trainable:
            raise RuntimeError("gradient buffer of %s is not bound" % self.alpha.name)
        if a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16:
            if existing[0] is not None or existing[1] is not None:
                raise NotImplementedError("gradient fan-in into a bf16 activation")
            g = dy if dy.dtype == torch.bfloat16 else ops.cast_to_bf16(dy)
            da = K.empty(a.shape, dtype=torch.bfloat16) if need[0] else None
            db = K.empty(b.shape, dtype=torch.bfloat16) if need[1] else None
            # a producer's activation derivative is applied here instead of in a multiply launch of its own: the
            # saved tensor, or recovered from the producer's stored output, which is this layer's input a / b
            mul, mal = [None, None], [0.0, 0.0]
            for k, (om, d_, inp) in enumerate(zip(outmuls, (da, db), (a, b))):
                if d_ is None or om is None or not om.has_derivative() or om.premultiplied:
                    continue
                if om.dact is not None:
                    mul[k] = om.dact
                elif om.yact is inp:
                    mal[k] = om.alpha
            if any(m is not None for m in mul):          # the kernel takes one kind of factor per call
                mal = [0.0, 0.0]
            for k, om in enumerate(outmuls):
                if mul[k] is not None or mal[k] > 0.0:
                    om.premultiplied = True
            ops.weighted_sum_bf16_bwd(g, a, b, self.alpha.value(), da, db, dalpha, ma=mul[0], mb=mul[1],
                                      act_alpha_a=mal[0], act_alpha_b=mal[1])
            return [(da, True) if da is not None else None, (db, True) if db is not None else None]
        mixed = (a.dtype, b.dtype)
        if a.dtype != b.dtype:
            a = a if a.dtype == torch.float32 else ops.cast_to_f32(a)
            b = b if b.dtype == torch.float32 else ops.cast_to_f32(b)
        da = db = None
        if need[0]:
            da = existing[0] if existing[0] is not None else K.empty(a.shape)
        if need[1]:
            db = existing[1] if existing[1] is not None else K.empty(b.shape)
        ops.weighted_sum_bwd(dy, a, b, self.alpha.value(), da, existing[0] is not None, db,
                             existing[1] is not None, dalpha)
        if da is not None and mixed[0] == torch.bfloat16:
            da = ops.cast_to_bf16(da)
        if db is not None and mixed[1] == torch.bfloat16:
            db = ops.cast_to_bf16(db)
        return [(da, True) if da is not None else None, (db, True) if db is not None else None]

    def get_config(self):
        cfg = super().get_config()
        cfg.update({"alpha": float(self.alpha.numpy())})
        return cfg


class MinibatchStdDev(Layer):
    """trainer/models.py:162-192.  The data is NHWC although the reference's comments assume
    NCHW, so the group statistic becomes an extra ROW: (N,H,W,C) -> (N,H+1,W,C)."""

    # The tensor that arrives holds this many independent batches one after the other (WGAN's merged passes:
    # the generated and the real batch in one pass); the group statistic never mixes them.
    sub_batches = 1

    def compute_output_shape(self, s):
        return (s[0], s[1] + 1, s[2], s[3])

    def forward(self, xs, save):
        x = xs[0]
        nt, h, w, c = x.shape
        sub = MinibatchStdDev.sub_batches
        if nt % sub:
            raise ValueError("MinibatchStdDev: %d samples do not split into %d batches" % (nt, sub))
        n = nt // sub
        g = min(4, n)
        if n % g:
            raise ValueError("MinibatchStdDev: batch %d is not divisible by the group size %d" % (n, g))
        y = K.empty((nt, h + 1, w, c))
        stat = K.empty((nt // g,))
        for k in range(sub):
            ops.mbstd_fwd(x[k * n:(k + 1) * n], y[k * n:(k + 1) * n], stat[k * (n // g):(k + 1) * (n // g)], n, h, w, c)
        return y, ((x, sub) if save else None)

    def backward(self, saved, dy, need, existing, wgrad):
        if not need[0]:
            return [None]
        x, sub = saved
        nt, h, w, c = x.shape
        n = nt // sub
        dx = K.empty(x.shape)
        for k in range(sub):
            ops.mbstd_bwd(dy[k * n:(k + 1) * n], x[k * n:(k + 1) * n], dx[k * n:(k + 1) * n], n, h, w, c)
        if existing[0] is not None:
            ops.add_inplace(existing[0], dx)
            return [(existing[0], True)]
        return [(dx, True)]


class _TanhPlusLinear(Layer):
    is_soft_rectifier = True
    alpha_value = 1.0

    def forward(self, xs, save):
        x = xs[0]
        y = K.empty(x.shape)
        ops.softrect_fwd(x, y, self.alpha_value)
        return y, (x if save else None)

    def backward(self, saved, dy, need, existing, wgrad):
        if not need[0]:
            return [None]
        dx = existing[0] if existing[0] is not None else K.empty(saved.shape)
        ops.softrect_bwd(dy, saved, dx, existing[0] is not None, self.alpha_value)
        return [(dx, True)]


class SoftRectifier(_TanhPlusLinear):
    """tanh(x) + x/start_alpha, trainer/models.py:199-211."""

    def __init__(self, start_alpha=20000.0, **kwargs):
        super().__init__(**kwargs)
        self.start_alpha = start_alpha
        self.alpha_value = float(start_alpha)

    def get_config(self):
        cfg = super().get_config()
        cfg.update({"start_alpha": self.start_alpha})
        return cfg


class StaticOptTanh(_TanhPlusLinear):
    """tanh(x) + x/alpha, trainer/models.py:214-225."""

    def __init__(self, alpha=1500.0, **kwargs):
        super().__init__(**kwargs)
        self.alpha = alpha
        self.alpha_value = float(alpha)

    def get_config(self):
        cfg = super().get_config()
        cfg.update({"alpha": self.alpha})
        return cfg


# --------------------------------------------------------------------------------------------
# discriminator (trainer/models.py:229-354)
# --------------------------------------------------------------------------------------------
# (StaticOptTanh alpha, SoftRectifier alpha) keyed on the height of the model being grown
_D_ALPHAS = {8: (60000.0, 40000.0), 16: (90000.0, 80000.0), 32: (120000.0, 120000.0),
             64: (150000.0, 150000.0), 128: (200000.0, 11000.0), 256: (350000.0, 200000.0)}


def _conv_act(t, filters, ksize, soft_alpha, strides=(1, 1), padding="same"):
    t = Conv2D(filters, ksize, strides=strides, padding=padding, kernel_initializer="he_normal")(t)
    return SoftRectifier(start_alpha=soft_alpha)(t)


def _replay_tail(old_model, t, n_input_layers, alpha):
    """Re-apply old_model.layers[n_input_layers:] to t, swapping the final StaticOptTanh for a
    new one (models.py:284-288 / 299-303)."""
    final = None
    for layer in old_model.layers[n_input_layers:]:
        if isinstance(layer, StaticOptTanh):
            final = StaticOptTanh(alpha=alpha)(t)
        else:
            t = layer(t)
    return final


def add_discriminator_block(old_model, n_input_layers=3):
    in_shape = list(old_model.input[0].shape)
    alpha, soft_alpha = _D_ALPHAS.get(in_shape[-3], (40000.0, 20000.0))
    in_image = Input(shape=(in_shape[-3] * 2, in_shape[-2] * 2, in_shape[-1]))
    # new from-audio 1x1 conv, then three conv groups; the last conv of the third group halves
    # the resolution (2x2 stride 2, 'valid'); a 1x1 conv restores 128 features
    t = _conv_act(in_image, 128, (1, 1), soft_alpha)
    for group in range(3):
        t = _conv_act(t, 32, (3, 3), soft_alpha)
        t = _conv_act(t, 32, (3, 3), soft_alpha)
        if group < 2:
            t = _conv_act(t, 32, (2, 2), soft_alpha)
        else:
            t = _conv_act(t, 32, (2, 2), soft_alpha, strides=(2, 2), padding="valid")
    block_new = _conv_act(t, 128, (1, 1), soft_alpha)
    straight = Model(in_image, _replay_tail(old_model, block_new, n_input_layers, alpha))
    # fade-in twin: the old from-audio conv + activation on the average-pooled input, blended
    downsample = AveragePooling2D()(in_image)
    block_old = old_model.layers[2](old_model.layers[1](downsample))
    blended = WeightedSum()([block_old, block_new])
    fadein = Model(in_image, _replay_tail(old_model, blended, n_input_layers, alpha))
    return [straight, fadein]


def define_discriminator(n_blocks, input_shape=(4, 750, 2)):
    in_image = Input(shape=input_shape)
    t = Conv2D(128, (1, 1), padding="same", kernel_initializer="he_normal")(in_image)
    t = SoftRectifier()(t)
    for group in range(3):
        for ksize in ((3, 3), (4, 4), (3, 3)):
            t = Conv2D(32, ksize, padding="same", kernel_initializer="he_normal")(t)
            t = SoftRectifier()(t)
    t = MinibatchStdDev()(t)
    t = Flatten()(t)
    t = Dense(1)(t)
    base = Model(in_image, StaticOptTanh(alpha=40000.0)(t))
    model_list = [[base, base]]
    for i in range(1, n_blocks):
        model_list.append(add_discriminator_block(model_list[i - 1][0]))
    return model_list


# --------------------------------------------------------------------------------------------
# generator (trainer/models.py:358-463)
# --------------------------------------------------------------------------------------------
def _g_tail(t):
    for units in (256, 64, 8, 2):
        t = Dense(units)(t)
    return t


def _register_fused_block(block_end, upsampling, for_sum_layer):
    """Describe the sub-graph block_end -> for_sum_layer of add_generator_block to the executor so
    that forwards without a tape run it as the single fused kernel mgx_gblock_fwd_f32."""
    nodes, t = [], for_sum_layer
    tail = []
    for _ in range(4):                       # Dense(2) <- Dense(8) <- Dense(64) <- Dense(256)
        tail.append(t.node)
        t = t.node.inputs[0]
    add_node = t.node
    branch_nodes = []
    for bt in add_node.inputs:               # Dense(128) <- ConvT(1,2) <- Dense(32) <- ConvT(2,1)
        chain, u = [], bt
        for _ in range(4):
            chain.append(u.node)
            u = u.node.inputs[0]
        branch_nodes.append(chain[::-1])
    feat_node = branch_nodes[0][0].inputs[0].node
    nodes = [feat_node] + [n for ch in branch_nodes for n in ch] + [add_node] + tail
    tail_layers = [n.layer for n in tail[::-1]]   # Dense(256), Dense(64), Dense(8), Dense(2)
    state = {}

    layer_fields = [(feat_node.layer, "k0", "b0", None)]
    for k, ch in enumerate(branch_nodes):
        for name_k, name_b, node in (("k1", "b1", ch[0]), ("d1", "c1", ch[1]), ("k2", "b2", ch[2]), ("d2", "c2", ch[3])):
            layer_fields.append((node.layer, name_k, name_b, k))
    for i, layer in enumerate(tail_layers):
        layer_fields.append((layer, "w%d" % (i + 1), "e%d" % (i + 1), None))

    def pointers(of):
        w = _cabi.GBlockWeights()
        for layer, name_k, name_b, k in layer_fields:
            pk, pb = of(layer.kernel).data_ptr(), of(layer.bias).data_ptr()
            if k is None:
                setattr(w, name_k, pk)
                setattr(w, name_b, pb)
            else:
                getattr(w, name_k)[k] = pk
                getattr(w, name_b)[k] = pb
        return w

    def weights():
        if "w" not in state:
            state["w"] = pointers(lambda v: v.value())
        return state["w"]

    # Every forward / backward call re-packs the block's weights into a scratch buffer before its kernel
    # reads them.  The streams of two lanes are not ordered against each other between fork() and join(), so
    # each lane packs into a buffer of its own: no lane shares writable scratch with another.
    def pack():
        key = ("pack", K.current_lane())
        if key not in state:
            state[key] = K.empty((ops.gblock_pack_floats(),))
        return state[key]

    def packb(x):
        key = ("packb", K.current_lane())
        if key not in state:
            state[key] = torch.empty((ops.gblock_pack_bf16_bytes(),), dtype=torch.uint8, device=x.device)
        return state[key]

    def run(x):
        w = weights()
        n, hl, wl, _ = x.shape
        y = K.empty((n, 2 * hl, 2 * wl, 2))
        if K.bf16_enabled():
            ops.gblock_fwd_bf16(w, packb(x), x, y, n, hl, wl)
        else:
            ops.gblock_fwd(w, pack(), x, y, n, hl, wl)
        return y

    def run_save(x):
        """Taped forward.  With enough memory the block's intermediate tensors are kept (one
        tile-major buffer, bf16 under the bf16 precision); above FusedGroup.side_budget_bytes only the
        2-channel input is kept and the backward re-computes them range by range."""
        w = weights()
        n, hl, wl, _ = x.shape
        bf = K.bf16_enabled()
        lean = bf and K.GBLOCK_LEAN_TAPE
        side = (ops.gblock_side_elems_bf16_lean if lean else ops.gblock_side_elems_bf16)(n * hl * wl) if bf \
            else ops.gblock_side_floats(n * hl * wl)
        budget = FusedGroup.side_budget_bytes
        side_bytes = (2 if bf else 4) * side
        if budget is None and side_bytes > FusedGroup.auto_budget_from_bytes:
            # no bound given: keep the tape only if it, the gradient side buffer of the same size the backward
            # will want and what the caller announced it will allocate in between (FusedGroup.reserve_bytes: the
            # discriminator's tape in the generator update) leave room (the full-resolution stage at the
            # reference's batch of 2 needs 2 x 94 GiB in float32); otherwise bound both to 1/16 of the device
            # memory and re-compute.  Decided once per shape: a later call (the capture of a HIP graph after the
            # eager warm-up, say) must take the path whose scratch sizes the first one established.
            key = ("bound", n, hl, wl, bf, lean)
            if key not in state:
                avail, total = K.available_bytes()
                need = 2 * side_bytes + FusedGroup.reserve_bytes
                state[key] = max(total // 16, 1 << 20) if need > (0.6 if not FusedGroup.reserve_bytes else 0.75) * avail \
                    else None
                FusedGroup.decisions.append((side_bytes, state[key]))
                del FusedGroup.decisions[:-64]
            budget = state[key]
        if budget is not None and side_bytes > budget:
            return run(x), (None, x, (n, hl, wl), bf, budget, lean)
        y = K.empty((n, 2 * hl, 2 * wl, 2))
        act = K.empty((side,), dtype=torch.bfloat16 if bf else torch.float32)
        if bf:
            ops.gblock_fwd_bf16(w, packb(x), x, y, n, hl, wl, act=act, lean=lean)
        else:
            ops.gblock_fwd(w, pack(), x, y, n, hl, wl, act=act)
        return y, (act, None, (n, hl, wl), bf, None, lean)

    def backward(saved, dy, need_weight_grads):
        act, x, (n, hl, wl), bf, budget, lean = saved
        w = weights()
        key = ("packtb" if bf else "packt", K.current_lane())
        if key not in state:
            state[key] = torch.empty((ops.gblock_packt_bf16_bytes(),), dtype=torch.uint8, device=dy.device) if bf \
                else K.empty((ops.gblock_packt_floats(),))
        packt = state[key]
        # lean (bf16, K.GBLOCK_LEAN_TAPE when the tape was written): side buffers without the four tensors the
        # weight-gradient kernels recompute from their neighbours
        fwd = (lambda **kw: ops.gblock_fwd_bf16(w, packb(dy), x, None, n, hl, wl, lean=lean, **kw)) if bf else \
              (lambda **kw: ops.gblock_fwd(w, pack(), x, None, n, hl, wl, **kw))
        if bf:
            dgrad = lambda *a, **kw: ops.gblock_dgrad_bf16(*a, lean=lean, **kw)
            wgrad = lambda a, d, g, cnt: ops.gblock_wgrad_bf16(a, d, g, cnt, weights=w if lean else None)
        else:
            dgrad, wgrad = ops.gblock_dgrad, ops.gblock_wgrad
        side_of = (ops.gblock_side_elems_bf16_lean if lean else ops.gblock_side_elems_bf16) if bf else ops.gblock_side_floats
        grads = None
        if need_weight_grads:
            for layer, _, _, _ in layer_fields:
                if layer.kernel.grad is None or layer.bias.grad is None:
                    raise RuntimeError("gradient buffers of %s are not bound" % layer.name)
            grads = pointers(lambda v: v.grad)
        dx = K.empty((n, hl, wl, 2))
        total = n * hl * wl
        if act is not None:
            delta = K.empty((side_of(total),), dtype=torch.bfloat16 if bf else torch.float32)
            dgrad(w, packt, dy, delta, dx, n, hl, wl)
            if grads is not None:
                wgrad(act, delta, grads, total)
            return dx
        # re-computation: ranges of positions sized so that both side buffers fit the budget
        esize = 2 if bf else 4
        per_pos = esize * side_of(128) // 128
        step = max(128, (budget // (2 * per_pos)) // 128 * 128)
        step = min(step, (total + 127) // 128 * 128)
        elems = side_of(step)
        bufs = K.side_buffers((elems + 1) // 2 if bf else elems)
        if bf:
            bufs = [b.view(torch.bfloat16) for b in bufs]
        for pos0 in range(0, total, step):
            cnt = min(step, total - pos0)
            if grads is not None:
                fwd(act=bufs[0], pos0=pos0, npos=cnt)
            dgrad(w, packt, dy, bufs[1], dx, n, hl, wl, pos0=pos0, npos=cnt)
            if grads is not None:
                wgrad(bufs[0], bufs[1], grads, cnt)
        return dx

    group = FusedGroup(block_end, for_sum_layer, nodes, [upsampling.node], run)
    group.run_save, group.backward = run_save, backward
    for_sum_layer.fused_group = group


def add_generator_block(old_model):
    block_end = old_model.layers[-2].output          # last Dense(2), before its LayerNormalization
    upsampling = UpSampling2D()(block_end)
    featured = Conv2D(128, (2, 2), strides=(2, 2), padding="valid", kernel_initializer="he_normal")(upsampling)
    branches = []
    for _ in range(6):
        g = Conv2DTranspose(16, (2, 1), strides=(2, 1), padding="valid", kernel_initializer="he_normal")(featured)
        g = Dense(32)(g)
        g = Conv2DTranspose(64, (1, 2), strides=(1, 2), padding="valid", kernel_initializer="he_normal")(g)
        branches.append(Dense(128)(g))
    for_sum_layer = _g_tail(Add()(branches))
    _register_fused_block(block_end, upsampling, for_sum_layer)
    model1 = Model(old_model.input, LayerNormalization(axis=[1, 2, 3])(for_sum_layer))
    merged = WeightedSum()([upsampling, for_sum_layer])
    model2 = Model(old_model.input, LayerNormalization(axis=[1, 2, 3])(merged))
    return [model1, model2]


def _register_head_group(ly0, featured, branches, summed):
    """Describe the sub-graph latent -> Add of define_generator to the executor: its first seventeen layers
    (Conv2D(128,(1,2)) and, per branch, ConvT(4) -> Dense(8) -> ConvT(16) -> Dense(32)) run as one launch
    (mgx_ghead_fwd_f32) and their gradients as two (mgx_ghead_bwd_f32); the ConvT(64,(1,15)), Dense(128) and Add
    that follow keep their own kernels."""
    feat_node = featured.node
    chains = []
    for bt in branches:
        chain, u = [], bt
        for _ in range(6):                  # Dense(128) <- ConvT(64) <- Dense(32) <- ConvT(16) <- Dense(8) <- ConvT(4)
            chain.append(u.node)
            u = u.node.inputs[0]
        if u is not featured:
            return
        chains.append(chain[::-1])
    add_node = summed.node
    nodes = [feat_node] + [n for ch in chains for n in ch] + [add_node]
    fields = [(feat_node.layer, "k0", "b0", None)]
    for k, ch in enumerate(chains):
        for name_k, name_b, node in (("k1", "b1", ch[0]), ("d1", "c1", ch[1]), ("k2", "b2", ch[2]), ("d2", "c2", ch[3])):
            fields.append((node.layer, name_k, name_b, k))
    expect = {"k0": (1, 2, 2, 128), "k1": (2, 1, 4, 128), "d1": (4, 8), "k2": (2, 1, 16, 8), "d2": (16, 32)}
    for layer, name_k, _, _ in fields:
        if tuple(layer.kernel.shape) != expect[name_k]:
            return                          # not the reference's head: leave it to the layer kernels
    tails = [(ch[4].layer, ch[5].layer) for ch in chains]          # ConvT(64,(1,15)), Dense(128)
    add_layer = add_node.layer
    state = {}

    def pointers(of):
        w = _cabi.GHeadWeights()
        for layer, name_k, name_b, k in fields:
            pk, pb = of(layer.kernel).data_ptr(), of(layer.bias).data_ptr()
            if k is None:
                setattr(w, name_k, pk)
                setattr(w, name_b, pb)
            else:
                getattr(w, name_k)[k] = pk
                getattr(w, name_b)[k] = pb
        return w

    def weights():
        if "w" not in state:
            state["w"] = pointers(lambda v: v.value())
        return state["w"]

    def forward(z, save):
        n = z.shape[0]
        ys = [K.empty((n, 4, 50, 32)) for _ in range(4)]
        ops.ghead_fwd(weights(), z, ys)
        outs, saved = [], []
        for (convt, dense), y in zip(tails, ys):
            t, s1 = convt.forward([y], save)
            t, s2 = dense.forward([t], save)
            outs.append(t)
            saved.append((s1, s2))
        out, s_add = add_layer.forward(outs, save)
        return out, ((z, saved, s_add) if save else None)

    def backward(saved_all, dy, need_weight_grads):
        z, saved, s_add = saved_all
        dbr = add_layer.backward(s_add, dy, [True] * 4, [None] * 4, need_weight_grads)
        dys = []
        for (convt, dense), (s1, s2), r in zip(tails, saved, dbr):
            g = dense.backward(s2, r[0], [True], [None], need_weight_grads)[0][0]
            dys.append(convt.backward(s1, g, [True], [None], need_weight_grads)[0][0])
        grads = None
        if need_weight_grads:
            for layer, _, _, _ in fields:
                if layer.kernel.grad is None or layer.bias.grad is None:
                    raise RuntimeError("gradient buffers of %s are not bound" % layer.name)
            grads = pointers(lambda v: v.grad)
        dz = K.empty(z.shape)
        ops.ghead_bwd(weights(), z, dys, grads, dz)
        return dz

    group = FusedGroup(ly0, summed, nodes, [], lambda z: forward(z, False)[0])
    group.run_save = lambda z: forward(z, True)
    group.backward = backward
    summed.fused_group = group


def define_generator(n_blocks):
    ly0 = Input(shape=(1, 100, 2))
    featured = Conv2D(128, (1, 2), strides=(1, 2), padding="valid", kernel_initializer="he_normal")(ly0)
    branches = []
    for _ in range(4):
        g = featured
        for filters, ksize, units in ((4, (2, 1), 8), (16, (2, 1), 32), (64, (1, 15), 128)):
            g = Conv2DTranspose(filters, ksize, strides=ksize, padding="valid", kernel_initializer="he_normal")(g)
            g = Dense(units)(g)
        branches.append(g)
    summed = Add()(branches)
    _register_head_group(ly0, featured, branches, summed)
    out = LayerNormalization(axis=[1, 2, 3])(_g_tail(summed))
    model = Model(ly0, out)
    model_list = [[model, model]]
    for i in range(1, n_blocks):
        model_list.append(add_generator_block(model_list[i - 1][0]))
    return model_list


# --------------------------------------------------------------------------------------------
# losses (trainer/models.py:468-484).  Host-callable versions for (N,1) logit tensors; the
# training step uses the same kernel to also emit d(loss)/d(logit).
# --------------------------------------------------------------------------------------------
def _loss(fake_logits, real_logits, mode):
    fake = fake_logits if isinstance(fake_logits, torch.Tensor) else torch.as_tensor(np.asarray(fake_logits, np.float32))
    real = real_logits if isinstance(real_logits, torch.Tensor) else torch.as_tensor(np.asarray(real_logits, np.float32))
    fake = fake.to(K.device(), torch.float32).contiguous().view(-1)
    real = real.to(K.device(), torch.float32).contiguous().view(-1)
    sums, out4 = K.empty((2,)), K.empty((4,))
    ops.logit_sums(fake, real, sums)
    ops.wgan_loss(sums, fake.numel(), mode, out4, None, None, 0)
    return out4[0]


def discriminator_loss(fake_logits, real_logits):
    return _loss(fake_logits, real_logits, 0)


def generator_loss(fake_logits, real_logits):
    return _loss(fake_logits, real_logits, 1)


# --------------------------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------------------------
class Adam:
    """keras.optimizers.Adam, the update of tf.keras OptimizerV2 (non-amsgrad).  As configured by
    define_composite (models.py:502-503) beta_1 = 0, so the first moment equals the gradient and is
    not stored; any other beta_1 (keras' default 0.9, the evaluator's optimiser) keeps one.
    One fused multi-tensor kernel updates every variable of a model."""

    def __init__(self, lr=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, learning_rate=None):
        self.lr = float(learning_rate if learning_rate is not None else lr)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self._state = None

    def _build(self, model: Model):
        layout = model.grad_layout()
        flat = model.bind_grads()
        dev = flat.device
        ptrs, ns, offs, blk_t, blk_s = [], [], [], [], []
        for i, (v, off, n) in enumerate(layout):
            ptrs.append(v.value().data_ptr()); ns.append(n); offs.append(off)
            for start in range(0, n, 4096):
                blk_t.append(i); blk_s.append(start)
        self._state = dict(
            model=model, v=torch.zeros_like(flat), m=torch.zeros_like(flat) if self.beta_1 != 0.0 else None,
            p=torch.tensor(ptrs, dtype=torch.int64, device=dev), n=torch.tensor(ns, dtype=torch.int64, device=dev),
            off=torch.tensor(offs, dtype=torch.int64, device=dev),
            blk_t=torch.tensor(blk_t, dtype=torch.int32, device=dev),
            blk_s=torch.tensor(blk_s, dtype=torch.int64, device=dev), nblocks=len(blk_t))

    def next_step_size(self) -> float:
        """Count one more update and return its lr_t = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t)."""
        self.iterations += 1
        return self.lr * math.sqrt(1.0 - self.beta_2 ** self.iterations) / (1.0 - self.beta_1 ** self.iterations)

    def apply_model_gradients(self, model: Model, lr_dev: Optional[torch.Tensor] = None):
        """Adam.apply_gradients(zip(grads, model.trainable_variables)) for gradients held in the
        model's flat gradient buffer.  `lr_dev` (a 1-element device tensor holding lr_t): the update
        reads its step size from there and the iteration count is the caller's business (a step
        captured into a HIP graph; WGAN writes next_step_size() into the tensor before each replay)."""
        if self._state is None:
            self._build(model)
        st = self._state
        if st["model"] is not model:
            raise RuntimeError("an Adam instance is bound to the first model it updates")
        if st["m"] is not None:
            if lr_dev is not None:
                raise NotImplementedError("device-resident step sizes are implemented for beta_1 = 0")
            ops.adam_multi_m(st["p"], st["n"], st["off"], model.bind_grads(), st["m"], st["v"], st["blk_t"],
                             st["blk_s"], st["nblocks"], self.next_step_size(), self.beta_1, self.beta_2, self.epsilon)
            return
        lr_t = lr_dev if lr_dev is not None else self.next_step_size()
        ops.adam_multi(st["p"], st["n"], st["off"], model.bind_grads(), st["v"], st["blk_t"], st["blk_s"],
                       st["nblocks"], lr_t, self.beta_2, self.epsilon)

    def set_second_moments(self, model: Model, values: Dict[str, np.ndarray], iterations: int):
        """Load the second-moment estimate of every trainable variable of `model` (by variable name) and the
        count of updates already applied: the state a resumed or externally warmed optimiser continues from."""
        if self._state is None:
            self._build(model)
        if self._state["model"] is not model:
            raise RuntimeError("an Adam instance is bound to the first model it updates")
        flat = np.zeros(self._state["v"].numel(), np.float32)
        for var, off, n in model.grad_layout():
            a = np.asarray(values[var.name], np.float32).reshape(-1)
            if a.size != n:
                raise ValueError("second moment of %s has %d elements, the variable %d" % (var.name, a.size, n))
            flat[off:off + n] = a
        self._state["v"].copy_(torch.from_numpy(flat))
        self.iterations = int(iterations)

    def second_moments(self) -> Dict[str, np.ndarray]:
        if self._state is None:
            return {}
        v = self._state["v"].cpu().numpy()
        return {var.name: v[off:off + n].reshape(var.shape).copy() for var, off, n in self._state["model"].grad_layout()}


# --------------------------------------------------------------------------------------------
# WGAN (trainer/models.py:42-158)
# --------------------------------------------------------------------------------------------
class History:
    def __init__(self):
        self.history: Dict[str, List[float]] = {}


class _MicroBatch:
    """Samples a * m + m0 + b (a < g, b < cnt) of a batch of g * m: cnt whole MinibatchStdDev groups."""
    __slots__ = ("g", "m", "m0", "cnt", "_index")

    def __init__(self, g, m, m0, cnt):
        self.g, self.m, self.m0, self.cnt, self._index = g, m, m0, cnt, None

    @property
    def index(self):
        """The sample numbers as a device tensor (only the kept-logits diagnostics use it)."""
        if self._index is None:
            idx = (np.arange(self.g)[:, None] * self.m + self.m0 + np.arange(self.cnt)[None, :]).reshape(-1)
            self._index = torch.as_tensor(idx, dtype=torch.int64, device=K.device())
        return self._index


class WGAN:
    def __init__(self, discriminator, generator, latent_dim, fade_in=False, discriminator_extra_steps=3,
                 gp_weight=10.0, trainable_alpha=True):
        self.discriminator = discriminator
        self.generator = generator
        self.latent_dim = latent_dim
        self.d_steps = discriminator_extra_steps
        self.gp_weight = gp_weight            # stored and never used, as in the reference
        self.actual_step = 0
        self.total_steps = 0
        self.fade_in = fade_in
        self.train_steps = None
        self.dist = None                      # parallel.DataParallel when sharded over GPUs
        self.last_losses = None
        self.micro_batch = None               # samples per micro-batch (multiple of 4); None = whole batch
        self.generator_chunk = None           # samples per generator backward chunk (see train_step)
        # models.py:126, 141 and 152 evaluate discriminator(real_images) three times with the same
        # weights (the discriminator only changes in its own update).  True evaluates it once and
        # re-uses the logit sum: same values bit for bit, two discriminator forwards less per step.
        # Off by default so that a step runs every pass the reference runs.
        self.reuse_real_logits = False
        # True: train_step captures the whole step into one HIP graph (per batch shape) the second
        # time it sees a shape and replays it from then on; see _graphed_step.  None (default): decided
        # per call, on for the small batches whose step is bound by the launch rate (up to GRAPH_AUTO_BATCH
        # samples; the reference's own batch sizes are 16 and fewer) whenever nothing in the way of a
        # capture is attached (see _graph_wanted).  False: never.
        self.use_graph = None
        # True: the discriminator step keeps its logits (last_logits = {"fake", "real"}: (n,) device tensors in
        # batch order) for comparison against a reference; off by default (one gather per step)
        self.keep_logits = False
        self.last_logits = None
        # None (default): batches of up to OVERLAP_AUTO_BATCH samples run every discriminator pass over the real
        # batch on a second stream beside the pass over the generated batch (see _overlap); True / False force it
        self.overlap_passes = None
        # True: whole batches of up to MERGE_AUTO_BATCH samples hand the discriminator the generated and the real
        # batch as one pass of 2n samples (see _merge: faster eagerly at the low stages, slower as a replayed graph)
        self.merge_passes = False
        # True (default): in the small-batch step (see _overlap) the generator forwards of metric block 0 and of the
        # generator update run on a third stream beside the discriminator update
        self.early_generator = True
        self._both: Dict = {}                 # (2n, H, W, 2) input buffers of the merged passes, by batch shape
        self._graphs: Dict = {}
        self._scalars = None                  # device [alpha, lr_t of D, lr_t of G] while capturing
        self._mb_cache: Dict = {}
        self._ones_cache: Dict = {}
        self._latents: List = []
        self._rng: Optional[torch.Generator] = None
        for model in (discriminator, generator):
            for layer in model.layers:
                if isinstance(layer, WeightedSum):
                    layer.alpha.trainable = bool(trainable_alpha)

    def compile(self, d_optimizer, g_optimizer, d_loss_fn=None, g_loss_fn=None):
        self.d_optimizer = d_optimizer
        self.g_optimizer = g_optimizer
        self.d_loss_fn = d_loss_fn
        self.g_loss_fn = g_loss_fn

    def set_train_steps(self, size):
        self.train_steps = size

    # -- latent vectors: drawn on the device, or injected for parity runs --
    def inject_latents(self, zs: Sequence):
        """Queue latent batches; the next draws of train_step consume them in order (D step,
        metric 0, G step, metric 1)."""
        self._latents.extend(zs)

    def seed(self, seed: int):
        self._rng = torch.Generator(device=K.device())
        self._rng.manual_seed(int(seed))

    def _draw(self, n):
        if self._latents:
            z = self._latents.pop(0)
            z = z if isinstance(z, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(z, dtype=np.float32))
            return z.to(K.device(), torch.float32).contiguous()
        return torch.randn((n,) + tuple(self.latent_dim), dtype=torch.float32, device=K.device(), generator=self._rng)

    # -- micro-batching: whole MinibatchStdDev groups (members m, M+m, 2M+m, 3M+m), g-major --
    def _micro_batches(self, n):
        """The micro-batches of a local batch of n as _MicroBatch records (None = the whole batch)."""
        mb = self.micro_batch
        if not mb or mb >= n:
            return [None]
        g = min(4, n)
        if n % g or mb % g:
            raise ValueError("batch %d / micro-batch %d must be multiples of the MinibatchStdDev group size %d"
                             % (n, mb, g))
        key = (n, mb)
        if key not in self._mb_cache:
            m, per = n // g, mb // g
            self._mb_cache[key] = [_MicroBatch(g, m, m0, min(per, m - m0)) for m0 in range(0, m, per)]
        return self._mb_cache[key]

    @staticmethod
    def _take(t, mbatch):
        return t if mbatch is None else ops.gather_groups(t, mbatch.g, mbatch.m, mbatch.m0, mbatch.cnt)

    def _ones(self, n):
        t = self._ones_cache.get(n)
        if t is None:
            t = torch.ones((n, 1), dtype=torch.float32, device=K.device())
            self._ones_cache[n] = t
        return t

    def _finish_sums(self, sums, mode, n_local, want_coef):
        """(all-reduce the logit sums) -> loss kernel.  Returns (out4, dfake[1], dreal[1]): the two
        coefficients are d(loss)/d(logit), uniform over the batch (trainer/models.py:468-484)."""
        n_global = n_local
        if self.dist is not None:
            self.dist.all_reduce(sums)
            n_global = n_local * self.dist.world_size
        out4 = K.empty((4,))
        dfake = K.empty((1,)) if want_coef else None
        dreal = K.empty((1,)) if (want_coef and mode == 0) else None
        ops.wgan_loss(sums, n_global, mode, out4, dfake, dreal, 1 if want_coef else 0)
        return out4, dfake, dreal

    def _metric_block(self, real, n, batches, real_sum=None, fake=None):
        """models.py:124-130 / 150-156: fresh latents, forward only.  `real_sum` (1 float on the
        device) replaces the real pass with an already known sum of real logits; `fake`: the generated batch of
        this block, already computed (whole batch only)."""
        D, G = self.discriminator, self.generator
        z = self._draw(n) if fake is None else None
        sums = K.zeros((2,))
        for idx in batches:
            lane = real_logits = None
            if real_sum is None and self._overlap(n):
                lane = K.fork()
                with lane:
                    real_logits = D(self._take(real, idx)).view(-1)
            fake_logits = D(G(self._take(z, idx)) if fake is None else fake)
            if lane is not None:
                lane.join()
            elif real_sum is None:
                real_logits = D(self._take(real, idx)).view(-1)
            ops.logit_sums(fake_logits.view(-1), real_logits, sums, accumulate=True)
        if real_sum is not None:
            sums[1:2].copy_(real_sum)
        out4, _, _ = self._finish_sums(sums, 0, n, False)
        return out4

    def train_step(self, real_images):
        """One G+D step (trainer/models.py:72-158).  Because d(loss)/d(logit) is the same constant
        for every fake (resp. real) sample, each micro-batch is back-propagated with unit seeds as
        soon as it has been forwarded; the fake-pass and real-pass gradient sums are scaled by the
        two constants once the batch-wide sign is known.  A step therefore costs the same number
        of passes with or without micro-batching, and needs one micro-batch of activations."""
        if isinstance(real_images, tuple):
            real_images = real_images[0]
        alpha = self._fade_alpha()              # validates set_train_steps before anything touches the device
        real = self.discriminator._to_device(real_images)
        if self._graph_wanted(real):
            return self._graphed_step(real)
        if alpha is not None:
            for layer in self._fade_layers():
                backend.set_value(layer.alpha, alpha)
        out = self._step_body(real)
        self.actual_step += 1
        return out

    GRAPH_AUTO_BATCH = 16
    OVERLAP_AUTO_BATCH = 16
    SCALAR_SLOTS = 8            # pinned slots for the per-step scalars of a replayed graph (see _graphed_step)
    # the automatic graph mode stops below the full-resolution stage (batch x rows x columns of the audio): a
    # captured step keeps its activations in a memory pool of its own, which at 256 x 48000 does not fit beside
    # the eager step's, and a step of a second gains nothing from replay
    GRAPH_AUTO_POSITIONS = 8 << 20

    def _overlap(self, n) -> bool:
        """Whether the discriminator passes over the real batch run on the second stream beside the passes
        over the generated batch: by default for whole batches small enough to leave the chip idle."""
        if self.dist is not None or self.micro_batch or ops.PROFILER is not None:
            return False
        if self.overlap_passes is not None:
            return bool(self.overlap_passes)
        return n <= self.OVERLAP_AUTO_BATCH

    MERGE_AUTO_BATCH = 16

    def _merge(self, n) -> bool:
        """Whether the discriminator sees the generated and the real batch as ONE pass of 2n samples (its only
        cross-sample layer, MinibatchStdDev, keeps them apart): half the launches of every discriminator pass,
        which is what a small batch's step time consists of.  The backward is seeded with the per-sample
        d(loss)/d(logit), so the gradient comes out combined.  Opt-in (`merge_passes = True`): it shortens the
        EAGER step of the reference's batch sizes at the low stages (stage 0, batch 16: 8.2 -> 6.7 ms; stage 1,
        batch 8: 15.0 -> 11.6), but the replayed graph, which is what such a step runs as by default, is faster
        with the two passes side by side on two streams (4.9 against 5.4 ms), and from stage 4 up the taped pass
        over the real half of the generator step costs more than the launches saved."""
        if (self.dist is not None or self.micro_batch or self.keep_logits or self.reuse_real_logits
                or self.generator_chunk or n > self.MERGE_AUTO_BATCH):
            return False
        return bool(self.merge_passes)

    def _optimizers_replayable(self) -> bool:
        """A captured step reads its Adam step sizes from device memory, which the update kernel offers for
        beta_1 = 0 only (define_composite's setting, models.py:502-503)."""
        return all(getattr(o, "beta_1", None) == 0.0 for o in (self.d_optimizer, self.g_optimizer))

    def _graph_wanted(self, real) -> bool:
        if self.use_graph is not None:
            return bool(self.use_graph)
        return (real.shape[0] <= self.GRAPH_AUTO_BATCH and real.numel() // 2 <= self.GRAPH_AUTO_POSITIONS
                and self.dist is None and self.d_steps == 1 and self._optimizers_replayable()
                and not self._latents and ops.PROFILER is None and not self.keep_logits
                and not self.micro_batch and not torch.cuda.is_current_stream_capturing())

    def _fade_alpha(self):
        """models.py:78-82: alpha = actual_step / (train_steps - 1) on fade-in models."""
        if not self.fade_in:
            return None
        if self.train_steps is None or self.train_steps < 2:
            raise ValueError("fade-in needs set_train_steps(n) with n >= 2 (alpha = step/(n-1))")
        return self.actual_step / float(self.train_steps - 1)

    def _fade_layers(self):
        return [layer for model in (self.discriminator, self.generator) for layer in model.layers
                if isinstance(layer, WeightedSum)]

    def _graphed_step(self, real):
        """The step as one HIP graph.  A small step is bound by the cost of enqueueing its ~10^3
        kernels from the host, not by the GPU; captured once, the whole step is a single launch.
        What changes from step to step lives in device memory the graph reads: the batch (copied
        into a fixed buffer), three scalars (fade-in alpha and the two Adam step sizes, written
        before every replay) and the state of the latent generator (advanced by the graph).  The
        first step of a batch shape runs eagerly (it sizes every cache and workspace), the second
        one is captured.  The returned metric tensors are the graph's own: the next step overwrites
        them."""
        if self.dist is not None:
            raise NotImplementedError("graph replay covers the single-GPU step; collectives are not captured")
        if self.d_steps != 1:
            raise NotImplementedError("graph replay is written for one discriminator update per step "
                                      "(define_composite's setting)")
        if self._latents:
            raise RuntimeError("injected latents cannot be used with use_graph (the graph draws its own)")
        if ops.PROFILER is not None:
            raise RuntimeError("per-launch profiling and graph capture exclude each other")
        if not self._optimizers_replayable():
            raise NotImplementedError("graph replay needs both optimisers with beta_1 = 0 (the update kernel reads "
                                      "its step size from device memory only without a first moment)")
        # a graph is valid for one batch shape, one set of step-shaping options and one generation of the
        # scratch buffers whose addresses its launches carry
        key = (tuple(real.shape), self.micro_batch, self.generator_chunk, self.reuse_real_logits,
               FusedGroup.side_budget_bytes, K.precision(), self._overlap(real.shape[0]),
               self._merge(real.shape[0]), self.early_generator)
        entry = self._graphs.get(key)
        alpha = self._fade_alpha()
        if entry is None:                       # first sight of this shape: ordinary step, caches warm up
            self._graphs[key] = {"graph": None}
            if alpha is not None:
                for layer in self._fade_layers():
                    backend.set_value(layer.alpha, alpha)
            out = self._step_body(real)
            self.actual_step += 1
            return out
        if entry["graph"] is not None and entry["generation"] != K.buffer_generation():
            # a scratch buffer moved since the capture (a larger shape, another model, clear_session): the
            # old graph would write through stale addresses, so it is dropped and this shape captured again
            entry["graph"] = None
        if entry["graph"] is None:
            if "real" not in entry:
                entry["real"] = torch.empty_like(real)
                entry["scal"] = torch.zeros((3,), dtype=torch.float32, device=real.device)
                # The per-step scalars travel through pinned host memory with an asynchronous copy.  The host may
                # run several replays ahead of the GPU (nothing in a replay waits for it), so one slot would be
                # rewritten before the copy of an earlier step has read it: a ring of slots, each guarded by an
                # event recorded behind its copy, and a slot is reused only after that event has passed.
                entry["host"] = torch.zeros((self.SCALAR_SLOTS, 3), dtype=torch.float32).pin_memory()
                entry["host_events"] = [None] * self.SCALAR_SLOTS
                entry["host_next"] = 0
            graph = torch.cuda.CUDAGraph()
            if self._rng is not None:
                graph.register_generator_state(self._rng)
            torch.cuda.synchronize()
            self._scalars = entry["scal"]
            generation = K.buffer_generation()
            try:
                with torch.cuda.graph(graph):
                    if alpha is not None:
                        for layer in self._fade_layers():
                            layer.alpha.value().copy_(entry["scal"][0:1])
                    entry["out"] = self._step_body(entry["real"])
                    # a stream forked inside the capture and not joined back makes hipStreamEndCapture fail (or
                    # crash): whatever the step left open is joined here
                    K.join_all()
            finally:
                self._scalars = None
            if K.buffer_generation() != generation:
                raise RuntimeError("a scratch buffer was reallocated while the step was being captured; the "
                                   "eager step before the capture must see the same shapes")
            entry["graph"], entry["generation"] = graph, generation
            entry["scratch"] = K.live_scratch()      # referenced as long as this graph may replay
        slot = entry["host_next"]
        entry["host_next"] = (slot + 1) % self.SCALAR_SLOTS
        if entry["host_events"][slot] is not None:
            entry["host_events"][slot].synchronize()
        host = entry["host"][slot]
        host[0] = 0.0 if alpha is None else alpha
        host[1] = self.d_optimizer.next_step_size()
        host[2] = self.g_optimizer.next_step_size()
        entry["scal"].copy_(host, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        entry["host_events"][slot] = ev
        entry["real"].copy_(real)
        entry["graph"].replay()
        self.actual_step += 1
        return entry["out"]

    def _step_body(self, real):
        """The device work of one step on a batch resident in HBM."""
        D, G = self.discriminator, self.generator
        scal = self._scalars
        n = real.shape[0]
        if self._merge(n):
            return self._step_body_merged(real)
        batches = self._micro_batches(n)

        # ---- discriminator update(s) ----
        early = (self.early_generator and self._overlap(n) and self.d_steps == 1 and batches == [None]
                 and not self.generator_chunk)
        lane_g = early_fake = early_gen = early_tape = None
        for _ in range(self.d_steps):
            z = self._draw(n)
            if early:
                # The generated batches of metric block 0 and of the generator update only need the generator as it
                # is now (the discriminator update does not touch it): they are drawn now, in the order the step
                # draws them anyway, and computed on a third stream beside the discriminator update, which leaves
                # CUs idle at these sizes.
                z_m0, z_g = self._draw(n), self._draw(n)
                lane_g = K.fork(2)
                with lane_g:
                    early_fake, _ = G.forward(z_m0, save=False)
                    early_gen, early_tape = G.forward(z_g, save=True)
            flat_f = D.bind_grads(0)
            flat_f.zero_()
            flat_r = D.bind_grads(1)
            flat_r.zero_()
            sums = K.zeros((2,))
            pending = pending_r = None
            kept = [] if self.keep_logits else None
            for bi, idx in enumerate(batches):
                zb, rb = self._take(z, idx), self._take(real, idx)
                ones = self._ones(rb.shape[0])
                lane = None
                if self._overlap(n):
                    # a small batch leaves most of the chip idle in every launch: the pass over the real batch
                    # (its own gradient buffer, its own scratch) runs on the second stream beside the pass
                    # over the generated batch; same kernels, same results
                    lane = K.fork()
                    with lane:
                        real_logits, tape = D.forward(rb, save=True)
                        D.bind_grads(1)
                        D.backward(tape, ones, need_input_grad=False)
                        del tape
                fake, _ = G.forward(zb, save=False)
                fake_logits, tape = D.forward(fake, save=True)
                del fake
                D.bind_grads(0)
                D.backward(tape, ones, need_input_grad=False)
                if lane is not None:
                    del tape
                    lane.join()
                    ops.logit_sums(fake_logits.view(-1), real_logits.view(-1), sums, accumulate=True)
                    if kept is not None:
                        kept.append((idx, fake_logits.view(-1).clone(), real_logits.view(-1).clone()))
                    continue
                if self.dist is not None and bi == len(batches) - 1:
                    # the fake-pass gradient sums are final: their all-reduce runs on RCCL's stream
                    # beside the real pass of this micro-batch (the combination below is linear, so
                    # reducing the two unit-seed sums separately gives the reduced gradient)
                    pending = self.dist.all_reduce_async(flat_f)
                real_logits, tape = D.forward(rb, save=True)
                D.bind_grads(1)
                D.backward(tape, ones, need_input_grad=False)
                del tape
                if self.dist is not None and bi == len(batches) - 1:
                    # the real-pass sums are final too: reduced beside the logit sums / loss kernel below
                    pending_r = self.dist.all_reduce_async(flat_r)
                ops.logit_sums(fake_logits.view(-1), real_logits.view(-1), sums, accumulate=True)
                if kept is not None:
                    kept.append((idx, fake_logits.view(-1).clone(), real_logits.view(-1).clone()))
            if kept is not None:
                self.last_logits = {}
                for j, key in ((1, "fake"), (2, "real")):
                    full = K.empty((n,))
                    for item in kept:
                        if item[0] is None:
                            full.copy_(item[j])
                        else:
                            full.index_copy_(0, item[0].index, item[j])
                    self.last_logits[key] = full
            out_d, cf, cr = self._finish_sums(sums, 0, n, True)
            if self.dist is not None:
                self.dist.wait_for(pending)
                self.dist.wait_for(pending_r)
            ops.combine_grads(flat_f, flat_f, flat_r, cf, cr)
            D.bind_grads(0)
            self.d_optimizer.apply_model_gradients(D, None if scal is None else scal[1:2])

        # ---- metric block 0 ----
        real_sum = None
        if self.reuse_real_logits:
            rs = K.zeros((2,))
            for idx in batches:
                ops.logit_sums(None, D(self._take(real, idx)).view(-1), rs, accumulate=True)
            real_sum = rs[1:2].clone()           # local sum; the all-reduce happens with the fake sums
        if early:
            lane_g.join()
        m0 = self._metric_block(real, n, batches, real_sum, fake=early_fake)
        del early_fake

        # ---- generator update ----
        z = None if early else self._draw(n)
        flat_g = G.bind_grads(0)
        flat_g.zero_()
        sums = K.zeros((2,))
        pending_g = None
        for bi, idx in enumerate(batches):
            zb, rb = (None if early else self._take(z, idx)), self._take(real, idx)
            nb = rb.shape[0]
            chunk = self.generator_chunk if (self.generator_chunk and self.generator_chunk < nb) else None
            lane = real_logits = None
            if real_sum is None and self._overlap(n):
                lane = K.fork()
                with lane:
                    real_logits = D(rb).view(-1)
            if early:
                gen, tape_g, early_gen, early_tape = early_gen, early_tape, None, None
            else:
                # what follows the generator's taped forward before its backward frees anything: the generated
                # batch, the discriminator's tape of it and the gradient that comes back
                FusedGroup.reserve_bytes = D.tape_bytes_estimate(nb) + 2 * rb.numel() * 4
                try:
                    gen, tape_g = G.forward(zb, save=chunk is None)
                finally:
                    FusedGroup.reserve_bytes = 0
            gen_logits, tape_d = D.forward(gen, save=True)
            del gen
            dgen = D.backward(tape_d, self._ones(nb), need_input_grad=True, need_weight_grads=False)
            del tape_d
            if chunk is None:
                G.backward(tape_g, dgen, need_input_grad=False)
            else:
                # The generator has no cross-sample op (its LayerNormalization is per sample), so
                # its taped forward + backward can run on fewer samples than the discriminator's
                # MinibatchStdDev groups allow: one extra generator forward buys a tape of `chunk`
                # samples instead of a micro-batch.
                for s0 in range(0, nb, chunk):
                    _, tape_g = G.forward(zb[s0:s0 + chunk], save=True)
                    G.backward(tape_g, dgen[s0:s0 + chunk], need_input_grad=False)
            del tape_g, dgen
            if self.dist is not None and bi == len(batches) - 1:
                # the unit-seed generator gradients are final: their all-reduce (the largest message of the
                # step) runs beside the discriminator forward on the real batch below; scaling by the loss
                # coefficient afterwards is the same linear map on every rank
                pending_g = self.dist.all_reduce_async(flat_g)
            if lane is not None:
                lane.join()
            elif real_sum is None:
                real_logits = D(rb).view(-1)
            ops.logit_sums(gen_logits.view(-1), real_logits, sums, accumulate=True)
        if real_sum is not None:
            sums[1:2].copy_(real_sum)
        out_g, cf, _ = self._finish_sums(sums, 1, n, True)
        if self.dist is not None:
            self.dist.wait_for(pending_g)
        ops.combine_grads(flat_g, flat_g, None, cf, None)
        self.g_optimizer.apply_model_gradients(G, None if scal is None else scal[2:3])

        # ---- metric block 1 ----
        m1 = self._metric_block(real, n, batches, real_sum)

        self.last_losses = {"d_loss": out_d[0], "g_loss": out_g[0]}
        return {"delta_0": m0[3], "cu_0": m0[1], "ci_0": m0[2], "delta_1": m1[3], "cu_1": m1[1], "ci_1": m1[2]}

    def _step_body_merged(self, real):
        """The step of `_step_body` for a small whole batch with every pair of discriminator passes (generated
        batch, real batch) run as one pass over 2n samples: 4 discriminator forwards and 2 backwards instead of 8
        and 3.  Same passes over the same data as trainer/models.py:72-158; the weight gradient of the
        discriminator update is accumulated over both halves in one sweep (seeds = the per-sample loss
        derivatives), so it differs from the two-sweep form by float32 summation order only."""
        D, G = self.discriminator, self.generator
        scal = self._scalars
        n = real.shape[0]
        key = tuple(real.shape)
        both = self._both.get(key)
        if both is None:
            both = self._both[key] = K.empty((2 * n,) + tuple(real.shape[1:]))
        both[n:].copy_(real)
        MinibatchStdDev.sub_batches = 2
        try:
            def logits_of(fake, save):
                both[:n].copy_(fake)
                lg, tape = D.forward(both, save=save)
                lv = lg.view(-1)
                sums = K.empty((2,))
                ops.logit_sums(lv[:n], lv[n:], sums, accumulate=False)
                return sums, tape

            def metric_block():
                fake, _ = G.forward(self._draw(n), save=False)
                sums, _ = logits_of(fake, False)
                out4, _, _ = self._finish_sums(sums, 0, n, False)
                return out4

            # ---- discriminator update(s) ----
            for _ in range(self.d_steps):
                z = self._draw(n)
                flat = D.bind_grads(0)
                flat.zero_()
                fake, _ = G.forward(z, save=False)
                sums, tape = logits_of(fake, True)
                del fake
                out_d = K.empty((4,))
                seeds = K.empty((2 * n, 1))
                ops.wgan_loss(sums, n, 0, out_d, seeds[:n], seeds[n:], n)
                D.backward(tape, seeds, need_input_grad=False)
                del tape
                self.d_optimizer.apply_model_gradients(D, None if scal is None else scal[1:2])
            m0 = metric_block()
            # ---- generator update ----
            z = self._draw(n)
            flat_g = G.bind_grads(0)
            flat_g.zero_()
            gen, tape_g = G.forward(z, save=True)
            sums, tape_d = logits_of(gen, True)
            del gen
            out_g = K.empty((4,))
            seeds = K.zeros((2 * n, 1))
            ops.wgan_loss(sums, n, 1, out_g, seeds[:n], None, n)
            dboth = D.backward(tape_d, seeds, need_input_grad=True, need_weight_grads=False)
            del tape_d
            G.backward(tape_g, dboth[:n], need_input_grad=False)
            del tape_g, dboth
            self.g_optimizer.apply_model_gradients(G, None if scal is None else scal[2:3])
            m1 = metric_block()
        finally:
            MinibatchStdDev.sub_batches = 1
        self.last_losses = {"d_loss": out_d[0], "g_loss": out_g[0]}
        return {"delta_0": m0[3], "cu_0": m0[1], "ci_0": m0[2], "delta_1": m1[3], "cu_1": m1[1], "ci_1": m1[2]}

    def fit(self, x, batch_size=32, epochs=1, callbacks=None, shuffle=True, verbose=0, initial_epoch=0,
            shuffle_seed=0):
        """keras.Model.fit for this custom step: per epoch a fresh permutation, full passes in
        batches, History holds the last batch's metrics of every epoch.  `initial_epoch` (Keras'
        argument of the same name) resumes a run: the permutations of the finished epochs are
        drawn and dropped so that the remaining epochs see the batches they would have seen.
        `shuffle_seed` seeds the permutation stream of this call.
        With a DataParallel attached (`self.dist`), `x` is the same full data set on every rank and
        `batch_size` the GLOBAL batch: every rank draws the same permutation and steps on its own
        MinibatchStdDev groups of each global batch (parallel.shard_indices), so R ranks walk the
        trajectory of one process; a global batch whose groups do not split over the ranks raises."""
        callbacks = list(callbacks or [])
        for cb in callbacks:
            cb.model = self
        hist = History()
        n = x.shape[0]
        rng = np.random.default_rng(shuffle_seed)
        for epoch in range(epochs):
            order = rng.permutation(n) if shuffle else np.arange(n)
            if epoch < initial_epoch:
                continue
            logs = None
            for i in range(0, n, batch_size):
                idx = order[i:i + batch_size]
                if self.dist is not None and self.dist.world_size > 1:
                    from .parallel import shard_indices
                    idx = idx[shard_indices(len(idx), self.dist.rank, self.dist.world_size)]
                logs = self.train_step(x[idx])
            logs = {k: float(v) for k, v in (logs or {}).items()}
            for k, v in logs.items():
                hist.history.setdefault(k, []).append(v)
            if verbose:
                print("epoch %d/%d " % (epoch + 1, epochs) + " ".join("%s=%.5f" % kv for kv in logs.items()))
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
        return hist


def define_composite(discriminators, generators, latent_dim, trainable_alpha=True):
    """One straight and one fade-in WGAN per stage, each compiled with its own pair of fresh Adam
    optimisers (lr 5e-4, beta_1 0, beta_2 0.99, epsilon 10e-8) — trainer/models.py:488-524."""
    model_list = []
    for d_models, g_models in zip(discriminators, generators):
        pair = []
        for twin, fade in ((0, False), (1, True)):
            gan = WGAN(discriminator=d_models[twin], generator=g_models[twin], latent_dim=latent_dim, fade_in=fade,
                       discriminator_extra_steps=1, trainable_alpha=trainable_alpha)
            gan.compile(d_optimizer=Adam(lr=0.0005, beta_1=0, beta_2=0.99, epsilon=10e-8),
                        g_optimizer=Adam(lr=0.0005, beta_1=0, beta_2=0.99, epsilon=10e-8),
                        g_loss_fn=generator_loss, d_loss_fn=discriminator_loss)
            pair.append(gan)
        model_list.append(pair)
    return model_list


class GANMonitor:
    """Epoch-end hook in the spirit of trainer/models.py:528-551: for straight-through models it
    draws `num_examples` batches of 10 clips from the generator (the inference call shape of
    trainer/utils.py:103-104), hands them to `sink(epoch, batch_index, audio)` when the epoch is a
    `save_every`-th one (the reference's cadence is 10, models.py:541) and, when an evaluator model
    is given, collects its class probabilities for all batches and records the epoch's Inception
    score (utils.py:121-127) in `inception_scores[epoch + 1]`.  The reference's cloud upload and
    .h5 export are outside this package."""

    def __init__(self, job_dir=None, evaluador=None, num_examples=20, latent_dim=(1, 100, 2), sink=None,
                 save_every=1):
        self.num_examples = num_examples
        self.latent_dim = latent_dim
        self.job_dir = job_dir
        self.evaluador = evaluador
        self.sink = sink
        self.save_every = int(save_every)
        self.inception_scores: Dict[int, float] = {}
        self.model = None

    def on_epoch_end(self, epoch, logs=None):
        if self.model is None or self.model.fade_in or (self.sink is None and self.evaluador is None):
            return
        save = self.sink is not None and (epoch + 1) % self.save_every == 0
        pred = []
        for i in range(self.num_examples):
            z = torch.randn((10,) + tuple(self.latent_dim), dtype=torch.float32, device=K.device())
            audio = self.model.generator(z, training=False)
            if save:
                self.sink(epoch, i, audio)
            if self.evaluador is not None:
                pred.append(self.evaluador(audio, training=False).cpu().numpy())
        if pred:
            from .evaluacion import calculate_inception_score
            self.inception_scores[epoch + 1] = calculate_inception_score(np.concatenate(pred, axis=0))
