"""The evaluator network and the Inception score computed from it: trainer/evaluacion.py:27-110 and
trainer/utils.py:65-76, inference side (the `evaluador.predict(gen_auds)` calls of
trainer/utils.py:98, 121).

The evaluator is a progressive family like the discriminator, but purely linear: 128-channel
convolutions without activations and a Dense(9, softmax) classifier head.  Its 128 -> 128
convolutions run on the library's general contraction kernel.  Fitting the evaluator
(load_evaluator's compile/fit with categorical cross-entropy, evaluacion.py:158-161) is not part of
this package: weights come from `Model.load_weights` / `set_weights_dict`.
"""
from __future__ import annotations

import numpy as np

from . import backend as K
from . import ops
from .graph import Input, Model
from .layers import Conv2D, Dense, Flatten


class SoftmaxDense(Dense):
    """Dense(units, activation='softmax') on a flattened input (evaluacion.py:97)."""

    def forward(self, xs, save, act_alpha: float = 0.0):
        if save:
            raise NotImplementedError("the evaluator head is inference-only")
        logits, _ = super().forward(xs, False, 0.0)
        y = K.empty(logits.shape)
        ops.softmax_rows(logits.contiguous(), y)
        return y, None

    def backward(self, saved, dy, need, existing, wgrad):
        raise NotImplementedError("the evaluator head is inference-only")


def _conv(t, ksize, **kw):
    return Conv2D(128, ksize, kernel_initializer="he_normal", **kw)(t)


def add_evaluator_block(old_model, n_input_layers=3):
    """evaluacion.py:27-65: a new 128-channel block at twice the resolution in front of the old
    model's layers from index `n_input_layers` on (so, as in the reference, the old model's input,
    its 1x1 feature convolution and its first 3x3 convolution are left out)."""
    in_shape = old_model.input.shape
    in_image = Input(shape=(in_shape[-3] * 2, in_shape[-2] * 2, in_shape[-1]))
    t = _conv(in_image, (1, 1), padding="same")
    for _ in range(2):                                   # convolution blocks 1 and 2
        t = _conv(t, (3, 3), padding="same")
        t = _conv(t, (3, 3), padding="same")
        t = _conv(t, (2, 2), padding="same")
    t = _conv(t, (3, 3), padding="same")                 # block 3 ends in the stride-2 reduction
    t = _conv(t, (3, 3), padding="same")
    t = _conv(t, (2, 2), strides=(2, 2), padding="valid")
    t = _conv(t, (1, 1), padding="same")
    final = None
    for layer in old_model.layers[n_input_layers:]:
        if isinstance(layer, Dense):
            final = layer(t)
        else:
            t = layer(t)
    return Model(in_image, final)


def define_evaluator(n_blocks, input_shape=(4, 750, 2)):
    """evaluacion.py:69-110: list of n_blocks models, one per resolution, sharing the old layers."""
    in_image = Input(shape=input_shape)
    t = _conv(in_image, (1, 1), padding="same")
    for _ in range(3):
        for ksize in ((3, 3), (4, 4), (3, 3)):
            t = _conv(t, ksize, padding="same")
    t = Flatten()(t)
    model_list = [Model(in_image, SoftmaxDense(9)(t))]
    for i in range(1, n_blocks):
        model_list.append(add_evaluator_block(model_list[i - 1]))
    return model_list


def calculate_inception_score(p_yx, eps=1e-16):
    """exp(mean_x KL(p(y|x) || p(y))) over a set of class-probability rows, trainer/utils.py:65-76
    (host arithmetic on the (n, 9) predictions)."""
    p_yx = np.asarray(p_yx.detach().cpu().numpy() if hasattr(p_yx, "detach") else p_yx, dtype=np.float64)
    p_y = p_yx.mean(axis=0, keepdims=True)
    kl = p_yx * (np.log(p_yx + eps) - np.log(p_y + eps))
    return float(np.exp(kl.sum(axis=1).mean()))
