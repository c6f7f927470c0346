"""The evaluator network and the Inception score computed from it: trainer/evaluacion.py:27-110 and
trainer/utils.py:65-76, inference side (the `evaluador.predict(gen_auds)` calls of
trainer/utils.py:98, 121).

The evaluator is a progressive family like the discriminator, but purely linear: 128-channel
convolutions without activations and a Dense(9, softmax) classifier head.  Its 128 -> 128
convolutions run on the library's general contraction kernel.  `load_evaluator` is the
download=False branch of evaluacion.py:114-166: fit the model of one resolution on (clips, one-hot
labels) with Adam (keras defaults) and categorical cross-entropy; the cloud download / upload of
the .h5 file is replaced by `Model.save` / `load_weights`.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import backend as K
from . import ops
from .graph import Input, Model
from .layers import Conv2D, Dense, Flatten


class SoftmaxDense(Dense):
    """Dense(units, activation='softmax') on a flattened input (evaluacion.py:97).  With a tape
    the layer returns its LOGITS: the fitting loop pairs them with the fused softmax +
    cross-entropy kernel, whose logit gradient is what backward() receives."""

    def forward(self, xs, save, act_alpha: float = 0.0):
        logits, saved = super().forward(xs, save, 0.0)
        if save:
            return logits, saved
        y = K.empty(logits.shape)
        ops.softmax_rows(logits.contiguous(), y)
        return y, None


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


# evaluacion.py:130-157: batch size and epochs per resolution
FIT_TABLE = {4: (64, 10), 8: (32, 20), 16: (16, 30), 32: (8, 40), 64: (4, 50), 128: (2, 60), 256: (2, 70)}


def labels_to_device(y) -> torch.Tensor:
    """One-hot labels (n, classes) as a contiguous float32 device tensor."""
    t = y if isinstance(y, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32))
    return t.to(device=K.device(), dtype=torch.float32).contiguous()


def evaluate(model, x, y, batch_size: int = 32) -> Dict[str, float]:
    """Mean categorical cross-entropy and accuracy over (x, y) - keras' evaluate / validation pass."""
    n = x.shape[0]
    loss, hits = 0.0, 0
    for i in range(0, n, batch_size):
        xb = model._to_device(x[i:i + batch_size])
        yb = labels_to_device(y[i:i + batch_size])
        logits, _ = model.forward(xb, save=True)          # with a tape the head returns logits
        prob, rows = K.empty(logits.shape), K.empty((logits.shape[0],))
        ops.softmax_xent(logits.contiguous(), yb, prob, rows, None)
        loss += float(rows.sum())
        hits += int((prob.argmax(dim=1) == yb.argmax(dim=1)).sum())
    return {"loss": loss / n, "accuracy": hits / n}


def fit_evaluator(model, x, y, epochs: int, batch_size: int, optimizer=None, validation_data=None,
                  shuffle_seed: int = 0, verbose: int = 0) -> Dict[str, List[float]]:
    """model.compile(optimizer="Adam", loss='categorical_crossentropy', metrics=["accuracy"]) +
    model.fit(X, y, epochs, batch_size, validation_data) of evaluacion.py:160-161.  Returns the
    history: per epoch the sample-weighted mean of the batch losses / accuracies (and val_*)."""
    from .models import Adam
    opt = optimizer or Adam()                              # keras defaults: 1e-3, 0.9, 0.999, 1e-7
    n = x.shape[0]
    rng = np.random.default_rng(shuffle_seed)
    hist: Dict[str, List[float]] = {"loss": [], "accuracy": []}
    y = np.ascontiguousarray(y, dtype=np.float32)
    for epoch in range(epochs):
        order = rng.permutation(n)
        loss_sum = torch.zeros((), dtype=torch.float32, device=K.device())
        hits = torch.zeros((), dtype=torch.int64, device=K.device())
        for i in range(0, n, batch_size):
            idx = order[i:i + batch_size]
            xb, yb = model._to_device(x[idx]), labels_to_device(y[idx])
            flat = model.bind_grads(0)
            flat.zero_()
            logits, tape = model.forward(xb, save=True)
            prob, rows, dlogits = K.empty(logits.shape), K.empty((logits.shape[0],)), K.empty(logits.shape)
            ops.softmax_xent(logits.contiguous(), yb, prob, rows, dlogits)
            model.backward(tape, dlogits, need_input_grad=False)
            opt.apply_model_gradients(model)
            loss_sum += rows.sum()
            hits += (prob.argmax(dim=1) == yb.argmax(dim=1)).sum()
        hist["loss"].append(float(loss_sum) / n)
        hist["accuracy"].append(int(hits) / n)
        if validation_data is not None:
            val = evaluate(model, validation_data[0], validation_data[1], batch_size)
            hist.setdefault("val_loss", []).append(val["loss"])
            hist.setdefault("val_accuracy", []).append(val["accuracy"])
        if verbose:
            print("epoch %d/%d " % (epoch + 1, epochs) + " ".join("%s=%.4f" % (k, v[-1]) for k, v in hist.items()))
    return hist


def load_evaluator(dimension, train_dataset, epochs: Optional[int] = None, n_blocks: int = 7, verbose: int = 0):
    """evaluacion.py:114-166 with download=False: build the family, pick the model of `dimension`
    (H, W), fit it on train_dataset = (X, one-hot y) with the batch size / epochs of the
    reference's table, validating on the training set as the reference does.  Returns the model
    (history in `model.fit_history`)."""
    batch_size, table_epochs = FIT_TABLE[int(dimension[0])]
    stage = int(dimension[0]).bit_length() - 3            # 4 -> 0, 8 -> 1, ...
    model = define_evaluator(max(n_blocks, stage + 1))[stage]
    x, y = train_dataset
    model.fit_history = fit_evaluator(model, x, y, table_epochs if epochs is None else epochs, batch_size,
                                      validation_data=(x, y), verbose=verbose)
    return model
