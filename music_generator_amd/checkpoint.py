"""Training checkpoints: one local .npz per save with everything a run needs to continue.

The reference only exports the two Keras models as .h5 every 10th epoch (trainer/models.py:541-550,
trainer/utils.py:148-156), never the optimiser state, and its fade-in models cannot be serialised
at all (WeightedSum.get_config calls the wrong super class, trainer/models.py:37-40).  Here a
checkpoint holds, for the whole progressive family built by define_discriminator /
define_generator / define_composite:

* every variable once (layers are shared by reference between stages and twins, so variables are
  numbered in a fixed traversal order instead of being saved per model),
* per WGAN: actual_step, total_steps, train_steps, the state of its latent generator when it was
  seeded, and for each of its two Adam optimisers the iteration count and the flat second-moment
  buffer (beta_1 = 0: there is no first moment),
* the position of the schedule (stage, fade-in flag, epochs finished in that phase).

Loading walks a freshly built family in the same order, so it needs no layer names and fails on
the first shape that does not match.
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

FORMAT = "mgx-checkpoint-1"


def _unique_variables(model_lists: Sequence[Sequence[Sequence]]) -> List:
    """Variables of [[straight, fade], ...] families in (family, stage, twin, layer, weight) order,
    each once."""
    seen, out = set(), []
    for family in model_lists:
        for pair in family:
            for model in pair:
                for layer in model.layers:
                    for v in layer.weights:
                        if id(v) not in seen:
                            seen.add(id(v))
                            out.append(v)
    return out


def _optimisers(gan):
    return (("d", gan.d_optimizer, gan.discriminator), ("g", gan.g_optimizer, gan.generator))


def save_checkpoint(path: str, d_models, g_models, gan_models, stage: int = 0, fade_in: bool = False,
                    epochs_done: int = 0, extra: Optional[Dict] = None) -> str:
    """Write the checkpoint atomically (temporary file + rename).  Returns the file name."""
    arrays: Dict[str, np.ndarray] = {}
    variables = _unique_variables([d_models, g_models])
    for i, v in enumerate(variables):
        arrays["var%05d" % i] = v.numpy()
    gans = []
    for s, pair in enumerate(gan_models):
        for t, gan in enumerate(pair):
            rec = {"stage": s, "twin": t, "actual_step": int(gan.actual_step), "total_steps": int(gan.total_steps),
                   "train_steps": None if gan.train_steps is None else int(gan.train_steps), "optimisers": {}}
            for tag, opt, _model in _optimisers(gan):
                has_v = opt._state is not None
                has_m = has_v and opt._state["m"] is not None
                rec["optimisers"][tag] = {"iterations": int(opt.iterations), "has_v": has_v, "has_m": has_m,
                                          "lr": opt.lr, "beta_1": opt.beta_1, "beta_2": opt.beta_2,
                                          "epsilon": opt.epsilon}
                if has_v:
                    arrays["adam_%d_%d_%s" % (s, t, tag)] = opt._state["v"].detach().cpu().numpy()
                if has_m:
                    arrays["adam_%d_%d_%s_m" % (s, t, tag)] = opt._state["m"].detach().cpu().numpy()
            rec["has_rng"] = gan._rng is not None
            if gan._rng is not None:                      # latent draws of a seeded WGAN continue where they stopped
                arrays["rng_%d_%d" % (s, t)] = gan._rng.get_state().cpu().numpy()
            gans.append(rec)
    meta = {"format": FORMAT, "n_variables": len(variables), "gans": gans,
            "position": {"stage": int(stage), "fade_in": bool(fade_in), "epochs_done": int(epochs_done)},
            "extra": extra or {}}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode("utf-8"), dtype=np.uint8)
    if not path.endswith(".npz"):
        path += ".npz"
    folder = os.path.dirname(path)
    if folder:
        os.makedirs(folder, exist_ok=True)
    tmp = path + ".tmp.npz"
    np.savez(tmp, **arrays)
    os.replace(tmp, path)
    return path


def load_checkpoint(path: str, d_models, g_models, gan_models) -> Dict:
    """Restore a family built with the same n_blocks.  Returns the saved schedule position
    {"stage", "fade_in", "epochs_done"} (plus "extra")."""
    if not path.endswith(".npz"):
        path += ".npz"
    with np.load(path, allow_pickle=False) as f:
        meta = json.loads(bytes(f["meta"]).decode("utf-8"))
        if meta.get("format") != FORMAT:
            raise ValueError("%s is not a %s file" % (path, FORMAT))
        variables = _unique_variables([d_models, g_models])
        if len(variables) != meta["n_variables"]:
            raise ValueError("checkpoint holds %d variables, the models have %d (different n_blocks?)"
                             % (meta["n_variables"], len(variables)))
        for i, v in enumerate(variables):
            a = f["var%05d" % i]
            if tuple(a.shape) != tuple(v.shape):
                raise ValueError("variable %d (%s): checkpoint shape %s, model shape %s"
                                 % (i, v.name, a.shape, v.shape))
            v.assign(a)
        for rec in meta["gans"]:
            gan = gan_models[rec["stage"]][rec["twin"]]
            gan.actual_step = rec["actual_step"]
            gan.total_steps = rec["total_steps"]
            gan.train_steps = rec["train_steps"]
            for tag, opt, model in _optimisers(gan):
                o = rec["optimisers"][tag]
                for hp in ("lr", "beta_1", "beta_2", "epsilon"):
                    if hp in o and abs(float(o[hp]) - float(getattr(opt, hp))) > 1e-12 * max(1.0, abs(float(o[hp]))):
                        raise ValueError("Adam of stage %d twin %d %s: checkpoint has %s = %r, the optimiser %r"
                                         % (rec["stage"], rec["twin"], tag, hp, o[hp], getattr(opt, hp)))
                if o["has_v"] and bool(o.get("has_m", False)) != (opt.beta_1 != 0.0):
                    raise ValueError("Adam of stage %d twin %d %s: the checkpoint %s a first moment, the optimiser "
                                     "(beta_1 = %g) %s one" % (rec["stage"], rec["twin"], tag,
                                                               "holds" if o.get("has_m") else "lacks", opt.beta_1,
                                                               "keeps" if opt.beta_1 != 0.0 else "does not keep"))
                opt.iterations = o["iterations"]
                if o["has_v"]:
                    if opt._state is None:
                        opt._build(model)
                    v = f["adam_%d_%d_%s" % (rec["stage"], rec["twin"], tag)]
                    if v.size != opt._state["v"].numel():
                        raise ValueError("Adam state of stage %d twin %d %s: %d values, expected %d"
                                         % (rec["stage"], rec["twin"], tag, v.size, opt._state["v"].numel()))
                    opt._state["v"].copy_(torch.from_numpy(np.ascontiguousarray(v)))
                    if o.get("has_m"):
                        m = f["adam_%d_%d_%s_m" % (rec["stage"], rec["twin"], tag)]
                        opt._state["m"].copy_(torch.from_numpy(np.ascontiguousarray(m)))
                elif opt._state is not None:
                    # zeroed in place: a captured HIP graph may hold the addresses of these tensors
                    opt._state["v"].zero_()
                    if opt._state["m"] is not None:
                        opt._state["m"].zero_()
            if rec.get("has_rng"):
                if gan._rng is None:
                    gan.seed(0)
                gan._rng.set_state(torch.from_numpy(np.ascontiguousarray(f["rng_%d_%d" % (rec["stage"], rec["twin"])])))
    out = dict(meta["position"])
    out["extra"] = meta.get("extra", {})
    return out


class CheckpointCallback:
    """Epoch-end hook: saves every `every`-th epoch (the reference's cadence is 10,
    trainer/models.py:541) into `<folder>/<H>-<W>/epoch<N>/<prefix>checkpoint.npz`, the directory
    scheme of trainer/utils.py:151-152."""

    def __init__(self, folder: str, d_models, g_models, gan_models, stage: int, fade_in: bool, every: int = 10,
                 prefix: str = ""):
        self.folder, self.every, self.prefix = folder, int(every), prefix
        self.d_models, self.g_models, self.gan_models = d_models, g_models, gan_models
        self.stage, self.fade_in = stage, fade_in
        self.model = None
        self.saved: List[str] = []

    def on_epoch_end(self, epoch, logs=None):
        if (epoch + 1) % self.every:
            return
        shape = self.g_models[self.stage][0].output_shape
        name = os.path.join(self.folder, "%d-%d" % (shape[-3], shape[-2]), "epoch%d" % (epoch + 1),
                            self.prefix + "checkpoint.npz")
        self.saved.append(save_checkpoint(name, self.d_models, self.g_models, self.gan_models, self.stage,
                                          self.fade_in, epoch + 1))
