"""Progressive training driver: the schedule of trainer/task.py:108-189 (per stage: fade-in fit,
then straight-through fit; batch sizes / epochs tables; fresh optimisers per phase because every
WGAN owns its own pair), fed from an in-memory array provider instead of the reference's cloud
dataset reader.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import models as M

# trainer/task.py:186-189
BATCH_SIZES = [16, 8, 4, 2, 2, 2, 2]
EPOCHS_NORM = [200, 220, 250, 280, 300, 350, 420]
EPOCHS_FADE = [10, 15, 32, 25, 30, 35, 40]
LATENT_DIM = (1, 100, 2)            # trainer/task.py:78


def synthetic_provider(n_clips: int, seed: int = 0) -> Callable[[int, int], np.ndarray]:
    """Stand-in for read_dataset (trainer/dataset.py:213-239): returns `n_clips` clips shaped
    (H, W, 2) ~ U(-1, 1) for the requested resolution."""
    def provide(h: int, w: int) -> np.ndarray:
        return np.random.default_rng(seed + h).uniform(-1.0, 1.0, (n_clips, h, w, 2)).astype(np.float32)
    return provide


def phase_steps(n_samples: int, n_batch: int, epochs: int) -> int:
    """n_steps = int(len/batch) * epochs, trainer/task.py:127,162,174."""
    return int(n_samples / n_batch) * epochs


def train(g_models, d_models, gan_models, provider, batch_sizes: Sequence[int] = BATCH_SIZES,
          epochs_norm: Sequence[int] = EPOCHS_NORM, epochs_fade: Sequence[int] = EPOCHS_FADE,
          callbacks_for: Optional[Callable[[int, bool], list]] = None, shuffle_seed: int = 0,
          verbose: int = 0, resume: Optional[Dict] = None) -> List[Dict[str, list]]:
    """trainer/task.py:108-184.  Returns one History.history per fit call, in call order.

    `resume` = the position checkpoint.load_checkpoint returns ({"stage", "fade_in", "epochs_done"}):
    phases before it are skipped (their data shuffles are still drawn, so later phases see the
    same order) and the phase it names continues after `epochs_done` epochs."""
    histories = []
    rng = np.random.default_rng(shuffle_seed)

    def phase_rank(stage, fade):
        return 2 * stage + (0 if fade or stage == 0 else 1)      # fade-in twin runs first within a stage

    start = (phase_rank(int(resume["stage"]), bool(resume["fade_in"])), int(resume.get("epochs_done", 0))) \
        if resume else (-1, 0)

    def run(gan, data, n_batch, epochs, stage, fade):
        rank = phase_rank(stage, fade)
        if rank < start[0]:
            return
        first = start[1] if rank == start[0] else 0
        gan.set_train_steps(phase_steps(data.shape[0], n_batch, epochs))
        cbs = callbacks_for(stage, fade) if callbacks_for else []
        hist = gan.fit(data, batch_size=n_batch, epochs=epochs, callbacks=cbs, verbose=verbose, initial_epoch=first)
        histories.append(hist.history)

    for i in range(len(g_models)):
        gen_shape = g_models[i][0].output_shape
        data = provider(gen_shape[-3], gen_shape[-2])
        n_batch = batch_sizes[i]
        limit = int(data.shape[0] / n_batch) * n_batch       # trim to a whole number of batches
        data = data[:limit]
        if i == 0:
            rng.shuffle(data)
            run(gan_models[0][0], data, n_batch, epochs_norm[0], 0, False)
            continue
        rng.shuffle(data)
        run(gan_models[i][1], data, n_batch, epochs_fade[i], i, True)      # fade-in twin first
        run(gan_models[i][0], data, n_batch, epochs_norm[i], i, False)     # then straight-through
    return histories


def build(n_blocks: int = 7, trainable_alpha: bool = True):
    """discriminators, generators, composite exactly as trainer/task.py:82-90."""
    discriminators = M.define_discriminator(n_blocks)
    generators = M.define_generator(n_blocks)
    composite = M.define_composite(discriminators, generators, LATENT_DIM, trainable_alpha=trainable_alpha)
    return discriminators, generators, composite
