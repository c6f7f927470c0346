"""Generated audio to WAV files: the sink side of trainer/utils.py:78-122.

A generated clip is an (H, W, 2) image of a stereo signal; the reference flattens it row-major to
(H*W, 2) samples, optionally normalises every channel to a peak of 1 (utils.py:108) and writes
32-bit float WAV at a sample rate equal to the image width W (utils.py:95, 118), under
`local_gen/<H>x<W>/<prefix><n>.wav`.  The cloud upload that follows each write in the reference
is not part of this package.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
from scipy.io import wavfile


def clip_to_signal(clip, normalise: bool = False) -> np.ndarray:
    """(H, W, 2) device tensor or array -> (H*W, 2) float32 samples (utils.py:84, 107-108).
    With `normalise`, each channel is divided by its own absolute peak; a silent channel (peak 0)
    is left as zeros instead of the reference's division by zero."""
    a = clip.detach().cpu().numpy() if hasattr(clip, "detach") else np.asarray(clip)
    if a.ndim != 3 or a.shape[-1] != 2:
        raise ValueError("expected one (H, W, 2) clip, got shape %s" % (a.shape,))
    sig = np.ascontiguousarray(a, dtype=np.float32).reshape(a.shape[0] * a.shape[1], 2)
    if normalise:
        peak = np.max(np.abs(sig), axis=0)
        sig = sig / np.where(peak > 0, peak, 1.0).astype(np.float32)
    return sig


def wav_path(out_dir: str, h: int, w: int, prefix: str, number: int) -> str:
    return os.path.join(out_dir, "%dx%d" % (h, w), "%s%d.wav" % (prefix, number))


def write_clip(path: str, clip, normalise: bool = False) -> str:
    """Write one clip; the sample rate is the image width."""
    sig = clip_to_signal(clip, normalise)
    folder = os.path.dirname(path)
    if folder:
        os.makedirs(folder, exist_ok=True)
    wavfile.write(path, int(clip.shape[1]), sig)
    return path


def write_examples(audio, prefix: str, iter_num: int, out_dir: str = "local_gen", n_examples: Optional[int] = None,
                   normalise: bool = False) -> List[str]:
    """generar_ejemplos (utils.py:78-96): clip i of the batch goes to `<prefix><i*iter_num>.wav`."""
    n = audio.shape[0] if n_examples is None else min(int(n_examples), audio.shape[0])
    h, w = int(audio.shape[1]), int(audio.shape[2])
    return [write_clip(wav_path(out_dir, h, w, prefix, i * iter_num), audio[i], normalise) for i in range(n)]


class WavSink:
    """`sink(epoch, batch_index, audio)` for models.GANMonitor: writes one peak-normalised clip of
    every batch it is handed (generar_ejemplo, utils.py:101-119, picks one of the 10 at random;
    `pick` fixes the choice, None draws it from `rng`)."""

    def __init__(self, out_dir: str = "local_gen", prefix: str = "", pick: Optional[int] = 0, seed: int = 0):
        self.out_dir, self.prefix, self.pick = out_dir, prefix, pick
        self.rng = np.random.default_rng(seed)
        self.written: List[str] = []

    def __call__(self, epoch: int, batch_index: int, audio):
        i = self.pick if self.pick is not None else int(self.rng.integers(0, max(1, audio.shape[0] - 1)))
        h, w = int(audio.shape[1]), int(audio.shape[2])
        name = wav_path(self.out_dir, h, w, "%sepoch%d_" % (self.prefix, epoch), batch_index)
        self.written.append(write_clip(name, audio[i], normalise=True))
