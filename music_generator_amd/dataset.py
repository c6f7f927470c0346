"""Local-directory dataset reader: read_dataset of trainer/dataset.py:213-239 without the cloud
download and resampling steps around it.

Layout on disk (dataset.py:220): `<root>/<format>/<H>-<W>/<folder>/<anything>.<format>` with folders
1..9 = the nine classes the evaluator is fitted on.  A clip holds H*W stereo frames.  The reference
reads a file with `audiofile.read`, which returns float32 samples shaped (channels, frames), and
reshapes that array AS IS to (H, W, 2) (dataset.py:224-226) - i.e. the first half of the image is the
left channel and the second half the right one, not interleaved frames.  That behaviour is kept
(`channels_first=True`); `channels_first=False` reshapes the interleaved (frames, 2) samples instead.
Only WAV is read here (scipy); integer PCM is scaled to [-1, 1) as audiofile does.
"""
from __future__ import annotations

import os
from typing import Callable, Tuple

import numpy as np
from scipy.io import wavfile

N_CLASSES = 9


def _read_wav(path: str) -> np.ndarray:
    """-> float32 (channels, frames)."""
    _, data = wavfile.read(path)
    if data.ndim == 1:
        data = data[:, None]
    if np.issubdtype(data.dtype, np.integer):
        data = data.astype(np.float32) / float(-np.iinfo(data.dtype).min)
    return np.ascontiguousarray(data.T, dtype=np.float32)


def read_dataset(dimension: Tuple[int, int], files_format: str = "wav", root: str = "local_ds",
                 channels_first: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """-> (clips (N, H, W, 2) float32, one-hot labels (N, 9) float32).  Files that cannot be read
    or reshaped are skipped; five failures in a row end a folder (dataset.py:233-236); a missing
    folder holds no clips."""
    if files_format != "wav":
        raise NotImplementedError("only WAV files are read (no audio decoding library beyond scipy here)")
    h, w = int(dimension[0]), int(dimension[1])
    data, labels = [], []
    for folder in range(N_CLASSES):
        directory = os.path.join(root, files_format, "%d-%d" % (h, w), str(folder + 1))
        if not os.path.isdir(directory):
            continue
        errors = 0
        for name in sorted(os.listdir(directory)):
            try:
                signal = _read_wav(os.path.join(directory, name))
                clip = np.reshape(signal if channels_first else signal.T, (h, w, 2))
            except Exception:                       # unreadable or wrong length, as the reference's bare except
                errors += 1
                if errors == 5:
                    break
                continue
            errors = 0
            data.append(clip)
            labels.append(folder)
    x = np.stack(data).astype(np.float32) if data else np.zeros((0, h, w, 2), np.float32)
    y = np.zeros((len(labels), N_CLASSES), np.float32)
    y[np.arange(len(labels)), labels] = 1.0
    return x, y


def directory_provider(root: str = "local_ds", files_format: str = "wav",
                       channels_first: bool = True) -> Callable[[int, int], np.ndarray]:
    """provider(h, w) for task.train: the clips of one resolution (trainer/task.py:113-114)."""
    def provide(h: int, w: int) -> np.ndarray:
        return read_dataset((h, w), files_format, root, channels_first)[0]
    return provide
