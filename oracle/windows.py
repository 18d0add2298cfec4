"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference dataset assembly.

Follows ``dataset.py:212-265`` (``_load_and_concatenate_all``: per-file crop to
``min(mel_T, label_T)``, concatenation along time ACROSS files) and ``dataset.py:267-317``
(``_create_windows``: 250-frame windows, 50-frame hop, tail windows zero-padded in the
spectrogram and background-padded in the labels, spectrogram permuted to [T, C, F]).

Pinned by the reference's recorded window counts:
  total frames 4470 -> 90, 3035 -> 61, 11470 -> 230, 5270 -> 106 windows.
"""
from __future__ import annotations

import numpy as np

WINDOW_FRAMES = int(120000 / 480)   # dataset.py:198  int(WINDOW_LENGTH / SPECTROGRAM_HOP_LENGTH) = 250
HOP_FRAMES = int(24000 / 480)       # dataset.py:199  = 50


def window_starts(total_frames: int, hop: int = HOP_FRAMES) -> np.ndarray:
    """dataset.py:271-315: ``start = 0; while start < total: ...; start += hop``."""
    starts = []
    s = 0
    while s < total_frames:
        starts.append(s)
        s += hop
    return np.asarray(starts, dtype=np.int64)


def crop_pair(spec: np.ndarray, labels: np.ndarray):
    """dataset.py:243-249.  spec [C, F, Tm], labels [Tl, ...] -> both cropped to min(Tm, Tl)."""
    t = min(spec.shape[2], labels.shape[0])
    return spec[:, :, :t], labels[:t]


def concatenate(specs, labels):
    """dataset.py:259-260."""
    return np.concatenate(specs, axis=2), np.concatenate(labels, axis=0)


def make_window(spec_cat: np.ndarray, labels_cat: np.ndarray, start: int,
                window: int = WINDOW_FRAMES, num_classes: int = 14):
    """One item exactly as ``__getitem__`` returns it (dataset.py:274-303,319-330):
    (spec float32 [window, C, F], labels float32 [window, G, num_classes])."""
    total = spec_cat.shape[2]
    end = start + window
    if end <= total:
        w_spec = spec_cat[:, :, start:end]
        w_lab = labels_cat[start:end]
    else:
        pad = window - (total - start)
        w_spec = spec_cat[:, :, start:]
        w_lab = labels_cat[start:]
        # dataset.py:293 hard-codes 4 channels for the pad; generalised to C here (identical at C=4)
        spec_pad = np.zeros((spec_cat.shape[0], spec_cat.shape[1], pad), dtype=w_spec.dtype)
        w_spec = np.concatenate([w_spec, spec_pad], axis=2)
        lab_pad = np.zeros((pad,) + labels_cat.shape[1:], dtype=w_lab.dtype)
        lab_pad[:, :, num_classes - 1] = 1.0
        w_lab = np.concatenate([w_lab, lab_pad], axis=0)
    return np.ascontiguousarray(np.transpose(w_spec, (2, 0, 1))), np.ascontiguousarray(w_lab)


def make_window_mask(mask_cat: np.ndarray, start: int, window: int = WINDOW_FRAMES) -> np.ndarray:
    """Same slicing on the compact uint16 class mask [T, G]; padded frames get mask 0, which
    expands to the background one-hot (dataset.py:298-299)."""
    total = mask_cat.shape[0]
    out = np.zeros((window, mask_cat.shape[1]), dtype=mask_cat.dtype)
    n = max(0, min(window, total - start))
    out[:n] = mask_cat[start:start + n]
    return out
