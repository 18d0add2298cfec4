"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference label rasteriser.

Follows ``dataset.py:60-119`` (``metadata_to_labels``) and ``utils.py:77-90``
(``polar_to_grid``).  Integer / index work: the bar is bit-exact.

Pinned by the reference's recorded known answers:
  polar_to_grid(-98, -16, 18, 36) == (7, 8)           SMR_SELD_2.ipynb:751
  labels for L=2145600 have shape [4470, 648, 14]      SMR_SELD_2.ipynb:663
  labels[0, 0] is the background one-hot               SMR_SELD_2.ipynb:663
and by tests/golden/polar_grid.npz (generated from the reference's own utils.polar_to_grid).
"""
from __future__ import annotations

import csv
import io

import numpy as np

FRAME_MS = 20            # dataset.py:69
META_FRAME_MS = 100      # dataset.py:70
FRAMES_PER_META = META_FRAME_MS // FRAME_MS   # dataset.py:71 -> 5
GRID_I, GRID_J = 18, 36  # 180//10, 360//10  (dataset.py:76-77, config.py:97)
NUM_CLASSES = 14


def polar_to_grid(phi, theta, I=GRID_I, J=GRID_J):
    """utils.py:77-90.  float64 normalise, clip, truncate."""
    phi_norm = (phi + 180.0) / 360.0
    theta_norm = (theta + 90.0) / 180.0
    j = int(np.clip(phi_norm * J, 0, J - 1))
    i = int(np.clip(theta_norm * I, 0, I - 1))
    return i, j


def total_label_frames(num_samples: int, sample_rate: int = 24000) -> int:
    """dataset.py:232 + dataset.py:73: ``audio_duration = L / sr`` (float64) then
    ``int((audio_duration * 1000) / 20)`` -- NOT integer division: under-counts by one at a
    few exact multiples of 480 (e.g. L=96480 -> 200)."""
    audio_duration = num_samples / sample_rate
    return int((audio_duration * 1000) / FRAME_MS)


def parse_metadata_csv(text_or_path) -> np.ndarray:
    """pd.read_csv(path, header=None) + int() casts of columns 0..4 (dataset.py:86-97).
    Returns int64 [R, 5] = (meta_frame, class, source, azimuth, elevation)."""
    if isinstance(text_or_path, (bytes, bytearray)):
        text_or_path = text_or_path.decode()
    if "\n" in str(text_or_path) or "," in str(text_or_path):
        fh = io.StringIO(str(text_or_path))
    else:
        fh = open(text_or_path, "r", newline="")
    rows = []
    with fh:
        for rec in csv.reader(fh):
            if not rec:
                continue
            rows.append([int(float(v)) for v in rec[:5]])
    return np.asarray(rows, dtype=np.int64).reshape(-1, 5)


def metadata_to_labels_loops(rows: np.ndarray, num_samples: int, sample_rate: int = 24000,
                             I: int = GRID_I, J: int = GRID_J,
                             num_classes: int = NUM_CLASSES) -> np.ndarray:
    """Line-for-line restatement of dataset.py:73-117 (pure Python loops; small cases only).
    Returns float32 [T, I*J, num_classes]."""
    total_frames = total_label_frames(num_samples, sample_rate)
    total_cells = I * J
    labels = np.zeros((total_frames, total_cells, num_classes), dtype=np.float32)
    active = [set() for _ in range(total_frames)]
    for row in rows:
        meta_frame, cls, _src, az, el = (int(v) for v in row[:5])
        start = meta_frame * FRAMES_PER_META
        end = min(start + FRAMES_PER_META, total_frames)
        i, j = polar_to_grid(az, el, I, J)
        cell = i * J + j
        for t in range(start, end):
            labels[t, cell, cls] = 1.0
            active[t].add(cell)
    for t in range(total_frames):
        for cell in range(total_cells):
            if cell not in active[t]:
                labels[t, cell, num_classes - 1] = 1.0
    return labels


def metadata_to_mask(rows: np.ndarray, num_samples: int, sample_rate: int = 24000,
                     I: int = GRID_I, J: int = GRID_J) -> np.ndarray:
    """Vectorised integer form: uint16 [T, I*J], bit c set <=> labels[t, cell, c] == 1 for an
    event row.  Background (bit 13) is NOT stored: it is implied by ``mask == 0``
    (dataset.py:114-117) -- unless an event row itself names class 13, which sets bit 13."""
    T = total_label_frames(num_samples, sample_rate)
    mask = np.zeros((T, I * J), dtype=np.uint16)
    for row in rows:
        meta_frame, cls, _src, az, el = (int(v) for v in row[:5])
        start = meta_frame * FRAMES_PER_META
        end = min(start + FRAMES_PER_META, T)
        if start >= end or start < 0:
            continue
        i, j = polar_to_grid(az, el, I, J)
        mask[start:end, i * J + j] |= np.uint16(1 << cls)
    return mask


def mask_to_dense(mask: np.ndarray, num_classes: int = NUM_CLASSES) -> np.ndarray:
    """uint16 [..., G] -> float32 [..., G, num_classes] exactly as dataset.py:110-117 leaves it."""
    bits = (mask[..., None] >> np.arange(num_classes, dtype=np.uint16)) & 1
    dense = bits.astype(np.float32)
    dense[..., num_classes - 1] = np.where(mask == 0, 1.0, dense[..., num_classes - 1])
    return dense


def synth_metadata(clip_idx: int, meta_frames: int = 600, seed_base: int = 1234) -> np.ndarray:
    """Seeded synthetic STARSS22-style metadata: 0-3 simultaneous
    sources per 100 ms frame, classes 0..12, integer az/el, with duplicate-cell rows and
    rows beyond the end of the audio (5t >= T) mixed in."""
    rng = np.random.default_rng(seed_base + clip_idx)
    rows = []
    for t in range(meta_frames):
        k = int(rng.integers(0, 4))
        for s in range(k):
            cls = int(rng.integers(0, 13))
            az = int(rng.integers(-180, 181))
            el = int(rng.integers(-90, 91))
            rows.append((t, cls, s, az, el))
            if rng.random() < 0.05:                       # second class in the same cell
                rows.append((t, int(rng.integers(0, 13)), s + 1, az, el))
    for t in (meta_frames, meta_frames + 3):              # rows past the end: silently dropped
        rows.append((t, 1, 0, 10, 10))
    return np.asarray(rows, dtype=np.int64).reshape(-1, 5)


def metadata_to_csv(rows: np.ndarray) -> str:
    return "".join(",".join(str(int(v)) for v in r) + "\n" for r in rows)


# --------------------------------------------------------------------------- Gaussian-region augmentation

def gaussian_source_noise(rows: np.ndarray, sigma_az: float = 5.0, sigma_el: float = 5.0, rng=None) -> np.ndarray:
    """smrl_seld_gaussian.py:426-437: one (az, el) normal draw per unique (class, source), in the sorted key
    order of ``df.groupby([1, 2])``; returns the box centre (az + noise, el + noise) of every row."""
    rng = np.random if rng is None else rng
    keys = sorted({(int(r[1]), int(r[2])) for r in rows})
    noise = {}
    for k in keys:
        noise[k] = (rng.normal(0, sigma_az), rng.normal(0, sigma_el))
    return np.asarray([(r[3] + noise[(int(r[1]), int(r[2]))][0], r[4] + noise[(int(r[1]), int(r[2]))][1])
                       for r in rows], dtype=np.float64).reshape(-1, 2)


def gaussian_labels_loops(rows: np.ndarray, centres: np.ndarray, num_samples: int, sample_rate: int = 24000,
                          I: int = GRID_I, J: int = GRID_J, num_classes: int = NUM_CLASSES,
                          sigma_azimuth: float = 5.0, sigma_elevation: float = 5.0) -> np.ndarray:
    """Line-for-line restatement of smrl_seld_gaussian.py:440-532 given the per-row box centres."""
    total_frames = total_label_frames(num_samples, sample_rate)
    labels = np.zeros((total_frames, I * J, num_classes), dtype=np.float32)
    active = [set() for _ in range(total_frames)]
    for row, (center_azimuth, center_elevation) in zip(rows, centres):
        metadata_frame, active_class = int(row[0]), int(row[1])
        start_frame = metadata_frame * FRAMES_PER_META
        end_frame = min(start_frame + FRAMES_PER_META, total_frames)
        elevation_min = max(center_elevation - 2 * sigma_elevation, -90)
        elevation_max = min(center_elevation + 2 * sigma_elevation, 90)
        affected = set()
        for grid_i in range(I):
            for grid_j in range(J):
                cell_elevation = -90 + (grid_i + 0.5) * (180.0 / I)
                cell_azimuth = -180 + (grid_j + 0.5) * (360.0 / J)
                diff = cell_azimuth - center_azimuth
                while diff > 180:
                    diff -= 360
                while diff < -180:
                    diff += 360
                if abs(diff) <= 2 * sigma_azimuth and elevation_min <= cell_elevation <= elevation_max:
                    affected.add(grid_i * J + grid_j)
        for cell_idx in affected:
            for t in range(start_frame, end_frame):
                labels[t, cell_idx, active_class] = 1.0
                active[t].add(cell_idx)
    for t in range(total_frames):
        for cell_idx in range(I * J):
            if cell_idx not in active[t]:
                labels[t, cell_idx, num_classes - 1] = 1.0
    return labels


def gaussian_mask(rows: np.ndarray, centres: np.ndarray, num_samples: int, sample_rate: int = 24000,
                  I: int = GRID_I, J: int = GRID_J, sigma_azimuth: float = 5.0, sigma_elevation: float = 5.0) -> np.ndarray:
    """Vectorised uint16 form of ``gaussian_labels_loops`` (same float64 comparisons, whole grid at once);
    used for the sizes the loops cannot finish in seconds."""
    total_frames = total_label_frames(num_samples, sample_rate)
    mask = np.zeros((total_frames, I * J), dtype=np.uint16)
    cell_el = -90 + (np.arange(I, dtype=np.float64) + 0.5) * (180.0 / I)
    cell_az = -180 + (np.arange(J, dtype=np.float64) + 0.5) * (360.0 / J)
    for row, (c_az, c_el) in zip(rows, np.asarray(centres, dtype=np.float64).reshape(-1, 2)):
        start = int(row[0]) * FRAMES_PER_META
        end = min(start + FRAMES_PER_META, total_frames)
        if start >= end:
            continue
        diff = cell_az - c_az
        for _ in range(4):                                  # the two while loops of :498-505 (|diff| < 4*360 here)
            diff = np.where(diff > 180, diff - 360, diff)
        for _ in range(4):
            diff = np.where(diff < -180, diff + 360, diff)
        az_ok = np.abs(diff) <= 2 * sigma_azimuth
        el_ok = (max(c_el - 2 * sigma_elevation, -90) <= cell_el) & (cell_el <= min(c_el + 2 * sigma_elevation, 90))
        box = (el_ok[:, None] & az_ok[None, :]).reshape(-1)
        mask[start:end, box] |= np.uint16(1 << int(row[1]))
    return mask
