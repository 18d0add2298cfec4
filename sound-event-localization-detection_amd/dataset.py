"""Dataset assembly for SELD training on MI355X (drop-in for the reference's ``dataset.py``).

Same public surface -- ``load_audio``, ``audio_to_mel_spectrogram``, ``metadata_to_labels``,
``load_files``, ``SELDDataset`` (``.I .J .total_cells``, ``len``, ``[i] -> (spec [250,C,64] f32,
labels [250,648,14] f32)``) -- so the reference's ``main.py`` runs unchanged, but the work is done
by the HIP kernels behind ``seld_native`` instead of torchaudio / pandas / Python loops:

  reference (dataset.py)                         here
  -------------------------------------------    -------------------------------------------------
  :18-25   torchaudio.load                       stdlib ``wave`` PCM reader (int16 kept as int16)
  :27-58   MelSpectrogram + AmplitudeToDB        seld_logmel_{f32,i16}: one fused kernel, written
                                                 time-major so a window is a contiguous slice
  :60-119  iterrows + T x 648 Python loops       seld_labels_rasterise -> uint16 class mask / cell
  :212-265 torch.cat of per-file tensors         one device timeline [sum T, C, 64] + [sum T, 648]
  :267-317 list of 250-frame dict views          window start table; seld_window_gather per batch

The dense [250,648,14] float labels the reference keeps in RAM (36 KB per frame) are only
materialised when an item is requested through ``__getitem__`` (stock DataLoader path); the
trainer's device feed uses the compact mask directly (``device_batch``).
"""
import csv
import logging
import wave
from glob import glob
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

import seld_native
from config import Config
from utils import polar_to_grid  # noqa: F401  (re-exported like the reference module)

logger = logging.getLogger("SMR_SELD")
config = Config()

FRAME_MS = 20
META_FRAME_MS = 100


# ------------------------------------------------------------------------------------ audio I/O

def _read_wav(audio_path):
    """PCM WAV -> (int array [C, L], sample_rate, bits).  16-bit stays int16 (fast path)."""
    with wave.open(str(audio_path), "rb") as wf:
        channels, width, rate, frames = wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()
        raw = wf.readframes(frames)
    if width == 2:
        data = np.frombuffer(raw, dtype="<i2")
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4")
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        data = (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
        data = np.where(data >= 1 << 23, data - (1 << 24), data)
    elif width == 1:
        data = np.frombuffer(raw, dtype=np.uint8).astype(np.int16) - 128
    else:
        raise ValueError(f"unsupported WAV sample width {width} in {audio_path}")
    return np.ascontiguousarray(data.reshape(-1, channels).T), rate, 8 * width


def load_audio(audio_path):
    """(waveform float32 [C, L] in [-1, 1), sample_rate) -- what ``torchaudio.load`` returns for a
    PCM file (dataset.py:18-25).  Warns when the file does not have 4 channels."""
    data, rate, bits = _read_wav(audio_path)
    waveform = torch.from_numpy(data.astype(np.float32) / float(1 << (bits - 1)))
    if waveform.shape[0] != 4:
        logger.warning(f"Expected 4 channels but got {waveform.shape[0]} channels in {audio_path}")
    return waveform, rate


def _compute_device():
    if not torch.cuda.is_available():
        raise seld_native.SeldNativeError(
            "the SELD feature / label kernels need a ROCm GPU: there is no CPU fallback in the product path")
    return torch.device("cuda", torch.cuda.current_device())


def audio_to_mel_spectrogram(waveform, sample_rate, n_fft=None, hop_length=None, n_mels=None):
    """[C, L] waveform -> log-mel dB [C, n_mels, 1 + L // hop] (dataset.py:27-58), computed by the fused
    gfx950 kernel.  The kernel is specialised for the reference's configuration (24 kHz, n_fft 960,
    hop 480, 64 mels: config.py:85-88); anything else raises.  The result lives where the input lives."""
    n_fft = config.SPECTROGRAM_N_FFT if n_fft is None else n_fft
    hop_length = config.SPECTROGRAM_HOP_LENGTH if hop_length is None else hop_length
    n_mels = config.N_MELS if n_mels is None else n_mels
    if (int(sample_rate), int(n_fft), int(hop_length), int(n_mels)) != (24000, 960, 480, 64):
        raise NotImplementedError(
            "the HIP log-mel kernel is built for sr=24000, n_fft=960, hop=480, n_mels=64 (config.py:85-88); "
            f"got sr={sample_rate}, n_fft={n_fft}, hop={hop_length}, n_mels={n_mels}")
    on_host = not waveform.is_cuda
    x = waveform.to(_compute_device()) if on_host else waveform
    if x.dtype not in (torch.float32, torch.int16):
        x = x.to(torch.float32)
    out = seld_native.logmel(x, layout="cft")
    return out.cpu() if on_host else out


# ------------------------------------------------------------------------------------ labels

def _read_metadata_rows(metadata_path):
    """``pd.read_csv(path, header=None)`` + the int() casts of dataset.py:93-97 -> int64 [R, 5]."""
    rows = []
    with open(metadata_path, "r", newline="") as fh:
        for rec in csv.reader(fh):
            if rec:
                rows.append([int(float(v)) for v in rec[:5]])
    return np.asarray(rows, dtype=np.int64).reshape(-1, 5)


def label_frame_count(audio_duration):
    """dataset.py:73, same float64 operation order (NOT an integer division)."""
    return int((audio_duration * 1000) / FRAME_MS)


def _grid_dims(I, J, cell_size_deg):
    if (I is None or J is None) and cell_size_deg is not None:
        return int(180 // cell_size_deg), int(360 // cell_size_deg)
    if I is None or J is None:
        raise ValueError("Either provide (I, J) or cell_size_deg for grid dimensions")
    return I, J


def metadata_to_mask(metadata_path, audio_duration, I=None, J=None, cell_size_deg=None, device=None):
    """Compact form of ``metadata_to_labels``: uint16 [T, I*J] on the GPU, bit c = class c active."""
    cell_size_deg = config.GRID_CELL_DEGREES if cell_size_deg is None else cell_size_deg
    I, J = _grid_dims(I, J, cell_size_deg)
    rows = _read_metadata_rows(metadata_path) if not isinstance(metadata_path, np.ndarray) else metadata_path
    total_frames = label_frame_count(audio_duration)
    device = _compute_device() if device is None else device
    return seld_native.rasterise_labels(torch.from_numpy(np.ascontiguousarray(rows)), total_frames, I, J,
                                        device=device), I, J


def metadata_to_labels(metadata_path, audio_duration, sample_rate=24000, I=None, J=None,
                       cell_size_deg=None, num_classes=14):
    """CSV metadata -> (labels float32 [T, I*J, num_classes], I, J) as dataset.py:60-119 builds it,
    rasterised on the GPU (integer path, bit-exact) and returned on the host like the reference."""
    mask, I, J = metadata_to_mask(metadata_path, audio_duration, I, J, cell_size_deg)
    return seld_native.expand_labels(mask, num_classes).cpu(), I, J


def augment_with_gaussian_mask(metadata_path, audio_duration, I=None, J=None, cell_size_deg=None,
                               sigma_azimuth=5.0, sigma_elevation=5.0, device=None, rng=None):
    """Compact form of ``augment_with_gaussian_noise``: uint16 [T, I*J] on the GPU.  The per-source normal
    draws (smrl_seld_gaussian.py:426-437) come from ``rng`` (default ``numpy.random`` like the reference)."""
    cell_size_deg = config.GRID_CELL_DEGREES if cell_size_deg is None else cell_size_deg
    I, J = _grid_dims(I, J, cell_size_deg)
    rows = _read_metadata_rows(metadata_path) if not isinstance(metadata_path, np.ndarray) else metadata_path
    total_frames = label_frame_count(audio_duration)
    device = _compute_device() if device is None else device
    centres = seld_native.gaussian_source_noise(rows, sigma_azimuth, sigma_elevation, rng=rng)
    return seld_native.rasterise_labels_gaussian(torch.from_numpy(np.ascontiguousarray(rows)), centres, total_frames,
                                                 I, J, sigma_azimuth, sigma_elevation, device=device), I, J


def augment_with_gaussian_noise(metadata_path, audio_duration, sample_rate=24000, I=None, J=None,
                                cell_size_deg=None, num_classes=14, sigma_azimuth=5.0, sigma_elevation=5.0, rng=None):
    """smrl_seld_gaussian.py:397-534 with the same signature and return value: every source paints its class
    into all grid cells whose centre lies in the +-2 sigma box around its (noise-shifted) direction."""
    mask, I, J = augment_with_gaussian_mask(metadata_path, audio_duration, I, J, cell_size_deg, sigma_azimuth,
                                            sigma_elevation, rng=rng)
    return seld_native.expand_labels(mask, num_classes).cpu(), I, J


# ------------------------------------------------------------------------------------ file lists

def load_files():
    """(train_audio, train_meta, test_audio, test_meta) lists (dataset.py:121-165)."""
    if not config.USE_FULL_DATASET:
        return ([str(config.TRAIN_AUDIO_PATH)], [str(config.TRAIN_META_PATH)],
                [str(config.TEST_AUDIO_PATH)], [str(config.TEST_META_PATH)])

    def pair(audio_dir, meta_dir):
        audio = sorted(glob(str(audio_dir / "*.wav")))
        meta = []
        for path in audio:
            candidate = meta_dir / f"{Path(path).stem}.csv"
            if not candidate.exists():
                raise FileNotFoundError(f"Metadata file not found: {candidate}")
            meta.append(str(candidate))
        return audio, meta

    sony_tr, tau_tr = pair(config.SONY_TRAIN_DIR, config.SONY_TRAIN_META_DIR), pair(config.TAU_TRAIN_DIR, config.TAU_TRAIN_META_DIR)
    sony_te, tau_te = pair(config.SONY_TEST_DIR, config.SONY_TEST_META_DIR), pair(config.TAU_TEST_DIR, config.TAU_TEST_META_DIR)
    return sony_tr[0] + tau_tr[0], sony_tr[1] + tau_tr[1], sony_te[0] + tau_te[0], sony_te[1] + tau_te[1]


# ------------------------------------------------------------------------------------ dataset

class _WindowTable:
    """``dataset.windows`` of the reference is a list of dicts of tensor views (dataset.py:305-311);
    this is the same thing computed on demand from the window start table."""

    def __init__(self, owner):
        self._owner = owner

    def __len__(self):
        return len(self._owner.window_starts)

    def __getitem__(self, idx):
        ds = self._owner
        spec, labels = ds[idx]
        start = int(ds.window_starts[idx])
        return {"spectrogram": spec, "labels": labels, "window_idx": int(idx), "start_frame": start,
                "end_frame": min(start + ds.window_length_frames, ds.total_frames)}


def save_compact_features(path, spec, mask):
    """Write one recording's compact features to ``path`` so that a concurrent reader sees nothing or the whole file.
    Data-parallel runs construct the dataset on every rank at once: each writer gets its OWN temporary file (same
    directory, so the rename stays on one file system) and renames it into place; the loser of the race replaces the
    winner's file with identical bytes."""
    import os
    import tempfile
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    fd, tmp = tempfile.mkstemp(prefix=path.stem + ".", suffix=".tmp.npz", dir=path.parent)
    try:
        with os.fdopen(fd, "wb") as handle:
            np.savez(handle, spec=spec, mask=mask)
        os.replace(tmp, path)
    except BaseException:
        if os.path.exists(tmp):
            os.unlink(tmp)
        raise


class SELDDataset(Dataset):
    """All recordings -> one concatenated timeline -> 5 s windows with 1 s hop (dataset.py:167-330).

    Construction runs entirely on the GPU: each file is uploaded once (int16 when the WAV is 16-bit),
    the fused kernel writes its log-mel frames straight into the shared time-major timeline
    ``spec_tm [total_frames, C, 64]`` and the rasteriser writes the uint16 class mask timeline
    ``mask_tm [total_frames, 648]``; both are cropped per file to min(mel frames, label frames)
    (dataset.py:243-249) and windows are cut across file boundaries like the reference does.
    """

    def __init__(self, audio_files, metadata_files, num_classes=14, device=None, keep_on_device=True,
                 use_gaussian_augmentation=None):
        assert len(audio_files) == len(metadata_files), \
            "Number of audio files must match number of metadata files"
        self._init_fields(num_classes, device)
        if use_gaussian_augmentation is not None:          # smrl_seld_gaussian.py:539-560 (train: True, test: False)
            self.use_gaussian_augmentation = bool(use_gaussian_augmentation)
        self.audio_files = audio_files
        self.metadata_files = metadata_files
        self.keep_on_device = keep_on_device

        logger.info("SELDDataset initialization started...")
        logger.info(f"  Files: {len(audio_files)} audio files")
        logger.info(f"  Grid: {self.I}x{self.J} = {self.total_cells} cells")
        logger.info(f"  Window: {self.window_length_frames} frames, hop {self.hop_length_frames} frames")
        self._build_timeline()
        self._build_window_table()
        logger.info(f"SELDDataset initialized with {len(self)} windows")

    # -- construction -------------------------------------------------------------------------
    def _cache_path(self, audio_path, metadata_path):
        """Where this recording's compact features live under Config.FEATURE_CACHE_DIR, or None (cache off; Gaussian
        label augmentation draws fresh noise per construction, smrl_seld_gaussian.py:397-534, so it is never cached).
        The name carries everything the arrays depend on: both files' size and mtime, the feature set and the signal
        parameters behind it (FFT size, hop, mel bands), the grid, the class count."""
        root = getattr(config, "FEATURE_CACHE_DIR", None)
        if not root or self.use_gaussian_augmentation:
            return None
        import hashlib
        a, m = Path(audio_path), Path(metadata_path)
        sa, sm = a.stat(), m.stat()
        key = "|".join(str(v) for v in (a.resolve(), sa.st_size, sa.st_mtime_ns, m.resolve(), sm.st_size, sm.st_mtime_ns,
                                        getattr(config, "FEATURE_SET", "logmel"), self.I, self.J, self.sample_rate,
                                        config.SPECTROGRAM_N_FFT, config.SPECTROGRAM_HOP_LENGTH, config.N_MELS,
                                        self.num_classes, getattr(config, "GRID_CELL_DEGREES", 10), "v2"))
        return Path(root) / f"{a.stem}.{hashlib.sha1(key.encode()).hexdigest()[:16]}.npz"

    def _file_features(self, audio_path, metadata_path):
        """One recording -> (spec_tm [T, C, 64] f32, mask [T, 648] u16) on the device, cropped to the
        common frame count (dataset.py:224-249).  With Config.FEATURE_CACHE_DIR the pair is kept on disk in its COMPACT
        form (1 KB of features + 1.3 KB of label mask per frame; the reference's dense tensors are 37 KB per frame) and a
        later construction uploads it instead of decoding and transforming the recording again."""
        cached = self._cache_path(audio_path, metadata_path)
        if cached is not None and cached.exists():
            with np.load(cached) as z:
                return torch.from_numpy(z["spec"]).to(self.device), torch.from_numpy(z["mask"]).to(self.device)
        spec, mask = self._file_features_uncached(audio_path, metadata_path)
        if cached is not None and not cached.exists():
            save_compact_features(cached, spec.cpu().numpy(), mask.cpu().numpy())
        return spec, mask

    def _file_features_uncached(self, audio_path, metadata_path):
        data, rate, bits = _read_wav(audio_path)
        if data.shape[0] != 4 and getattr(config, "FEATURE_SET", "logmel") != "logmel_gcc":
            logger.warning(f"Expected 4 channels but got {data.shape[0]} channels in {audio_path}")
        if bits == 16:
            pcm = torch.from_numpy(data).to(self.device)                       # int16: half the PCIe / HBM bytes
        else:
            pcm = torch.from_numpy(data.astype(np.float32) / float(1 << (bits - 1))).to(self.device)
        return self._features_from_pcm(pcm, rate, _read_metadata_rows(metadata_path))

    def _features_from_pcm(self, pcm, rate, rows):
        if int(rate) != self.sample_rate:
            raise NotImplementedError(f"sample rate {rate} != {self.sample_rate}: the feature kernel is built for 24 kHz")
        # [F, C_total, 64]: the reference's per-channel log-mel, optionally followed by the north-star additions
        # (FOA intensity vectors / GCC-PHAT, csrc/spatial.hip) as extra input channels
        spec = seld_native.spatial_features(pcm, getattr(config, "FEATURE_SET", "logmel"))
        audio_duration = pcm.shape[1] / rate                                   # dataset.py:232 (float64)
        if self.use_gaussian_augmentation:                                      # smrl_seld_gaussian.py:608-618
            mask, _, _ = augment_with_gaussian_mask(rows, audio_duration, self.I, self.J,
                                                    sigma_azimuth=config.GAUSSIAN_SIGMA_AZIMUTH,
                                                    sigma_elevation=config.GAUSSIAN_SIGMA_ELEVATION,
                                                    device=self.device, rng=self._label_rng)
        else:
            mask, _, _ = metadata_to_mask(rows, audio_duration, self.I, self.J, device=self.device)
        frames = min(spec.shape[0], mask.shape[0])                             # dataset.py:243-249
        return spec[:frames], mask[:frames]

    def _build_timeline(self):
        specs, masks = [], []
        for idx, (audio_path, metadata_path) in enumerate(zip(self.audio_files, self.metadata_files)):
            try:
                spec, mask = self._file_features(audio_path, metadata_path)
            except Exception as exc:
                logger.error(f"Error processing file {idx} ({audio_path}): {exc}")
                raise
            specs.append(spec)
            masks.append(mask)
        self._set_timeline(torch.cat(specs, dim=0), torch.cat(masks, dim=0))

    def _set_timeline(self, spec_tm, mask_tm):
        self.spec_tm = spec_tm.contiguous()            # [total, C, 64] float32 (device)
        self.mask_tm = mask_tm.contiguous()            # [total, 648]  uint16  (device)
        self.total_frames = int(self.spec_tm.shape[0])
        self.n_channels = int(self.spec_tm.shape[1])
        # host mirrors for the stock DataLoader path (worker processes must not touch the GPU)
        self._spec_host = self.spec_tm.cpu()
        self._mask_host = self.mask_tm.cpu().numpy()
        if not self.keep_on_device:
            self.spec_tm = self.mask_tm = None
        logger.info(f"Concatenated data: {self.total_frames} total frames, {self.n_channels} channels")

    def _build_window_table(self):
        """dataset.py:271-315: start = 0; while start < total: ...; start += hop."""
        self.window_starts = np.arange(0, self.total_frames, self.hop_length_frames, dtype=np.int64)
        self.windows = _WindowTable(self)
        logger.info(f"Created {len(self.window_starts)} windows")

    @classmethod
    def from_pcm(cls, clips, metadata_rows, sample_rate=24000, num_classes=14, device=None,
                 use_gaussian_augmentation=None):
        """Build from in-memory PCM tensors ([C, L] float32 / int16 each) and parsed metadata rows
        (int [R, 5] each) -- used by the benchmark and the tests, same code path as files."""
        self = cls.__new__(cls)
        SELDDataset._init_fields(self, num_classes, device)
        if use_gaussian_augmentation is not None:
            self.use_gaussian_augmentation = bool(use_gaussian_augmentation)
        specs, masks = [], []
        for pcm, rows in zip(clips, metadata_rows):
            spec, mask = self._features_from_pcm(pcm.to(self.device), sample_rate, np.asarray(rows))
            specs.append(spec)
            masks.append(mask)
        self._set_timeline(torch.cat(specs, dim=0), torch.cat(masks, dim=0))
        self._build_window_table()
        return self

    def _init_fields(self, num_classes, device):
        self.audio_files, self.metadata_files = [], []
        self.sample_rate = config.SR
        self.n_fft, self.spectrogram_hop_length, self.n_mels = config.SPECTROGRAM_N_FFT, config.SPECTROGRAM_HOP_LENGTH, config.N_MELS
        self.cell_size_deg = config.GRID_CELL_DEGREES
        self.num_classes = num_classes
        self.I, self.J = int(180 // self.cell_size_deg), int(360 // self.cell_size_deg)
        self.total_cells = self.I * self.J
        self.window_length_samples, self.hop_length_samples = config.WINDOW_LENGTH, config.HOP_LENGTH
        self.window_length_frames = int(self.window_length_samples / self.spectrogram_hop_length)
        self.hop_length_frames = int(self.hop_length_samples / self.spectrogram_hop_length)
        self.device = torch.device(device) if device is not None else _compute_device()
        self.keep_on_device = True
        self._label_rng = np.random.RandomState(config.SEED) if config.SEED is not None else None
        self.use_gaussian_augmentation = bool(config.GAUSSIAN_AUGMENT)

    # -- reference-shaped views ---------------------------------------------------------------
    @property
    def concatenated_spectrograms(self):
        """[C, n_mels, total] view of the timeline (dataset.py:259)."""
        return self._spec_host.permute(1, 2, 0)

    @property
    def concatenated_labels(self):
        """Dense [total, 648, 14] labels (dataset.py:260) -- 36 KB per frame, built on request only."""
        return torch.from_numpy(_expand_mask_host(self._mask_host, self.num_classes))

    # -- Dataset protocol ---------------------------------------------------------------------
    def __len__(self):
        return len(self.window_starts)

    def __getitem__(self, idx):
        """(spec float32 [250, C, 64], labels float32 [250, 648, 14]) as CPU tensors (dataset.py:319-330),
        picklable across DataLoader workers; tail windows are zero / background padded (:282-299)."""
        if idx < 0:
            idx += len(self)
        start = int(self.window_starts[idx])
        w = self.window_length_frames
        n = min(w, self.total_frames - start)
        spec = torch.zeros((w, self.n_channels, self.n_mels), dtype=torch.float32)
        spec[:n] = self._spec_host[start:start + n]
        mask = np.zeros((w, self.total_cells), dtype=np.uint16)
        mask[:n] = self._mask_host[start:start + n]
        return spec, torch.from_numpy(_expand_mask_host(mask, self.num_classes))

    # -- device feed (used by trainer when DEVICE_FEED is on) -----------------------------------
    def device_batch(self, indices, out=None):
        """Window indices -> (spec [B, 250, C, 64] f32, mask [B, 250, 648] u16) on the device, gathered
        by seld_window_gather straight from the device timeline (no host round trip, no dense labels).
        ``out``: callable (spec_shape, spec_dtype, mask_shape, mask_dtype) -> (spec_buffer, mask_buffer) or None --
        the static input buffers of a captured training step (seld_graph.GraphedTrainStep.static_inputs)."""
        if self.spec_tm is None:
            raise RuntimeError("device_batch needs keep_on_device=True")
        starts = torch.as_tensor(self.window_starts[np.asarray(indices, dtype=np.int64)])
        w = self.window_length_frames
        bufs = None
        if out is not None:
            bufs = out((len(starts), w) + tuple(self.spec_tm.shape[1:]), self.spec_tm.dtype,
                       (len(starts), w) + tuple(self.mask_tm.shape[1:]), self.mask_tm.dtype)
        spec = seld_native.gather_windows(self.spec_tm, starts, w, out=bufs[0] if bufs else None)
        mask = seld_native.gather_windows(self.mask_tm, starts, w, out=bufs[1] if bufs else None)
        return spec, mask


def _expand_mask_host(mask, num_classes):
    """Host-side expansion for the stock DataLoader path: uint16 [..., G] -> float32 [..., G, M] with the
    reference's background rule (dataset.py:110-117).  Pure indexing, no arithmetic."""
    bits = (mask[..., None] >> np.arange(num_classes, dtype=np.uint16)) & np.uint16(1)
    dense = bits.astype(np.float32)
    dense[..., num_classes - 1][mask == 0] = 1.0
    return dense
