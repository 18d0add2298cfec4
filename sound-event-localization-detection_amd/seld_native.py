"""ctypes binding of ``libseld_hip.so`` (C ABI: ``include/seld_hip.h``).

PyTorch is used for device memory and streams only; every entry point below hands raw
device pointers (``tensor.data_ptr()``) and the current HIP stream to the hand-written
gfx950 kernels.  There is NO CPU fallback: if the shared library is missing or a GPU call
fails, a ``RuntimeError`` is raised.
"""
from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
# SELD_HIP_LIB: developer override to load an experimental build of the SAME library (A/B kernel experiments)
LIB_PATH = Path(os.environ["SELD_HIP_LIB"]).resolve() if os.environ.get("SELD_HIP_LIB") else _HERE / "libseld_hip.so"

N_FFT = 960
HOP = 480
N_BINS = 481
N_MELS = 64

_lock = threading.Lock()
_lib = None
_initialised_devices = set()

_i64 = ctypes.c_int64
_int = ctypes.c_int
_ptr = ctypes.c_void_p


class SeldNativeError(RuntimeError):
    pass


def _declare(lib):
    lib.seld_last_error.restype = ctypes.c_char_p
    lib.seld_last_error.argtypes = []
    lib.seld_version.restype = _int
    lib.seld_init.argtypes = [_int]
    lib.seld_shutdown.argtypes = []
    lib.seld_set_mel_filterbank.argtypes = [_ptr]
    lib.seld_set_window.argtypes = [_ptr]
    lib.seld_default_tables.argtypes = [_ptr] * 5
    lib.seld_num_frames.restype = _i64
    lib.seld_num_frames.argtypes = [_i64]
    for name in ("seld_logmel_f32", "seld_logmel_i16"):
        getattr(lib, name).argtypes = [_ptr, _i64, _i64, _i64, _ptr, _int, _ptr]
    for name in ("seld_logmel_f32_strided", "seld_logmel_i16_strided"):
        getattr(lib, name).argtypes = [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr]
    for name in ("seld_stft_f32", "seld_stft_i16"):
        getattr(lib, name).argtypes = [_ptr, _i64, _i64, _i64, _ptr, _ptr]
    lib.seld_foa_intensity.argtypes = [_ptr, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr]
    for fn in (lib.seld_logmel_iv_f32, lib.seld_logmel_iv_i16):
        fn.argtypes = [_ptr, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr]
    for fn in (lib.seld_logmel_spectrum_f32, lib.seld_logmel_spectrum_i16):
        fn.argtypes = [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr]
        fn.restype = ctypes.c_int
    for fn in (lib.seld_logmel_phasors_f32, lib.seld_logmel_phasors_i16):
        fn.argtypes = [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr, _ptr]
    lib.seld_phasor_pitch.restype = ctypes.c_int64
    lib.seld_gcc_phat_q15.argtypes = [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr]
    lib.seld_gcc_table_host.argtypes = [_ptr]
    lib.seld_gcc_phat.argtypes = [_ptr, _i64, _i64, _i64, _ptr, _i64, _i64, _i64, _i64, _ptr]
    lib.seld_labels_rasterise.argtypes = [_ptr, _i64, _i64, _int, _int, _ptr, _ptr]
    lib.seld_labels_expand.argtypes = [_ptr, _i64, _int, _ptr, _ptr]
    lib.seld_labels_rasterise_box.argtypes = [_ptr, _ptr, _i64, _i64, _int, _int, ctypes.c_double, ctypes.c_double, _ptr,
                                              _ptr]
    lib.seld_window_gather.argtypes = [_ptr, _i64, _i64, _ptr, _i64, _i64, _ptr, _ptr]
    lib.seld_softmax_mse_workspace_bytes.restype = _i64
    lib.seld_softmax_mse_workspace_bytes.argtypes = []
    lib.seld_softmax_mse.argtypes = [_ptr, _int, _ptr, _ptr, _i64, _int, ctypes.c_float, _ptr, _ptr, _ptr, _ptr]
    lib.seld_scale_by_device_scalar.argtypes = [_ptr, _int, _i64, _ptr, _ptr]
    lib.seld_smr_loss_workspace_bytes.restype = _i64
    lib.seld_smr_loss_workspace_bytes.argtypes = [_i64, _i64]
    lib.seld_smr_loss.argtypes = [_ptr, _int, _ptr, _ptr, _i64, _int, _int, _int, ctypes.c_float, ctypes.c_float,
                                  ctypes.c_float, _ptr, _ptr, _ptr, _ptr]
    lib.seld_multi_cast.argtypes = [_ptr, _ptr, _ptr, _int, _int, _ptr]
    lib.seld_stream_delay.argtypes = [_i64, _ptr]
    lib.seld_gru_fold_bias.argtypes = [_ptr, _ptr, _i64, _ptr, _int, _ptr, _ptr]
    lib.seld_gru_bias_grads.argtypes = [_ptr, _i64, _i64, _ptr, _ptr, _ptr]
    lib.seld_sum_chunks.argtypes = [_ptr, _int, _i64, _i64, _ptr, _int, _ptr]
    lib.seld_gru_dwhh_finish.argtypes = [_ptr, _ptr, _int, _i64, _i64, _ptr, _int, _ptr]
    lib.seld_column_sums.argtypes = [_ptr, _int, _i64, _i64, _ptr, _ptr, _int, _ptr]
    _pp, _pi64, _pi32 = ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int32)
    _f = ctypes.c_float
    lib.seld_multi_adam.argtypes = [_pp, _pi32, _pp, _pp, _pp, _pp, _pi64, _int, _ptr, _ptr, _f, _f, _f, _f, _f, _ptr]
    lib.seld_multi_sum_chunks.argtypes = [_pp, _pp, _pi64, _pi32, _pi32, _int, _ptr]
    lib.seld_multi_column_sums_scratch.argtypes = [_pi64, _pi64, _int, _pi64]
    lib.seld_multi_column_sums.argtypes = [_pp, _pp, _pi64, _pi64, _pi32, _int, _ptr, _i64, _ptr]
    lib.seld_column_sums_blocks.argtypes = [_i64, _i64]
    lib.seld_column_sums_blocks.restype = ctypes.c_int64
    lib.seld_conv_weight_flip_transpose.argtypes = [_ptr, _int, _i64, _i64, _ptr, _ptr]
    lib.seld_layernorm_supported.argtypes = [_i64]
    lib.seld_layernorm_workspace_floats.restype = _i64
    lib.seld_layernorm_workspace_floats.argtypes = [_i64, _i64]
    lib.seld_layernorm_forward.argtypes = [_ptr, _int, _i64, _i64, _ptr, _ptr, ctypes.c_float, _int, _ptr, _ptr, _ptr]
    lib.seld_layernorm_backward.argtypes = [_ptr, _ptr, _int, _i64, _i64, _ptr, _ptr, _ptr, _int, _ptr, _ptr, _ptr,
                                            _ptr, _ptr]
    lib.seld_conv_tail_workspace_floats.restype = _i64
    lib.seld_conv_tail_workspace_floats.argtypes = [_int]
    lib.seld_conv_tail_forward.argtypes = [_ptr, _ptr, _int, _i64, _int, _int, _ptr, _ptr, _ptr, _ptr, ctypes.c_float,
                                           ctypes.c_float, _int, _ptr, _ptr, _ptr, _ptr, _ptr]
    lib.seld_conv_tail_backward.argtypes = [_ptr, _ptr, _ptr, _int, _i64, _int, _int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                            _ptr, _ptr]
    lib.seld_dwconv1d.argtypes = [_ptr, _int, _ptr, _ptr, _i64, _i64, _int, _int, _int, _ptr, _ptr]
    lib.seld_dwconv1d_wgrad.argtypes = [_ptr, _ptr, _int, _i64, _i64, _int, _int, _ptr, _ptr]
    lib.seld_dwconv1d_wgrad_rows.argtypes = [_i64, _i64]
    lib.seld_dwconv1d_wgrad_rows.restype = ctypes.c_int64
    lib.seld_gru_to_tile.argtypes = [_ptr, _int, _i64, _i64, _int, _ptr, _ptr]
    lib.seld_gru_from_pair_tile.argtypes = [_ptr, _int, _i64, _i64, _ptr, _ptr, _ptr]
    lib.seld_gru_previous_state.argtypes = [_ptr, _int, _i64, _i64, _ptr, _ptr]
    lib.seld_gru_tile_rows.restype = _i64
    lib.seld_gru_tile_rows.argtypes = []
    lib.seld_gru_forward.argtypes = [_ptr, _int, _ptr, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr]
    lib.seld_gru_backward.argtypes = [_ptr, _ptr, _ptr, _int, _ptr, _i64, _i64, _i64, _ptr, _ptr, _ptr]
    return lib


def load_library():
    """Load (once) and return the ctypes handle.  Fails loudly when the .so is absent."""
    global _lib
    if _lib is not None:                     # hot path: ~40 calls per optimiser iteration on an enqueue-bound step
        return _lib
    with _lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise SeldNativeError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    f"or `make -C {_HERE / 'csrc'}`.  There is no CPU fallback.")
            lib = _declare(ctypes.CDLL(str(LIB_PATH)))
            global GRU_TILE
            GRU_TILE = int(lib.seld_gru_tile_rows())       # geometry of THIS build, before anything sizes a buffer
            _lib = lib
        return _lib


def check(rc: int, what: str = "libseld_hip"):
    if rc != 0:
        msg = load_library().seld_last_error()
        raise SeldNativeError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_ptr(device) -> ctypes.c_void_p:
    """The caller's current stream on ``device`` as a hipStream_t.  The raw accessor skips the Stream wrapper object
    (~5 us per call; two dozen launches per iteration take their stream here)."""
    if _raw_stream is not None:
        index = device.index if isinstance(device, torch.device) else torch.device(device).index
        return ctypes.c_void_p(_raw_stream(index if index is not None else torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def mel_filterbank() -> torch.Tensor:
    """fp32 HTK filterbank [481, 64] built with the same torch ops as
    ``torchaudio.functional.melscale_fbanks(481, 0, 12000, 64, 24000, norm=None, mel_scale='htk')``
    (the constant inside the MelScale the reference instantiates at dataset.py:38-43)."""
    import math
    all_freqs = torch.linspace(0, 24000 // 2, N_BINS)
    m_max = 2595.0 * math.log10(1.0 + 12000.0 / 700.0)
    m_pts = torch.linspace(0.0, m_max, N_MELS + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    rising = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    falling = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(rising, falling), min=0.0).contiguous()


def ensure_init(device) -> int:
    """seld_init for ``device`` (idempotent) and upload of the torch-built mel table."""
    if isinstance(device, torch.device) and device.index in _initialised_devices:     # hot path, no lock
        return device.index
    device = torch.device(device)
    if device.type != "cuda":
        raise SeldNativeError(f"the SELD HIP path needs a ROCm device, got {device}")
    index = device.index if device.index is not None else torch.cuda.current_device()
    lib = load_library()
    with _lock:
        if index in _initialised_devices:
            return index
    with _device_guard(index):
        check(lib.seld_init(index), "seld_init")
        fb = mel_filterbank()
        check(lib.seld_set_mel_filterbank(ctypes.c_void_p(fb.data_ptr())), "seld_set_mel_filterbank")
        win = torch.hann_window(N_FFT, periodic=True, dtype=torch.float32).contiguous()
        check(lib.seld_set_window(ctypes.c_void_p(win.data_ptr())), "seld_set_window")
    with _lock:
        _initialised_devices.add(index)
    return index


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device_guard(index: int):
    """``torch.cuda.device(index)`` only when a switch is needed: one process drives one GPU (one rank per GPU), so
    the current device is almost always right and the context manager's two runtime calls (~15 us of host time per
    kernel launch on a launch-bound model) are skipped."""
    return _NO_GUARD if torch.cuda.current_device() == index else torch.cuda.device(index)


def num_frames(num_samples: int) -> int:
    return 1 + int(num_samples) // HOP


def logmel(pcm: torch.Tensor, layout: str = "cft", out: torch.Tensor | None = None) -> torch.Tensor:
    """Fused log-mel on the GPU.  ``pcm``: [N, C, L] or [C, L], float32 in [-1, 1) or int16.

    layout 'cft' -> [N, C, 64, F] (reference layout, dataset.py:53)
    layout 'tcf' -> [N, F, C, 64] (time-major; windows are contiguous slices)
    """
    squeeze = pcm.dim() == 2
    if squeeze:
        pcm = pcm.unsqueeze(0)
    if pcm.dim() != 3:
        raise ValueError("pcm must be [N, C, L] or [C, L]")
    if not pcm.is_cuda:
        raise SeldNativeError("logmel: pcm must live on the GPU (no CPU fallback in the product path)")
    if pcm.dtype not in (torch.float32, torch.int16):
        raise TypeError(f"logmel: pcm dtype must be float32 or int16, got {pcm.dtype}")
    pcm = pcm.contiguous()
    n, c, length = pcm.shape
    index = ensure_init(pcm.device)
    frames = num_frames(length)
    code = {"cft": 0, "tcf": 1}[layout]
    shape = (n, c, N_MELS, frames) if code == 0 else (n, frames, c, N_MELS)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=pcm.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError(f"logmel: out must be contiguous float32 {shape}")
    lib = load_library()
    fn = lib.seld_logmel_f32 if pcm.dtype == torch.float32 else lib.seld_logmel_i16
    with _device_guard(index):
        check(fn(ctypes.c_void_p(pcm.data_ptr()), n, c, length, ctypes.c_void_p(out.data_ptr()), code,
                 _stream_ptr(pcm.device)), "seld_logmel")
    return out[0] if squeeze else out


# --------------------------------------------------------------------------- labels / windows

GRID_I, GRID_J = 18, 36
NUM_CLASSES = 14


def _p(t: torch.Tensor | None):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def rasterise_labels(events: torch.Tensor, total_frames: int, I: int = GRID_I, J: int = GRID_J,
                     device=None, out: torch.Tensor | None = None) -> torch.Tensor:
    """dataset.py:60-119 on the GPU.  ``events``: int [R, 5] rows (meta_frame, class, source, az, el).
    Returns the compact class mask, uint16 [total_frames, I*J] (bit c <=> class c active)."""
    events = torch.as_tensor(events)
    if events.numel() and (events.dim() != 2 or events.shape[1] < 5):
        raise ValueError("events must be [R, >=5]")
    if events.numel() and not events.is_cuda:      # host rows are validated; device rows are trusted (no sync)
        cls = events[:, 1]
        if int(cls.max()) >= NUM_CLASSES or int(cls.min()) < 0:
            raise IndexError("metadata class index out of range for 14 classes (dataset.py:110)")
    device = torch.device(device) if device is not None else (events.device if events.is_cuda else None)
    if device is None:
        raise SeldNativeError("rasterise_labels: pass device= (no CPU fallback)")
    index = ensure_init(device)
    ev = events[:, :5].to(device=device, dtype=torch.int32).contiguous() if events.numel() else \
        torch.zeros((0, 5), dtype=torch.int32, device=device)
    if out is None:
        out = torch.empty((total_frames, I * J), dtype=torch.uint16, device=device)
    elif tuple(out.shape) != (total_frames, I * J) or out.dtype != torch.uint16 or not out.is_contiguous():
        raise ValueError("rasterise_labels: out must be a contiguous uint16 [total_frames, I*J] tensor")
    with _device_guard(index):
        check(load_library().seld_labels_rasterise(_p(ev), ev.shape[0], total_frames, I, J, _p(out),
                                                   _stream_ptr(device)), "seld_labels_rasterise")
    return out


def gaussian_source_noise(events, sigma_az: float = 5.0, sigma_el: float = 5.0, rng=None):
    """One (azimuth, elevation) normal draw per unique (class, source) pair, in the sorted-key order of
    ``df.groupby([1, 2])`` (smrl_seld_gaussian.py:426-437).  Returns float64 [R, 2]: the box centre of every row.
    ``rng``: a numpy Generator / RandomState; default ``numpy.random`` (the reference never seeds)."""
    import numpy as np
    ev = np.asarray(events)[:, :5].astype(np.int64)
    rng = np.random if rng is None else rng
    keys = sorted({(int(c), int(s)) for c, s in ev[:, 1:3]})
    noise = {k: (rng.normal(0, sigma_az), rng.normal(0, sigma_el)) for k in keys}
    centres = np.empty((ev.shape[0], 2), dtype=np.float64)
    for r, row in enumerate(ev):
        dn = noise[(int(row[1]), int(row[2]))]
        centres[r] = (row[3] + dn[0], row[4] + dn[1])
    return centres


def rasterise_labels_gaussian(events, centres, total_frames: int, I: int = GRID_I, J: int = GRID_J,
                              sigma_az: float = 5.0, sigma_el: float = 5.0, device=None) -> torch.Tensor:
    """smrl_seld_gaussian.py:397-534 on the GPU: uint16 class mask [total_frames, I*J]."""
    events = torch.as_tensor(events)
    if events.numel() and not events.is_cuda:
        cls = events[:, 1]
        if int(cls.max()) >= NUM_CLASSES or int(cls.min()) < 0:
            raise IndexError("metadata class index out of range for 14 classes")
    device = torch.device(device) if device is not None else (events.device if events.is_cuda else None)
    if device is None:
        raise SeldNativeError("rasterise_labels_gaussian: pass device= (no CPU fallback)")
    index = ensure_init(device)
    ev = events[:, :5].to(device=device, dtype=torch.int32).contiguous() if events.numel() else \
        torch.zeros((0, 5), dtype=torch.int32, device=device)
    ctr = torch.as_tensor(centres, dtype=torch.float64).to(device).contiguous().reshape(-1, 2)
    if ctr.shape[0] != ev.shape[0]:
        raise ValueError("one box centre per metadata row")
    out = torch.empty((total_frames, I * J), dtype=torch.uint16, device=device)
    with _device_guard(index):
        check(load_library().seld_labels_rasterise_box(_p(ev), _p(ctr), ev.shape[0], total_frames, I, J, float(sigma_az),
                                                       float(sigma_el), _p(out), _stream_ptr(device)),
              "seld_labels_rasterise_box")
    return out


def expand_labels(mask: torch.Tensor, num_classes: int = NUM_CLASSES) -> torch.Tensor:
    """uint16 [..., G] -> float32 [..., G, num_classes] (dataset.py:110-117 semantics)."""
    if not mask.is_cuda or mask.dtype != torch.uint16:
        raise SeldNativeError("expand_labels: mask must be a uint16 GPU tensor")
    mask = mask.contiguous()
    index = ensure_init(mask.device)
    out = torch.empty(tuple(mask.shape) + (num_classes,), dtype=torch.float32, device=mask.device)
    with _device_guard(index):
        check(load_library().seld_labels_expand(_p(mask), mask.numel(), num_classes, _p(out),
                                                _stream_ptr(mask.device)), "seld_labels_expand")
    return out


def gather_windows(src: torch.Tensor, starts: torch.Tensor, window: int, out: torch.Tensor | None = None) -> torch.Tensor:
    """dataset.py:267-317: src [T, ...] (time-major rows) -> [B, window, ...], zero padded past T.  ``out``: write into
    this contiguous tensor of that shape (the static input buffer of a captured training step)."""
    if not src.is_cuda:
        raise SeldNativeError("gather_windows: src must be a GPU tensor")
    src = src.contiguous()
    row_bytes = src[0].numel() * src.element_size() if src.shape[0] else 0
    index = ensure_init(src.device)
    starts = starts.to(device=src.device, dtype=torch.int64).contiguous()
    shape = (starts.numel(), window) + tuple(src.shape[1:])
    if out is None:
        out = torch.empty(shape, dtype=src.dtype, device=src.device)
    elif tuple(out.shape) != shape or out.dtype != src.dtype or out.device != src.device or not out.is_contiguous():
        raise ValueError(f"gather_windows: out must be a contiguous {src.dtype} tensor of shape {shape} on {src.device}")
    if row_bytes == 0:
        raise ValueError("gather_windows: empty source")
    with _device_guard(index):
        check(load_library().seld_window_gather(_p(src), src.shape[0], row_bytes, _p(starts), starts.numel(),
                                                window, _p(out), _stream_ptr(src.device)), "seld_window_gather")
    return out


# --------------------------------------------------------------------------- fused loss

_workspaces = {}


def _workspace(device) -> torch.Tensor:
    key = (device.type, device.index)
    ws = _workspaces.get(key)
    if ws is None:
        ws = torch.empty(load_library().seld_softmax_mse_workspace_bytes(), dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def softmax_mse(logits: torch.Tensor, labels: torch.Tensor, grad_scale: float | None = None):
    """loss.py:43-54 fused.  ``labels``: uint16 mask [...cells] or dense float32 [...cells, 14].
    Returns (loss scalar tensor, grad or None); grad has the dtype/shape of ``logits``."""
    if not logits.is_cuda:
        raise SeldNativeError("softmax_mse: logits must be a GPU tensor")
    if logits.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("softmax_mse: logits must be float32 or bfloat16")
    logits = logits.contiguous()
    m = logits.shape[-1]
    n_cells = logits.numel() // m
    index = ensure_init(logits.device)
    labels = labels.contiguous()
    if labels.dtype == torch.uint16:
        if labels.numel() != n_cells:
            raise ValueError("softmax_mse: mask shape does not match logits")
        mask_p, dense_p = _p(labels), None
    else:
        if labels.dtype != torch.float32 or labels.numel() != logits.numel():
            raise ValueError("softmax_mse: dense labels must be float32 with the logits' shape")
        mask_p, dense_p = None, _p(labels)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits) if grad_scale is not None else None
    with _device_guard(index):
        check(load_library().seld_softmax_mse(_p(logits), int(logits.dtype == torch.bfloat16), mask_p, dense_p,
                                              n_cells, m, float(grad_scale or 0.0), _p(loss), _p(grad),
                                              _p(_workspace(logits.device)), _stream_ptr(logits.device)),
              "seld_softmax_mse")
    return loss[0], grad


def smr_loss(logits: torch.Tensor, labels: torch.Tensor, grid, w_class: float, w_aiur: float, w_cl: float,
             want_grad: bool):
    """smrl_seld_gaussian.py:946-1072 fused (csrc/loss3.hip).  logits [..., G, 14] with G = grid[0] * grid[1]; labels:
    uint16 mask [..., G] or dense float32 [..., G, 14].  Returns (terms [4] fp32 = total, mse, aiur, cl; grad like logits
    or None)."""
    if not logits.is_cuda:
        raise SeldNativeError("smr_loss: logits must be a GPU tensor")
    if logits.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("smr_loss: logits must be float32 or bfloat16")
    rows, cols = int(grid[0]), int(grid[1])
    m = logits.shape[-1]
    cells = rows * cols
    if logits.dim() < 2 or logits.shape[-2] != cells:
        raise ValueError(f"smr_loss: logits must be [..., {cells}, {m}] for a {rows} x {cols} grid")
    logits = logits.contiguous()
    frames = logits.numel() // (cells * m)
    labels = labels.contiguous()
    if labels.dtype == torch.uint16:
        if labels.numel() != frames * cells:
            raise ValueError("smr_loss: mask shape does not match logits")
        mask_p, dense_p = _p(labels), None
    else:
        if labels.dtype != torch.float32 or labels.numel() != logits.numel():
            raise ValueError("smr_loss: dense labels must be float32 with the logits' shape")
        mask_p, dense_p = None, _p(labels)
    index = ensure_init(logits.device)
    lib = load_library()
    out = torch.empty(4, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits) if want_grad else None
    ws = torch.empty(lib.seld_smr_loss_workspace_bytes(frames, cells), dtype=torch.uint8, device=logits.device)
    with _device_guard(index):
        check(lib.seld_smr_loss(_p(logits), int(logits.dtype == torch.bfloat16), mask_p, dense_p, frames, rows, cols, m,
                                float(w_class), float(w_aiur), float(w_cl), _p(out), _p(grad), _p(ws),
                                _stream_ptr(logits.device)), "seld_smr_loss")
    return out, grad


_unit_gradients = {}        # device index -> a persistent ones scalar handed to autograd as the loss's upstream gradient


def unit_gradient(device) -> torch.Tensor:
    """The ones scalar ``loss.backward()`` would create with a fill kernel, kept per device: a caller that seeds the
    backward pass with it (``torch.autograd.backward(loss, unit_gradient(device))``) saves that launch, and
    ``scale_by_device_scalar_`` recognises it by address and skips its own (the upstream gradient IS one)."""
    index = device.index if device.index is not None else torch.cuda.current_device()
    t = _unit_gradients.get(index)
    if t is None:
        t = _unit_gradients[index] = torch.ones((), dtype=torch.float32, device=device)
    return t


def scale_by_device_scalar_(data: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """data *= scale (a one-element fp32 GPU tensor) in place, without reading the scalar on the host; a scale of
    exactly 1.0 costs one empty launch.  Falls back to torch when the layout does not fit the kernel."""
    if scale.is_cuda and scale.numel() == 1:
        unit = _unit_gradients.get(scale.device.index)
        if unit is not None and scale.data_ptr() == unit.data_ptr():
            return data
    if (not data.is_cuda or data.dtype not in (torch.float32, torch.bfloat16) or not data.is_contiguous()
            or data.numel() % 8 or data.data_ptr() % 16 or scale.numel() != 1):
        return data.mul_(scale.to(data.dtype))
    scale = scale.reshape(1).to(device=data.device, dtype=torch.float32)
    with _device_guard(ensure_init(data.device)):
        check(load_library().seld_scale_by_device_scalar(_p(data), int(data.dtype == torch.bfloat16), data.numel(),
                                                         _p(scale), _stream_ptr(data.device)),
              "seld_scale_by_device_scalar")
    return data


def layernorm_supported(d: int) -> bool:
    return bool(load_library().seld_layernorm_supported(int(d)))


def layernorm_forward(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float, relu: bool):
    """x [..., D] contiguous fp32 / bf16; weight, bias [D] fp32 -> (y like x, mean_rstd [rows, 2] fp32)."""
    d = x.shape[-1]
    rows = x.numel() // d
    y = torch.empty_like(x)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=x.device)
    with _device_guard(ensure_init(x.device)):
        check(load_library().seld_layernorm_forward(_p(x), int(x.dtype == torch.bfloat16), rows, d, _p(weight), _p(bias),
                                                    float(eps), int(relu), _p(y), _p(stats), _stream_ptr(x.device)),
              "seld_layernorm_forward")
    return y, stats


def layernorm_backward(x: torch.Tensor, dy: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                       stats: torch.Tensor, relu: bool):
    """-> (dx like x, dweight [D] fp32, dbias [D] fp32)."""
    d = x.shape[-1]
    rows = x.numel() // d
    lib = load_library()
    dx = torch.empty_like(x)
    dweight = torch.empty((d,), dtype=torch.float32, device=x.device)
    dbias = torch.empty((d,), dtype=torch.float32, device=x.device)
    work = torch.empty((lib.seld_layernorm_workspace_floats(rows, d),), dtype=torch.float32, device=x.device)
    with _device_guard(ensure_init(x.device)):
        check(lib.seld_layernorm_backward(_p(x), _p(dy), int(x.dtype == torch.bfloat16), rows, d, _p(weight), _p(bias),
                                          _p(stats), int(relu), _p(dx), _p(dweight), _p(dbias), _p(work),
                                          _stream_ptr(x.device)), "seld_layernorm_backward")
    return dx, dweight, dbias


def stream_delay(device, nanoseconds: int) -> None:
    """Hold the CURRENT stream of ``device`` for ``nanoseconds`` (seld_stream_delay)."""
    with _device_guard(ensure_init(device)):
        check(load_library().seld_stream_delay(int(nanoseconds), _stream_ptr(device)), "seld_stream_delay")


def _dense_like(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Same shape and strides, and the storage is walked exactly once (contiguous in some memory format)."""
    if a.shape != b.shape or a.device != b.device:
        return False
    if any(sa != sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1):     # size-1 dims: any stride
        return False
    return a.is_contiguous() or (a.dim() == 4 and a.is_contiguous(memory_format=torch.channels_last))


def multi_cast(srcs, dsts, cache=None) -> bool:
    """dsts[i] <- srcs[i] for lists of GPU tensors, bf16 -> fp32 or fp32 -> bf16 (all pairs the same direction), in one
    launch per 96 tensors (csrc/cast.hip).  Returns False -- having done nothing -- when a pair does not share its
    memory layout, so that the caller can fall back to the framework's copies.  ``cache``: a dict owned by a caller
    that passes the SAME parameter lists every iteration (the optimiser): the layout checks and the descriptor arrays
    are reused while every address is unchanged (the step is enqueue-bound: this is ~80 us of host time per call)."""
    srcs, dsts = list(srcs), list(dsts)
    if not srcs:
        return True
    if cache is not None:
        key = tuple(t.data_ptr() for t in srcs) + tuple(t.data_ptr() for t in dsts)
        hit = cache.get("key") == key
        if hit:
            src, dst, lengths, n, to_float = cache["args"]
            device = srcs[0].device
            with _device_guard(ensure_init(device)):
                check(load_library().seld_multi_cast(src, dst, lengths, n, to_float, _stream_ptr(device)),
                      "seld_multi_cast")
            return True
    to_float = srcs[0].dtype == torch.bfloat16
    want = (torch.bfloat16, torch.float32) if to_float else (torch.float32, torch.bfloat16)
    for s, d in zip(srcs, dsts):
        if s.dtype != want[0] or d.dtype != want[1] or not s.is_cuda or not _dense_like(s, d):
            return False
    n = len(srcs)
    src = (ctypes.c_void_p * n)(*[s.data_ptr() for s in srcs])
    dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d in dsts])
    lengths = (ctypes.c_int64 * n)(*[s.numel() for s in srcs])
    device = srcs[0].device
    if cache is not None:
        cache["key"], cache["args"] = key, (src, dst, lengths, n, int(to_float))
    with _device_guard(ensure_init(device)):
        check(load_library().seld_multi_cast(src, dst, lengths, n, int(to_float), _stream_ptr(device)), "seld_multi_cast")
    return True


def multi_adam(grads, params, exp_avgs, exp_avg_sqs, lows, lr: torch.Tensor, step: torch.Tensor, beta1: float, beta2: float,
               eps: float, weight_decay: float, grad_scale: float = 1.0, cache=None) -> bool:
    """One Adam update (csrc/adam.hip; the arithmetic of torch's fused Adam, L2 weight decay) of fp32 ``params`` from
    ``grads`` (bf16 or fp32, same layout as their parameter), rewriting the bf16 working copies ``lows[i]`` (or None) from
    the new values; ``lr`` / ``step`` are fp32 device scalars (``step`` already incremented).  Returns False -- having done
    nothing -- when a tensor does not qualify (layout / dtype), so that the caller can take the framework's path.
    ``cache``: dict of a caller that passes the same lists every iteration (descriptor arrays reused while no address
    changed)."""
    n = len(params)
    if n == 0:
        return True
    key = None
    if cache is not None:
        key = tuple(t.data_ptr() for t in grads) + tuple(t.data_ptr() for t in params) + \
            tuple(0 if t is None else t.data_ptr() for t in lows)
        if cache.get("key") == key:
            args = cache["args"]
        else:
            args = None
    else:
        args = None
    device = params[0].device
    if args is None:
        for g, p, m, v, lo in zip(grads, params, exp_avgs, exp_avg_sqs, lows):
            if not (p.is_cuda and p.dtype == torch.float32 and m.dtype == torch.float32 and v.dtype == torch.float32
                    and g.dtype in (torch.float32, torch.bfloat16) and _dense_like(g, p) and _dense_like(m, p)
                    and _dense_like(v, p) and (lo is None or (lo.dtype == torch.bfloat16 and _dense_like(lo, p)))):
                return False
        args = ((ctypes.c_void_p * n)(*[g.data_ptr() for g in grads]),
                (ctypes.c_int32 * n)(*[int(g.dtype == torch.bfloat16) for g in grads]),
                (ctypes.c_void_p * n)(*[p.data_ptr() for p in params]),
                (ctypes.c_void_p * n)(*[m.data_ptr() for m in exp_avgs]),
                (ctypes.c_void_p * n)(*[v.data_ptr() for v in exp_avg_sqs]),
                (ctypes.c_void_p * n)(*[0 if lo is None else lo.data_ptr() for lo in lows]),
                (ctypes.c_int64 * n)(*[p.numel() for p in params]))
        if cache is not None:
            cache["key"], cache["args"] = key, args
    if not (lr.is_cuda and step.is_cuda and lr.dtype == torch.float32 and step.dtype == torch.float32):
        return False
    with _device_guard(ensure_init(device)):
        check(load_library().seld_multi_adam(*args, n, _p(lr), _p(step), float(beta1), float(beta2), float(eps),
                                             float(weight_decay), float(grad_scale), _stream_ptr(device)), "seld_multi_adam")
    return True


# --------------------------------------------------------------------------- CNN block tail (BN + ReLU + pool)

def conv_tail_supported(channels: int) -> bool:
    return channels % 8 == 0 and 256 % (channels // 8) == 0


def _tail_view(x: torch.Tensor):
    """4-D activation in channels-last memory order -> (rows, C); raises when the memory order is anything else."""
    if x.dim() != 4 or not x.is_cuda or x.dtype not in (torch.float32, torch.bfloat16):
        raise SeldNativeError("conv_tail: x must be a 4-D GPU tensor of float32 or bfloat16")
    if not x.is_contiguous(memory_format=torch.channels_last):
        raise SeldNativeError("conv_tail: x must be in channels-last memory order")
    b, c, t, f = x.shape
    return b * t * f, c


def conv_tail_forward(x, weight, bias, running_mean, running_var, momentum, eps, training, pool, residual=None):
    """model_crnn.py:5-17 after the convolution: BatchNorm2d -> ReLU -> MaxPool2d((1, pool)) on a channels-last
    [B, C, T, F] activation; with ``residual`` (pool = 1): BatchNorm2d -> + residual -> ReLU (resnet50_model.py:30-52).
    Returns (y [B, C, T, F // pool] channels-last, mean_invstd [2, C], scale_shift [2, C])."""
    rows, c = _tail_view(x)
    if residual is not None:
        if pool != 1 or residual.shape != x.shape or residual.dtype != x.dtype:
            raise ValueError("conv_tail: the residual must match x and needs pool = 1")
        _tail_view(residual)
    b, _, t, f = x.shape
    width = 2 if pool == 2 else 1                   # pool = 4: no pooling, SiLU instead of ReLU (BatchNorm1d -> Swish)
    if f % width:
        raise ValueError("conv_tail: the frequency extent must be a multiple of the pool width")
    index = ensure_init(x.device)
    y = torch.empty((b, c, t, f // width), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    stats = torch.empty((2, 2, c), dtype=torch.float32, device=x.device)
    lib = load_library()
    ws = torch.empty(lib.seld_conv_tail_workspace_floats(c), dtype=torch.float32, device=x.device)
    with _device_guard(index):
        check(lib.seld_conv_tail_forward(_p(x), _p(residual), int(x.dtype == torch.bfloat16), rows, c, pool, _p(weight),
                                         _p(bias),
                                         _p(running_mean), _p(running_var), float(momentum), float(eps), int(training),
                                         _p(y), _p(stats[0]), _p(stats[1]), _p(ws), _stream_ptr(x.device)),
              "seld_conv_tail_forward")
    return y, stats[0], stats[1]


def conv_tail_backward(x, dy, mean_invstd, scale_shift, pool, residual=None):
    """-> (dx like x, dweight [C] fp32, dbias [C] fp32[, dresidual like x when ``residual`` is given])."""
    rows, c = _tail_view(x)
    if dy.dtype != x.dtype or not dy.is_contiguous(memory_format=torch.channels_last):
        dy = dy.to(x.dtype).contiguous(memory_format=torch.channels_last)
    index = ensure_init(x.device)
    dx = torch.empty_like(x, memory_format=torch.channels_last)
    dres = torch.empty_like(x, memory_format=torch.channels_last) if residual is not None else None
    dwb = torch.empty((2, c), dtype=torch.float32, device=x.device)
    lib = load_library()
    ws = torch.empty(lib.seld_conv_tail_workspace_floats(c), dtype=torch.float32, device=x.device)
    with _device_guard(index):
        check(lib.seld_conv_tail_backward(_p(x), _p(residual), _p(dy), int(x.dtype == torch.bfloat16), rows, c, pool,
                                          _p(mean_invstd), _p(scale_shift), _p(dx), _p(dres), _p(dwb[0]), _p(dwb[1]),
                                          _p(ws), _stream_ptr(x.device)), "seld_conv_tail_backward")
    if residual is not None:
        return dx, dwb[0], dwb[1], dres
    return dx, dwb[0], dwb[1]


# --------------------------------------------------------------------------- depthwise Conv1d (Conformer conv module)

def dwconv1d_supported(d: int, k: int) -> bool:
    return d % 64 == 0 and 1 <= k <= 31 and k % 2 == 1


def dwconv1d(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None, flip: bool = False) -> torch.Tensor:
    """Depthwise Conv1d over time on channels-last activations: x [B, T, D], weight [D, K] (fp32), bias [D] or None."""
    if not x.is_cuda or x.dim() != 3 or x.dtype not in (torch.float32, torch.bfloat16):
        raise SeldNativeError("dwconv1d: x must be a GPU tensor [B, T, D] of float32 or bfloat16")
    x = x.contiguous()
    b, t, d = x.shape
    w = weight.to(torch.float32).contiguous()
    bias = None if bias is None else bias.to(torch.float32).contiguous()
    y = torch.empty_like(x)
    with _device_guard(ensure_init(x.device)):
        check(load_library().seld_dwconv1d(_p(x), int(x.dtype == torch.bfloat16), _p(w), _p(bias), b, t, d, w.shape[1],
                                           int(flip), _p(y), _stream_ptr(x.device)), "seld_dwconv1d")
    return y


def dwconv1d_wgrad(x: torch.Tensor, dy: torch.Tensor, k: int):
    """-> (dweight [D, K] fp32, dbias [D] fp32) of the depthwise convolution."""
    x, dy = x.contiguous(), dy.to(x.dtype).contiguous()
    b, t, d = x.shape
    rows = int(load_library().seld_dwconv1d_wgrad_rows(b, t))          # one per batch row and 50-step time chunk
    partial = torch.empty((rows, d, 32), dtype=torch.float32, device=x.device)
    with _device_guard(ensure_init(x.device)):
        check(load_library().seld_dwconv1d_wgrad(_p(x), _p(dy), int(x.dtype == torch.bfloat16), b, t, d, k, _p(partial),
                                                 _stream_ptr(x.device)), "seld_dwconv1d_wgrad")
    total = partial.sum(dim=0)
    return total[:, :k], total[:, 31]


# --------------------------------------------------------------------------- GRU recurrence

GRU_H = 256
GRU_TILE = 4          # sequences per workgroup; set from seld_gru_tile_rows() by load_library()


def _tile_geometry():
    """(sequences per tile, lanes sharing a sequence, units per lane) of the loaded library (8, 2, 4) or (4, 4, 2).
    Without a built library (CPU-only checks of the torch layout definitions) the default build's geometry."""
    if LIB_PATH.exists():
        load_library()
    parts = 16 // GRU_TILE
    return GRU_TILE, parts, 8 // parts


def to_tile(x: torch.Tensor, ns: int) -> torch.Tensor:
    """[B, T, 2, ns, 256] -> the kernels' tile layout [tiles, T, 2, 8(w), ns, 4(q), parts, seqs, units] (batch
    zero-padded to whole tiles); the torch definition of what seld_gru_to_tile does.  Hidden unit
    u = 32*w + 16*s + 4*q + i with the flat index 4*s + i = part*units + j; row b = seqs*tile + seq; the lane
    dimensions (q, part, seq) are the wavefront lane q*16 + part*seqs + seq."""
    seqs, parts, units = _tile_geometry()
    b, t = x.shape[0], x.shape[1]
    tiles = (b + seqs - 1) // seqs
    if tiles * seqs != b:
        x = torch.cat((x, x.new_zeros((tiles * seqs - b,) + tuple(x.shape[1:]))), dim=0)
    x = x.reshape(tiles, seqs, t, 2, ns, 8, 2, 4, 4)                 # tile, seq, T, dir, slot, w, s, q, i
    x = x.permute(0, 2, 3, 5, 4, 7, 6, 8, 1)                         # tile, T, dir, w, slot, q, s, i, seq
    x = x.reshape(tiles, t, 2, 8, ns, 4, parts, units, seqs)         # (s, i) -> (part, j)
    return x.permute(0, 1, 2, 3, 4, 5, 6, 8, 7).contiguous()         # tile, T, dir, w, slot, q, part, seq, j


def from_tile(x: torch.Tensor, batch: int) -> torch.Tensor:
    """Inverse of to_tile: [tiles, T, 2, 8, ns, 4, parts, seqs, units] -> [batch, T, 2, ns, 256]."""
    seqs, parts, units = _tile_geometry()
    tiles, t, ns = x.shape[0], x.shape[1], x.shape[4]
    y = x.permute(0, 7, 1, 2, 4, 3, 5, 6, 8).reshape(tiles, seqs, t, 2, ns, 8, 4, 2, 4)   # .., w, q, s, i
    return y.permute(0, 1, 2, 3, 4, 5, 7, 6, 8).reshape(tiles * seqs, t, 2, ns, GRU_H)[:batch]


def from_pair_tile(x: torch.Tensor, batch: int):
    """The pair-slot layout of the backward kernel's output, [tiles, T, 2, 8(w), 2(pair slot), 4(q), parts, seqs,
    2(member), units] with slots (da_r|da_z), (da_n|da_n*r) -> (dgi [batch, T, 2, 3, 256] = (da_r, da_z, da_n),
    dghn [batch, T, 2, 256] = da_n*r); the torch definition of what seld_gru_from_pair_tile does."""
    seqs, parts, units = _tile_geometry()
    tiles, t = x.shape[0], x.shape[1]
    y = x.permute(0, 7, 1, 2, 4, 8, 3, 5, 6, 9).reshape(tiles, seqs, t, 2, 4, 8, 4, 2, 4)  # .., slot, w, q, s, i
    y = y.permute(0, 1, 2, 3, 4, 5, 7, 6, 8).reshape(tiles * seqs, t, 2, 4, GRU_H)[:batch]
    return y[:, :, :, :3].contiguous(), y[:, :, :, 3].contiguous()


def to_tile_device(x: torch.Tensor, ns: int) -> torch.Tensor:
    """``to_tile`` by the HIP permute kernel (x: contiguous GPU tensor [B, T, 2, ns, 256], bf16 or fp32)."""
    seqs, parts, units = _tile_geometry()
    b, t = x.shape[0], x.shape[1]
    tiles = (b + seqs - 1) // seqs
    x = x.contiguous()
    out = torch.empty((tiles, t, 2, 8, ns, 4, parts, seqs, units), dtype=x.dtype, device=x.device)
    with _device_guard(ensure_init(x.device)):
        check(load_library().seld_gru_to_tile(_p(x), x.element_size(), b, t, ns, _p(out), _stream_ptr(x.device)),
              "seld_gru_to_tile")
    return out


def from_pair_tile_device(x: torch.Tensor, batch: int):
    """``from_pair_tile`` by the HIP permute kernel."""
    t = x.shape[1]
    dgi = torch.empty((batch, t, 2, 3, GRU_H), dtype=x.dtype, device=x.device)
    dghn = torch.empty((batch, t, 2, GRU_H), dtype=x.dtype, device=x.device)
    with _device_guard(ensure_init(x.device)):
        check(load_library().seld_gru_from_pair_tile(_p(x), x.element_size(), batch, t, _p(dgi), _p(dghn),
                                                     _stream_ptr(x.device)), "seld_gru_from_pair_tile")
    return dgi, dghn


def gru_previous_state(y: torch.Tensor) -> torch.Tensor:
    """y [B, T, 2*256] (h_t of both directions) -> h_{t-1} in each direction's own time order, same shape; zero at the
    first step (t = 0 forward, t = T-1 reverse)."""
    b, t, _ = y.shape
    y = y.contiguous()
    out = torch.empty_like(y)
    with _device_guard(ensure_init(y.device)):
        check(load_library().seld_gru_previous_state(_p(y), y.element_size(), b, t, _p(out), _stream_ptr(y.device)),
              "seld_gru_previous_state")
    return out


def gru_forward(gi: torch.Tensor, w_hh: torch.Tensor, b_hn: torch.Tensor, need_saved: bool):
    """gi [B, T, 2, 3H] (fp32 / bf16; must already include b_ih and the r/z part of b_hh), w_hh [2, 3H, H],
    b_hn [2, H] (n-gate recurrent bias) -> (y [B, T, 2H], saved (tile layout, opaque) or None).  y is a view of
    the first B rows of a buffer padded to whole 8-sequence tiles."""
    if not gi.is_cuda:
        raise SeldNativeError("gru_forward: tensors must live on the GPU")
    b, t, two, g3 = gi.shape
    h = g3 // 3
    if two != 2 or h != GRU_H or gi.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("gru_forward: gi must be [B, T, 2, 768] float32 or bfloat16")
    index = ensure_init(gi.device)
    gi = gi.contiguous()                    # read by the kernel as it is: the input GEMM's own output
    tiles = (b + GRU_TILE - 1) // GRU_TILE
    w = w_hh.to(torch.bfloat16).contiguous()
    bias = b_hn.to(torch.float32).contiguous()
    if tuple(bias.shape) != (2, h):
        raise ValueError("gru_forward: b_hn must be [2, H]")
    y = torch.empty((tiles * GRU_TILE, t, 2 * h), dtype=gi.dtype, device=gi.device)
    saved_dtype = torch.float16 if gi.dtype == torch.bfloat16 else torch.float32      # see include/seld_hip.h
    saved = torch.empty((tiles, t, 2, 8, 2, 64, 2, 8 * GRU_TILE // 16), dtype=saved_dtype, device=gi.device) \
        if need_saved else None
    with _device_guard(index):
        check(load_library().seld_gru_forward(_p(gi), int(gi.dtype == torch.bfloat16), _p(w), _p(bias), b, t,
                                              h, _p(y), _p(saved), _stream_ptr(gi.device)), "seld_gru_forward")
    return y[:b], saved


def gru_backward(dy: torch.Tensor, saved: torch.Tensor, y: torch.Tensor, w_hh: torch.Tensor, raw_bias: bool = False):
    """dy [B, T, 2H], the forward's (saved, y) -> (dgi [B, T, 2, 3, H] = (da_r, da_z, da_n), dghn [B, T, 2, H] =
    da_n*r, both of dy's dtype, dbias [2, 4, H] fp32 = the four slots summed over batch and time; with ``raw_bias``
    the per-tile sums [tiles, 2, 4, H] as the kernel left them, for ``gru_bias_grads``)."""
    b, t, h2 = dy.shape
    h = h2 // 2
    index = ensure_init(dy.device)
    if saved.dtype != (torch.float16 if dy.dtype == torch.bfloat16 else torch.float32):
        raise ValueError("gru_backward: saved activations do not belong to a forward pass of this dtype")
    if y.dtype != dy.dtype or tuple(y.shape) != (b, t, h2):
        raise ValueError("gru_backward: y must be the forward output matching dy")
    dy_tile = to_tile_device(dy.reshape(b, t, 2, 1, h), 1)
    tiles = dy_tile.shape[0]
    if tiles * GRU_TILE != b:               # the kernel reads whole tiles of y (h_{t-1}); pad rows are never used
        y = torch.cat((y, y.new_zeros((tiles * GRU_TILE - b, t, h2))), dim=0)
    y = y.contiguous()
    w_t = w_hh.to(torch.bfloat16).transpose(1, 2).contiguous()            # [2, H, 3H]
    dg_tile = torch.empty((tiles, t, 2, 8, 2, 4, 16 // GRU_TILE, GRU_TILE, 2, 8 * GRU_TILE // 16), dtype=dy.dtype,
                          device=dy.device)
    dbias = torch.empty((tiles, 2, 4, h), dtype=torch.float32, device=dy.device)
    with _device_guard(index):
        check(load_library().seld_gru_backward(_p(dy_tile), _p(saved), _p(y), int(dy.dtype == torch.bfloat16),
                                               _p(w_t), tiles, t, h, _p(dg_tile), _p(dbias), _stream_ptr(dy.device)),
              "seld_gru_backward")
    dgi, dghn = from_pair_tile_device(dg_tile, b)
    if raw_bias:
        return dgi, dghn, dbias
    return dgi, dghn, dbias.sum(dim=0) if tiles > 1 else dbias[0]


# --------------------------------------------------------------------------- glue kernels (csrc/glue.hip)

def _is_bf16(t: torch.Tensor) -> int:
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"float32 or bfloat16 expected, got {t.dtype}")
    return int(t.dtype == torch.bfloat16)


def gru_fold_bias(b_ih: torch.Tensor, b_hh: torch.Tensor, dtype: torch.dtype):
    """b_ih, b_hh: fp32 [2*3H] (forward rows, then reverse) -> (gi_bias [6H] in ``dtype`` = b_ih + the r / z rows of b_hh,
    b_hn [2, H] fp32): one launch instead of clone, fill, add, cast, slice-copy."""
    if not (b_ih.is_cuda and b_ih.dtype == torch.float32 and b_hh.dtype == torch.float32 and b_ih.is_contiguous()
            and b_hh.is_contiguous() and b_ih.numel() == b_hh.numel() and b_ih.numel() % 6 == 0):
        raise SeldNativeError("gru_fold_bias: b_ih, b_hh must be contiguous fp32 GPU tensors of 2 * 3H elements")
    h = b_ih.numel() // 6
    gi_bias = torch.empty(6 * h, dtype=dtype, device=b_ih.device)
    b_hn = torch.empty((2, h), dtype=torch.float32, device=b_ih.device)
    with _device_guard(ensure_init(b_ih.device)):
        check(load_library().seld_gru_fold_bias(_p(b_ih), _p(b_hh), h, _p(gi_bias), _is_bf16(gi_bias), _p(b_hn),
                                                _stream_ptr(b_ih.device)), "seld_gru_fold_bias")
    return gi_bias, b_hn


def gru_bias_grads(partial: torch.Tensor):
    """The backward recurrence's per-tile sums [tiles, 2, 4, H] fp32 -> (db_ih [2*3H], db_hh [2*3H]) fp32."""
    tiles, two, four, h = partial.shape
    if two != 2 or four != 4 or partial.dtype != torch.float32 or not partial.is_contiguous():
        raise SeldNativeError("gru_bias_grads: partial must be contiguous fp32 [tiles, 2, 4, H]")
    db = torch.empty((2, 6 * h), dtype=torch.float32, device=partial.device)
    with _device_guard(ensure_init(partial.device)):
        check(load_library().seld_gru_bias_grads(_p(partial), tiles, h, _p(db[0]), _p(db[1]),
                                                 _stream_ptr(partial.device)), "seld_gru_bias_grads")
    return db[0], db[1]


def sum_chunks(partial: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[...] = partial.sum(dim=0) accumulated in fp32 (partial [chunks, ...] contiguous bf16 / fp32; out contiguous)."""
    if not (partial.is_cuda and partial.is_contiguous() and out.is_contiguous()
            and tuple(partial.shape[1:]) == tuple(out.shape)):
        raise SeldNativeError("sum_chunks: partial [chunks, ...] and out [...] must be contiguous with matching shapes")
    with _device_guard(ensure_init(partial.device)):
        check(load_library().seld_sum_chunks(_p(partial), _is_bf16(partial), partial.shape[0], out.numel(), _p(out),
                                             _is_bf16(out), _stream_ptr(partial.device)), "seld_sum_chunks")
    return out


def column_sums_supported(g: torch.Tensor, out: torch.Tensor) -> bool:
    return (g.is_cuda and g.dim() == 2 and g.is_contiguous() and out.is_contiguous() and g.shape[1] % 8 == 0
            and g.dtype in (torch.float32, torch.bfloat16) and out.dtype in (torch.float32, torch.bfloat16)
            and out.numel() == g.shape[1] and g.data_ptr() % 16 == 0 and g.shape[0] > 0)


def column_sums(g: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[n] = sum_r g[r, n] accumulated in fp32 (the bias gradient of a Linear from its [rows, N] output gradient)."""
    if not column_sums_supported(g, out):
        raise SeldNativeError("column_sums: g [rows, N] contiguous bf16 / fp32 with N % 8 == 0, out [N] contiguous")
    lib = load_library()
    rows, n = g.shape
    blocks = int(lib.seld_column_sums_blocks(rows, n))
    partial = torch.empty((blocks, n), dtype=torch.float32, device=g.device)
    with _device_guard(ensure_init(g.device)):
        check(lib.seld_column_sums(_p(g), _is_bf16(g), rows, n, _p(partial), _p(out), _is_bf16(out), _stream_ptr(g.device)),
              "seld_column_sums")
    return out


def multi_sum_chunks(pairs) -> None:
    """``out[...] = partial.sum(dim=0)`` for every (partial [chunks, ...], out [...]) pair in ONE launch per 64 pairs
    (csrc/glue.hip multi_sum_chunks_kernel; fp32 accumulation in chunk order, like ``sum_chunks``)."""
    pairs = list(pairs)
    if not pairs:
        return
    for partial, out in pairs:
        if not (partial.is_cuda and partial.is_contiguous() and out.is_contiguous()
                and tuple(partial.shape[1:]) == tuple(out.shape) and partial.dtype in (torch.float32, torch.bfloat16)
                and out.dtype in (torch.float32, torch.bfloat16)):
            raise SeldNativeError("multi_sum_chunks: partial [chunks, ...] / out [...] contiguous bf16 or fp32")
    n = len(pairs)
    src = (ctypes.c_void_p * n)(*[p.data_ptr() for p, _ in pairs])
    dst = (ctypes.c_void_p * n)(*[o.data_ptr() for _, o in pairs])
    counts = (ctypes.c_int64 * n)(*[o.numel() for _, o in pairs])
    chunks = (ctypes.c_int32 * n)(*[p.shape[0] for p, _ in pairs])
    flags = (ctypes.c_int32 * n)(*[_is_bf16(p) | (_is_bf16(o) << 1) for p, o in pairs])
    device = pairs[0][1].device
    with _device_guard(ensure_init(device)):
        check(load_library().seld_multi_sum_chunks(src, dst, counts, chunks, flags, n, _stream_ptr(device)),
              "seld_multi_sum_chunks")


_column_scratch = {}      # device index -> fp32 scratch for the row-block partials, grown on demand
_retired_scratch = []


def multi_column_sums(pairs) -> None:
    """``out[n] = sum_r g[r, n]`` for every (g [rows, N], out [N]) pair in TWO launches per 40 pairs (csrc/glue.hip
    multi_column_partials_kernel / multi_column_finish_kernel: row blocks in parallel, added in a fixed order)."""
    pairs = list(pairs)
    if not pairs:
        return
    for g, out in pairs:
        if not column_sums_supported(g, out):
            raise SeldNativeError("multi_column_sums: g [rows, N] contiguous bf16 / fp32 with N % 8 == 0, out [N] contiguous")
    lib = load_library()
    n = len(pairs)
    src = (ctypes.c_void_p * n)(*[g.data_ptr() for g, _ in pairs])
    dst = (ctypes.c_void_p * n)(*[o.data_ptr() for _, o in pairs])
    rows = (ctypes.c_int64 * n)(*[g.shape[0] for g, _ in pairs])
    cols = (ctypes.c_int64 * n)(*[g.shape[1] for g, _ in pairs])
    flags = (ctypes.c_int32 * n)(*[_is_bf16(g) | (_is_bf16(o) << 1) for g, o in pairs])
    need_f = ctypes.c_int64(0)
    check(lib.seld_multi_column_sums_scratch(rows, cols, n, ctypes.byref(need_f)), "seld_multi_column_sums_scratch")
    device = pairs[0][0].device
    index = ensure_init(device)
    scratch = _column_scratch.get(index)
    if scratch is None or scratch.numel() < need_f.value:
        if torch.cuda.is_current_stream_capturing():
            raise SeldNativeError("multi_column_sums: the scratch must be sized by an eager call before a graph capture")
        if scratch is not None:
            _retired_scratch.append(scratch)          # a captured graph may still hold its address: never handed back
        scratch = torch.empty(max(need_f.value, 1 << 20), dtype=torch.float32, device=device)
        _column_scratch[index] = scratch
    with _device_guard(index):
        check(lib.seld_multi_column_sums(src, dst, rows, cols, flags, n, _p(scratch), scratch.numel(), _stream_ptr(device)),
              "seld_multi_column_sums")


def gru_dwhh_finish(p_gi: torch.Tensor, p_n: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """p_gi [chunks, 6H, 2H], p_n [chunks, 2H, 2H] (chunked products against both directions' h_prev) -> out [2, 3H, H]."""
    chunks = p_gi.shape[0]
    h = out.shape[2]
    if not (p_gi.is_contiguous() and p_n.is_contiguous() and out.is_contiguous() and p_gi.dtype == p_n.dtype
            and tuple(p_gi.shape) == (chunks, 6 * h, 2 * h) and tuple(p_n.shape) == (chunks, 2 * h, 2 * h)
            and tuple(out.shape) == (2, 3 * h, h)):
        raise SeldNativeError("gru_dwhh_finish: shapes must be [chunks, 6H, 2H], [chunks, 2H, 2H] -> [2, 3H, H]")
    with _device_guard(ensure_init(out.device)):
        check(load_library().seld_gru_dwhh_finish(_p(p_gi), _p(p_n), _is_bf16(p_gi), chunks, h, _p(out), _is_bf16(out),
                                                  _stream_ptr(out.device)), "seld_gru_dwhh_finish")
    return out


def conv_weight_flip_transpose(w: torch.Tensor) -> torch.Tensor:
    """w [O, I, 3, 3] in channels-last memory -> [I, O, 3, 3] in channels-last memory with wt[i,o,r,s] = w[o,i,2-r,2-s]
    (= ``w.transpose(0, 1).flip(2, 3).contiguous(memory_format=torch.channels_last)`` in one launch)."""
    o, i, kh, kw = w.shape
    if not (w.is_cuda and kh == 3 and kw == 3 and w.dtype in (torch.float32, torch.bfloat16)
            and w.is_contiguous(memory_format=torch.channels_last)):
        raise SeldNativeError("conv_weight_flip_transpose: 3x3 weights in channels-last memory expected")
    wt = torch.empty((i, o, 3, 3), dtype=w.dtype, device=w.device, memory_format=torch.channels_last)
    with _device_guard(ensure_init(w.device)):
        check(load_library().seld_conv_weight_flip_transpose(_p(w), w.element_size(), o, i, _p(wt),
                                                             _stream_ptr(w.device)), "seld_conv_weight_flip_transpose")
    return wt


# --------------------------------------------------------------------------- STFT / spatial features

def _prep_pcm(pcm):
    squeeze = pcm.dim() == 2
    if squeeze:
        pcm = pcm.unsqueeze(0)
    if pcm.dim() != 3 or not pcm.is_cuda or pcm.dtype not in (torch.float32, torch.int16):
        raise SeldNativeError("pcm must be a GPU tensor [N, C, L] (or [C, L]) of float32 or int16")
    return pcm.contiguous(), squeeze


def stft(pcm: torch.Tensor) -> torch.Tensor:
    """Complex STFT (960 / 480 / periodic Hann / center, reflect): [N, C, L] -> complex64 [N, C, F, 481]
    (frame-major; ``.transpose(-1, -2)`` gives torch.stft's [.., 481, F])."""
    pcm, squeeze = _prep_pcm(pcm)
    n, c, length = pcm.shape
    index = ensure_init(pcm.device)
    out = torch.empty((n, c, num_frames(length), N_BINS, 2), dtype=torch.float32, device=pcm.device)
    fn = load_library().seld_stft_f32 if pcm.dtype == torch.float32 else load_library().seld_stft_i16
    with _device_guard(index):
        check(fn(_p(pcm), n, c, length, _p(out), _stream_ptr(pcm.device)), "seld_stft")
    spec = torch.view_as_complex(out)
    return spec[0] if squeeze else spec


def spatial_features(pcm: torch.Tensor, kind: str) -> torch.Tensor:
    """Time-major feature tensor [N, F, C_total, 64] (float32) for one of
      'logmel'      C_total = C                       (the reference's features)
      'logmel_iv'   C_total = 4 + 3   (FOA: log-mel + mel-projected intensity vectors; needs C = 4)
      'logmel_gcc'  C_total = C + C(C-1)/2            (MIC: log-mel + GCC-PHAT, 2 <= C <= 8)."""
    pcm, squeeze = _prep_pcm(pcm)
    n, c, length = pcm.shape
    frames = num_frames(length)
    if kind == "logmel":
        out = logmel(pcm, layout="tcf")
        return out[0] if squeeze else out
    if kind == "logmel_iv":
        if c != 4:
            raise ValueError("FOA intensity vectors need 4 channels (W first)")
        extra = 3
    elif kind == "logmel_gcc":
        if not 2 <= c <= 8:
            raise ValueError("GCC-PHAT supports 2..8 channels")
        extra = c * (c - 1) // 2
    else:
        raise ValueError(f"unknown feature kind {kind!r}")
    index = ensure_init(pcm.device)
    total = c + extra
    out = torch.empty((n, frames, total, N_MELS), dtype=torch.float32, device=pcm.device)
    s_n, s_t, s_c, s_m = frames * total * N_MELS, total * N_MELS, N_MELS, 1
    lib = load_library()
    stream = _stream_ptr(pcm.device)
    import os
    mode = os.environ.get("SELD_GCC", "mfma")                # developer A/B: "fft" = the FFT kernel, "spectra" = the
    if kind == "logmel_gcc" and mode not in ("fft", "spectra"):     # matrix-core kernel fed with complex64 spectra
        # ("planar" is read by the library: the Q15 kernel that unpacks the words to fp32 in LDS)
        # default: the log-mel pass also writes every bin's phasor X / |X| as Q15 pairs (4 B per bin instead of the 8 B of
        # the complex64 spectrum: half the HBM bytes on both sides) and the matrix-core GCC-PHAT kernel reads those
        pitch = int(lib.seld_phasor_pitch())
        with _device_guard(index):
            phasors = torch.empty((n, c, frames, pitch), dtype=torch.int32, device=pcm.device)
            fn = lib.seld_logmel_phasors_f32 if pcm.dtype == torch.float32 else lib.seld_logmel_phasors_i16
            check(fn(_p(pcm), n, c, length, _p(out), s_n, s_c, s_m, s_t, _p(phasors), stream), "seld_logmel_phasors")
            tail = ctypes.c_void_p(out.data_ptr() + c * N_MELS * 4)          # channel offset c
            check(lib.seld_gcc_phat_q15(_p(phasors), n, c, frames, tail, s_n, s_c, s_m, s_t, stream), "seld_gcc_phat_q15")
        return out[0] if squeeze else out
    if kind == "logmel_iv" and os.environ.get("SELD_FOA", "fused") != "spectra":
        # default: ONE kernel -- the four channels of a clip meet in one workgroup, the spectra stay in LDS
        # (csrc/logmel.hip logmel_iv_kernel; SELD_FOA=spectra: the two-kernel form through complex64 spectra, developer A/B)
        fn = lib.seld_logmel_iv_f32 if pcm.dtype == torch.float32 else lib.seld_logmel_iv_i16
        with _device_guard(index):
            check(fn(_p(pcm), n, length, _p(out), s_n, s_c, s_m, s_t, stream), "seld_logmel_iv")
        return out[0] if squeeze else out
    with _device_guard(index):
        # one pass over the PCM: the log-mel channels and the spectra the spatial kernels read
        spec = torch.empty((n, c, frames, N_FFT // 2 + 1, 2), dtype=torch.float32, device=pcm.device)
        fn = lib.seld_logmel_spectrum_f32 if pcm.dtype == torch.float32 else lib.seld_logmel_spectrum_i16
        check(fn(_p(pcm), n, c, length, _p(out), s_n, s_c, s_m, s_t, _p(spec), stream), "seld_logmel_spectrum")
        tail = ctypes.c_void_p(out.data_ptr() + c * N_MELS * 4)              # channel offset c
        if kind == "logmel_iv":
            check(lib.seld_foa_intensity(_p(spec), n, frames, tail, s_n, s_c, s_m, s_t, stream), "seld_foa_intensity")
        else:
            check(lib.seld_gcc_phat(_p(spec), n, c, frames, tail, s_n, s_c, s_m, s_t, stream), "seld_gcc_phat")
    return out[0] if squeeze else out
