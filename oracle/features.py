"""Oracle (TEST INFRASTRUCTURE): CPU restatement of the reference feature extractor.

Reference call sites: ``dataset.py:27-58`` (``audio_to_mel_spectrogram``):
``torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, hop_length, n_mels)`` applied
per channel (``dataset.py:47-50``), concatenated (``dataset.py:53``), then
``torchaudio.transforms.AmplitudeToDB()`` (``dataset.py:56``).

torchaudio (``requirements.txt:6``, ``torchaudio>=2.0.0``) is NOT vendored in the
reference and is not installable here, so its arithmetic is restated from its documented
defaults:

  Spectrogram : win_length = n_fft, window = hann_window(n_fft) (periodic), center=True,
                pad_mode='reflect', power=2.0, normalized=False, onesided -> n_fft//2+1 bins
  MelScale    : f_min=0, f_max=sr/2, mel_scale='htk', norm=None,
                mel = fb^T . |X|^2   with triangular fb built from
                all_freqs = linspace(0, sr//2, n_freqs), m_pts = linspace(m(f_min), m(f_max), n_mels+2)
  AmplitudeToDB: stype='power' -> 10*log10(clamp(x, 1e-10)) - 10*log10(max(1e-10, 1.0)); top_db=None

VALUE PARITY WITH TORCHAUDIO ITSELF IS UNPINNED (only the frame count is recorded by the
reference: [4, 64, 4471] for L=2145600, SMR_SELD_2.ipynb:518-519).
An independent third-party restatement (transformers.audio_utils) agrees to 1.3e-6 dB with one filterbank and 4.3e-5 dB
with its own float64-built one (tests/test_oracle_cpu.py): corroboration, not a pin.

Two restatements are provided:
  * ``logmel_torch``  -- fp32, ``torch.stft`` based: what the reference's CPU path computes.
  * ``logmel_f64``    -- float64 numpy: the arbiter when two fp32 implementations disagree
                         on bins that sit at fp32 rounding-noise level.
"""
from __future__ import annotations

import math

import numpy as np
import torch

SR = 24000          # config.py:88
N_FFT = 960         # config.py:85  int(0.04*24000)
HOP = 480           # config.py:86  int(0.02*24000)
N_MELS = 64         # config.py:87
AMIN = 1e-10        # torchaudio AmplitudeToDB default
GCC_SILENCE_POWER = 1e-12   # gcc_phat_f64: |X|^2 at or below this is a silent bin (north-star addition, no reference)


def n_frames(num_samples: int, hop: int = HOP) -> int:
    """center=True STFT frame count: 1 + L // hop (pinned: L=2145600 -> 4471)."""
    return 1 + num_samples // hop


def hz_to_mel_htk(f: float) -> float:
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_filterbank_htk(n_freqs: int = N_FFT // 2 + 1, f_min: float = 0.0,
                       f_max: float = SR / 2, n_mels: int = N_MELS,
                       sample_rate: int = SR) -> torch.Tensor:
    """fp32 triangular filterbank [n_freqs, n_mels] (torchaudio ``melscale_fbanks``,
    mel_scale='htk', norm=None), the matrix ``MelScale`` multiplies |X|^2 with."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = hz_to_mel_htk(f_min)
    m_max = hz_to_mel_htk(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]                       # (n_mels+1)
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)  # (n_freqs, n_mels+2)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


def stft_torch(pcm: torch.Tensor, n_fft: int = N_FFT, hop: int = HOP) -> torch.Tensor:
    """[..., L] fp32 -> complex64 [..., n_fft//2+1, 1+L//hop] (torchaudio Spectrogram's stft)."""
    window = torch.hann_window(n_fft, periodic=True, dtype=pcm.dtype)
    return torch.stft(pcm, n_fft=n_fft, hop_length=hop, win_length=n_fft, window=window,
                      center=True, pad_mode="reflect", normalized=False, onesided=True,
                      return_complex=True)


def power_spectrogram_torch(pcm: torch.Tensor, n_fft: int = N_FFT, hop: int = HOP) -> torch.Tensor:
    return stft_torch(pcm, n_fft, hop).abs().pow(2.0)


def logmel_torch(pcm: torch.Tensor, sample_rate: int = SR, n_fft: int = N_FFT, hop: int = HOP,
                 n_mels: int = N_MELS) -> torch.Tensor:
    """dataset.py:27-58 restated.  pcm [C, L] fp32 -> [C, n_mels, 1+L//hop] fp32 (dB)."""
    pcm = pcm.to(torch.float32)
    fb = mel_filterbank_htk(n_fft // 2 + 1, 0.0, sample_rate / 2, n_mels, sample_rate)
    chans = []
    for c in range(pcm.shape[0]):                         # dataset.py:47-50 per-channel loop
        spec = power_spectrogram_torch(pcm[c:c + 1], n_fft, hop)          # (1, F, T)
        mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)  # (1, n_mels, T)
        chans.append(mel)
    mel = torch.cat(chans, dim=0)                         # dataset.py:53
    return 10.0 * torch.log10(torch.clamp(mel, min=AMIN))  # dataset.py:56, db_multiplier = 0


# --------------------------------------------------------------------------- float64 arbiter

def _reflect_pad_np(x: np.ndarray, pad: int) -> np.ndarray:
    return np.pad(x, [(0, 0)] * (x.ndim - 1) + [(pad, pad)], mode="reflect")


def stft_f64(pcm: np.ndarray, n_fft: int = N_FFT, hop: int = HOP) -> np.ndarray:
    """[..., L] -> complex128 [..., n_fft//2+1, 1+L//hop], same conventions as stft_torch."""
    x = _reflect_pad_np(np.asarray(pcm, dtype=np.float64), n_fft // 2)
    L = pcm.shape[-1]
    T = 1 + L // hop
    n = np.arange(n_fft)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    idx = hop * np.arange(T)[:, None] + n[None, :]        # (T, n_fft)
    frames = x[..., idx] * window                         # (..., T, n_fft)
    spec = np.fft.rfft(frames, axis=-1)                   # (..., T, F)
    return np.swapaxes(spec, -1, -2)


def logmel_f64(pcm: np.ndarray, sample_rate: int = SR, n_fft: int = N_FFT, hop: int = HOP,
               n_mels: int = N_MELS, return_mel: bool = False):
    """float64 arbiter.  The filterbank is the fp32 one (it is a constant of the method)."""
    fb = mel_filterbank_htk(n_fft // 2 + 1, 0.0, sample_rate / 2, n_mels, sample_rate)
    fb = fb.numpy().astype(np.float64)
    spec = stft_f64(pcm, n_fft, hop)
    power = spec.real ** 2 + spec.imag ** 2               # (C, F, T)
    mel = np.einsum("cft,fm->cmt", power, fb)
    db = 10.0 * np.log10(np.maximum(mel, AMIN))
    return (db, mel) if return_mel else db


# --------------------------------------------------------------------------- synthetic inputs

def synth_pcm(clip_idx: int, channels: int = 4, num_samples: int = 240000,
              kind: str = "noise") -> torch.Tensor:
    """Deterministic synthetic PCM.

    kind='noise' : N(0, 0.1^2) clipped to [-1, 1), seed 1234 + clip_idx
    kind='tones' : multi-tone + 25% silence + a click, exercises the -100 dB floor
    """
    g = torch.Generator().manual_seed(1234 + clip_idx)
    if kind == "noise":
        x = torch.randn(channels, num_samples, generator=g) * 0.1
        return x.clamp_(-1.0, 1.0 - 2.0 ** -15).to(torch.float32)
    if kind == "tones":
        t = torch.arange(num_samples, dtype=torch.float64) / SR
        x = torch.zeros(channels, num_samples, dtype=torch.float64)
        for c in range(channels):
            for f, a in ((220.0 * (c + 1), 0.3), (1870.0 + 13.0 * c, 0.05), (9000.0 - 101.0 * c, 0.01)):
                x[c] += a * torch.sin(2 * math.pi * f * t + 0.1 * c)
        q = num_samples // 4
        x[:, q:2 * q] = 0.0                                # exact digital silence -> -100 dB
        x[:, 3 * q] = 0.9                                  # click
        return x.to(torch.float32)
    raise ValueError(kind)


def pcm_to_int16(pcm: torch.Tensor) -> torch.Tensor:
    """float [-1,1) -> int16 the way a 16-bit WAV stores it (load_audio, dataset.py:18-25,
    returns int16/32768 as float32)."""
    return torch.clamp(torch.round(pcm * 32768.0), -32768, 32767).to(torch.int16)


def int16_to_pcm(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.float32) / 32768.0


# --------------------------------------------------------------------------- spatial features (A15 / A16)
# The reference has NO intensity-vector / GCC-PHAT code: these are this project's own CPU
# statements of the DCASE SELD-baseline definitions.  Parity vs the reference: none possible.

def foa_intensity_f64(pcm: np.ndarray, eps: float = 1e-8) -> np.ndarray:
    """[4, L] (W first) -> [3, 64, F]: mel-projected, energy-normalised active intensity vectors."""
    spec = stft_f64(pcm)                                            # [4, 481, F]
    w, xyz = spec[0], spec[1:]
    inten = np.real(np.conj(w)[None] * xyz)
    energy = eps + np.abs(w) ** 2 + (np.abs(xyz) ** 2).sum(0) / 3.0
    fb = mel_filterbank_htk().numpy().astype(np.float64)
    return np.einsum("cft,fm->cmt", inten / energy[None], fb)


def gcc_phat_f64(pcm: np.ndarray, n_lags: int = N_MELS) -> np.ndarray:
    """[C, L] -> [C(C-1)/2, 64, F]: cc = irfft(exp(1j*angle(conj(X_m) X_n))), lags -32..31.
    A bin of a SILENT channel has the phase factor 1.  Silent means |X|^2 <= GCC_SILENCE_POWER (1e-12: below the
    noise floor of any recording, above what a transform that packs two frames leaves in an all-zero frame -- about
    1e-7 of its neighbour's amplitude).  Without the rule numpy's angle() of a zero depends on the SIGNS of its zero
    components (angle(-0.0 + 0j) = pi), an artefact no definition of the feature intends, and the phase of rounding
    residue would be compared.  The kernel (csrc/spatial.hip) implements the same rule."""
    spec = stft_f64(pcm)                                            # [C, 481, F]
    power = spec.real ** 2 + spec.imag ** 2
    c = spec.shape[0]
    out = []
    for m in range(c):
        for n in range(m + 1, c):
            r = np.conj(spec[m]) * spec[n]
            phase = np.exp(1j * np.angle(r))
            phase[(power[m] <= GCC_SILENCE_POWER) | (power[n] <= GCC_SILENCE_POWER)] = 1.0
            cc = np.fft.irfft(phase, n=N_FFT, axis=0)                       # [960, F]
            out.append(np.concatenate((cc[-n_lags // 2:], cc[:n_lags // 2]), axis=0))
    return np.stack(out)
