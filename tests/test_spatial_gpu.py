"""GPU tests of the north-star feature additions that have NO reference implementation (SURVEY.md F4, section 8
A14-A16): explicit STFT, FOA intensity vectors, GCC-PHAT.  The oracle is this project's own float64 statement
of the DCASE SELD-baseline definitions (oracle/features.py) -- parity with the reference is not claimable."""
import numpy as np
import pytest
import torch

from oracle import features as ofeat
from logmel_checks import assert_logmel_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("num_samples", [481, 4800, 24123])
def test_stft_matches_torch_stft(gpu_device, num_samples):
    import seld_native
    pcm = ofeat.synth_pcm(3, 4, num_samples, "noise")
    got = seld_native.stft(pcm.to(gpu_device)).cpu()                      # [4, F, 481]
    ref = ofeat.stft_torch(pcm).transpose(-1, -2)                         # [4, F, 481]
    assert tuple(got.shape) == tuple(ref.shape) == (4, 1 + num_samples // 480, 481)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-6 * scale
    ref64 = np.swapaxes(ofeat.stft_f64(pcm.numpy()), -1, -2)
    assert np.abs(got.numpy() - ref64).max() <= 2e-6 * scale
    # the log-mel kernel squares exactly this spectrum
    mel = torch.einsum("cfk,km->cmf", got.abs() ** 2, ofeat.mel_filterbank_htk())
    db = 10 * torch.log10(mel.clamp(min=1e-10))
    assert_logmel_close(seld_native.logmel(pcm.to(gpu_device)).cpu().numpy(), db.numpy())


@pytest.mark.parametrize("path", ["fused", "spectra"])
def test_foa_intensity_vectors(gpu_device, path, monkeypatch):
    """Both FOA paths: the one-kernel pass (default: the four channels of a clip in one workgroup, spectra in LDS only) and
    the two-kernel form through complex64 spectra in HBM (SELD_FOA=spectra)."""
    import seld_native
    monkeypatch.setenv("SELD_FOA", path)
    pcm = ofeat.synth_pcm(4, 4, 24000, "noise")
    pcm[1] = 0.7 * pcm[0] + 0.3 * pcm[1]                                  # correlate X with W: non-trivial vectors
    feat = seld_native.spatial_features(pcm.to(gpu_device), "logmel_iv").cpu()      # [F, 7, 64]
    assert tuple(feat.shape) == (51, 7, 64)
    assert_logmel_close(feat[:, :4].permute(1, 2, 0).numpy(), ofeat.logmel_torch(pcm).numpy())
    ref = ofeat.foa_intensity_f64(pcm.numpy())                            # [3, 64, F]
    got = feat[:, 4:].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() <= 1e-4                                # values are O(1) per mel band
    assert np.abs(ref[0]).max() > 0.1                                     # the planted correlation shows up


@pytest.mark.parametrize("kernel", ["mfma", "planar", "spectra", "fft"])
@pytest.mark.parametrize("channels", [8, 7, 6, 4, 2])
def test_gcc_phat(gpu_device, channels, kernel, monkeypatch):
    """The GCC-PHAT paths (csrc/spatial.hip): the matrix-core kernel fed with the log-mel pass's Q15 phasors, products by
    integer dot products on the packed words (default); the one that unpacks the words to fp32 in LDS (SELD_GCC=planar);
    the same fed with complex64 spectra (SELD_GCC=spectra: what seld_gcc_phat does on an exported STFT); and the FFT-based
    one that serves outputs whose lag stride is not 1 (SELD_GCC=fft).  7 and 6 channels = 21 and 15 pairs: either side of
    the 16-pair tile boundary."""
    import seld_native
    monkeypatch.setenv("SELD_GCC", kernel)
    pcm = ofeat.synth_pcm(6, channels, 12000 + 17, "noise")
    pcm[1, 7:] = pcm[0, :-7]                                              # channel 1 = channel 0 delayed by 7 samples
    feat = seld_native.spatial_features(pcm.to(gpu_device), "logmel_gcc").cpu()
    pairs = channels * (channels - 1) // 2
    frames = 1 + pcm.shape[1] // 480
    assert tuple(feat.shape) == (frames, channels + pairs, 64)
    assert_logmel_close(feat[:, :channels].permute(1, 2, 0).numpy(), ofeat.logmel_torch(pcm).numpy())
    ref = ofeat.gcc_phat_f64(pcm.numpy())                                 # [pairs, 64, F]
    got = feat[:, channels:].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() <= 1e-4
    # pair (0, 1): the peak sits at lag +7 (index 32 + 7) in the interior frames
    assert (got[0, :, 3:-3].argmax(axis=0) == 39).all()


def test_batched_spatial_features(gpu_device):
    import seld_native
    pcm = torch.stack([ofeat.synth_pcm(i, 4, 9600, "noise") for i in range(3)]).to(gpu_device)
    iv = seld_native.spatial_features(pcm, "logmel_iv")
    gcc = seld_native.spatial_features(pcm, "logmel_gcc")
    assert tuple(iv.shape) == (3, 21, 7, 64) and tuple(gcc.shape) == (3, 21, 10, 64)
    for i in range(3):
        assert torch.equal(iv[i], seld_native.spatial_features(pcm[i], "logmel_iv"))
        assert torch.equal(gcc[i], seld_native.spatial_features(pcm[i], "logmel_gcc"))


def test_fused_foa_pass_runs_that_cross_clip_boundaries(gpu_device):
    """40 clips of 13 iterations = 520 (clip, iteration) pairs on 256 CUs: every workgroup of the one-kernel FOA pass walks
    3 consecutive pairs, most of them across a clip boundary (last iteration of one clip, first -- reflected -- of the next).
    Each clip alone is one pair per workgroup: the two must agree bit for bit."""
    import seld_native
    pcm = torch.stack([ofeat.synth_pcm(40 + i, 4, 24000, "noise") for i in range(40)]).to(gpu_device)
    batched = seld_native.spatial_features(pcm, "logmel_iv")
    assert tuple(batched.shape) == (40, 51, 7, 64) and torch.isfinite(batched).all()
    for i in range(40):
        assert torch.equal(batched[i], seld_native.spatial_features(pcm[i], "logmel_iv")), i


@pytest.mark.parametrize("kernel", ["mfma", "planar", "spectra", "fft"])
def test_gcc_phat_with_a_silent_channel(gpu_device, kernel, monkeypatch):
    """X = 0 must give R / |R| = 1 (np.exp(1j * np.angle(0))): pairs with the dead microphone are a unit pulse at lag 0,
    the other pairs are untouched.  (The kernel stores a zero phasor for such a channel and switches, per frame, to the
    variant that turns zero products into 1 -- csrc/spatial.hip.)"""
    import seld_native
    monkeypatch.setenv("SELD_GCC", kernel)
    pcm = ofeat.synth_pcm(11, 4, 9600 + 5, "noise")
    pcm[2] = 0.0
    pcm[:, 4800:6000] = 0.0                                               # and a stretch of digital silence in all of them
    feat = seld_native.spatial_features(pcm.to(gpu_device), "logmel_gcc").cpu()
    ref = ofeat.gcc_phat_f64(pcm.numpy())                                 # [6, 64, F]
    got = feat[:, 4:].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() <= 1e-4
    dead = [1, 3, 5]                                                      # pairs (0,2), (1,2), (2,3) in lexicographic order
    pulse = np.zeros(64)
    pulse[32] = 1.0
    assert np.abs(got[dead] - pulse[None, :, None]).max() <= 2e-5         # (frames with a silent bin take the exact route)


def test_gcc_phat_full_size_clip_and_kernel_variants(gpu_device, monkeypatch):
    """BASELINE configs[3] sizes: a 60 s 8-channel clip (3001 frames x 28 pairs) through the default matrix-core kernel --
    a planted delay shows as the peak of its pair in every interior frame, 500 frames spread over the clip agree with the
    float64 oracle computed on their own excerpts, and the log-mel channels written by the same pass equal the plain
    log-mel call bit for bit."""
    import seld_native
    pcm = ofeat.synth_pcm(21, 8, 1_440_000, "noise")
    pcm[5, 11:] = pcm[2, :-11]                                            # channel 5 = channel 2 delayed by 11 samples
    dev = pcm.to(gpu_device)
    feat = seld_native.spatial_features(dev, "logmel_gcc")                # [3001, 36, 64]
    assert tuple(feat.shape) == (3001, 36, 64)
    assert torch.equal(feat[:, :8], seld_native.logmel(dev, layout="tcf"))
    gcc = feat[:, 8:].cpu().numpy()                                       # [F, 28, 64]
    pair_25 = [p for p, (m, n) in enumerate((m, n) for m in range(8) for n in range(m + 1, 8)) if (m, n) == (2, 5)][0]
    assert (gcc[2:-2, pair_25].argmax(axis=1) == 32 + 11).all()
    assert np.isfinite(gcc).all() and np.abs(gcc).max() <= 1.0 + 1e-4
    for start in (0, 480 * 1458, 1_440_000 - 48_000):                     # excerpts (hop-aligned) whose interior frames are the clip's
        excerpt = pcm[:, start:start + 48_000]
        ref = ofeat.gcc_phat_f64(excerpt.numpy())                         # [28, 64, 101]
        f0 = start // 480
        got = gcc[f0 + 2:f0 + 99].transpose(1, 2, 0)                      # frames whose 960 samples lie inside the excerpt
        assert np.abs(got - ref[:, :, 2:99]).max() <= 1e-4


@pytest.mark.parametrize("path", ["fused", "spectra"])
def test_foa_intensity_full_size_clip(gpu_device, path, monkeypatch):
    """The FOA feature set at workload size: a 60 s 4-channel clip (3001 frames) through both FOA paths -- excerpts
    spread over the clip agree with the float64 oracle computed on their own samples (interior frames), a planted W-X
    correlation shows in channel 0, and the log-mel channels of the same pass equal the plain log-mel call: bit for bit in
    the two-kernel form (the same kernel template), to the log-mel bar in the one-kernel form (another function body: the
    compiler fuses the transforms' multiply-adds differently -- last-bit differences, printed)."""
    import seld_native
    monkeypatch.setenv("SELD_FOA", path)
    pcm = ofeat.synth_pcm(31, 4, 1_440_000, "noise")
    pcm[1] = 0.6 * pcm[0] + 0.4 * pcm[1]
    dev = pcm.to(gpu_device)
    feat = seld_native.spatial_features(dev, "logmel_iv")                 # [3001, 7, 64]
    assert tuple(feat.shape) == (3001, 7, 64)
    plain = seld_native.logmel(dev, layout="tcf")
    if path == "spectra":
        assert torch.equal(feat[:, :4], plain)
    else:
        print(f"fused FOA pass vs plain log-mel kernel: max |dB difference| = {(feat[:, :4] - plain).abs().max().item():.3e}, "
              f"{(feat[:, :4] != plain).float().mean().item():.3%} of the values differ")
        assert_logmel_close(feat[:, :4].permute(1, 2, 0).cpu().numpy(), plain.permute(1, 2, 0).cpu().numpy())
    iv = feat[:, 4:].cpu().numpy()                                        # [F, 3, 64]
    assert np.isfinite(iv).all() and np.abs(iv[:, 0]).mean() > 0.05
    for start in (0, 480 * 1458, 1_440_000 - 48_000):
        excerpt = pcm[:, start:start + 48_000]
        ref = ofeat.foa_intensity_f64(excerpt.numpy())                    # [3, 64, 101]
        f0 = start // 480
        got = iv[f0 + 2:f0 + 99].transpose(1, 2, 0)                       # frames whose 960 samples lie inside the excerpt
        assert np.abs(got - ref[:, :, 2:99]).max() <= 1e-4
