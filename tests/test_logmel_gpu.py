"""GPU parity tests for the fused log-mel kernel (through the C ABI, seld_native -> libseld_hip.so).

Oracle: oracle/features.py (restatement of dataset.py:27-58 + torchaudio defaults).
Tolerance (BASELINE.json north_star): <= 1e-4 abs in dB.  Value parity against torchaudio
itself is UNPINNED (not installable here); the reference only records the frame count.
"""
import numpy as np
import pytest
import torch

from oracle import features as ofeat
from logmel_checks import assert_logmel_close

pytestmark = pytest.mark.gpu

TOL_DB = 1e-4


def _gpu_logmel(pcm, device, layout="cft"):
    import seld_native
    return seld_native.logmel(pcm.to(device), layout=layout).cpu()


@pytest.mark.parametrize("num_samples", [481, 960, 24000, 24123, 96480, 240000])
def test_noise_matches_oracle(gpu_device, num_samples):
    pcm = ofeat.synth_pcm(5, 4, num_samples, "noise")
    got = _gpu_logmel(pcm, gpu_device)
    ref = ofeat.logmel_torch(pcm)
    assert got.shape == ref.shape == (4, 64, 1 + num_samples // 480)
    assert_logmel_close(got.numpy(), ref.numpy())


def test_reference_recorded_frame_count(gpu_device):
    # SMR_SELD_2.ipynb:518-519: load_audio -> [4, 2145600]; audio_to_mel_spectrogram -> [4, 64, 4471]
    import seld_native
    assert seld_native.num_frames(2145600) == 4471
    pcm = torch.zeros(4, 2145600)
    got = _gpu_logmel(pcm, gpu_device)
    assert tuple(got.shape) == (4, 64, 4471)
    assert (got == -100.0).all()          # silence -> clamp(1e-10) -> exactly -100 dB


def test_golden_int16_vector(gpu_device, golden_dir):
    z = np.load(golden_dir / "logmel_noise_1s.npz")
    pcm = torch.from_numpy(z["pcm_i16"])
    got = _gpu_logmel(pcm, gpu_device)                       # int16 entry point
    assert_logmel_close(got.numpy(), z["logmel_from_i16"])
    got_f = _gpu_logmel(ofeat.int16_to_pcm(pcm), gpu_device)  # same samples through the f32 entry point
    # same arithmetic, but the compiler may contract mul+add into fma differently in the two instantiations
    assert (got - got_f).abs().max().item() <= 2e-5


def test_time_major_layout_is_a_permutation(gpu_device):
    pcm = ofeat.synth_pcm(9, 3, 50000, "noise")
    a = _gpu_logmel(pcm, gpu_device, "cft")
    b = _gpu_logmel(pcm, gpu_device, "tcf")
    assert tuple(b.shape) == (1 + 50000 // 480, 3, 64)
    assert torch.equal(a.permute(2, 0, 1), b)


def test_batched_equals_per_clip_and_channels(gpu_device):
    import seld_native
    pcm = torch.stack([ofeat.synth_pcm(i, 8, 30000, "noise") for i in range(3)])   # [3, 8, L], 8-ch MIC shape
    batched = seld_native.logmel(pcm.to(gpu_device)).cpu()
    for i in range(3):
        assert torch.equal(batched[i], _gpu_logmel(pcm[i], gpu_device))
    assert_logmel_close(batched[1].numpy(), ofeat.logmel_torch(pcm[1]).numpy())


def test_tones_and_silence_against_float64(gpu_device):
    """Multi-tone + digital silence + click.  Far from the tones the mel power sits at fp32
    rounding-noise level, where ANY two fp32 FFTs (torch's included) disagree by far more than
    1e-4 dB; there the check is an absolute power error bound relative to the frame's energy,
    and the GPU must be no worse than the fp32 CPU path is against float64."""
    pcm = ofeat.synth_pcm(0, 4, 48000, "tones")
    got = _gpu_logmel(pcm, gpu_device).double().numpy()
    ref64, mel64 = ofeat.logmel_f64(pcm.numpy(), return_mel=True)
    ref32 = ofeat.logmel_torch(pcm).double().numpy()
    got_pow = 10.0 ** (got / 10.0)
    frame_peak = mel64.max(axis=1, keepdims=True)                 # per (channel, frame)
    strong = mel64 >= 1e-4 * np.maximum(frame_peak, 1e-30)
    assert np.abs(got - ref64)[strong].max() <= TOL_DB
    abs_err = np.abs(got_pow - np.maximum(mel64, 1e-10))
    assert (abs_err <= 2e-6 * frame_peak + 1e-12).all()
    err_cpu32 = np.abs(10.0 ** (ref32 / 10.0) - np.maximum(mel64, 1e-10))
    assert abs_err.max() <= 4.0 * err_cpu32.max() + 1e-12
    silent = np.all(pcm.numpy() == 0.0, axis=0)                   # frames fully inside the silent quarter
    t_sil = [t for t in range(got.shape[2]) if silent[max(0, 480 * (t - 1)):480 * (t + 1)].all()]
    assert len(t_sil) > 5 and (got[:, :, t_sil] == -100.0).all()


def test_full_size_properties_60s_clip(gpu_device):
    """BASELINE workload size (4 ch, 60 s @ 24 kHz): size-independent properties.
    (1) gain: x -> 2x adds 10*log10(4) dB exactly where not floored (power-of-two scaling is exact
        in fp32, so the difference is the rounding of log10 only);
    (2) shift: delaying by 16 hops (one frame group, so every frame keeps its lane slot) moves
        interior frames by 16, bit-exactly;
    (3) a spot-check of 200 random frames against the oracle on those frames' samples."""
    import seld_native
    L = 1440000
    pcm = ofeat.synth_pcm(1, 4, L, "noise").to(gpu_device)
    a = seld_native.logmel(pcm)
    assert tuple(a.shape) == (4, 64, 3001)
    b = seld_native.logmel(pcm * 2.0)
    assert (b - a - 10.0 * np.log10(4.0)).abs().max().item() <= 2e-5
    shifted = torch.cat([torch.zeros(4, 16 * 480, device=gpu_device), pcm[:, :-16 * 480]], dim=1)
    c = seld_native.logmel(shifted)
    # groups 0 and 187 run the reflect-capable load path (a different instantiation, whose fma
    # contraction may differ): compare frames that use the interior path in both runs
    assert torch.equal(c[:, :, 32:2976], a[:, :, 16:2960])
    assert (c[:, :, 18:3000] - a[:, :, 2:2984]).abs().max().item() <= 2e-5
    rng = np.random.default_rng(0)
    frames = np.sort(rng.choice(np.arange(2, 2998), size=200, replace=False))
    host = pcm.cpu()
    a_host = a.cpu()
    for t in frames[:50]:
        seg = host[:, 480 * (t - 2): 480 * (t + 3)]                # frames t-1..t+1 fully inside
        ref = ofeat.logmel_torch(seg)[:, :, 2:3]
        assert_logmel_close(a_host[:, :, t:t + 1].numpy(), ref.numpy())
    # and the whole 60 s clip against the oracle (the bar of logmel_checks: 1e-4 dB on every band within 40 dB
    # of its frame's peak, fp32-noise-floor bound below that)
    worst, weak_share = assert_logmel_close(a_host.numpy(), ofeat.logmel_torch(host).numpy())
    assert weak_share < 1e-3


def test_rejects_bad_arguments(gpu_device):
    import seld_native
    with pytest.raises(seld_native.SeldNativeError):
        seld_native.logmel(torch.zeros(1, 4, 480, device=gpu_device))     # reflect pad needs L > 480
    with pytest.raises(seld_native.SeldNativeError):
        seld_native.logmel(torch.zeros(4, 1000))                            # CPU tensor: no fallback
    with pytest.raises(TypeError):
        seld_native.logmel(torch.zeros(4, 1000, dtype=torch.float64, device=gpu_device))
