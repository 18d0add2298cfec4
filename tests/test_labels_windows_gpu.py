"""GPU parity (bit-exact) for the label rasteriser, mask expansion and window gather.
Oracle: oracle/labels.py + oracle/windows.py (restating dataset.py:60-119, :212-317, utils.py:77-90)."""
import numpy as np
import pytest
import torch

from oracle import labels as olab
from oracle import windows as owin

pytestmark = pytest.mark.gpu


def test_polar_grid_table_bit_exact(gpu_device, golden_dir):
    """Every integer direction through the GPU rasteriser vs the table produced by the
    reference's own utils.polar_to_grid."""
    import seld_native
    z = np.load(golden_dir / "polar_grid.npz")
    az, el = np.meshgrid(z["az"], z["el"], indexing="ij")
    n = az.size
    # one event per metadata frame, class 0: frame 5*r holds exactly one set cell
    rows = np.stack([np.arange(n), np.zeros(n, int), np.zeros(n, int), az.ravel(), el.ravel()], 1)
    mask = seld_native.rasterise_labels(torch.from_numpy(rows), 5 * n, device=gpu_device).cpu().numpy()
    cells = mask[::5].argmax(1)
    assert (mask[::5] > 0).sum(1).tolist() == [1] * n
    expect = (z["i"].astype(int) * 36 + z["j"].astype(int)).ravel()
    assert np.array_equal(cells, expect)
    assert np.array_equal(mask[0::5], mask[4::5])


@pytest.mark.parametrize("num_samples,meta_frames", [(24000 * 3 + 100, 32), (96480, 45), (240000, 100), (1440000, 600)])
def test_rasteriser_matches_oracle(gpu_device, num_samples, meta_frames):
    import seld_native
    rows = olab.synth_metadata(7, meta_frames=meta_frames)
    T = olab.total_label_frames(num_samples)
    got = seld_native.rasterise_labels(torch.from_numpy(rows), T, device=gpu_device)
    ref = olab.metadata_to_mask(rows, num_samples)
    assert got.dtype == torch.uint16 and tuple(got.shape) == (T, 648)
    assert np.array_equal(got.cpu().numpy(), ref)
    dense = seld_native.expand_labels(got).cpu().numpy()
    assert np.array_equal(dense, olab.mask_to_dense(ref))
    if T * 648 <= 200 * 648:          # the line-for-line loops are slow: small cases only
        assert np.array_equal(dense, olab.metadata_to_labels_loops(rows, num_samples))


def test_rasteriser_edge_cases(gpu_device):
    import seld_native
    empty = seld_native.rasterise_labels(torch.zeros((0, 5), dtype=torch.int64), 50, device=gpu_device)
    assert (empty.cpu().numpy() == 0).all()
    assert (seld_native.expand_labels(empty).cpu().numpy()[..., 13] == 1).all()
    # rows entirely past the end are dropped; a row straddling the end is truncated (dataset.py:103)
    rows = np.array([[9, 2, 0, 0, 0], [10, 3, 0, 0, 0], [400, 4, 0, 0, 0]])
    T = 48
    got = seld_native.rasterise_labels(torch.from_numpy(rows), T, device=gpu_device).cpu().numpy()
    assert olab.total_label_frames(T * 480) == T                  # the oracle takes SAMPLES: 48 frames of 480
    assert np.array_equal(got, olab.metadata_to_mask(rows, T * 480))
    cell = 9 * 36 + 18
    assert (got[45:48, cell] == (1 << 2)).all() and got[:45].sum() == 0
    with pytest.raises(IndexError):
        seld_native.rasterise_labels(torch.tensor([[0, 14, 0, 0, 0]]), 10, device=gpu_device)


def test_window_gather_matches_oracle(gpu_device):
    import seld_native
    rng = np.random.default_rng(3)
    total = 1310
    spec_tm = rng.standard_normal((total, 4, 64)).astype(np.float32)          # time-major [T, C, F]
    mask = ((rng.random((total, 648)) < 0.01) * rng.integers(1, 1 << 13, (total, 648))).astype(np.uint16)
    starts = owin.window_starts(total)
    assert len(starts) == 27
    d_spec = torch.from_numpy(spec_tm).to(gpu_device)
    d_mask = torch.from_numpy(mask).to(gpu_device)
    w_spec = seld_native.gather_windows(d_spec, torch.from_numpy(starts), 250).cpu().numpy()
    w_mask = seld_native.gather_windows(d_mask, torch.from_numpy(starts), 250)
    dense = seld_native.expand_labels(w_mask).cpu().numpy()
    spec_cft = np.ascontiguousarray(spec_tm.transpose(1, 2, 0))               # reference layout [C, F, T]
    labels_dense = olab.mask_to_dense(mask)
    for b, s in enumerate(starts):
        ref_spec, ref_lab = owin.make_window(spec_cft, labels_dense, int(s))
        assert np.array_equal(w_spec[b], ref_spec)
        assert np.array_equal(dense[b], ref_lab)
    # shuffled / repeated starts (what a shuffling sampler produces)
    perm = torch.tensor([26, 0, 13, 13, 25])
    again = seld_native.gather_windows(d_spec, torch.from_numpy(starts)[perm], 250).cpu().numpy()
    assert np.array_equal(again, w_spec[perm.numpy()])


def test_reference_window_counts_on_device(gpu_device):
    for total, expected in [(4470, 90), (3035, 61)]:
        assert len(owin.window_starts(total)) == expected


@pytest.mark.parametrize("num_samples,meta_frames,seed", [(24000 * 2 + 100, 20, 1), (240000, 100, 2), (1440000, 600, 3)])
def test_gaussian_rasteriser_matches_oracle(gpu_device, num_samples, meta_frames, seed):
    """smrl_seld_gaussian.py:397-534: bit-exact cell sets for the +-2 sigma boxes (float64 comparisons)."""
    import seld_native
    rows = olab.synth_metadata(seed, meta_frames=meta_frames)
    centres = olab.gaussian_source_noise(rows, rng=np.random.RandomState(seed))
    mine = seld_native.gaussian_source_noise(rows, rng=np.random.RandomState(seed))
    assert np.array_equal(centres, mine)                    # same draw order as the reference's groupby([1, 2])
    T = olab.total_label_frames(num_samples)
    got = seld_native.rasterise_labels_gaussian(torch.from_numpy(rows), centres, T, device=gpu_device)
    assert got.dtype == torch.uint16 and tuple(got.shape) == (T, 648)
    assert np.array_equal(got.cpu().numpy(), olab.gaussian_mask(rows, centres, num_samples))
    if T <= 200:
        dense = seld_native.expand_labels(got).cpu().numpy()
        assert np.array_equal(dense, olab.gaussian_labels_loops(rows, centres, num_samples))


def test_gaussian_rasteriser_edge_cases(gpu_device):
    import seld_native
    empty = seld_native.rasterise_labels_gaussian(torch.zeros((0, 5), dtype=torch.int64), np.zeros((0, 2)), 50,
                                                  device=gpu_device)
    assert (empty.cpu().numpy() == 0).all()
    # seam / pole cases with zero noise, other sigmas, centres exactly on a cell-centre boundary
    rows = np.array([[0, 3, 0, 175, 0], [1, 4, 0, -178, 85], [2, 5, 1, 10, -88], [3, 6, 2, 5, 5], [40, 1, 0, 0, 0]])
    centres = rows[:, 3:5].astype(np.float64)
    for sa, se in [(5.0, 5.0), (2.5, 20.0), (0.0, 0.0), (90.0, 45.0)]:
        got = seld_native.rasterise_labels_gaussian(torch.from_numpy(rows), centres, 48, sigma_az=sa, sigma_el=se,
                                                    device=gpu_device).cpu().numpy()
        assert np.array_equal(got, olab.gaussian_mask(rows, centres, 48 * 480, sigma_azimuth=sa, sigma_elevation=se))
    with pytest.raises(IndexError):
        seld_native.rasterise_labels_gaussian(torch.tensor([[0, 14, 0, 0, 0]]), np.zeros((1, 2)), 10, device=gpu_device)


def test_dataset_gaussian_augmentation_flag(gpu_device):
    """SELDDataset(use_gaussian_augmentation=True) (smrl_seld_gaussian.py:539-618) paints boxes, the default does not."""
    from dataset import SELDDataset
    rows = olab.synth_metadata(9, meta_frames=100)
    pcm = torch.zeros((4, 240000), dtype=torch.int16)
    plain = SELDDataset.from_pcm([pcm], [rows], device=gpu_device)
    aug = SELDDataset.from_pcm([pcm], [rows], device=gpu_device, use_gaussian_augmentation=True)
    a, b = (plain.mask_tm != 0).sum().item(), (aug.mask_tm != 0).sum().item()
    assert b > 2 * a > 0
