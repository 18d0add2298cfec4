"""GPU tests of the dataset surface (dataset.py mirror) on synthetic WAV + CSV files, and of the full
main.py call sequence (load files -> SELDDataset x2 -> DataLoader x2 -> train_model -> test_model) on a tiny
configuration.  Oracle: oracle/{features,labels,windows}.py."""
import wave
from pathlib import Path

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader

from oracle import features as ofeat
from oracle import labels as olab
from oracle import windows as owin
from logmel_checks import assert_logmel_close

pytestmark = pytest.mark.gpu


def _write_clip(folder: Path, name: str, clip_idx: int, num_samples: int, meta_frames: int):
    pcm = ofeat.pcm_to_int16(ofeat.synth_pcm(clip_idx, 4, num_samples, "noise")).numpy()
    with wave.open(str(folder / f"{name}.wav"), "wb") as wf:
        wf.setnchannels(4)
        wf.setsampwidth(2)
        wf.setframerate(24000)
        wf.writeframes(np.ascontiguousarray(pcm.T).tobytes())
    rows = olab.synth_metadata(clip_idx, meta_frames=meta_frames)
    (folder / f"{name}.csv").write_text(olab.metadata_to_csv(rows))
    return pcm, rows


@pytest.fixture(scope="module")
def clips(tmp_path_factory):
    folder = tmp_path_factory.mktemp("seld_clips")
    specs = [("fold3_room21_mix001", 0, 24000 * 7 + 200, 70), ("fold3_room21_mix002", 1, 96480, 45),
             ("fold4_room23_mix001", 2, 24000 * 6, 60)]
    out = {}
    for name, idx, n, mf in specs:
        pcm, rows = _write_clip(folder, name, idx, n, mf)
        out[name] = dict(wav=str(folder / f"{name}.wav"), csv=str(folder / f"{name}.csv"), pcm=pcm, rows=rows)
    return out


def _oracle_timeline(items):
    specs, labels = [], []
    for it in items:
        x = ofeat.int16_to_pcm(torch.from_numpy(it["pcm"]))
        spec = ofeat.logmel_torch(x).numpy()
        lab = olab.mask_to_dense(olab.metadata_to_mask(it["rows"], it["pcm"].shape[1]))
        s, l = owin.crop_pair(spec, lab)
        specs.append(s)
        labels.append(l)
    return owin.concatenate(specs, labels)


def test_function_surface(gpu_device, clips):
    import dataset
    c = clips["fold3_room21_mix002"]
    waveform, sr = dataset.load_audio(c["wav"])
    assert sr == 24000 and waveform.dtype == torch.float32 and tuple(waveform.shape) == (4, 96480)
    assert torch.equal(waveform, ofeat.int16_to_pcm(torch.from_numpy(c["pcm"])))
    mel = dataset.audio_to_mel_spectrogram(waveform, sr)
    assert not mel.is_cuda and tuple(mel.shape) == (4, 64, 202)
    assert_logmel_close(mel.numpy(), ofeat.logmel_torch(waveform).numpy())
    labels, I, J = dataset.metadata_to_labels(c["csv"], waveform.shape[1] / sr, sample_rate=sr)
    assert (I, J) == (18, 36) and tuple(labels.shape) == (200, 648, 14)          # float rounding trap: 200, not 201
    assert np.array_equal(labels.numpy(), olab.metadata_to_labels_loops(c["rows"], 96480))
    with pytest.raises(NotImplementedError):
        dataset.audio_to_mel_spectrogram(waveform, sr, n_fft=1024)


def test_dataset_matches_reference_assembly(gpu_device, clips):
    import dataset
    items = [clips["fold3_room21_mix001"], clips["fold3_room21_mix002"]]
    ds = dataset.SELDDataset([i["wav"] for i in items], [i["csv"] for i in items], num_classes=14)
    spec_cat, lab_cat = _oracle_timeline(items)
    total = spec_cat.shape[2]
    assert (ds.I, ds.J, ds.total_cells) == (18, 36, 648)
    assert ds.total_frames == total == 350 + 200
    starts = owin.window_starts(total)
    assert len(ds) == len(starts) == 11 and len(ds.windows) == 11
    assert tuple(ds.concatenated_spectrograms.shape) == (4, 64, total)
    for idx in (0, 3, 6, 10):                       # window 3 straddles the file boundary, 7..10 are padded
        spec, labels = ds[idx]
        ref_spec, ref_lab = owin.make_window(spec_cat, lab_cat, int(starts[idx]))
        assert tuple(spec.shape) == (250, 4, 64) and tuple(labels.shape) == (250, 648, 14)
        n_real = min(250, total - int(starts[idx]))                     # the zero-padded tail is compared exactly below
        assert_logmel_close(spec.numpy()[:n_real], ref_spec[:n_real], mel_axis=-1)
        assert np.array_equal(spec.numpy() == 0.0, ref_spec == 0.0)              # zero pad, not -100 dB
        assert np.array_equal(labels.numpy(), ref_lab)
    w = ds.windows[10]
    assert w["start_frame"] == 500 and w["end_frame"] == total and w["window_idx"] == 10
    spec_b, mask_b = ds.device_batch([10, 0, 3])
    assert tuple(spec_b.shape) == (3, 250, 4, 64) and mask_b.dtype == torch.uint16
    for k, idx in enumerate((10, 0, 3)):
        spec, labels = ds[idx]
        assert torch.equal(spec_b[k].cpu(), spec)
        assert np.array_equal(olab.mask_to_dense(mask_b[k].cpu().numpy()), labels.numpy())
    batch = next(iter(DataLoader(ds, batch_size=4, shuffle=True, num_workers=2, pin_memory=True)))
    assert tuple(batch[0].shape) == (4, 250, 4, 64) and tuple(batch[1].shape) == (4, 250, 648, 14)


def test_main_py_call_sequence(gpu_device, clips, tmp_path):
    """What the reference's main.py does (main.py:41-104), on a tiny CRNN, both batch sources."""
    import dataset
    import trainer
    cfg = trainer.config
    names = ("MODEL_TYPE", "CRNN_CNN_CHANNELS", "CRNN_RNN_HIDDEN", "NUM_EPOCHS", "BATCH_SIZE", "SEED", "OUTPUT_PATH",
             "CHECKPOINT_PATH", "DEVICE_FEED")
    saved = {k: getattr(cfg, k) for k in names}
    try:
        _main_py_call_sequence(cfg, dataset, trainer, clips, tmp_path)
    finally:                                   # the config object is shared by every test of the session
        for k, v in saved.items():
            setattr(cfg, k, v)


def _main_py_call_sequence(cfg, dataset, trainer, clips, tmp_path):
    cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS, cfg.CRNN_RNN_HIDDEN = "crnn", [8, 8, 16, 16], 16
    cfg.NUM_EPOCHS, cfg.BATCH_SIZE, cfg.SEED = 2, 4, 0
    cfg.OUTPUT_PATH, cfg.CHECKPOINT_PATH = tmp_path / "outputs", tmp_path / "checkpoints"
    cfg.OUTPUT_PATH.mkdir()
    cfg.CHECKPOINT_PATH.mkdir()
    train_c = [clips["fold3_room21_mix001"], clips["fold3_room21_mix002"]]
    test_c = [clips["fold4_room23_mix001"]]
    train_ds = dataset.SELDDataset([c["wav"] for c in train_c], [c["csv"] for c in train_c], num_classes=cfg.NUM_CLASSES)
    test_ds = dataset.SELDDataset([c["wav"] for c in test_c], [c["csv"] for c in test_c], num_classes=cfg.NUM_CLASSES)
    train_loader = DataLoader(train_ds, batch_size=cfg.BATCH_SIZE, shuffle=True, num_workers=2, pin_memory=True)
    test_loader = DataLoader(test_ds, batch_size=cfg.BATCH_SIZE, shuffle=False, num_workers=2, pin_memory=True)
    losses = {}
    for feed in (True, False):
        cfg.DEVICE_FEED = feed
        model, history = trainer.train_model(train_loader=train_loader, test_loader=test_loader,
                                             num_epochs=cfg.NUM_EPOCHS, batch_size=cfg.BATCH_SIZE,
                                             learning_rate=cfg.LEARNING_RATE, device=torch.device("cuda"))
        assert history["total_epochs"] == 2 and np.isfinite(history["best_test_loss"])
        losses[feed] = history["test_losses"]
        results = trainer.test_model(test_loader=test_loader, model_path=cfg.CHECKPOINT_PATH / "best_model.pth",
                                     batch_size=cfg.BATCH_SIZE, device=torch.device("cuda"), num_visualizations=2,
                                     save_visualizations=True)
        for key in ("test_loss", "class_mse", "overall_accuracy", "non_bg_accuracy", "num_frames_with_events",
                    "visualizations", "checkpoint_epoch"):
            assert key in results
        assert results["num_frames_with_events"] > 0 and len(results["visualizations"]) == 2
        _check_test_model_against_the_reference_metrics(cfg, trainer, test_ds, test_loader, results)
    cfg.DEVICE_FEED = True
    # both batch sources train the same model on the same windows (different shuffles / bf16 noise): same ballpark
    assert abs(losses[True][-1] - losses[False][-1]) <= 0.2 * max(losses[True][-1], losses[False][-1])


def _check_test_model_against_the_reference_metrics(cfg, trainer, test_ds, test_loader, results):
    """test_model's on-device reductions (argmax accuracies, event counts from the compact mask, frames with events)
    against the reference's own formulas on dense host arrays (oracle/evalmetrics.py: trainer.py:542-556, 621-635):
    same checkpoint, every test window through the stock DataLoader path, all predictions and labels collected."""
    from oracle import evalmetrics
    dev = torch.device("cuda")
    ckpt = torch.load(cfg.CHECKPOINT_PATH / "best_model.pth", weights_only=False)          # written by this test run
    model = trainer.prepare_model_for_device(trainer.build_model((test_ds.I, test_ds.J)), dev)
    model.load_state_dict(ckpt["model_state_dict"])
    model.eval()
    preds, labels, losses = [], [], []
    with torch.no_grad():
        for spec, lab in test_loader:                        # CPU tensors, dense labels (dataset.py:319-330)
            with trainer.autocast_context(dev):
                out = model(spec.to(dev)).float()
            losses.append(torch.nn.functional.mse_loss(torch.softmax(out, -1), lab.to(dev)).item())
            preds.append(out.cpu().numpy())
            labels.append(lab.numpy())
    preds, labels = np.concatenate(preds), np.concatenate(labels)
    overall, non_bg, active, cells = evalmetrics.accuracies(preds, labels, cfg.NUM_CLASSES)
    assert abs(results["overall_accuracy"] - overall) <= 1e-3 and abs(results["non_bg_accuracy"] - non_bg) <= 1e-3
    frames = evalmetrics.frames_with_events(labels, cfg.NUM_CLASSES)
    assert results["num_frames_with_events"] == len(frames)
    assert abs(results["test_loss"] - float(np.mean(losses))) <= 1e-4 * float(np.mean(losses))   # trainer.py:524-525
    for viz in results["visualizations"]:
        assert (viz["window_idx"], viz["time_idx"], viz["num_active"]) in set(frames)


def test_mic_array_gcc_feature_set_end_to_end(gpu_device, tmp_path):
    """BASELINE configs[3] in miniature: 8-channel MIC recordings, FEATURE_SET = 'logmel_gcc' (8 log-mel + 28 GCC-PHAT
    channels, csrc/spatial.hip), a CRNN whose input width follows the dataset, one training epoch + evaluation.
    The first 8 feature channels are the reference's per-channel log-mel (dataset.py:27-58): checked against the
    oracle; the GCC-PHAT channels have no upstream counterpart (self-oracle: tests/test_spatial_gpu.py)."""
    import dataset
    import trainer
    cfg = trainer.config
    saved = (cfg.FEATURE_SET, cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS, cfg.CRNN_RNN_HIDDEN, cfg.NUM_EPOCHS, cfg.BATCH_SIZE,
             cfg.OUTPUT_PATH, cfg.CHECKPOINT_PATH)
    try:
        # config.py is edited in place upstream (class attributes); dataset.py and trainer.py each hold an instance
        type(cfg).FEATURE_SET = "logmel_gcc"
        cfg.FEATURE_SET, cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS, cfg.CRNN_RNN_HIDDEN = "logmel_gcc", "crnn", [8, 8, 16, 16], 16
        cfg.NUM_EPOCHS, cfg.BATCH_SIZE, cfg.SEED = 1, 4, 0
        cfg.OUTPUT_PATH, cfg.CHECKPOINT_PATH = tmp_path / "outputs", tmp_path / "checkpoints"
        cfg.OUTPUT_PATH.mkdir()
        cfg.CHECKPOINT_PATH.mkdir()
        files = []
        for idx, n in enumerate((24000 * 6 + 100, 24000 * 5)):
            pcm = ofeat.pcm_to_int16(ofeat.synth_pcm(20 + idx, 8, n, "noise")).numpy()
            wav, csv_path = tmp_path / f"mic{idx}.wav", tmp_path / f"mic{idx}.csv"
            with wave.open(str(wav), "wb") as wf:
                wf.setnchannels(8)
                wf.setsampwidth(2)
                wf.setframerate(24000)
                wf.writeframes(np.ascontiguousarray(pcm.T).tobytes())
            csv_path.write_text(olab.metadata_to_csv(olab.synth_metadata(20 + idx, meta_frames=50)))
            files.append((str(wav), str(csv_path), pcm))
        ds = dataset.SELDDataset([f[0] for f in files], [f[1] for f in files], num_classes=cfg.NUM_CLASSES)
        assert ds.n_channels == 8 + 28
        spec, labels = ds[0]
        assert tuple(spec.shape) == (250, 36, 64) and tuple(labels.shape) == (250, 648, 14)
        ref = ofeat.logmel_torch(ofeat.int16_to_pcm(torch.from_numpy(files[0][2])))            # [8, 64, F]
        assert_logmel_close(spec[:, :8].permute(1, 2, 0), ref[:, :, :250])
        assert torch.isfinite(spec).all()
        loader = DataLoader(ds, batch_size=cfg.BATCH_SIZE, shuffle=True)
        model, history = trainer.train_model(train_loader=loader, test_loader=loader, device=torch.device("cuda"))
        assert trainer.unwrap(model).cnn_blocks[0].conv.in_channels == 36
        assert history["total_epochs"] == 1 and np.isfinite(history["best_test_loss"])
        results = trainer.test_model(test_loader=loader, model_path=cfg.CHECKPOINT_PATH / "best_model.pth",
                                     device=torch.device("cuda"), num_visualizations=1, save_visualizations=False)
        assert np.isfinite(results["test_loss"])
    finally:
        type(cfg).FEATURE_SET = "logmel"
        (cfg.FEATURE_SET, cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS, cfg.CRNN_RNN_HIDDEN, cfg.NUM_EPOCHS, cfg.BATCH_SIZE,
         cfg.OUTPUT_PATH, cfg.CHECKPOINT_PATH) = saved


def test_feature_cache_round_trip(gpu_device, clips, tmp_path, monkeypatch):
    """Config.FEATURE_CACHE_DIR: the second construction uploads the compact per-recording
    arrays instead of running the feature / label kernels, and the dataset is bit-identical; a changed recording
    invalidates its entry; Gaussian label augmentation (fresh noise per construction) is never cached."""
    import dataset
    import seld_native
    cfg = dataset.config
    items = [clips["fold3_room21_mix001"], clips["fold3_room21_mix002"]]
    wavs, csvs = [c["wav"] for c in items], [c["csv"] for c in items]
    plain = dataset.SELDDataset(wavs, csvs)
    monkeypatch.setattr(cfg, "FEATURE_CACHE_DIR", str(tmp_path / "cache"), raising=False)
    first = dataset.SELDDataset(wavs, csvs)
    files = sorted((tmp_path / "cache").glob("*.npz"))
    assert len(files) == 2 and all(f.stat().st_size < 2.6e3 * plain.total_frames for f in files)   # compact: ~2.3 KB / frame
    calls = []
    real = seld_native.spatial_features
    monkeypatch.setattr(seld_native, "spatial_features", lambda *a, **k: calls.append(1) or real(*a, **k))
    second = dataset.SELDDataset(wavs, csvs)
    assert not calls                                                      # nothing was recomputed
    for ds in (first, second):
        assert torch.equal(ds.spec_tm, plain.spec_tm) and torch.equal(ds.mask_tm, plain.mask_tm)
        assert np.array_equal(ds.window_starts, plain.window_starts)
    # a recording that changed on disk gets a new entry
    import os
    st = os.stat(wavs[0])
    os.utime(wavs[0], ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000_000))
    dataset.SELDDataset(wavs[:1], csvs[:1])
    assert calls and len(list((tmp_path / "cache").glob("*.npz"))) == 3
    del calls[:]
    dataset.SELDDataset(wavs[:1], csvs[:1], use_gaussian_augmentation=True)
    assert calls and len(list((tmp_path / "cache").glob("*.npz"))) == 3
