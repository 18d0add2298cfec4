"""CPU tests of the epoch loop (trainer.py mirror) incl. the data-parallel path with 2 gloo ranks.
A tiny stand-in dataset with the SELDDataset surface (.I .J .total_cells, items (spec, labels)) feeds the
stock-DataLoader branch, so no GPU / HIP code is involved."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.multiprocessing as mp
from torch.utils.data import DataLoader, Dataset

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "sound-event-localization-detection_amd"


class TinySeld(Dataset):
    I, J, total_cells = 3, 4, 12

    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.spec = torch.randn(n, 10, 4, 64, generator=g) * 10 - 20
        cls = torch.randint(0, 14, (n, 10, 12), generator=g)
        self.labels = torch.nn.functional.one_hot(cls, 14).float()

    def __len__(self):
        return self.spec.shape[0]

    def __getitem__(self, i):
        return self.spec[i], self.labels[i]


def _configure(trainer, tmp):
    cfg = trainer.config
    cfg.MODEL_TYPE = "crnn"
    cfg.CRNN_CNN_CHANNELS = [4, 8, 8, 16]
    cfg.CRNN_RNN_HIDDEN = 8
    cfg.NUM_EPOCHS = 3
    cfg.BATCH_SIZE = 4
    cfg.SAVE_EVERY_N_EPOCHS = 1
    cfg.KEEP_LAST_N_CHECKPOINTS = 2
    cfg.SEED = 0
    cfg.OUTPUT_PATH = Path(tmp) / "outputs"
    cfg.CHECKPOINT_PATH = Path(tmp) / "checkpoints"
    cfg.OUTPUT_PATH.mkdir(parents=True, exist_ok=True)
    cfg.CHECKPOINT_PATH.mkdir(parents=True, exist_ok=True)
    return cfg


def test_shard_indices_matches_distributed_sampler_semantics():
    import trainer
    order = list(range(10))
    parts = [trainer.shard_indices(order, r, 4) for r in range(4)]
    assert all(len(p) == 3 for p in parts)                       # padded to 12 by wrapping
    assert sorted(sum(parts, []))[:10] == sorted(order + [0, 1])[:10]
    assert trainer.shard_indices(order, 0, 1) == order


def test_train_and_test_model_single_process(tmp_path):
    import trainer
    cfg = _configure(trainer, tmp_path)
    train_loader = DataLoader(TinySeld(12, 1), batch_size=4, shuffle=True)
    test_loader = DataLoader(TinySeld(8, 2), batch_size=4, shuffle=False)
    model, history = trainer.train_model(train_loader, test_loader, device=torch.device("cpu"))
    assert set(history) >= {"train_losses", "test_losses", "best_train_loss", "best_test_loss", "best_epoch",
                            "total_epochs", "config"}
    assert history["total_epochs"] == 3 and len(history["train_losses"]) == 3
    assert history["config"]["grid_size"] == (3, 4)
    best = cfg.CHECKPOINT_PATH / "best_model.pth"
    assert best.exists()
    ckpt = torch.load(best, weights_only=False)
    assert set(ckpt) == {"epoch", "model_state_dict", "optimizer_state_dict", "train_loss", "test_loss", "config"}
    assert "rnn.weight_ih_l0_reverse" in ckpt["model_state_dict"] and "fnn.4.bias" in ckpt["model_state_dict"]
    assert len(list(cfg.CHECKPOINT_PATH.glob("checkpoint_epoch_*.pth"))) == 2        # keep-last-N
    assert list(cfg.OUTPUT_PATH.glob("loss_curves_*.png")) and list(cfg.OUTPUT_PATH.glob("training_history_*.pth"))
    results = trainer.test_model(test_loader, model_path=best, device=torch.device("cpu"), num_visualizations=2)
    assert set(results) >= {"test_loss", "class_mse", "overall_accuracy", "non_bg_accuracy", "num_frames_with_events",
                            "visualizations", "checkpoint_epoch"}
    assert len(results["visualizations"]) == 2 and results["visualizations"][0]["save_path"].exists()


def _ddp_worker(rank, world, tmp, port, queue):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    for p in (str(ROOT), str(PKG)):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    import trainer
    _configure(trainer, tmp)
    train_loader = DataLoader(TinySeld(8, 1), batch_size=2, shuffle=False)
    test_loader = DataLoader(TinySeld(4, 2), batch_size=2, shuffle=False)
    model, history = trainer.train_model(train_loader, test_loader, device=torch.device("cpu"))
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    feed = trainer.make_feed(train_loader, torch.device("cpu"), rank, world)
    seen = sorted(i for batch in feed._batches_of_rank(1) for i in batch)
    queue.put((rank, flat.double().sum().item(), flat.abs().double().sum().item(), history["train_losses"],
               history["test_losses"], history["best_epoch"], history["config"], seen))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_two_ranks_gloo(tmp_path):
    """World size 2 on CPU (gloo): gradients are averaged by DDP, the per-epoch loss sums are all-reduced, so
    both ranks end with identical weights and identical history; only rank 0 writes files."""
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, str(tmp_path), port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(queue.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, s0, a0, tr0, te0, be0, cfg0, seen0), (_, s1, a1, tr1, te1, be1, cfg1, seen1) = results
    assert s0 == s1 and a0 == a1
    assert tr0 == tr1 and te0 == te1 and be0 == be1
    # the epoch really is sharded inside train_model: 8 windows, batch 2 -> 4 batches in one process, 2 per rank here,
    # and the two ranks' index sets are disjoint and cover the dataset
    assert cfg0["world_size"] == 2 and cfg0["batches_per_rank"] == 2 and cfg0["batch_source"] == "LoaderFeed"
    assert cfg1["batches_per_rank"] == 2
    assert not set(seen0) & set(seen1) and sorted(seen0 + seen1) == list(range(8))
    assert (Path(tmp_path) / "checkpoints" / "best_model.pth").exists()


def _cache_writer(path, seed, barrier, queue):
    for p in (str(ROOT), str(PKG)):
        if p not in sys.path:
            sys.path.insert(0, p)
    import numpy as np
    import dataset
    rng = np.random.default_rng(0)                       # every rank computes the same features
    spec = rng.standard_normal((400, 4, 64)).astype(np.float32)
    mask = rng.integers(0, 1 << 14, (400, 648)).astype(np.uint16)
    barrier.wait()
    try:
        for _ in range(20):                              # hammer the same final name from every process
            dataset.save_compact_features(path, spec, mask)
            with np.load(path) as z:
                assert np.array_equal(z["spec"], spec) and np.array_equal(z["mask"], mask)
        queue.put((seed, "ok"))
    except Exception as exc:      # noqa: BLE001
        queue.put((seed, f"{type(exc).__name__}: {exc}"))


def test_feature_cache_writers_do_not_collide(tmp_path):
    """Config.FEATURE_CACHE_DIR under data parallelism: every rank builds the dataset and writes the SAME cache entry at
    the same time (round 2 shared one temporary name: the second rename raised FileNotFoundError).  Four processes write
    and re-read one entry 20 times each; nobody fails, no temporary file is left behind."""
    ctx = mp.get_context("spawn")
    queue, barrier = ctx.Queue(), ctx.Barrier(4)
    target = tmp_path / "cache" / "clip.0123456789abcdef.npz"
    procs = [ctx.Process(target=_cache_writer, args=(str(target), i, barrier, queue)) for i in range(4)]
    for p in procs:
        p.start()
    results = sorted(queue.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in results] == ["ok"] * 4, results
    assert sorted(f.name for f in target.parent.iterdir()) == [target.name]


def test_checkpoint_holds_plain_adam_state(tmp_path):
    """A capturable Adam keeps lr / step on the device; the checkpoint must hold what the reference's plain Adam writes
    (trainer.py:278-285): a float learning rate, host step counts."""
    import trainer
    model = torch.nn.Linear(4, 3)
    opt = torch.optim.Adam(model.parameters(), lr=torch.tensor(1e-3), capturable=False)
    model(torch.randn(2, 4)).sum().backward()
    opt.step()
    payload = trainer.checkpoint_payload(3, model, opt, 0.5, 0.6)
    group = payload["optimizer_state_dict"]["param_groups"][0]
    assert isinstance(group["lr"], float) and abs(group["lr"] - 1e-3) < 1e-9
    for entry in payload["optimizer_state_dict"]["state"].values():
        assert not entry["step"].is_cuda
    assert isinstance(opt.param_groups[0]["lr"], torch.Tensor)          # the live optimiser keeps its tensor ...
    live = next(iter(opt.state.values()))
    assert payload["optimizer_state_dict"]["state"][0] is not live      # ... and its own state dicts (not the payload's)
    torch.save(payload, tmp_path / "c.pth")
    again = torch.load(tmp_path / "c.pth", weights_only=False)
    fresh = torch.optim.Adam(torch.nn.Linear(4, 3).parameters(), lr=1.0)
    fresh.load_state_dict(again["optimizer_state_dict"])               # a plain Adam resumes from it
    assert fresh.param_groups[0]["lr"] == group["lr"]


def test_loader_feed_refuses_samplers_it_cannot_shard():
    import trainer
    from torch.utils.data import WeightedRandomSampler
    ds = TinySeld(8, 1)
    loader = DataLoader(ds, batch_size=2, sampler=WeightedRandomSampler([1.0] * 8, 8))
    with pytest.raises(ValueError, match="Random / Sequential"):
        trainer.LoaderFeed(loader, torch.device("cpu"), rank=0, world=2)
    trainer.LoaderFeed(loader, torch.device("cpu"), rank=0, world=1)    # single process: the loader is used as it is
