"""Rank body of tests/test_ddp_gpu.py::test_train_model_shards_the_device_feed (run under torch.distributed.run).

Every rank builds the SAME small SELDDataset on the shared GPU (features, labels, windows by the HIP kernels), wraps it
in the stock DataLoader main.py builds (main.py:60-74) and calls trainer.train_model -- so the epoch loop itself picks
DeviceFeed, shards the window order over the ranks, trains through DistributedDataParallel (gloo rehearsal: the ranks
share one GPU) and all-reduces the epoch sums.  Rank r prints one JSON line with what the test compares."""
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "sound-event-localization-detection_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402


def main():
    out_dir = Path(sys.argv[1])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    import dataset
    import trainer
    from oracle import features as ofeat          # seeded synthetic inputs only (test infrastructure)
    from oracle import labels as olab
    cfg = trainer.config
    cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS = "crnn", [16, 16, 32, 32]      # HIP BiGRU (hidden 256), small encoder
    cfg.NUM_EPOCHS, cfg.BATCH_SIZE, cfg.SEED = 2, 3, 5
    cfg.SAVE_EVERY_N_EPOCHS = 1
    cfg.OUTPUT_PATH, cfg.CHECKPOINT_PATH = out_dir / "outputs", out_dir / "checkpoints"
    for d in (cfg.OUTPUT_PATH, cfg.CHECKPOINT_PATH):
        d.mkdir(parents=True, exist_ok=True)
    clips = [ofeat.synth_pcm(i, 4, 24000 * 7 + 480 * i, "noise") for i in range(2)]        # 14 s -> 14 windows
    rows = [olab.synth_metadata(i, meta_frames=70) for i in range(2)]
    train = dataset.SELDDataset.from_pcm(clips, rows, device=device)
    test = dataset.SELDDataset.from_pcm(clips[:1], rows[:1], device=device)
    train_loader = DataLoader(train, batch_size=cfg.BATCH_SIZE, shuffle=True)
    test_loader = DataLoader(test, batch_size=cfg.BATCH_SIZE, shuffle=False)
    torch.manual_seed(100 + rank)                 # the ranks initialise DIFFERENT weights, like the unseeded reference
    cfg.SEED = None
    feed = trainer.DeviceFeed(train_loader, device, rank, world, seed=7)
    shard = [int(i) for i in feed._order(1)]
    model, history = trainer.train_model(train_loader, test_loader, device=device)
    flat = torch.cat([p.detach().flatten().double() for p in model.parameters()])
    print("RANKLINE " + json.dumps({
        "rank": rank, "windows": len(train), "shard_epoch1": shard, "param_sum": flat.sum().item(),
        "param_abs": flat.abs().sum().item(), "train_losses": history["train_losses"],
        "test_losses": history["test_losses"], "config": {k: v for k, v in history["config"].items() if k != "grid_size"},
        "wrote_checkpoint": (cfg.CHECKPOINT_PATH / "best_model.pth").exists()}), flush=True)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
