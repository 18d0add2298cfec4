"""Developer tool: N steady-state CRNN optimiser iterations (run once to populate MIOpen's find cache, then
again under rocprofv3 --kernel-trace --stats for a clean per-kernel breakdown)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import trainer
torch.backends.cudnn.benchmark = True
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
trainer.config.MODEL_TYPE = sys.argv[2] if len(sys.argv) > 2 else "crnn"
torch.manual_seed(0)
model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), dev).train()
crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
trainer.enable_master_weights(model, dev)
opt = trainer.make_optimizer(model, 1e-3, dev)
x = torch.randn(32, 250, 4, 64, device=dev) * 20 - 30
mask = torch.zeros(32, 250, 648, dtype=torch.uint16, device=dev)
for _ in range(3):
    trainer.train_step(model, crit, opt, x, mask, dev)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        trainer.train_step(model, crit, opt, x, mask, dev)
    t_issue = time.perf_counter() - t0                  # host done enqueueing (never blocks: no sync in a step)
    torch.cuda.synchronize()
    print(f"{trainer.config.MODEL_TYPE}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms / iteration over {n} iterations; "
          f"host enqueue time {t_issue / n * 1e3:.2f} ms / iteration", flush=True)
