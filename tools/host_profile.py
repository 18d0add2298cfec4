"""Developer tool: where does the HOST time of a CRNN optimiser iteration go?  cProfile over N steady-state iterations
(the step is enqueue-bound once the kernels are short: every microsecond of Python / dispatcher time is wall time)."""
import cProfile
import pstats
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import trainer
torch.backends.cudnn.benchmark = True
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
trainer.config.MODEL_TYPE = sys.argv[2] if len(sys.argv) > 2 else "crnn"
torch.manual_seed(0)
model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), dev).train()
crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
trainer.enable_master_weights(model, dev)
opt = trainer.make_optimizer(model, 1e-3, dev)
x = torch.randn(32, 250, 4, 64, device=dev) * 20 - 30
mask = torch.zeros(32, 250, 648, dtype=torch.uint16, device=dev)
for _ in range(5):
    trainer.train_step(model, crit, opt, x, mask, dev)
torch.cuda.synchronize()
prof = cProfile.Profile()
prof.enable()
for _ in range(n):
    trainer.train_step(model, crit, opt, x, mask, dev)
prof.disable()
torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(60)
