"""Developer tool: which framework op (with input shapes) launches which GPU kernel in one steady-state CRNN
optimiser iteration -- torch.profiler over 3 iterations, grouped by op + shapes, sorted by device time."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
from torch.profiler import profile, ProfilerActivity
import trainer
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
trainer.config.MODEL_TYPE = sys.argv[1] if len(sys.argv) > 1 else "crnn"
torch.manual_seed(0)
model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), dev).train()
crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
trainer.enable_master_weights(model, dev)
opt = trainer.make_optimizer(model, 1e-3, dev)
x = torch.randn(32, 250, 4, 64, device=dev) * 20 - 30
mask = torch.zeros(32, 250, 648, dtype=torch.uint16, device=dev)
for _ in range(4):
    trainer.train_step(model, crit, opt, x, mask, dev)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(3):
        trainer.train_step(model, crit, opt, x, mask, dev)
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=70,
                                                          max_name_column_width=50, max_shapes_column_width=90))
