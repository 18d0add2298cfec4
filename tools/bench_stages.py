"""What does cutting the captured iteration cost on one GPU?  Replays the data-parallel step's graphs without collectives
(seld_graph.GraphedTrainStep(split=True), exactly what N ranks replay) for the CRNN at batch 32 and prints the duration of
every graph.  SELD_CUT_LEVELS=0 / 1 / 2: no cut (one backward graph + the update graph), the primary cut, both cuts.
developer tool: python tools/bench_stages.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import seld_cut
import seld_graph
import trainer

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
cfg = trainer.config
cfg.MODEL_TYPE = sys.argv[1] if len(sys.argv) > 1 else "crnn"
torch.manual_seed(0)
model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), dev).train()
trainer.enable_master_weights(model, dev)
crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
opt = trainer.make_optimizer(model, 1e-3, dev, capturable=True)
x = torch.randn(32, 250, 4, 64, device=dev) * 20 - 30
m = torch.zeros(32, 250, 648, dtype=torch.uint16, device=dev)
for split in (False, True):
    step = seld_graph.GraphedTrainStep(model, crit, opt, dev, autocast=lambda: trainer.autocast_context(dev), split=split)
    for _ in range(seld_graph.WARMUP + 3):
        step(x, m)
    step.timing = True
    for _ in range(30):
        step(x, m)
    seg = step.segment_ms()
    print(f"{cfg.MODEL_TYPE} cut levels {seld_cut.levels} split={split}: graphs {['%.3f' % s for s in seg]} ms, iteration {sum(seg):.3f} ms",
          flush=True)
    step.close()
