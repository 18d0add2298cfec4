"""Developer experiment: eager vs hipGraph replay of one CRNN optimiser iteration."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import trainer, seld_native
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
cfg = trainer.config
cfg.MODEL_TYPE = "crnn"
torch.manual_seed(0)
model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), dev).train()
crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-4, fused=True, capturable=True)
x = torch.randn(32, 250, 4, 64, device=dev) * 20 - 30
mask = torch.zeros(32, 250, 648, dtype=torch.uint16, device=dev)

def it():
    opt.zero_grad(set_to_none=False)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = model(x)
    l, _ = crit.loss_tensor(y, mask)
    l.backward()
    opt.step()
    return l

def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print(f"eager: {timeit(it):.2f} ms / iteration")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): it()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = it()
    print(f"graph: {timeit(g.replay):.2f} ms / iteration   loss={loss.item():.5f}")
except Exception as e:
    print("graph capture failed:", repr(e)[:500])
