"""Log-mel kernel micro-benchmark (developer tool; also the target of the rocprofv3 --pmc passes).
usage: python tools/bench_logmel.py [n_clips] [reps] [dtype f32|i16] [layout tcf|cft]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import seld_native

n_clips = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dtype = sys.argv[3] if len(sys.argv) > 3 else "f32"
layout = sys.argv[4] if len(sys.argv) > 4 else "tcf"
dev = torch.device("cuda:0")
L = 1440000
x = torch.randn(n_clips, 4, L, device=dev) * 0.1
if dtype == "i16":
    x = (x * 32768).clamp(-32768, 32767).to(torch.int16)
shape = (n_clips, 3001, 4, 64) if layout == "tcf" else (n_clips, 4, 64, 3001)
out = torch.empty(shape, device=dev)
for _ in range(3):
    seld_native.logmel(x, layout=layout, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    seld_native.logmel(x, layout=layout, out=out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
byt = n_clips * (4 * L * x.element_size() + 4 * 64 * 3001 * 4)
print(f"logmel {dtype} {layout} clips={n_clips}: {ms * 1e3:.1f} us/launch {ms * 1e3 / n_clips:.2f} us/clip "
      f"{byt / ms / 1e6:.1f} GB/s algorithmic ({byt / 1e6:.1f} MB/launch)")
