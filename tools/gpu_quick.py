"""Quick GPU sanity + timing of the log-mel kernel (developer tool)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import seld_native
from oracle import features as F

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count, "CUs")
pcm = F.synth_pcm(0, 4, 240000)
got = seld_native.logmel(pcm.to(dev)).cpu()
ref = F.logmel_torch(pcm)
print("parity 10s clip max|d|:", (got - ref).abs().max().item())
for n_clips in (1, 8, 32):
    x = (torch.randn(n_clips, 4, 1440000, device=dev) * 0.1)
    out = torch.empty(n_clips, 3001, 4, 64, device=dev)
    for layout in ("tcf", "cft"):
        o = out if layout == "tcf" else torch.empty(n_clips, 4, 64, 3001, device=dev)
        for _ in range(3):
            seld_native.logmel(x, layout=layout, out=o)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            seld_native.logmel(x, layout=layout, out=o)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        byt = n_clips * (4 * 1440000 * 4 + 4 * 64 * 3001 * 4)
        print(f"clips={n_clips:3d} layout={layout}: {ms*1e3:9.1f} us/launch  {ms*1e3/n_clips:8.1f} us/clip  {byt/ms/1e6:8.1f} GB/s")
