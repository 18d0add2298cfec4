"""Spatial-feature micro-benchmark (developer tool; target of rocprofv3 --kernel-trace --stats).
usage: python tools/bench_spatial.py [kind logmel_gcc|logmel_iv] [channels] [n_clips] [reps]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import seld_native

kind = sys.argv[1] if len(sys.argv) > 1 else "logmel_gcc"
channels = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_clips = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
dev = torch.device("cuda:0")
L = 1440000
x = torch.randn(n_clips, channels, L, device=dev) * 0.1
for _ in range(2):
    out = seld_native.spatial_features(x, kind)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    out = seld_native.spatial_features(x, kind)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
byt = n_clips * (channels * L * 4 + out.shape[2] * 64 * out.shape[1] * 4)
print(f"{kind} {channels}ch clips={n_clips}: {ms:.3f} ms/call {byt / ms / 1e6:.1f} GB/s algorithmic ({byt / 1e6:.1f} MB/call), "
      f"frac {byt / ms / 1e6 / 8000:.4f}")
