"""Developer experiment: does MIOpen's solver tuning (MIOPEN_FIND_ENFORCE=3, user perf-db kept in the tree) speed up
the encoder's 3x3 convolutions (model_crnn.py:5-17) at the workload's shapes?  Times forward, data gradient (as a forward
convolution with transposed weights, model_crnn._Conv3x3) and weight gradient of the four blocks, bf16 channels-last,
batch 32 x 250 frames.  usage: python tools/tune_miopen.py [tag]   (environment decides tuned / untuned)"""
import os
import sys
import time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
B, T = 32, 250
tag = sys.argv[1] if len(sys.argv) > 1 else "run"


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


total = 0.0
t_start = time.time()
for cin, cout, freq in [(4, 64, 64), (64, 128, 32), (128, 256, 16), (256, 512, 8)]:
    x = torch.randn(B, cin, T, freq, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, cout, T, freq, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    wt = w.transpose(0, 1).flip(2, 3).contiguous(memory_format=torch.channels_last)
    fwd = timeit(lambda: F.conv2d(x, w, padding=1))
    dgrad = timeit(lambda: F.conv2d(dy, wt, padding=1)) if cin > 4 else 0.0
    wgrad = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                               (False, True, False)))
    total += fwd + dgrad + wgrad
    print(f"[{tag}] conv {cin:3d}->{cout:3d} F={freq:2d}: forward {fwd:7.1f} us, data gradient {dgrad:7.1f} us, weight gradient "
          f"{wgrad:7.1f} us", flush=True)
print(f"[{tag}] total {total:.1f} us per iteration; wall {time.time() - t_start:.0f} s; MIOPEN_FIND_ENFORCE="
      f"{os.environ.get('MIOPEN_FIND_ENFORCE')} MIOPEN_USER_DB_PATH={os.environ.get('MIOPEN_USER_DB_PATH')}")
