"""Developer experiment: the encoder's first convolution (4 -> 64 channels, model_crnn.py:5-17) with its input channels
zero-padded to 8: does MIOpen then pick a solver without the 17 us output fill (SubTensorOpWithScalar1d) it needs now?"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
B, T, Fq = 32, 250, 64


def timeit(fn, reps=30, warm=5):
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


for cin in (4, 8, 16):
    x = torch.randn(B, cin, T, Fq, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, cin, 3, 3, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, 64, T, Fq, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    fwd = timeit(lambda: F.conv2d(x, w, padding=1))
    wgrad = timeit(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1,
                                                               (False, True, False)))
    print(f"conv {cin:2d}->64 F=64 bf16 channels-last: forward {fwd:6.1f} us, weight gradient {wgrad:6.1f} us (HIP events, back to back)")
