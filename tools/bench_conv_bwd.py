"""Developer experiment: data gradient of the 3x3 encoder convolutions -- MIOpen's backward-data solver vs the same
quantity computed as a FORWARD convolution with the transposed, flipped weights (bf16, channels-last, batch 32)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")


def timeit(fn, reps=20, warm=5):
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


for cin, cout, f in ((64, 128, 32), (128, 256, 16), (256, 512, 8)):
    x = torch.randn(32, cin, 250, f, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(32, cout, 250, f, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    mask_in, mask_w = (True, False, False), (False, True, False)
    bwd = lambda m: torch.ops.aten.convolution_backward(dy, x, w, None, (1, 1), (1, 1), (1, 1), False, (0, 0), 1, m)
    t_fwd = timeit(lambda: F.conv2d(x, w, padding=1))
    t_bd = timeit(lambda: bwd(mask_in))
    t_bw = timeit(lambda: bwd(mask_w))
    wt = w.transpose(0, 1).flip(2, 3).contiguous(memory_format=torch.channels_last)
    t_alt = timeit(lambda: F.conv2d(dy, w.transpose(0, 1).flip(2, 3).contiguous(memory_format=torch.channels_last), padding=1))
    ref = bwd(mask_in)[0]
    alt = F.conv2d(dy, wt, padding=1)
    err = (ref.float() - alt.float()).abs().max().item() / ref.float().abs().max().item()
    print(f"{cin:3d}->{cout:3d} F={f:2d}: fwd {t_fwd:6.1f} us | bwd-data {t_bd:6.1f} us vs as-forward {t_alt:6.1f} us (rel diff {err:.1e}) | "
          f"bwd-weight {t_bw:6.1f} us", flush=True)
