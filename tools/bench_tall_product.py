"""Developer experiment: chunk count of seld_linear.tall_product (a^T c with 8000 rows) for the shapes of the CRNN."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
dev = torch.device("cuda:0")


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


def product(a, c, chunks, out_dtype):
    n = a.shape[0]
    if chunks == 1:
        return a.t() @ c
    av = a.unflatten(0, (chunks, n // chunks)).transpose(1, 2)
    cv = c.unflatten(0, (chunks, n // chunks))
    return torch.sum(torch.bmm(av, cv), dim=0, dtype=out_dtype)


for g, k in ((1536, 2048), (1536, 512), (512, 512), (9072, 512), (512, 256), (1024, 512)):
    a = torch.randn(8000, g, device=dev).to(torch.bfloat16)
    c = torch.randn(8000, k, device=dev).to(torch.bfloat16)
    out = [f"G={g} K={k}:"]
    for chunks in (1, 2, 4, 5, 8, 10, 16, 20):
        if 8000 % chunks:
            continue
        out.append(f"{chunks}:{timeit(lambda: product(a, c, chunks, torch.bfloat16)):.0f}us")
    print(" ".join(out), flush=True)
