"""Developer micro-benchmark: column sums of a tall bf16 matrix (the bias gradient of every Linear: 8000 rows)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import seld_native
dev = torch.device("cuda:0")


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    for _ in range(3):
        seld_native.stream_delay(dev, 1_000_000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


ones = torch.ones(1, 8000, device=dev, dtype=torch.bfloat16)
for n in (256, 512, 1024, 2048, 9072):
    g = torch.randn(8000, n, device=dev).to(torch.bfloat16)
    out = torch.empty(n, device=dev, dtype=torch.bfloat16)
    a = timeit(lambda: torch.sum(g, dim=0, dtype=torch.bfloat16, out=out))
    b = timeit(lambda: torch.mm(ones, g))
    c = timeit(lambda: torch.mv(g.t(), ones[0]))
    ref = g.float().sum(0)
    err_b = (torch.mm(ones, g)[0].float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"N={n:5d}: torch.sum {a:6.1f} us   ones @ g {b:6.1f} us (rel err {err_b:.1e})   mv {c:6.1f} us")
for n in (256, 512, 1024, 2048, 9072):
    g = torch.randn(8000, n, device=dev).to(torch.bfloat16)
    out = torch.empty(n, device=dev, dtype=torch.bfloat16)
    d = timeit(lambda: seld_native.column_sums(g, out))
    print(f"N={n:5d}: seld column_sums {d:6.1f} us (incl. the partial allocation and two launches)")
