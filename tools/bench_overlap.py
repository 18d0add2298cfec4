"""Developer tool: does a weight-gradient GEMM on a side stream really run BESIDE the BiGRU backward recurrence
(16 of 256 CUs), and does the recurrence suffer?  HIP events only (rocprofv3's kernel trace serialises queues).
usage: python tools/bench_overlap.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import seld_native
from seld_linear import tall_product

B, T, H = 32, 250, 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
gi = (torch.randn(B, T, 2, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.rand(2, 3 * H, H, device=dev) * 2 - 1) / 16
bn = torch.zeros(2, H, device=dev)
dy = torch.randn(B, T, 2 * H, device=dev).to(torch.bfloat16)
y, saved = seld_native.gru_forward(gi, w, bn, True)
g2 = torch.randn(B * T, 9072, device=dev).to(torch.bfloat16)
x2 = torch.randn(B * T, 512, device=dev).to(torch.bfloat16)
dw = torch.empty(9072, 512, dtype=torch.bfloat16, device=dev)
print("stream priority range (least, greatest):", torch.cuda.Stream.priority_range())


def recurrence():
    seld_native.gru_backward(dy, saved, y, w)


def side_job():
    tall_product(g2, x2, out=dw)
    torch.sum(g2, dim=0, dtype=torch.float32)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def both(side, delay=0, jobs=1):
    main = torch.cuda.current_stream(dev)

    def fn():
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if delay:
                seld_native.stream_delay(dev, delay)
            for _ in range(jobs):
                side_job()
        recurrence()
        main.wait_stream(side)
    return fn


t_rec = timeit(recurrence)
t_job = timeit(side_job)
print(f"recurrence alone {t_rec:.0f} us (with its layout converters); side job alone {t_job:.0f} us; serial {t_rec + t_job:.0f} us")
for prio in (0,):
    side = torch.cuda.Stream(device=dev, priority=prio)
    for delay in (0, 15000, 25000, 40000):
        for jobs in (1, 2):
            t = timeit(both(side, delay, jobs))
            print(f"side priority {prio:2d} head start {delay:6d} ns, {jobs} job(s): together {t:.0f} us "
                  f"(serial would be {t_rec + jobs * t_job:.0f})")
