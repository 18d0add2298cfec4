"""Would running the post-CNN part of a CRNN iteration as two sub-batches on two streams hide the GEMMs under the
recurrences?  (VERDICT r02, next #4.)  Measured on the forward chain of model_crnn.py:106-125 with the real kernels at
batch 32 x 250 frames -- input projection GEMM -> BiGRU recurrence (csrc/gru.hip) x 2 layers -> head GEMMs -- as
  (a) one batch of 32 on one stream (what the captured iteration does),
  (b) two sub-batches of 16 on two streams, the second started right behind the first,
HIP events around 20 repetitions each.  The recurrence's latency does not depend on the batch (4 sequences per CU either
way), so (b) can only win what the half-size GEMMs of one sub-batch hide under the other's recurrence, minus the lag of
the second sub-batch.  developer tool: python tools/bench_subbatch.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch
import torch.nn.functional as F
import seld_native as nat

dev = torch.device("cuda:0")
B, T, H = 32, 250, 256
torch.manual_seed(0)
bf = torch.bfloat16
x = torch.randn(B, T, 2048, device=dev).to(bf)
w_ih0 = (torch.randn(6 * H, 2048, device=dev) * 0.02).to(bf)
w_ih1 = (torch.randn(6 * H, 2 * H, device=dev) * 0.05).to(bf)
bias = torch.zeros(6 * H, device=dev).to(bf)
w_hh = (torch.rand(2, 3 * H, H, device=dev) * 2 - 1) / 16
b_hn = torch.zeros(2, H, device=dev)
w1 = (torch.randn(512, 512, device=dev) * 0.05).to(bf)
w2 = (torch.randn(9072, 512, device=dev) * 0.05).to(bf)


def chain(xs):
    gi = F.linear(xs, w_ih0, bias)
    y, _ = nat.gru_forward(gi.view(xs.shape[0], T, 2, 3 * H), w_hh, b_hn, True)
    gi = F.linear(y, w_ih1, bias)
    y, _ = nat.gru_forward(gi.view(xs.shape[0], T, 2, 3 * H), w_hh, b_hn, True)
    return F.linear(torch.relu(F.linear(y, w1)), w2)


def timed(fn, reps=20):
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


side = torch.cuda.Stream()


def split():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    a = chain(x[:B // 2])
    with torch.cuda.stream(side):
        b = chain(x[B // 2:])
    main.wait_stream(side)
    return a, b


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


full_us = timed(graphed(lambda: chain(x)))
split_us = timed(graphed(split))
half_us = timed(graphed(lambda: chain(x[:B // 2])))
print(f"post-CNN forward chain, B=32 T=250 (graph replay, HIP events): one batch {full_us:.0f} us; two sub-batches of 16 on two "
      f"streams {split_us:.0f} us; one sub-batch of 16 alone {half_us:.0f} us")
