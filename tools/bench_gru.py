"""GRU recurrence micro-benchmark (developer tool; target of rocprofv3 --pmc passes).
usage: python tools/bench_gru.py [batch] [steps] [reps]"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import os
import torch
import seld_native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = int(sys.argv[2]) if len(sys.argv) > 2 else 250
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
H = 256
dev = torch.device("cuda:0")
torch.manual_seed(0)
gi = (torch.randn(B, T, 2, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.rand(2, 3 * H, H, device=dev) * 2 - 1) / 16
bn = torch.zeros(2, H, device=dev)
dy = torch.randn(B, T, 2 * H, device=dev).to(torch.bfloat16)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


y, saved = seld_native.gru_forward(gi, w, bn, True)
t_f = timeit(lambda: seld_native.gru_forward(gi, w, bn, True))
t_b = timeit(lambda: seld_native.gru_backward(dy, saved, y, w))
# kernel-only timings: pre-tiled operands straight through the C ABI
lib = seld_native.load_library()
P = seld_native._p
tiles = (B + seld_native.GRU_TILE - 1) // seld_native.GRU_TILE
wb = w.to(torch.bfloat16).contiguous()
wt = w.to(torch.bfloat16).transpose(1, 2).contiguous()
yk = torch.empty((tiles * seld_native.GRU_TILE, T, 2 * H), dtype=torch.bfloat16, device=dev)
dy_tile = seld_native.to_tile_device(dy.reshape(B, T, 2, 1, H), 1)
dg_tile = torch.empty((tiles, T, 2, 8, 2, 64, 2, seld_native.GRU_TILE // 2), dtype=torch.bfloat16, device=dev)
st = seld_native._stream_ptr(dev)
k_f = timeit(lambda: lib.seld_gru_forward(P(gi), 1, P(wb), P(bn), B, T, H, P(yk), P(saved), st))
k_i = timeit(lambda: lib.seld_gru_forward(P(gi), 1, P(wb), P(bn), B, T, H, P(yk), None, st))
dbias = torch.empty((tiles, 2, 4, H), device=dev)
k_b = timeit(lambda: lib.seld_gru_backward(P(dy_tile), P(saved), P(yk), 1, P(wt), tiles, T, H, P(dg_tile), P(dbias), st))
print(f"[{seld_native.GRU_TILE} sequences per tile] kernels only: forward {k_f * 1e3:.0f} us ({k_f * 1e3 / T:.2f} us/step)  forward(no save) {k_i * 1e3:.0f} us "
      f"({k_i * 1e3 / T:.2f} us/step)  backward {k_b * 1e3:.0f} us ({k_b * 1e3 / T:.2f} us/step)")
flop = 2 * B * T * 2 * H * 3 * H * 2          # per kernel (both directions)
print(f"gru B={B} T={T}: forward {t_f * 1e3:.0f} us ({t_f * 1e3 / T:.2f} us/step, {flop / t_f / 1e9:.1f} TFLOP/s) "
      f"backward {t_b * 1e3:.0f} us ({t_b * 1e3 / T:.2f} us/step)")
