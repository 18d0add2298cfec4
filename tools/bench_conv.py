"""Developer tool: time the CRNN pieces (conv blocks in several dtype/layout variants, GRU, head, loss)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "sound-event-localization-detection_amd")]
import torch, torch.nn as nn, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")

def timeit(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

B, T = 32, 250
shapes = [(4, 64, 64), (64, 128, 32), (128, 256, 16), (256, 512, 8)]
for cin, cout, freq in shapes:
    for name, dtype, cl, padc in (("bf16 nhwc", torch.bfloat16, True, 0), ("bf16 nhwc pad8", torch.bfloat16, True, 8),
                                  ("bf16 nchw", torch.bfloat16, False, 0), ("fp32 nchw", torch.float32, False, 0),
                                  ("fp32 nhwc", torch.float32, True, 0)):
        if padc and cin % 8 == 0: continue
        ci = cin if not padc else ((cin + 7) // 8) * 8
        conv = nn.Conv2d(ci, cout, 3, padding=1, bias=False).to(dev).to(dtype)
        x = torch.randn(B, ci, T, freq, device=dev, dtype=dtype, requires_grad=True)
        if cl:
            conv = conv.to(memory_format=torch.channels_last)
            x = x.detach().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        def step():
            y = conv(x)
            y.backward(y)
        try:
            ms = timeit(step)
            flops = 3 * 2 * B * T * freq * cout * cin * 9
            print(f"conv {cin:3d}->{cout:3d} F={freq:2d} {name:15s}: {ms:8.3f} ms fwd+bwd  {flops / ms / 1e9:8.1f} TFLOP/s")
        except Exception as e:
            print(f"conv {cin}->{cout} {name}: FAILED {e}")

import trainer, model_crnn, seld_rnn, loss as loss_mod
for amp in (True, False):
    for fused in (True, False):
        seld_rnn.SeldGRU.fused_enabled = fused
        torch.manual_seed(0)
        m = model_crnn.SELD_CRNN().to(dev).to(memory_format=torch.channels_last).train()
        crit = loss_mod.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
        opt = torch.optim.Adam(m.parameters(), lr=1e-3, fused=True)
        x = torch.randn(B, T, 4, 64, device=dev) * 20 - 30
        mask = torch.zeros(B, T, 648, dtype=torch.uint16, device=dev)
        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
                y = m(x)
            l, _ = crit.loss_tensor(y, mask)
            l.backward(); opt.step()
        ms = timeit(step, reps=5, warm=3)
        print(f"CRNN train step bs=32 amp={amp} fused_gru={fused}: {ms:.2f} ms  ({39.4 * 32 / ms:.1f} TFLOP/s)")
        # module breakdown (forward only, eval of pieces)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            f = model_crnn.run_cnn_encoder(m.cnn_blocks, x)
            t_cnn = timeit(lambda: model_crnn.run_cnn_encoder(m.cnn_blocks, x), 5, 2)
            t_rnn = timeit(lambda: m.rnn(f), 5, 2)
            r, _ = m.rnn(f)
            t_head = timeit(lambda: m.fnn(r), 5, 2)
        print(f"   fwd only: cnn {t_cnn:.2f} ms, rnn {t_rnn:.2f} ms, head {t_head:.2f} ms")
