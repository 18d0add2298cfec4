"""Developer experiment: operand-layout variants of the three head GEMMs (8000 x 512 x 9072, bf16)."""
import sys
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


for m, k, n in ((8000, 512, 9072), (8000, 2048, 1536), (8000, 512, 1536), (8000, 512, 512)):
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = torch.randn(n, k, device=dev).to(torch.bfloat16)
    wt = w.t().contiguous()
    g = torch.randn(m, n, device=dev).to(torch.bfloat16)
    b = torch.randn(n, device=dev).to(torch.bfloat16)
    gf = 2.0 * m * k * n / 1e9
    res = {
        "fwd x@W^T (linear)": timeit(lambda: torch.nn.functional.linear(x, w, b)),
        "fwd x@Wt (addmm NN)": timeit(lambda: torch.addmm(b, x, wt)),
        "dgrad g@W (NN)": timeit(lambda: g @ w),
        "dgrad g@Wt^T (NT)": timeit(lambda: g @ wt.t()),
        "wgrad g^T@x (TN)": timeit(lambda: g.t() @ x),
        "wgrad (x^T@g)^T (TN other way)": timeit(lambda: (x.t() @ g)),
    }
    print(f"M={m} K={k} N={n} ({gf:.1f} GF): " + " | ".join(f"{a}: {t:.1f} us ({gf / t * 1e3:.0f} TF/s)" for a, t in res.items()), flush=True)
