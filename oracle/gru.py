"""Oracle (TEST INFRASTRUCTURE): CPU restatement of one bidirectional GRU layer, PyTorch gate
equations (the nn.GRU the reference instantiates at model_crnn.py:65-72), written as an explicit
time loop so the HIP kernel's bf16 rounding points can be reproduced:

  exact=True  : plain fp32 math == torch.nn.GRU (used to pin this restatement to nn.GRU itself)
  exact=False : W_hh and the h fed to the recurrent matmul are rounded to bf16 (what the MFMA sees);
                accumulation, gates and the carried state stay fp32 -- the kernel's arithmetic model.
"""
import torch


def _round(t, exact):
    return t if exact else t.to(torch.bfloat16).to(torch.float32)


def bigru_layer(x, w_ih, b_ih, w_hh, b_hh, exact=False, gi_dtype=torch.float32):
    """x [B,T,In]; w_ih [2][3H,In]; b_ih [2][3H]; w_hh [2][3H,H]; b_hh [2][3H] (index 0 forward, 1 reverse).
    Returns y [B,T,2H].  Differentiable (rounding casts pass gradients straight through)."""
    b, t, _ = x.shape
    h_size = w_hh[0].shape[1]
    outs = []
    for d in range(2):
        gi = torch.nn.functional.linear(x.to(gi_dtype), w_ih[d].to(gi_dtype), b_ih[d].to(gi_dtype)).float()
        w = _round(w_hh[d].float(), exact)
        h = x.new_zeros(b, h_size, dtype=torch.float32)
        ys = [None] * t
        order = range(t) if d == 0 else range(t - 1, -1, -1)
        for tt in order:
            gh = _round(h, exact) @ w.t() + b_hh[d].float()
            i_r, i_z, i_n = gi[:, tt].chunk(3, dim=-1)
            h_r, h_z, h_n = gh.chunk(3, dim=-1)
            r = torch.sigmoid(i_r + h_r)
            z = torch.sigmoid(i_z + h_z)
            n = torch.tanh(i_n + r * h_n)
            h = (1.0 - z) * n + z * h
            ys[tt] = h
        outs.append(torch.stack(ys, dim=1))
    return torch.cat(outs, dim=-1)
