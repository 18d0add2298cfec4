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


# --------------------------------------------------------------------------------------------------------------
# The recurrence alone, in the arithmetic model of the bf16 build of csrc/gru.hip (what `seld_gru_forward` /
# `seld_gru_backward` compute between the host GEMMs), forward AND an explicit backward restatement.
#
# Rounding points of the kernels (low=True: the bf16 build; low=False: the fp32 build, which still feeds the MFMA
# bf16 operands; operands=None removes those too -- plain fp32, used to pin the explicit backward to autograd):
#   forward : gi arrives in bf16; W_hh and the h operand of  gh = W_hh h  are bf16, accumulation / gates / the carried
#             state fp32; y = bf16(h); saved r, z, n, gh_n (+ b_hn) = IEEE fp16.
#   backward: r, z, n, gh_n from the fp16 saves; h_{t-1} is re-read from y (bf16); dy bf16; all products fp32;
#             dgh = (da_r, da_z, da_n r) goes through the MFMA as bf16 against bf16 W_hh^T; the carried dh stays
#             fp32; outputs da_r, da_z, da_n, da_n r are written in bf16; the four bias sums add the UNROUNDED values.
# Gate equations: torch.nn.GRU (the module the reference builds at model_crnn.py:65-72).

def _q(t, dtype):
    return t if dtype is None else t.to(dtype).to(torch.float32)


def recurrence_forward(gi, w_hh, b_hn, low=True, operands=torch.bfloat16):
    """gi [B,T,2,3H] fp32 values as the kernel reads them (b_ih and the r/z halves of b_hh folded in);
    w_hh [2,3H,H]; b_hn [2,H].  Returns y [B,T,2H] and the saved gates (r, z, n, g = gh_n + b_hn), each [B,T,2,H]."""
    b, t = gi.shape[0], gi.shape[1]
    hs = w_hh.shape[2]
    data = torch.bfloat16 if low else None
    save = torch.float16 if low else None
    y = torch.zeros(b, t, 2, hs)
    saved = [torch.zeros(b, t, 2, hs) for _ in range(4)]
    for d in range(2):
        w = _q(w_hh[d].float(), operands)
        h = torch.zeros(b, hs)
        for tt in (range(t) if d == 0 else range(t - 1, -1, -1)):
            gh = _q(h, operands) @ w.t()
            i_r, i_z, i_n = gi[:, tt, d].float().chunk(3, dim=-1)
            h_r, h_z, h_n = gh.chunk(3, dim=-1)
            g = h_n + b_hn[d].float()
            r = torch.sigmoid(i_r + h_r)
            z = torch.sigmoid(i_z + h_z)
            n = torch.tanh(i_n + r * g)
            h = z * (h - n) + n
            y[:, tt, d] = _q(h, data)
            for dst, src in zip(saved, (r, z, n, g)):
                dst[:, tt, d] = _q(src, save)
    return y.reshape(b, t, 2 * hs), saved


def recurrence_backward(dy, saved, y, w_hh, low=True, operands=torch.bfloat16):
    """dy [B,T,2H]; saved = (r, z, n, g) and y as the forward pass left them (already rounded); w_hh [2,3H,H].
    Returns dgi [B,T,2,3,H] = (da_r, da_z, da_n), dghn [B,T,2,H] = da_n r (both rounded like the kernel's stores)
    and dbias [2,4,H] = sums over (B,T) of the unrounded (da_r, da_z, da_n, da_n r)."""
    b, t = dy.shape[0], dy.shape[1]
    hs = w_hh.shape[2]
    data = torch.bfloat16 if low else None
    r_s, z_s, n_s, g_s = saved
    yv = y.reshape(b, t, 2, hs).float()
    dyv = dy.reshape(b, t, 2, hs).float()
    dgi = torch.zeros(b, t, 2, 3, hs)
    dghn = torch.zeros(b, t, 2, hs)
    dbias = torch.zeros(2, 4, hs, dtype=torch.float64)
    for d in range(2):
        w = _q(w_hh[d].float(), operands)                          # [3H, H]
        dh = torch.zeros(b, hs)
        order = list(range(t) if d == 0 else range(t - 1, -1, -1))
        for k in range(t - 1, -1, -1):                             # reverse of the forward processing order
            tt = order[k]
            r, z, n, g = r_s[:, tt, d], z_s[:, tt, d], n_s[:, tt, d], g_s[:, tt, d]
            hprev = yv[:, order[k - 1], d] if k > 0 else torch.zeros(b, hs)
            dtot = dyv[:, tt, d] + dh
            dn = dtot * (1.0 - z)
            dz = dtot * (hprev - n)
            da_n = dn * (1.0 - n * n)
            da_z = dz * z * (1.0 - z)
            da_r = da_n * g * r * (1.0 - r)
            dg_n = da_n * r
            for slot, v in enumerate((da_r, da_z, da_n, dg_n)):
                dbias[d, slot] += v.double().sum(dim=0)
            dgi[:, tt, d, 0], dgi[:, tt, d, 1], dgi[:, tt, d, 2] = _q(da_r, data), _q(da_z, data), _q(da_n, data)
            dghn[:, tt, d] = _q(dg_n, data)
            dgh = torch.cat((_q(da_r, operands), _q(da_z, operands), _q(dg_n, operands)), dim=-1)
            dh = dtot * z + dgh @ w
    return dgi, dghn, dbias.float()


def recurrence_autograd(gi, w_hh, b_hn, dy):
    """Plain fp32 autograd through the same recurrence (no rounding at all): the gradients the explicit backward
    above restates.  Returns (y, d/d gi [B,T,2,3H], d/d w_hh, d/d b_hn)."""
    gi = gi.detach().float().clone().requires_grad_(True)
    w_hh = w_hh.detach().float().clone().requires_grad_(True)
    b_hn = b_hn.detach().float().clone().requires_grad_(True)
    b, t = gi.shape[0], gi.shape[1]
    hs = w_hh.shape[2]
    outs = []
    for d in range(2):
        h = torch.zeros(b, hs)
        ys = [None] * t
        for tt in (range(t) if d == 0 else range(t - 1, -1, -1)):
            gh = h @ w_hh[d].t()
            i_r, i_z, i_n = gi[:, tt, d].chunk(3, dim=-1)
            h_r, h_z, h_n = gh.chunk(3, dim=-1)
            r = torch.sigmoid(i_r + h_r)
            z = torch.sigmoid(i_z + h_z)
            n = torch.tanh(i_n + r * (h_n + b_hn[d]))
            h = z * (h - n) + n
            ys[tt] = h
        outs.append(torch.stack(ys, dim=1))
    y = torch.cat(outs, dim=-1)
    (y * dy.float()).sum().backward()
    return y.detach(), gi.grad, w_hh.grad, b_hn.grad
