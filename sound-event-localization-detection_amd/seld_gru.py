"""Autograd wrapper of the persistent HIP BiGRU recurrence (csrc/gru.hip) behind ``nn.GRU``'s
parameters (model_crnn.py:65-72: GRU(2048, 256, num_layers=2, batch_first, bidirectional, dropout)).

Per layer:   gi = x [W_ih ; W_ih_reverse]^T + [b_ih ; b_ih_reverse]         one GEMM (hipBLASLt / MFMA)
             y  = recurrence(gi, W_hh, b_hh)                                 one launch, both directions
backward:    dg = recurrence_backward(dy, saved gates)                       one launch
             dx, dW_ih, db_ih, dW_hh, db_hh                                   five GEMMs / reductions
Inter-layer dropout as in nn.GRU (training only).  Gate order r | z | n (PyTorch).
"""
import torch
import torch.nn.functional as F

import seld_native

HIDDEN = 256


def applicable(module, x):
    return (module.hidden_size == HIDDEN and module.bidirectional and module.batch_first and module.bias
            and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and module.proj_size == 0)


class _BiGRULayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, b_ih, w_hh, b_hh):
        """x [B,T,In]; w_ih [6H,In]; b_ih [6H]; w_hh [2,3H,H]; b_hh [2,3H]."""
        low = torch.is_autocast_enabled() or x.dtype == torch.bfloat16
        cdt = torch.bfloat16 if low else torch.float32
        with torch.autocast(device_type="cuda", enabled=False):
            xc = x.to(cdt)
            # the r / z recurrent biases commute with the sigmoid argument: fold them into the GEMM's bias
            fold = b_hh.clone()
            fold[:, 2 * HIDDEN:] = 0
            gi = F.linear(xc, w_ih.to(cdt), (b_ih + fold.reshape(-1)).to(cdt))   # [B, T, 6H]
            b, t, _ = gi.shape
            need = x.requires_grad or w_ih.requires_grad or w_hh.requires_grad
            y, saved = seld_native.gru_forward(gi.view(b, t, 2, 3 * HIDDEN), w_hh, b_hh[:, 2 * HIDDEN:], need)
        ctx.save_for_backward(xc, w_ih, w_hh, y, saved if saved is not None else torch.empty(0))
        ctx.cdt = cdt
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, w_ih, w_hh, y, saved = ctx.saved_tensors
        cdt = ctx.cdt
        b, t, _ = y.shape
        h = HIDDEN
        with torch.autocast(device_type="cuda", enabled=False):
            # one un-tiling pass; everything below reads strided views of it (no further copies of the 8000-row
            # gradient matrices): slots (r, z, n) = d/d(gi), slots (r, z, n*r) = d/d(gh)
            dg = seld_native.gru_backward(dy.to(y.dtype), saved, y, w_hh)           # [B, T, 2, 4, H]
            d2 = dg.view(b * t, 2, 4 * h)
            x2 = xc.reshape(b * t, -1)
            w = w_ih.to(cdt)
            dx = d2[:, 0, :3 * h] @ w[:3 * h]
            dx.addmm_(d2[:, 1, :3 * h], w[3 * h:])
            dw_ih = torch.cat((d2[:, 0, :3 * h].t() @ x2, d2[:, 1, :3 * h].t() @ x2), dim=0).float()
            sums = torch.sum(d2, dim=0, dtype=torch.float32)                      # [2, 4H]
            db_ih = sums[:, :3 * h].reshape(-1)
            # h_{t-1} of the forward recurrence: y shifted by one step in each direction's own time order
            yv = y.view(b, t, 2, h)
            h_prev = torch.zeros_like(yv)
            h_prev[:, 1:, 0] = yv[:, :-1, 0]
            h_prev[:, :-1, 1] = yv[:, 1:, 1]
            full = _reduce_rows(d2, h_prev.view(b * t, 2, h))                     # [2, 4H, H]
            dw_hh = torch.cat((full[:, :2 * h], full[:, 3 * h:]), dim=1)
            db_hh = torch.cat((sums[:, :2 * h], sums[:, 3 * h:]), dim=1)
        dx = dx.view_as(xc)
        return dx.to(dy.dtype) if dx.dtype != dy.dtype and not torch.is_autocast_enabled() else dx, \
            dw_ih.to(w_ih.dtype), db_ih.to(w_ih.dtype), dw_hh.to(w_hh.dtype), db_hh.to(w_hh.dtype)


def _reduce_rows(a, c, chunks=8):
    """sum_n a[n, d, :]^T c[n, d, :] -> [D, Ga, Gc] fp32.  The output is tiny (2 x 1024 x 256) and the reduction
    long (B*T = 8000 rows): split the rows into chunks so the batched GEMM has enough tiles to fill the GPU,
    then add the partial products in fp32."""
    n = a.shape[0]
    while chunks > 1 and n % chunks:
        chunks //= 2
    a = a.view(chunks, n // chunks, a.shape[1], a.shape[2])
    c = c.view(chunks, n // chunks, c.shape[1], c.shape[2])
    return torch.einsum("sndg,sndh->sdgh", a, c).float().sum(dim=0)


def bigru_forward(module, x):
    """Drop-in for ``nn.GRU.forward(x)`` with h0 = 0: returns (output [B,T,2H], h_n [2*layers,B,H])."""
    out = x
    finals = []
    for layer in range(module.num_layers):
        p = lambda name: getattr(module, f"{name}_l{layer}")                      # noqa: E731
        pr = lambda name: getattr(module, f"{name}_l{layer}_reverse")             # noqa: E731
        w_ih = torch.cat((p("weight_ih"), pr("weight_ih")), dim=0)
        b_ih = torch.cat((p("bias_ih"), pr("bias_ih")), dim=0)
        w_hh = torch.stack((p("weight_hh"), pr("weight_hh")), dim=0)
        b_hh = torch.stack((p("bias_hh"), pr("bias_hh")), dim=0)
        out = _BiGRULayer.apply(out, w_ih, b_ih, w_hh, b_hh)
        finals += [out[:, -1, :HIDDEN], out[:, 0, HIDDEN:]]
        if module.training and module.dropout > 0 and layer + 1 < module.num_layers:
            out = F.dropout(out, p=module.dropout, training=True)
    return out, torch.stack(finals, dim=0)
