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
import seld_overlap
from seld_linear import chunk_count, tall_chunks, tall_product

HIDDEN = 256


def applicable(module, x):
    # fp32 runs (no autocast) keep the stock fp32 nn.GRU so that the "<= 1e-3 relative on logits" bar of fp32 parity
    # runs holds: the HIP recurrence feeds bf16 operands to the MFMA.  ``module.allow_fp32 = True`` opts in.
    low = torch.is_autocast_enabled() or x.dtype in (torch.bfloat16, torch.float16)
    if not low and not getattr(module, "allow_fp32", False):
        return False
    return (module.hidden_size == HIDDEN and module.bidirectional and module.batch_first and module.bias
            and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16, torch.float16) and module.proj_size == 0)


class _BiGRULayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, b_ih, w_hh, b_hh, overlap=False, feature_cf=None):
        """x [B,T,In]; w_ih [6H,In]; b_ih [6H]; w_hh [2,3H,H]; b_hh [2,3H].  ``overlap``: the parameters are
        ``seld_overlap.defer`` aliases -- their weight gradients may be produced on the side stream.
        ``feature_cf = (C, F)``: x's features are ordered (frequency, channel) while w_ih's columns are (channel,
        frequency): the columns are permuted here, and dW_ih is un-permuted by the SAME job that computes it -- so that
        no kernel of the main stream reads a weight gradient the side stream may still be writing."""
        ctx.overlap = overlap
        ctx.feature_cf = feature_cf
        if feature_cf is not None:
            c, f = feature_cf
            w_ih = w_ih.view(w_ih.shape[0], c, f).transpose(1, 2).reshape(w_ih.shape[0], f * c)
        low = torch.is_autocast_enabled() or x.dtype == torch.bfloat16
        cdt = torch.bfloat16 if low else torch.float32
        with torch.autocast(device_type="cuda", enabled=False):
            xc = x.to(cdt)
            # the r / z recurrent biases commute with the sigmoid argument: they ride on the GEMM's bias
            if b_ih.dtype == torch.float32 and b_hh.dtype == torch.float32 and b_ih.is_contiguous() and b_hh.is_contiguous():
                gi_bias, b_hn = seld_native.gru_fold_bias(b_ih, b_hh.reshape(-1), cdt)        # one launch
            else:
                fold = b_hh.clone()
                fold[:, 2 * HIDDEN:] = 0
                gi_bias, b_hn = (b_ih + fold.reshape(-1)).to(cdt), b_hh[:, 2 * HIDDEN:]
            gi = F.linear(xc, w_ih.to(cdt), gi_bias)                               # [B, T, 6H]
            b, t, _ = gi.shape
            need = x.requires_grad or w_ih.requires_grad or w_hh.requires_grad
            y, saved = seld_native.gru_forward(gi.view(b, t, 2, 3 * HIDDEN), w_hh, b_hn, need)
        ctx.save_for_backward(xc, w_ih, w_hh, y, saved if saved is not None else torch.empty(0))
        ctx.cdt = cdt
        ctx.dtypes = (w_ih.dtype, b_ih.dtype, w_hh.dtype, b_hh.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, w_ih, w_hh, y, saved = ctx.saved_tensors
        cdt = ctx.cdt
        b, t, _ = y.shape
        h = HIDDEN
        n = b * t
        with torch.autocast(device_type="cuda", enabled=False):
            dyc = dy.to(y.dtype)
            # weight-gradient jobs queued by the layers above start now, beside this recurrence (seld_overlap)
            seld_overlap.launch_pending(dy.device)
            dgi, dghn, dbias = seld_native.gru_backward(dyc, saved, y, w_hh, raw_bias=True)
            dgi2 = dgi.view(n, 6 * h)                                             # d/d(gi), both directions
            x2 = xc.reshape(n, -1)
            t_wih, t_bih, t_whh, t_bhh = ctx.dtypes

            dw_ih = torch.empty((6 * h, x2.shape[1]), dtype=t_wih, device=dy.device)
            dw_hh = torch.empty((2, 3 * h, h), dtype=t_whh, device=dy.device)

            def weight_grads():
                if ctx.feature_cf is None:
                    tall_product(dgi2, x2, out=dw_ih)                            # [6H, In]
                else:                                                            # columns back to (channel, frequency)
                    c, f = ctx.feature_cf
                    perm = tall_product(dgi2, x2, out_dtype=t_wih)
                    dw_ih.view(6 * h, c, f).copy_(perm.view(6 * h, f, c).transpose(1, 2))
                # h_{t-1} of the forward recurrence: y shifted by one step in each direction's own time order
                hp = seld_native.gru_previous_state(y).view(n, 2 * h)
                # d/d(gh) = (da_r, da_z, da_n * r).  Two well-shaped products instead of eight skinny ones: all of
                # dgi and dghn against both directions' h_prev; the wanted blocks are those with matching directions
                # (the cross-direction blocks and dgi's n rows are computed and dropped: ~6 GFLOP, cheaper than the
                # launches).  The chunk sums and the block extraction are one kernel (seld_gru_dwhh_finish).
                chunks = chunk_count(n, 6 * h, 2 * h)
                p_gi = tall_chunks(dgi2, hp, chunks)                                # [chunks, (dir, gate, unit), (dir', unit')]
                p_n = tall_chunks(dghn.view(n, 2 * h), hp, chunks)                  # [chunks, (dir, unit), (dir', unit')]
                seld_native.gru_dwhh_finish(p_gi, p_n, dw_hh)

            dx = (dgi2 @ w_ih.to(cdt)).view_as(xc)
            if ctx.overlap:
                # on the side stream, beside the recurrence of the layer below (seld_overlap.launch_pending there)
                seld_overlap.submit(dy.device, [dgi, dghn, y, xc, dw_ih, dw_hh], weight_grads)
            elif seld_overlap.conv_wgrad_side and seld_overlap.enabled and dy.is_cuda:
                # layer 0 under the captured step: beside the convolution backward (joined by the stepper; carried over
                # to the next backward stage when the data-parallel step cuts the pass below this layer)
                seld_overlap.launch_now(dy.device, [dgi, dghn, y, xc, dw_ih, dw_hh], weight_grads, last_of_stage=True,
                                        outputs=[dw_ih, dw_hh])
            else:
                weight_grads()
            db_ih, db_hh = seld_native.gru_bias_grads(dbias)                       # [6H], [6H] fp32, one launch
            db_hh = db_hh.view(2, 3 * h)
        # (fresh aliases of dw_ih / dw_hh: a queued job keeps its tensors referenced, and autograd clones a gradient that
        # is referenced elsewhere -- before the job has filled it)
        return dx.to(dy.dtype) if dx.dtype != dy.dtype and not torch.is_autocast_enabled() else dx, \
            dw_ih.view_as(dw_ih), db_ih.to(t_bih), dw_hh.view_as(dw_hh), db_hh.to(t_bhh), None, None


from seld_pack import adjacent as _adjacent, join as _join, pack as _pack   # noqa: E402  (shared with the attention layers)


def pack_parameters(module):
    """Re-home each layer's (forward, reverse) parameter pairs in one buffer per kind so that the per-iteration
    concatenations in ``bigru_forward`` are views.  Call after the module is on its device and before the optimiser
    / DDP wrapper are built (``trainer.prepare_model_for_device`` does); ``state_dict`` keys and values unchanged."""
    with torch.no_grad():
        for layer in range(module.num_layers):
            for kind in ("weight_ih", "bias_ih", "weight_hh", "bias_hh"):
                _pack((getattr(module, f"{kind}_l{layer}"), getattr(module, f"{kind}_l{layer}_reverse")))


def bigru_forward(module, x, feature_cf=None, overlap=False, need_hn=True):
    """Drop-in for ``nn.GRU.forward(x)`` with h0 = 0: returns (output [B,T,2H], h_n [2*layers,B,H]; None unless
    ``need_hn`` -- the CRNN never looks at it and the stack is one more launch).
    ``feature_cf = (C, F)``: the input features are ordered (frequency, channel) -- f * C + c -- instead of the
    parameters' (channel, frequency) order c * F + f; layer 0's W_ih columns are permuted to match.
    ``overlap``: weight gradients of layers >= 1 on the side stream (seld_overlap)."""
    overlap = overlap and seld_overlap.active(x)
    params = []
    for layer in range(module.num_layers):
        p = lambda name: getattr(module, f"{name}_l{layer}")                      # noqa: E731
        pr = lambda name: getattr(module, f"{name}_l{layer}_reverse")             # noqa: E731
        group = (_join(p("weight_ih"), pr("weight_ih")), _join(p("bias_ih"), pr("bias_ih")),
                 _join(p("weight_hh"), pr("weight_hh")), _join(p("bias_hh"), pr("bias_hh")))
        # layers >= 1: their weight gradients hide under the recurrence of the layer below.  The identity node must
        # exist BEFORE layer 0's node (see seld_overlap); layer 0's own would only compete with the convolutions.
        params.append(seld_overlap.defer(*group) if overlap and layer > 0 else group)
    out = x
    finals = []
    for layer in range(module.num_layers):
        w_ih, b_ih, w_hh, b_hh = params[layer]
        out = _BiGRULayer.apply(out, w_ih, b_ih, w_hh.view(2, 3 * HIDDEN, HIDDEN), b_hh.view(2, 3 * HIDDEN),
                                overlap and layer > 0, feature_cf if layer == 0 else None)
        if need_hn:
            finals += [out[:, -1, :HIDDEN], out[:, 0, HIDDEN:]]
        if module.training and module.dropout > 0 and layer + 1 < module.num_layers:
            out = F.dropout(out, p=module.dropout, training=True)
    return out, (torch.stack(finals, dim=0) if need_hn else None)
