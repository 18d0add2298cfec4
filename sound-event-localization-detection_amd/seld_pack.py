"""Parameters that are multiplied together live side by side.

Several layers of the models apply two or three weight matrices to the SAME input -- the forward / reverse directions of
``nn.GRU`` (model_crnn.py:65-72), the q / k / v projections of ``MultiHeadSelfAttention`` (model_conformer.py:31-69) -- as
separate GEMMs on separate ``nn.Parameter``s.  ``pack`` re-homes such a group in ONE buffer (values, names and
``state_dict`` unchanged: every parameter becomes a slice of it) and ``join`` then hands the whole buffer to a single GEMM
as a view: no concatenation copy forward, and the gradient of the joined matrix splits back into views of one tensor."""
import torch


class _Joined(torch.autograd.Function):
    """torch.cat(tensors, 0) for parameters that are ADJACENT slices of one buffer: the concatenation is a view."""

    @staticmethod
    def forward(ctx, *tensors):
        a = tensors[0]
        ctx.rows = [t.shape[0] for t in tensors]
        out = a.new_empty(0)
        out.set_(a.untyped_storage(), a.storage_offset(), (sum(ctx.rows),) + tuple(a.shape[1:]))
        return out

    @staticmethod
    def backward(ctx, grad):
        return tuple(grad.split(ctx.rows, dim=0))


def adjacent(*tensors):
    a = tensors[0]
    offset = a.storage_offset()
    for t in tensors:
        if not (t.dtype == a.dtype and t.is_contiguous() and t.shape[1:] == a.shape[1:]
                and t.untyped_storage().data_ptr() == a.untyped_storage().data_ptr() and t.storage_offset() == offset):
            return False
        offset += t.numel()
    return True


def join(*tensors):
    """``torch.cat(tensors, dim=0)``; a view when the tensors were packed."""
    return _Joined.apply(*tensors) if adjacent(*tensors) else torch.cat(tensors, dim=0)


def pack(tensors):
    """Re-home the parameters (same dtype, same trailing shape) as consecutive slices of one new buffer."""
    tensors = list(tensors)
    if adjacent(*tensors):
        return
    with torch.no_grad():
        flat = torch.cat([t.detach().reshape(-1) for t in tensors])
        offset = 0
        for t in tensors:
            n = t.numel()
            t.data = flat[offset:offset + n].view_as(t)
            offset += n
