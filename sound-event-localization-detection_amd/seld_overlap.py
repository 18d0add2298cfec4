"""Weight gradients on a side HIP stream, hidden under the BiGRU backward recurrences.

The backward recurrence of a BiGRU layer (csrc/gru.hip) is a persistent kernel on 16 of the 256 CUs (one per
direction and 4-sequence tile at batch 32) for ~365 us, and the chain head -> layer 1 -> layer 0 is serial: the
other 240 CUs idle.  The WEIGHT gradients of the layers just before it on that chain (the head's Linears, then
layer 1's dW_ih / dW_hh) are not on the chain: nothing needs them before the optimiser (or DDP's bucket of that
parameter).  ``SeldLinear`` / ``seld_gru`` enqueue them on a side stream; the main stream goes on with the data
gradient and the next recurrence.

Ordering without touching the autograd engine: ``defer(...)`` routes the parameters through an identity node that
is created EARLY in the forward pass (before GRU layer 0), so the engine -- highest sequence number first -- runs
its backward LATE: after layer 0's recurrence, before the convolution stack.  That node is where the main stream
waits for the side stream, so everything downstream of it (AccumulateGrad, DDP's reducer hooks and their bucket
copies / all-reduce, which then overlap the convolution backward) sees finished gradients on the stream it expects.
If the engine ever ordered it differently the result would still be correct -- only the overlap would be lost.
"""
import os

import torch

enabled = os.environ.get("SELD_OVERLAP", "1") != "0"      # Config.OVERLAP_WEIGHT_GRADS via trainer.prepare_model_for_device
_streams = {}


def side_stream(device):
    index = device.index if device.index is not None else torch.cuda.current_device()
    if index not in _streams:
        _streams[index] = torch.cuda.Stream(device=index)
    return _streams[index]


def active(x):
    return enabled and x.is_cuda and torch.is_grad_enabled()


class _Deferred(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *params):
        ctx.device = params[0].device
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        torch.cuda.current_stream(ctx.device).wait_stream(side_stream(ctx.device))
        return grads


def defer(*params):
    """Aliases of ``params`` whose gradients may be produced on the side stream (``fork``)."""
    return _Deferred.apply(*params)


def defer_linear(*modules):
    """Give each ``SeldLinear`` a deferred alias of its parameters for the forward pass in flight."""
    for m in modules:
        ps = (m.weight,) if m.bias is None else (m.weight, m.bias)
        m.__dict__["_deferred"] = defer(*ps)


class fork:
    """``with fork(device, inputs) as f: ... f.outputs(...)``: run the block on the side stream once the main stream's
    work enqueued so far is done.  ``inputs`` were allocated on the main stream and are read here, ``outputs`` are
    allocated here and consumed on the main stream: both are recorded with the caching allocator."""

    def __init__(self, device, *inputs):
        self.main = torch.cuda.current_stream(device)
        self.side = side_stream(device)
        self.inputs = inputs
        self.ctx = torch.cuda.stream(self.side)

    def __enter__(self):
        self.side.wait_stream(self.main)
        self.ctx.__enter__()
        return self

    def outputs(self, *tensors):
        for t in tensors:
            if t is not None:
                t.record_stream(self.main)

    def __exit__(self, *exc):
        self.ctx.__exit__(*exc)
        for t in self.inputs:
            if t is not None:
                t.record_stream(self.side)
        return False
