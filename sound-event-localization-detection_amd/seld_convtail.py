"""Autograd wrapper of the fused CNN-block tail (csrc/convtail.hip): BatchNorm2d -> ReLU -> MaxPool2d((1, 2)) of
model_crnn.py:5-17 in one statistics pass + one apply pass forward, one reduction pass + one apply pass backward;
also BatchNorm2d -> ReLU and BatchNorm2d -> (+ shortcut) -> ReLU of the ResNet bottlenecks (resnet50_model.py:30-52).
Only the convolution output (and the shortcut) is kept for the backward pass (the unfused modules also keep the
BatchNorm output and the int64 pooling indices)."""
import torch
import torch.nn as nn

import seld_native

enabled = True        # flipped by the trainer from Config.FUSED_CONV_TAIL


def bn_applicable(bn, x):
    """Can the fused kernels stand in for ``relu(bn(x))`` on this activation?"""
    if not (enabled and x.is_cuda and x.dim() == 4 and type(bn) is nn.BatchNorm2d and bn.affine
            and bn.track_running_stats and bn.momentum is not None and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    if not seld_native.conv_tail_supported(x.shape[1]) or not x.is_contiguous(memory_format=torch.channels_last):
        return False
    # eval-mode BatchNorm with gradients flowing is not a training configuration of the reference: stock path
    return bn.training or not (torch.is_grad_enabled() and x.requires_grad)


def applicable(block, x):
    """``block``: a ConvBlock; ``x``: the convolution output."""
    pool = block.pool
    if pool is not None:
        k = pool.kernel_size if isinstance(pool.kernel_size, tuple) else (pool.kernel_size,) * 2
        s = pool.stride if isinstance(pool.stride, tuple) else (pool.stride,) * 2
        if tuple(k) != (1, 2) or tuple(s) != (1, 2) or pool.padding not in (0, (0, 0)) or pool.ceil_mode \
                or x.dim() != 4 or x.shape[3] % 2:
            return False
    return bn_applicable(block.bn, x)


class _ConvTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, running_mean, running_var, momentum, eps, training, pool):
        y, mean_invstd, scale_shift = seld_native.conv_tail_forward(x, weight, bias, running_mean, running_var,
                                                                    momentum, eps, training, pool, residual)
        ctx.save_for_backward(x, mean_invstd, scale_shift, residual if residual is not None else x.new_empty(0))
        ctx.pool = pool
        ctx.has_residual = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean_invstd, scale_shift, residual = ctx.saved_tensors
        if ctx.has_residual:
            dx, dweight, dbias, dres = seld_native.conv_tail_backward(x, dy, mean_invstd, scale_shift, ctx.pool, residual)
        else:
            dx, dweight, dbias = seld_native.conv_tail_backward(x, dy, mean_invstd, scale_shift, ctx.pool)
            dres = None
        return dx, dres, dweight, dbias, None, None, None, None, None, None


_collected = None          # the num_batches_tracked buffers of the fused tails run inside a ``batched_counters`` block


class batched_counters:
    """``with batched_counters():`` around an encoder's forward pass: the ``num_batches_tracked += 1`` of every fused
    BatchNorm tail inside becomes ONE multi-tensor add at exit (4 launches per CRNN iteration, 53 per ResNet-50 one,
    ~5 us apiece on a GPU-bound step); same values, nothing reads the counters in between (momentum is a constant)."""

    def __enter__(self):
        global _collected
        self.outer, _collected = _collected, []
        return self

    def __exit__(self, *exc):
        global _collected
        mine, _collected = _collected, self.outer
        if mine and exc[0] is None:
            torch._foreach_add_(mine, 1)
        return False


def bn_relu(bn, x, pool=1, residual=None):
    """relu(bn(x)) [-> MaxPool2d((1, 2)) when pool = 2]; with ``residual``: relu(bn(x) + residual)."""
    if bn.training:
        if _collected is not None:
            _collected.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
        residual = residual.contiguous(memory_format=torch.channels_last)
    return _ConvTail.apply(x, residual, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps,
                           bn.training, pool)


def conv_tail(block, x):
    return bn_relu(block.bn, x, 2 if block.pool is not None else 1)


def bn1d_silu_applicable(bn, x):
    """Can the fused kernels stand in for ``silu(bn(x))`` on a [rows, D] activation (BatchNorm1d -> Swish of the
    Conformer convolution module, model_conformer.py:71-96, evaluated channels-last by seld_dwconv)?"""
    if not (enabled and x.is_cuda and x.dim() == 2 and x.is_contiguous() and type(bn) is nn.BatchNorm1d and bn.affine
            and bn.track_running_stats and bn.momentum is not None and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    if not seld_native.conv_tail_supported(x.shape[1]):
        return False
    return bn.training or not (torch.is_grad_enabled() and x.requires_grad)


def bn1d_silu(bn, x):
    """silu(bn(x)) for x [rows, D]: the tail kernels in mode 4 (statistics pass + apply pass; the backward pass recomputes
    the normalised value and the SiLU derivative from x), on the [rows, D, 1, 1] channels-last view of the same memory."""
    if bn.training:
        if _collected is not None:
            _collected.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    rows, d = x.shape
    x4 = x.view(rows, 1, 1, d).permute(0, 3, 1, 2)
    y4 = _ConvTail.apply(x4, None, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, bn.training, 4)
    return y4.permute(0, 2, 3, 1).reshape(rows, d)
