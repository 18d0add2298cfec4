"""Autograd wrapper of the fused LayerNorm [-> ReLU] (csrc/layernorm.hip) behind ``nn.LayerNorm``'s parameters.

The heads of all three models are  Linear -> LayerNorm -> ReLU -> Dropout -> Linear  (model_crnn.py:77-83,
model_conformer.py:117-127, resnet50_model.py:80-91).  Under bf16 autocast the stock modules cast the Linear's bf16
output to fp32, normalise, clip and drop on fp32 tensors and cast back for the next Linear; here the activation keeps
its dtype (statistics and arithmetic in fp32 inside the kernel) and only x and two statistics per row are kept for the
backward pass.  Dropout stays the framework's (its random stream is the reference's)."""
import torch
import torch.nn as nn

import seld_native

enabled = True        # Config.FUSED_LAYERNORM via trainer.prepare_model_for_device


def applicable(ln, x):
    return (enabled and x.is_cuda and type(ln) is nn.LayerNorm and ln.elementwise_affine and ln.bias is not None
            and len(ln.normalized_shape) == 1 and x.shape[-1] == ln.normalized_shape[0]
            and x.dtype in (torch.float32, torch.bfloat16) and seld_native.layernorm_supported(x.shape[-1])
            and ln.weight.dtype == torch.float32 and x.numel() > 0)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu):
        xc = x.contiguous()
        y, stats = seld_native.layernorm_forward(xc, weight, bias, eps, relu)
        ctx.save_for_backward(xc, weight, bias, stats)
        ctx.relu = relu
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        xc, weight, bias, stats = ctx.saved_tensors
        dx, dweight, dbias = seld_native.layernorm_backward(xc, dy.to(xc.dtype).contiguous(), weight, bias, stats,
                                                            ctx.relu)
        return dx, dweight, dbias, None, None


def layer_norm(ln, x, relu=False):
    """``ln(x)`` (``relu(ln(x))`` with ``relu``) in x's dtype."""
    return _LayerNorm.apply(x, ln.weight, ln.bias, ln.eps, relu)


def norm(ln, x):
    """``ln(x)`` for the LayerNorms inside the Conformer blocks (model_conformer.py:10-29,31-69,71-113): the fused kernel
    keeps the activation's dtype -- under bf16 autocast the stock module casts to fp32, normalises and the next Linear
    casts back (two framework copies around every one of the 10 / 20 LayerNorms of a Conformer / ResNet50-Conformer
    iteration) -- statistics and arithmetic in fp32 either way."""
    if x.is_cuda and applicable(ln, x):
        return layer_norm(ln, x, relu=False)
    return ln(x)


def run_head(head, x):
    """``head(x)`` for the  Linear, LayerNorm, ReLU, Dropout, Linear  Sequential of the three models."""
    if (len(head) == 5 and isinstance(head[1], nn.LayerNorm) and isinstance(head[2], nn.ReLU)
            and isinstance(head[3], nn.Dropout)):
        y = head[0](x)
        if applicable(head[1], y):
            return head[4](head[3](layer_norm(head[1], y, relu=True)))
        return head[4](head[3](head[2](head[1](y))))
    return head(x)
