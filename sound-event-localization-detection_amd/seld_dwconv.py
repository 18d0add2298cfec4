"""Autograd wrapper of the depthwise Conv1d kernels (csrc/dwconv.hip) and the channels-last evaluation of the
Conformer convolution module (model_conformer.py:71-96) built on it: the module's two pointwise Conv1d layers are
Linear layers on [B, T, D] (same parameters, kernel-size-1 weights squeezed), GLU / BatchNorm1d / Swish act on the
last dimension, and no [B, T, D] <-> [B, D, T] transposes are needed."""
import os

import torch
import torch.nn.functional as F

import seld_native
from seld_linear import _Linear

# "auto" (default): on for wide modules (d_model >= 512) only.  Measured A/B, same box, back to back: the bs-32
# ResNet50-Conformer (d_model 512, 4 blocks, 80 % GPU-busy) gains 3 % (22.1 / 22.3 clips/s on vs 21.6 / 21.5 off), while
# the bs-32 Conformer (d_model 256, host-launch bound at 55 % GPU-busy) LOSES 7 % (61.6 / 72.2 on vs 65.3 / 78.0 off):
# this path trades GPU time for a little more host work.  Config.FUSED_DWCONV (True / False / "auto") via
# trainer.prepare_model_for_device, or SELD_DWCONV=1 / 0, override.
_env = os.environ.get("SELD_DWCONV")
enabled = "auto" if _env is None else (_env == "1")


class _DepthwiseConv1d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        """x [B, T, D]; weight [D, 1, K] (the nn.Conv1d parameter); bias [D] or None."""
        w2 = weight.reshape(weight.shape[0], weight.shape[-1])
        y = seld_native.dwconv1d(x, w2, bias)
        ctx.save_for_backward(x, w2)
        ctx.meta = (weight.shape, weight.dtype, None if bias is None else bias.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        shape, w_dtype, b_dtype = ctx.meta
        dy = dy.to(x.dtype).contiguous()
        dx = seld_native.dwconv1d(dy, w2, None, flip=True) if ctx.needs_input_grad[0] else None
        dw, db = seld_native.dwconv1d_wgrad(x, dy, w2.shape[1])
        return dx, dw.reshape(shape).to(w_dtype), None if b_dtype is None else db.to(b_dtype)


def applicable(module, x):
    dw = module.depthwise_conv
    k = dw.kernel_size[0]
    on = enabled is True or (enabled == "auto" and dw.in_channels >= 512)
    return (on and x.is_cuda and x.dim() == 3 and x.dtype in (torch.float32, torch.bfloat16)
            and dw.groups == dw.in_channels == dw.out_channels and dw.stride == (1,) and dw.dilation == (1,)
            and dw.padding == ((k - 1) // 2,) and dw.padding_mode == "zeros"
            and seld_native.dwconv1d_supported(dw.in_channels, k)
            and module.pointwise_conv1.kernel_size == (1,) and module.pointwise_conv2.kernel_size == (1,))


def conv_module_forward(module, x):
    """ConformerConvModule.forward on [B, T, D] without leaving that layout."""
    b, t, d = x.shape
    import seld_layernorm
    y = seld_layernorm.norm(module.layer_norm, x)
    p1, p2, dw = module.pointwise_conv1, module.pointwise_conv2, module.depthwise_conv
    y = _Linear.apply(y, p1.weight.squeeze(-1), p1.bias)                    # [B, T, 2D]
    y = F.glu(y, dim=-1)
    low = torch.is_autocast_enabled()
    if low:
        y = y.to(torch.get_autocast_dtype("cuda"))
    y = _DepthwiseConv1d.apply(y, dw.weight, dw.bias)
    import seld_convtail
    y2 = y.reshape(b * t, d)
    if type(module.swish).__name__ in ("Swish", "SiLU") and seld_convtail.bn1d_silu_applicable(module.batch_norm, y2):
        y = seld_convtail.bn1d_silu(module.batch_norm, y2).view(b, t, d)    # BatchNorm1d -> Swish in two passes each way
    else:
        y = module.swish(module.batch_norm(y2).view(b, t, d))               # BatchNorm1d over (B, T) per channel
    y = _Linear.apply(y, p2.weight.squeeze(-1), p2.bias)
    return x + module.dropout(y)
