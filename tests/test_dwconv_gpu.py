"""GPU parity for the depthwise Conv1d kernels (csrc/dwconv.hip) and the channels-last Conformer convolution module
built on them, against the stock modules the reference composes (model_conformer.py:71-96).  Floating point: fp32
<= 1e-5 relative; bf16 bounded against the fp32 result."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("b,t,d,k", [(2, 17, 64, 7), (3, 250, 256, 31), (1, 5, 128, 31), (4, 100, 512, 15)])
def test_depthwise_conv_matches_nn_conv1d(gpu_device, b, t, d, k):
    import seld_native
    torch.manual_seed(0)
    conv = nn.Conv1d(d, d, k, padding=(k - 1) // 2, groups=d).to(gpu_device)
    x = torch.randn(b, t, d, device=gpu_device, requires_grad=True)
    y_ref = conv(x.transpose(1, 2)).transpose(1, 2)
    go = torch.randn_like(y_ref)
    y_ref.backward(go)
    w2 = conv.weight.detach().reshape(d, k)
    y = seld_native.dwconv1d(x.detach(), w2, conv.bias.detach())
    assert (y - y_ref).abs().max().item() <= 1e-5 * y_ref.abs().max().item()
    dx = seld_native.dwconv1d(go.contiguous(), w2, None, flip=True)
    assert (dx - x.grad).abs().max().item() <= 1e-5 * x.grad.abs().max().item()
    dw, db = seld_native.dwconv1d_wgrad(x.detach(), go, k)
    assert (dw - conv.weight.grad.reshape(d, k)).abs().max().item() <= 2e-5 * conv.weight.grad.abs().max().item()
    assert (db - conv.bias.grad).abs().max().item() <= 2e-5 * conv.bias.grad.abs().max().item()
    # bf16 activations: same kernels, bf16 loads / stores
    yb = seld_native.dwconv1d(x.detach().to(torch.bfloat16), w2, conv.bias.detach())
    assert yb.dtype == torch.bfloat16 and (yb.float() - y_ref).abs().max().item() <= 2e-2 * y_ref.abs().max().item()


def test_conformer_conv_module_channels_last_equals_stock(gpu_device):
    import seld_dwconv
    from model_conformer import ConformerConvModule
    torch.manual_seed(1)
    fused = ConformerConvModule(256, kernel_size=31, dropout=0.0).to(gpu_device)
    stock = ConformerConvModule(256, kernel_size=31, dropout=0.0).to(gpu_device)
    stock.load_state_dict(fused.state_dict())
    x = torch.randn(4, 50, 256, device=gpu_device)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    was = seld_dwconv.enabled
    try:
        seld_dwconv.enabled = True
        assert seld_dwconv.applicable(fused, xa)
        y = fused(xa)
        seld_dwconv.enabled = False
        y_ref = stock(xb)
    finally:
        seld_dwconv.enabled = was
    assert (y - y_ref).abs().max().item() <= 1e-4 * y_ref.abs().max().item()
    go = torch.randn_like(y)
    y.backward(go)
    y_ref.backward(go)
    assert (xa.grad - xb.grad).abs().max().item() <= 1e-4 * xb.grad.abs().max().item()
    for (n, p), (_, q) in zip(fused.named_parameters(), stock.named_parameters()):
        assert p.grad.shape == q.grad.shape
        if n == "depthwise_conv.bias":          # feeds a training-mode BatchNorm: its exact gradient is zero, both are noise
            assert p.grad.abs().max().item() <= 1e-3 and q.grad.abs().max().item() <= 1e-3
            continue
        assert (p.grad - q.grad).abs().max().item() <= 5e-4 * (q.grad.abs().max().item() + 1e-6), n
    for (n, p), (_, q) in zip(fused.named_buffers(), stock.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-4, atol=1e-5), n
