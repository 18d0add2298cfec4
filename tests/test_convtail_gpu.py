"""GPU parity for the fused CNN-block tail (csrc/convtail.hip): BatchNorm2d -> ReLU -> MaxPool2d((1, 2)) of
model_crnn.py:5-17, forward and backward, through the C ABI.

Reference: the stock torch modules the upstream ConvBlock is made of (nn.BatchNorm2d, nn.ReLU, nn.MaxPool2d) in
fp32 on the same device -- a floating-point kernel, so the bar is a tolerance: fp32 build <= 2e-5 relative;
bf16 build within one bf16 ulp of the fp32 result on the forward, gradients <= 2 % of the tensor's scale
(the unfused bf16 modules themselves are no closer).  Running statistics follow nn.BatchNorm2d's momentum rule.
"""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _stock(c, pool, device):
    bn = nn.BatchNorm2d(c).to(device)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, device=device) + 0.5)
        bn.bias.copy_(torch.randn(c, device=device) * 0.3)
    layers = [bn, nn.ReLU()]
    if pool == 2:
        layers.append(nn.MaxPool2d((1, 2)))
    return bn, nn.Sequential(*layers)


def _mostly_close(got, want, rel, max_bad_frac=1e-5):
    """All but a vanishing fraction of the elements within rel * scale: an element whose two pooling candidates (or
    whose distance to the ReLU threshold) differ by a rounding error may legitimately route its gradient differently."""
    s = want.abs().max().item() + 1e-6
    bad = ((got.float() - want.float()).abs() > rel * s).float().mean().item()
    return bad <= max_bad_frac


def _inputs(shape, device, seed, offset=0.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    b, c, t, f = shape
    x = torch.randn(shape, generator=g) * (torch.rand(1, c, 1, 1, generator=g) * 3 + 0.2) + \
        torch.randn(1, c, 1, 1, generator=g) * 2 + offset
    return x.to(device).contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("shape,pool", [((2, 64, 5, 8), 2), ((3, 128, 7, 16), 2), ((1, 8, 1, 2), 2),
                                        ((2, 512, 9, 4), 2), ((2, 256, 3, 6), 1), ((4, 64, 250, 64), 2)])
def test_fp32_forward_backward_match_stock_modules(gpu_device, shape, pool):
    import seld_native
    torch.manual_seed(1)
    bn, ref = _stock(shape[1], pool, gpu_device)
    x = _inputs(shape, gpu_device, 3).requires_grad_(True)
    rm0, rv0 = bn.running_mean.clone(), bn.running_var.clone()
    y_ref = ref(x)
    go = torch.randn_like(y_ref)
    y_ref.backward(go)
    rm, rv = rm0.clone(), rv0.clone()
    y, mean_invstd, scale_shift = seld_native.conv_tail_forward(x.detach(), bn.weight.detach(), bn.bias.detach(), rm, rv,
                                                                bn.momentum, bn.eps, True, pool)
    assert y.shape == y_ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    scale = y_ref.abs().max().item() + 1e-6
    assert _mostly_close(y, y_ref, 2e-5, 0.0) or (y - y_ref).abs().max().item() <= 2e-5 * scale
    assert torch.allclose(rm, bn.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv, bn.running_var, rtol=1e-5, atol=1e-6)
    dx, dw, db = seld_native.conv_tail_backward(x.detach(), go.contiguous(memory_format=torch.channels_last),
                                                mean_invstd, scale_shift, pool)
    assert _mostly_close(dx, x.grad, 1e-4), "dx"
    for got, want, name in ((dw, bn.weight.grad, "dweight"), (db, bn.bias.grad, "dbias")):
        s = want.abs().max().item() + 1e-6
        assert (got - want).abs().max().item() <= 2e-4 * s, name


def test_large_mean_does_not_cancel(gpu_device):
    """mean >> std: the shifted accumulation keeps the variance (E[x^2] - E[x]^2 in fp32 would lose it)."""
    import seld_native
    shape = (4, 64, 50, 16)
    bn, ref = _stock(64, 2, gpu_device)
    x = _inputs(shape, gpu_device, 5, offset=3000.0)
    y_ref = ref.double()(x.double()).float()          # float64 arbiter: the stock fp32 kernel itself cancels here
    bn.float()
    y, _, _ = seld_native.conv_tail_forward(x, bn.weight.detach(), bn.bias.detach(), torch.zeros(64, device=gpu_device),
                                            torch.ones(64, device=gpu_device), 0.1, bn.eps, True, 2)
    assert (y - y_ref).abs().max().item() <= 2e-3 * (y_ref.abs().max().item() + 1e-6)


def test_eval_mode_uses_running_statistics(gpu_device):
    import seld_native
    bn, ref = _stock(128, 2, gpu_device)
    with torch.no_grad():
        bn.running_mean.normal_()
        bn.running_var.uniform_(0.5, 2.0)
    ref.eval()
    x = _inputs((2, 128, 6, 8), gpu_device, 7)
    rm, rv = bn.running_mean.clone(), bn.running_var.clone()
    y, _, _ = seld_native.conv_tail_forward(x, bn.weight.detach(), bn.bias.detach(), rm, rv, 0.1, bn.eps, False, 2)
    assert (y - ref(x)).abs().max().item() <= 2e-5 * ref(x).abs().max().item()
    assert torch.equal(rm, bn.running_mean) and torch.equal(rv, bn.running_var)


# (8, 16, 250, 64) ... (8, 32, 250, 8): the four blocks of the CRNN_CNN_CHANNELS = [16, 16, 32, 32], batch-8 model of
# tests/test_master_weights_gpu.py -- two and four threads per row, the narrowest geometry the kernels accept beside C = 8
@pytest.mark.parametrize("shape", [(2, 64, 11, 8), (32, 64, 250, 64), (32, 512, 250, 8), (8, 16, 250, 64),
                                   (8, 16, 250, 32), (8, 32, 250, 16), (8, 32, 250, 8)])
def test_bf16_block_matches_fp32_reference(gpu_device, shape):
    """The ConvBlock module path under bf16: fused tail vs the stock modules in fp32 on the same bf16 conv output."""
    import seld_native
    c = shape[1]
    bn, ref = _stock(c, 2, gpu_device)
    xb = _inputs(shape, gpu_device, 11).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    x32 = xb.float().requires_grad_(True)
    y_ref = ref(x32)
    go = torch.randn_like(y_ref).to(torch.bfloat16)
    y_ref.backward(go.float())
    rm, rv = torch.zeros(c, device=gpu_device), torch.ones(c, device=gpu_device)
    y, mean_invstd, scale_shift = seld_native.conv_tail_forward(xb, bn.weight.detach(), bn.bias.detach(), rm, rv, 0.1,
                                                                bn.eps, True, 2)
    assert y.dtype == torch.bfloat16
    err = (y.float() - y_ref).abs()
    assert (err <= y_ref.abs() * 2 ** -7 + 1e-3).all()                      # within one bf16 ulp
    assert torch.allclose(rm, bn.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(rv, bn.running_var, rtol=1e-4, atol=1e-5)
    dx, dw, db = seld_native.conv_tail_backward(xb, go.contiguous(memory_format=torch.channels_last), mean_invstd,
                                                scale_shift, 2)
    # elements whose bf16-rounded pair ties or straddles zero may route differently from the fp32 reference:
    # compare in aggregate (relative L2) and the parameter gradients
    rel = (dx.float() - x32.grad).norm().item() / (x32.grad.norm().item() + 1e-12)
    assert rel <= 2e-2, rel
    for got, want, name in ((dw, bn.weight.grad, "dweight"), (db, bn.bias.grad, "dbias")):
        assert (got - want).abs().max().item() <= 2e-2 * (want.abs().max().item() + 1e-6), name


def test_convblock_module_uses_fused_tail_and_trains(gpu_device):
    """ConvBlock (model_crnn.py:5-17) end to end: same outputs / gradients / buffers as the stock composition."""
    import seld_convtail
    from model_crnn import ConvBlock
    torch.manual_seed(3)
    fused = ConvBlock(4, 64, pool_size=(1, 2)).to(gpu_device).to(memory_format=torch.channels_last)
    stock = ConvBlock(4, 64, pool_size=(1, 2)).to(gpu_device).to(memory_format=torch.channels_last)
    stock.load_state_dict(fused.state_dict())
    x = torch.randn(3, 4, 20, 16, device=gpu_device).contiguous(memory_format=torch.channels_last)
    assert seld_convtail.applicable(fused, fused.conv(x))
    y = fused(x)
    seld_convtail.enabled = False
    try:
        y_ref = stock(x)
    finally:
        seld_convtail.enabled = True
    assert (y - y_ref).abs().max().item() <= 1e-4 * y_ref.abs().max().item()
    go = torch.randn_like(y)
    y.backward(go)
    y_ref.backward(go)
    for (n, p), (_, q) in zip(fused.named_parameters(), stock.named_parameters()):
        assert (p.grad - q.grad).abs().max().item() <= 2e-4 * (q.grad.abs().max().item() + 1e-6), n
    for (n, p), (_, q) in zip(fused.named_buffers(), stock.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-5, atol=1e-6), n
    # eval mode: running statistics, no buffer update
    fused.eval(), stock.eval()
    with torch.no_grad():
        assert (fused(x) - stock.pool(stock.act(stock.bn(stock.conv(x))))).abs().max().item() <= 1e-4


def test_rejects_unsupported_layouts(gpu_device):
    import seld_native
    x = torch.randn(2, 64, 4, 8, device=gpu_device)                          # NCHW contiguous: not channels-last
    with pytest.raises(seld_native.SeldNativeError):
        seld_native.conv_tail_forward(x, None, None, None, None, 0.1, 1e-5, True, 2)
    assert not seld_native.conv_tail_supported(12) and not seld_native.conv_tail_supported(8 * 3)
    assert seld_native.conv_tail_supported(64) and seld_native.conv_tail_supported(512)


@pytest.mark.parametrize("shape,dtype", [((2, 64, 5, 8), torch.float32), ((3, 256, 50, 16), torch.float32),
                                         ((8, 256, 250, 16), torch.bfloat16)])
def test_residual_mode_matches_bottleneck_tail(gpu_device, shape, dtype):
    """relu(bn3(y) + shortcut) of a ResNet bottleneck (resnet50_model.py:30-52): forward, dx, d(shortcut), dW, db."""
    import seld_native
    c = shape[1]
    bn = nn.BatchNorm2d(c).to(gpu_device)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(c, device=gpu_device) + 0.5)
        bn.bias.copy_(torch.randn(c, device=gpu_device) * 0.3)
    x = _inputs(shape, gpu_device, 21).to(dtype).contiguous(memory_format=torch.channels_last)
    res = _inputs(shape, gpu_device, 22).to(dtype).contiguous(memory_format=torch.channels_last)
    x32, r32 = x.float().requires_grad_(True), res.float().requires_grad_(True)
    y_ref = torch.relu(bn(x32) + r32)
    go = torch.randn_like(y_ref).to(dtype)
    y_ref.backward(go.float())
    rm, rv = torch.zeros(c, device=gpu_device), torch.ones(c, device=gpu_device)
    y, mi, ss = seld_native.conv_tail_forward(x, bn.weight.detach(), bn.bias.detach(), rm, rv, 0.1, bn.eps, True, 1,
                                              residual=res)
    dx, dw, db, dres = seld_native.conv_tail_backward(x, go.contiguous(memory_format=torch.channels_last), mi, ss, 1,
                                                      residual=res)
    if dtype == torch.float32:
        assert _mostly_close(y, y_ref, 2e-5)
        assert _mostly_close(dx, x32.grad, 1e-4) and _mostly_close(dres, r32.grad, 1e-6)
        tol = 2e-4
    else:
        assert ((y.float() - y_ref).abs() <= y_ref.abs() * 2 ** -6 + 2e-2).float().mean().item() >= 0.9999
        # bf16: an element whose rounded pre-activation lands on the other side of zero routes differently from the
        # fp32 reference; the aggregate error is what is bounded (the unfused bf16 modules behave the same way)
        assert (dx.float() - x32.grad).norm().item() <= 5e-2 * x32.grad.norm().item()
        assert (dres.float() - r32.grad).norm().item() <= 5e-2 * r32.grad.norm().item()
        for got, want, name in ((dw, bn.weight.grad, "dweight"), (db, bn.bias.grad, "dbias")):
            assert (got - want).norm().item() <= 5e-2 * want.norm().item(), name
        return
    for got, want, name in ((dw, bn.weight.grad, "dweight"), (db, bn.bias.grad, "dbias")):
        assert (got - want).abs().max().item() <= tol * (want.abs().max().item() + 1e-6), name


def test_resnet_bottleneck_uses_fused_tails(gpu_device):
    import seld_convtail
    from resnet50_model import Bottleneck
    torch.manual_seed(5)
    down = nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256))
    blk = Bottleneck(64, 64, 1, down).to(gpu_device).to(memory_format=torch.channels_last)
    ref = Bottleneck(64, 64, 1, nn.Sequential(nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256)))
    ref = ref.to(gpu_device).to(memory_format=torch.channels_last)
    ref.load_state_dict(blk.state_dict())
    x = torch.randn(2, 64, 10, 8, device=gpu_device).contiguous(memory_format=torch.channels_last)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = blk(xa)
    seld_convtail.enabled = False
    try:
        y_ref = ref(xb)
    finally:
        seld_convtail.enabled = True
    assert (y - y_ref).abs().max().item() <= 1e-4 * y_ref.abs().max().item()
    go = torch.randn_like(y)
    y.backward(go)
    y_ref.backward(go)
    assert (xa.grad - xb.grad).abs().max().item() <= 5e-4 * xb.grad.abs().max().item()
    for (n, p), (_, q) in zip(blk.named_parameters(), ref.named_parameters()):
        assert (p.grad - q.grad).abs().max().item() <= 5e-4 * (q.grad.abs().max().item() + 1e-6), n
    for (n, p), (_, q) in zip(blk.named_buffers(), ref.named_buffers()):
        assert torch.allclose(p.float(), q.float(), rtol=1e-4, atol=1e-5), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,d", [(8000, 256), (8000, 512), (1000, 64), (37, 128)])
def test_bn1d_silu_matches_stock_modules(gpu_device, rows, d, dtype):
    """Tail mode 4 (csrc/convtail.hip): BatchNorm1d -> Swish of the Conformer convolution module (model_conformer.py:71-96)
    against the stock modules in fp32 on the same device -- output, running statistics, dx, dweight, dbias."""
    import seld_convtail
    torch.manual_seed(rows + d)
    x = (torch.randn(rows, d, device=gpu_device) * 1.5 + 0.3)
    dy = torch.randn(rows, d, device=gpu_device)
    ref_bn = nn.BatchNorm1d(d).to(gpu_device)
    with torch.no_grad():
        ref_bn.weight.uniform_(0.5, 1.5)
        ref_bn.bias.uniform_(-0.5, 0.5)
    bn = nn.BatchNorm1d(d).to(gpu_device)
    bn.load_state_dict(ref_bn.state_dict())
    xr = x.clone().requires_grad_(True)
    yr = torch.nn.functional.silu(ref_bn(xr))
    yr.backward(dy)
    xl = x.to(dtype).requires_grad_(True)
    assert seld_convtail.bn1d_silu_applicable(bn, xl)
    y = seld_convtail.bn1d_silu(bn, xl)
    y.backward(dy.to(dtype))
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    scale = yr.abs().max().item()
    assert (y.float() - yr).abs().max().item() <= tol * scale
    assert torch.allclose(bn.running_mean, ref_bn.running_mean, rtol=1e-5 if dtype == torch.float32 else 1e-2, atol=1e-3)
    assert torch.allclose(bn.running_var, ref_bn.running_var, rtol=1e-4 if dtype == torch.float32 else 2e-2, atol=1e-3)
    assert int(bn.num_batches_tracked) == 1
    rel = lambda a, b: ((a.float() - b).norm() / b.norm().clamp_min(1e-12)).item()
    gtol = 1e-4 if dtype == torch.float32 else 3e-2
    assert rel(xl.grad, xr.grad) <= gtol
    assert rel(bn.weight.grad, ref_bn.weight.grad) <= gtol
    assert rel(bn.bias.grad, ref_bn.bias.grad) <= gtol
