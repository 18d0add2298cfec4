"""Fused LayerNorm [-> ReLU] (csrc/layernorm.hip) against torch.nn.functional.layer_norm in float64 / float32:
forward, all three gradients, every supported width, ragged row counts, bf16 and fp32 activations."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _reference(x, w, b, eps, relu, dy):
    x64 = x.double().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True)
    y = F.layer_norm(x64, (x.shape[-1],), w64, b64, eps)
    if relu:
        y = torch.relu(y)
    y.backward(dy.double())
    return y.detach(), x64.grad, w64.grad, b64.grad


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("d", [256, 512, 1024, 2048])
@pytest.mark.parametrize("rows", [1, 5, 8000])
def test_fp32_matches_float64_reference(rows, d, relu):
    import seld_native
    torch.manual_seed(rows + d)
    x = torch.randn(rows, d, device=DEV) * 3 + 1.5
    w = torch.randn(d, device=DEV) * 0.5 + 1
    b = torch.randn(d, device=DEV) * 0.2
    dy = torch.randn(rows, d, device=DEV)
    y, stats = seld_native.layernorm_forward(x, w, b, 1e-5, relu)
    dx, dw, db = seld_native.layernorm_backward(x, dy, w, b, stats, relu)
    ry, rdx, rdw, rdb = _reference(x, w, b, 1e-5, relu, dy)
    # a pre-activation within rounding of zero may be clipped on one side only: exclude |z| < 1e-5 from dx / masks
    assert (y.double() - ry).abs().max().item() <= 2e-5
    z = F.layer_norm(x.double(), (d,), w.double(), b.double(), 1e-5)
    clean_rows = ((z.abs() > 1e-5) | (not relu)).all(dim=1)
    assert clean_rows.float().mean().item() > 0.9
    assert (dx.double() - rdx)[clean_rows].abs().max().item() <= 2e-4
    scale = max(1.0, rows ** 0.5)
    # column sums: an element whose ReLU decision is ambiguous may be counted on one side only
    ambiguous = (z.abs() <= 1e-5) & relu
    xhat = (x.double() - x.double().mean(dim=1, keepdim=True)) / (x.double().var(dim=1, unbiased=False, keepdim=True) + 1e-5).sqrt()
    slack_w = (ambiguous * (dy.double() * xhat).abs()).sum(dim=0)
    slack_b = (ambiguous * dy.double().abs()).sum(dim=0)
    assert ((dw.double() - rdw).abs() <= 2e-4 * scale + slack_w).all()
    assert ((db.double() - rdb).abs() <= 2e-4 * scale + slack_b).all()
    assert torch.allclose(stats[:, 0].double(), x.double().mean(dim=1), atol=1e-5)


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("d", [512, 1024])
def test_bf16_activations_round_once(d, relu):
    """bf16 in / out: the result is the fp32 computation on the bf16 input, rounded once on the way out."""
    import seld_native
    torch.manual_seed(d)
    rows = 8000
    x = (torch.randn(rows, d, device=DEV) * 2).bfloat16()
    w = torch.randn(d, device=DEV) * 0.5 + 1
    b = torch.randn(d, device=DEV) * 0.2
    dy = torch.randn(rows, d, device=DEV).bfloat16()
    y, stats = seld_native.layernorm_forward(x, w, b, 1e-5, relu)
    dx, dw, db = seld_native.layernorm_backward(x, dy, w, b, stats, relu)
    assert y.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16
    ry, rdx, rdw, rdb = _reference(x.float(), w, b, 1e-5, relu, dy.float())
    assert (y.double() - ry).abs().max().item() <= 2 ** -8 * ry.abs().max().item() + 1e-6
    z = F.layer_norm(x.double(), (d,), w.double(), b.double(), 1e-5)
    clean_rows = ((z.abs() > 1e-5) | (not relu)).all(dim=1)
    err = (dx.double() - rdx)[clean_rows].abs().max().item()
    assert err <= 2 ** -8 * rdx.abs().max().item() + 1e-6
    assert (dw.double() - rdw).abs().max().item() <= 2e-2
    assert (db.double() - rdb).abs().max().item() <= 2e-2


def test_module_path_under_autocast_and_head_rewrite():
    """run_head == the stock Sequential (eval mode: no dropout randomness), forward and parameter gradients."""
    import seld_layernorm
    from seld_linear import SeldLinear
    torch.manual_seed(3)
    head = nn.Sequential(SeldLinear(512, 512), nn.LayerNorm(512), nn.ReLU(), nn.Dropout(0.3), SeldLinear(512, 96)).to(DEV)
    head.eval()
    x = torch.randn(4, 250, 512, device=DEV)
    grads = []
    outs = []
    for fused in (False, True):
        seld_layernorm.enabled = fused
        head.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = seld_layernorm.run_head(head, x)
        y.float().square().mean().backward()
        outs.append(y.float())
        grads.append({k: p.grad.float().clone() for k, p in head.named_parameters()})
    seld_layernorm.enabled = True
    assert (outs[0] - outs[1]).abs().max().item() <= 2e-2 * outs[0].abs().max().item()
    for k in grads[0]:
        ref = grads[0][k]
        assert (grads[1][k] - ref).abs().max().item() <= 3e-2 * ref.abs().max().item() + 1e-7, k


def test_unsupported_width_and_empty_input_use_the_stock_module():
    import seld_layernorm
    ln = nn.LayerNorm(384).to(DEV)
    assert not seld_layernorm.applicable(ln, torch.randn(3, 384, device=DEV))
    ln = nn.LayerNorm(512).to(DEV)
    assert not seld_layernorm.applicable(ln, torch.randn(0, 512, device=DEV))
    assert seld_layernorm.applicable(ln, torch.randn(2, 512, device=DEV))
