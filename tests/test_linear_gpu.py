"""GPU parity for SeldLinear (seld_linear.py): the split-K weight gradient against stock nn.Linear autograd
(model_crnn.py:77-83, model_conformer.py:28-51).  Floating point: fp32 <= 1e-5 relative; bf16 autocast within the
rounding of the stock bf16 path (both compared with an fp32 reference)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,fin,fout,bias", [((32, 250), 512, 512, True), ((8000,), 256, 1024, True),
                                                ((3, 7), 64, 9072, True), ((32, 250), 2048, 256, False)])
def test_fp32_matches_stock_linear(gpu_device, rows, fin, fout, bias):
    from seld_linear import SeldLinear
    torch.manual_seed(0)
    mine = SeldLinear(fin, fout, bias=bias).to(gpu_device)
    stock = nn.Linear(fin, fout, bias=bias).to(gpu_device)
    stock.load_state_dict(mine.state_dict())
    x = torch.randn(*rows, fin, device=gpu_device, requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    y, yr = mine(x), stock(xr)
    assert torch.allclose(y, yr, rtol=1e-5, atol=1e-5)
    go = torch.randn_like(y)
    y.backward(go)
    yr.backward(go)
    for a, b, name in [(x.grad, xr.grad, "dx"), (mine.weight.grad, stock.weight.grad, "dW")] + \
            ([(mine.bias.grad, stock.bias.grad, "db")] if bias else []):
        scale = b.abs().max().item() + 1e-12
        assert (a - b).abs().max().item() <= 2e-5 * scale, name


def test_bf16_autocast_tracks_fp32_reference(gpu_device):
    from seld_linear import SeldLinear, tall_product
    torch.manual_seed(1)
    mine = SeldLinear(512, 512).to(gpu_device)
    x = torch.randn(32, 250, 512, device=gpu_device, requires_grad=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = mine(x)
    assert y.dtype == torch.bfloat16
    go = torch.randn_like(y)
    y.backward(go)
    ref_w = go.float().reshape(-1, 512).t() @ x.detach().reshape(-1, 512)
    assert mine.weight.grad.dtype == torch.float32
    assert (mine.weight.grad - ref_w).abs().max().item() <= 2e-2 * ref_w.abs().max().item()
    # the chunked product itself, strided operands (row stride > width), odd shapes fall back to one GEMM
    a = torch.randn(8000, 2048, device=gpu_device)[:, :768]
    c = torch.randn(8000, 512, device=gpu_device)[:, :256]
    assert torch.allclose(tall_product(a, c), a.t() @ c, rtol=1e-4, atol=1e-2)
    a = torch.randn(77, 33, device=gpu_device)
    c = torch.randn(77, 5, device=gpu_device)
    assert torch.allclose(tall_product(a, c), a.t() @ c, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("rows,cols", [(8000, 256), (8000, 512), (8000, 9072), (37, 8), (1, 1024), (4099, 2048)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_column_sums_match_fp64(gpu_device, rows, cols, dtype):
    """csrc/glue.hip column sums (the bias gradient of every Linear): fp32 accumulation, deterministic, against a float64
    sum of the same (rounded) inputs; output in either dtype."""
    import seld_native
    torch.manual_seed(rows + cols)
    g = torch.randn(rows, cols, device=gpu_device).to(dtype)
    ref = g.double().sum(0)
    scale = g.double().abs().sum(0).clamp_min(1e-6)
    for out_dtype in (torch.float32, torch.bfloat16):
        out = torch.empty(cols, dtype=out_dtype, device=gpu_device)
        seld_native.column_sums(g, out)
        tol = 2e-6 if out_dtype == torch.float32 else 4e-3
        assert ((out.double() - ref).abs() / scale).max().item() <= tol if out_dtype == torch.float32 else \
            ((out.double() - ref).abs() <= 4e-3 * ref.abs() + 1e-2).all()
        again = torch.empty_like(out)
        seld_native.column_sums(g, again)
        assert torch.equal(out, again)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_qkv_projection_matches_three_linears(gpu_device, dtype):
    """MultiHeadSelfAttention (model_conformer.py:31-69) with its q / k / v projections as ONE GEMM on packed parameters
    (seld_pack.py) against the same module running three Linears: outputs and every gradient agree (same dot products,
    possibly another summation order inside the library GEMM)."""
    from model_conformer import MultiHeadSelfAttention
    torch.manual_seed(0)
    attn = MultiHeadSelfAttention(256, n_heads=4, dropout=0.0).to(gpu_device)
    attn.pack_parameters()
    x = torch.randn(8, 250, 256, device=gpu_device)
    results = {}
    for fused in (True, False):
        MultiHeadSelfAttention.fused_qkv = fused
        for p in attn.parameters():
            p.grad = None
        xin = x.clone().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = attn(xin)
        y.float().square().mean().backward()
        results[fused] = (y.detach().float(), xin.grad.float(), {n: p.grad.float().clone() for n, p in attn.named_parameters()})
    MultiHeadSelfAttention.fused_qkv = True
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
    (y1, dx1, g1), (y0, dx0, g0) = results[True], results[False]
    assert (y1 - y0).abs().max().item() <= tol * y0.abs().max().item()
    assert (dx1 - dx0).norm().item() <= tol * dx0.norm().item()
    for n in g0:
        assert (g1[n] - g0[n]).norm().item() <= tol * max(g0[n].norm().item(), 1e-12), n
