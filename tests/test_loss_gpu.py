"""GPU parity for the fused softmax+MSE loss (loss.py:43-54) against values produced by the
reference's own loss.py (tests/golden/loss_golden.npz) and against plain torch fp32."""
import numpy as np
import pytest
import torch

from oracle import labels as olab

pytestmark = pytest.mark.gpu


def test_matches_reference_loss_golden(gpu_device, golden_dir):
    import seld_native
    z = np.load(golden_dir / "loss_golden.npz")
    logits = torch.from_numpy(z["logits"]).to(gpu_device)
    labels = torch.from_numpy(z["labels"]).to(gpu_device)
    n = logits.numel()
    loss, grad = seld_native.softmax_mse(logits, labels, grad_scale=2.0 / n)
    ref = float(z["mse"])
    assert abs(loss.item() - ref) <= 1e-5 * abs(ref)                       # bar: <= 1e-3 rel
    g = grad.cpu().numpy()
    assert np.abs(g - z["mse_grad"]).max() <= 1e-3 * np.abs(z["mse_grad"]).max()
    assert np.allclose(g, z["mse_grad"], rtol=2e-3, atol=1e-11)
    loss2, none = seld_native.softmax_mse(logits, labels)
    assert none is None and loss2.item() == loss.item()                     # deterministic


def test_mask_labels_equal_dense_labels(gpu_device):
    import seld_native
    g = torch.Generator().manual_seed(3)
    B, T = 3, 250
    logits = (torch.randn(B, T, 648, 14, generator=g) * 3).to(gpu_device)
    mask_np = olab.metadata_to_mask(olab.synth_metadata(5, 150), 750 * 480)[:750].reshape(B, T, 648)
    mask = torch.from_numpy(mask_np).to(gpu_device)
    dense = torch.from_numpy(olab.mask_to_dense(mask_np)).to(gpu_device)
    scale = 2.0 / logits.numel()
    l1, g1 = seld_native.softmax_mse(logits, mask, grad_scale=scale)
    l2, g2 = seld_native.softmax_mse(logits, dense, grad_scale=scale)
    assert l1.item() == l2.item() and torch.equal(g1, g2)
    lg = logits.clone().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(torch.softmax(lg, -1), dense)
    ref.backward()
    assert abs(l1.item() - ref.item()) <= 1e-5 * ref.item()
    assert (g1 - lg.grad).abs().max().item() <= 1e-3 * lg.grad.abs().max().item()


def test_bf16_logits_and_ragged_tail(gpu_device):
    import seld_native
    g = torch.Generator().manual_seed(4)
    for n_cells in (1, 255, 257, 648 * 7 + 2):
        logits = (torch.randn(n_cells, 14, generator=g) * 2).to(gpu_device)
        mask = torch.randint(0, 1 << 13, (n_cells,), generator=g).to(torch.uint16).to(gpu_device)
        dense = seld_native.expand_labels(mask) if n_cells % 2 == 0 else \
            torch.from_numpy(olab.mask_to_dense(mask.cpu().numpy())).to(gpu_device)
        for dt in (torch.float32, torch.bfloat16):
            lg = logits.to(dt)
            loss, grad = seld_native.softmax_mse(lg, mask, grad_scale=1.0)
            lref = lg.float().clone().requires_grad_(True)
            ref = ((torch.softmax(lref, -1) - dense) ** 2).sum() / (2.0)   # d/dz of (1/2)*sum -> matches scale 1
            ref.backward()
            mean_ref = ((torch.softmax(lref, -1) - dense) ** 2).mean()
            assert abs(loss.item() - mean_ref.item()) <= 1e-5 * mean_ref.item() + 1e-9
            assert grad.dtype == dt
            tol = 1e-5 if dt == torch.float32 else 8e-3
            assert (grad.float() - lref.grad).abs().max().item() <= tol * max(1.0, lref.grad.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_module_backward_with_and_without_upstream_scale(gpu_device, dtype):
    """SMRSELDLoss through autograd: loss.backward() (upstream gradient exactly 1: the scaling kernel must leave the
    gradient untouched) and (0.37 * loss).backward() (scaled in place by the device scalar)."""
    import loss as loss_mod
    g = torch.Generator().manual_seed(9)
    logits = (torch.randn(2, 30, 648, 14, generator=g) * 2).to(gpu_device).to(dtype)
    mask = torch.randint(0, 1 << 13, (2, 30, 648), generator=g).to(torch.uint16).to(gpu_device)
    dense = loss_mod.mask_to_dense(mask, 14)
    crit = loss_mod.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
    for factor in (1.0, 0.37):
        a = logits.clone().requires_grad_(True)
        total, _ = crit.loss_tensor(a, mask)
        (total * factor if factor != 1.0 else total).backward()
        b = logits.float().clone().requires_grad_(True)
        ref = torch.nn.functional.mse_loss(torch.softmax(b, -1), dense) * factor
        ref.backward()
        tol = 1e-4 if dtype == torch.float32 else 1e-2
        assert (a.grad.float() - b.grad).abs().max().item() <= tol * b.grad.abs().max().item(), factor


@pytest.mark.parametrize("n_cells", [1, 257, 2048 * 256 + 1, 324000, 648000])
def test_every_variant_at_ragged_sizes(gpu_device, n_cells):
    """All eight instantiations (bf16 / fp32 logits x mask / dense labels x value-only / value + gradient) on sizes
    whose last tile is ragged, against plain torch.  Regression test for a hipcc miscompile (64-bit min() of the
    tile extent lowered to v_cmp + s_cselect without the VCC->SCC copy) that made the value-only bf16 kernels --
    the ones evaluation uses -- treat the ragged tile as a full one."""
    import seld_native
    import loss as loss_mod
    g = torch.Generator().manual_seed(n_cells)
    logits = (torch.randn(n_cells, 14, generator=g) * 2).to(gpu_device)
    mask = ((torch.rand(n_cells, generator=g) < 0.05).to(torch.int32) * (1 << 3)).to(torch.uint16).to(gpu_device)
    dense = loss_mod.mask_to_dense(mask, 14)
    # poison whatever lies behind the tensors: an over-read must show
    for dtype in (torch.bfloat16, torch.float32):
        lg = logits.to(dtype)
        ref = torch.nn.functional.mse_loss(torch.softmax(lg.float(), -1), dense).item()
        for labels in (mask, dense):
            value, none = seld_native.softmax_mse(lg, labels)
            value2, grad = seld_native.softmax_mse(lg, labels, grad_scale=1.0)
            assert none is None and grad.shape == lg.shape
            assert abs(value.item() - ref) <= 1e-5 * ref, (dtype, labels.dtype, value.item(), ref)
            assert abs(value2.item() - ref) <= 1e-5 * ref, (dtype, labels.dtype, value2.item(), ref)
