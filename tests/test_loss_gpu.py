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


# ---- the three-term loss of smrl_seld_gaussian.py:946-1072 (csrc/loss3.hip) ----------------------------------------

def _three_term_reference(logits, dense, grid, w):
    """The reference's own composition, restated with this package's loss.py (its AIUR / CL methods are pinned to the
    reference's loss.py by tests/golden/loss_golden.npz: `aiur_on_probs`, `cl_on_probs`)."""
    import loss as loss_mod
    crit = loss_mod.SMRSELDLoss("mse", *w, grid_size=grid)
    lg = logits.detach().float().clone().requires_grad_(True)
    probs = torch.softmax(lg, -1)
    mse = torch.nn.functional.mse_loss(probs, dense)
    aiur = crit.aiur_loss(probs, dense)
    cl = crit.converging_localization_loss(probs, dense)
    total = w[0] * mse + w[1] * aiur + w[2] * cl
    total.backward()
    return torch.stack((total, mse, aiur, cl)).detach(), lg.grad


def test_three_term_matches_reference_loss_golden(gpu_device, golden_dir):
    import seld_native
    z = np.load(golden_dir / "loss_golden.npz")
    logits = torch.from_numpy(z["logits"]).to(gpu_device)
    dense = torch.from_numpy(z["labels"]).to(gpu_device)
    w = (1.0, 0.5, 0.25)
    terms, grad = seld_native.smr_loss(logits, dense, (18, 36), *w, want_grad=True)
    mse, aiur, cl = float(z["mse"]), float(z["aiur_on_probs"]), float(z["cl_on_probs"])      # the reference's numbers
    got = terms.cpu().tolist()
    assert abs(got[1] - mse) <= 1e-5 * mse and abs(got[2] - aiur) <= 1e-6 and abs(got[3] - cl) <= 1e-5 * abs(cl) + 1e-9
    assert abs(got[0] - (w[0] * mse + w[1] * aiur + w[2] * cl)) <= 1e-6
    ref_terms, ref_grad = _three_term_reference(logits, dense, (18, 36), w)
    assert (grad - ref_grad).abs().max().item() <= 1e-4 * ref_grad.abs().max().item()
    # the class-term part of the gradient is the reference's own (mse_grad), the rest is the CL term
    only_mse, g_mse = seld_native.smr_loss(logits, dense, (18, 36), 1.0, 0.0, 0.0, want_grad=True)
    assert np.abs(g_mse.cpu().numpy() - z["mse_grad"]).max() <= 1e-3 * np.abs(z["mse_grad"]).max()
    # deterministic, and the compact mask gives the same numbers as the dense labels it encodes
    again, grad2 = seld_native.smr_loss(logits, dense, (18, 36), *w, want_grad=True)
    assert torch.equal(again, terms) and torch.equal(grad2, grad)
    bits = (dense == 1.0).to(torch.int32) * (1 << torch.arange(14, device=gpu_device, dtype=torch.int32))
    mask = bits.sum(-1)
    mask = torch.where(mask == (1 << 13), torch.zeros_like(mask), mask).to(torch.uint16)       # background rule: mask 0
    t3, g3 = seld_native.smr_loss(logits, mask, (18, 36), *w, want_grad=True)
    assert torch.equal(t3, terms) and torch.equal(g3, grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("frames,grid", [(1, (18, 36)), (7, (18, 36)), (500, (18, 36)), (33, (5, 7)), (4, (3, 3))])
def test_three_term_every_variant(gpu_device, dtype, frames, grid):
    """fp32 / bf16 logits x mask / dense labels x value-only / value + gradient against torch, with frames that have no
    event at all (IoU 1, no CL contribution), multi-hot cells, and grids small enough that a 256-cell tile spans many
    frames (the per-frame event counts then go straight to global counters)."""
    import seld_native
    import loss as loss_mod
    g = torch.Generator().manual_seed(frames * 100 + grid[0])
    cells = grid[0] * grid[1]
    logits = (torch.randn(1, frames, cells, 14, generator=g) * 2).to(gpu_device).to(dtype)        # [B, T, G, M]
    cls = torch.randint(0, 13, (frames, cells), generator=g).to(torch.int32)
    mask = torch.where(torch.rand(frames, cells, generator=g) < 0.06, torch.ones(1, dtype=torch.int32) << cls,
                       torch.zeros(1, dtype=torch.int32))
    mask = mask | torch.where(torch.rand(frames, cells, generator=g) < 0.01, torch.full((1,), 1 << 5, dtype=torch.int32),
                              torch.zeros(1, dtype=torch.int32))                 # a few second classes in a cell
    mask[::3] = 0                                                               # every third frame has no events
    mask = mask.to(torch.uint16).to(gpu_device).unsqueeze(0)
    dense = loss_mod.mask_to_dense(mask, 14)
    w = (1.0, 0.7, 1.3)
    ref_terms, ref_grad = _three_term_reference(logits, dense, grid, w)
    for labels in (mask, dense):
        terms, none = seld_native.smr_loss(logits, labels, grid, *w, want_grad=False)
        terms2, grad = seld_native.smr_loss(logits, labels, grid, *w, want_grad=True)
        assert none is None and torch.equal(terms, terms2) and grad.dtype == dtype
        assert (terms - ref_terms).abs().max().item() <= 2e-5 * max(1.0, ref_terms.abs().max().item()), (terms, ref_terms)
        tol = 1e-4 if dtype == torch.float32 else 8e-3
        assert (grad.float() - ref_grad).abs().max().item() <= tol * ref_grad.abs().max().item()


def test_three_term_module_and_config5_training_step(gpu_device):
    """BASELINE configs[4] on one GPU: ResNet50-Conformer + Gaussian label augmentation + the three-term loss under bf16,
    through the captured training step; the breakdown carries the reference's three keys."""
    import dataset
    import loss as loss_mod
    import trainer
    from oracle import features as ofeat
    from oracle import labels as olab
    cfg = trainer.config
    saved = (cfg.MODEL_TYPE, cfg.SEED)
    cfg.MODEL_TYPE, cfg.SEED = "resnet_conformer", 3
    try:
        clips = [ofeat.synth_pcm(i, 4, 24000 * 8, "noise") for i in range(2)]
        rows = [olab.synth_metadata(i, meta_frames=80) for i in range(2)]
        ds = dataset.SELDDataset.from_pcm(clips, rows, device=gpu_device, use_gaussian_augmentation=True)
        plain = dataset.SELDDataset.from_pcm(clips, rows, device=gpu_device, use_gaussian_augmentation=False)
        assert int((ds.mask_tm != 0).sum()) > 2 * int((plain.mask_tm != 0).sum())          # boxes, not single cells
        torch.manual_seed(0)
        model = trainer.prepare_model_for_device(trainer.build_model((ds.I, ds.J)), gpu_device).train()
        trainer.enable_master_weights(model, gpu_device)
        crit = loss_mod.SMRSELDLoss("mse", 1.0, 1.0, 1.0, grid_size=(ds.I, ds.J), three_term=True)
        opt = trainer.make_optimizer(model, 1e-3, gpu_device, capturable=trainer.graph_step_enabled(gpu_device))
        step = trainer.make_stepper(model, crit, opt, gpu_device)
        spec, mask = ds.device_batch([0, 3])
        losses = []
        for _ in range(6):
            total, term = step(spec, mask)
            losses.append(total.item())
        assert all(np.isfinite(losses)) and losses[-1] < losses[0]
        t = crit.last_terms.cpu().tolist()
        assert abs(t[0] - (t[1] + t[2] + t[3])) <= 1e-5 and 0 < t[1] < 1 and 0 <= t[2] <= 1
        with torch.no_grad(), trainer.autocast_context(gpu_device):
            total, breakdown = crit(model.eval()(spec), mask)
        assert set(breakdown) == {"class_mse", "aiur", "cl"}
        if hasattr(step, "close"):
            step.close()
    finally:
        cfg.MODEL_TYPE, cfg.SEED = saved
