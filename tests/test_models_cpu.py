"""CPU tests: the host-side model / loss mirrors against golden vectors produced by the reference's own
modules (tests/golden/make_golden.py imports model_crnn.py, model_conformer.py, resnet50_model.py, loss.py
from a scratch copy of the reference).  Bar (north_star): <= 1e-3 rel on logits; observed ~1e-6."""
import numpy as np
import pytest
import torch


def _load_sd(model, z):
    sd = {k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd::")}
    model.load_state_dict(sd, strict=True)       # key names / shapes are part of the contract


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_crnn_small_matches_reference_logits(golden_dir):
    import model_crnn
    z = np.load(golden_dir / "crnn_small.npz")
    m = model_crnn.SELD_CRNN(n_channels=4, n_mels=64, grid_size=(3, 4), num_classes=14, cnn_channels=[4, 8, 8, 16],
                             rnn_hidden=8, rnn_layers=2, dropout=0.3)
    _load_sd(m, z)
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(z["x"]))
    assert tuple(y.shape) == (2, 12, 12, 14)
    assert _rel(y.numpy(), z["logits_eval"]) <= 1e-3
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"])


def test_conformer_small_matches_reference_logits(golden_dir):
    import model_conformer
    z = np.load(golden_dir / "conformer_small.npz")
    m = model_conformer.SELD_Conformer(n_channels=4, n_mels=64, grid_size=(3, 4), num_classes=14,
                                       cnn_channels=[4, 8, 8, 16], conf_d_model=16, conf_n_heads=4, conf_n_layers=2,
                                       conf_kernel_size=7, dropout=0.3)
    _load_sd(m, z)
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(z["x"]))
    assert _rel(y.numpy(), z["logits_eval"]) <= 1e-3


@pytest.mark.parametrize("name,n_params", [("crnn", 11194864), ("conformer", 9909488), ("resnet_conformer", 59666928)])
def test_full_size_models_match_reference(golden_dir, name, n_params):
    """Same state_dict keys and shapes, same parameter count, same seeded initialisation (the modules are
    constructed in the reference's order, so torch.manual_seed gives identical weights) and logits."""
    import model_conformer
    import model_crnn
    import resnet50_model
    ctor = {"crnn": model_crnn.SELD_CRNN, "conformer": model_conformer.SELD_Conformer,
            "resnet_conformer": resnet50_model.SELD_ResNet50_Conformer}[name]
    z = np.load(golden_dir / "models_full.npz")
    torch.manual_seed(1234)
    m = ctor().eval()
    sd = m.state_dict()
    assert list(sd.keys()) == list(z[f"{name}::keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(z[f"{name}::shapes"])
    assert sum(p.numel() for p in m.parameters()) == n_params == int(z[f"{name}::n_params"])
    assert np.allclose([float(v.double().abs().sum()) for v in sd.values()], z[f"{name}::abs_sum"], rtol=1e-9)
    with torch.no_grad():
        y = m(torch.from_numpy(z["x_tiny"]))
    assert tuple(y.shape) == (1, 4, 648, 14)                     # verify_dims.py contract: [B,T,C,F] -> [B,T,648,14]
    assert _rel(y[:, :, ::37, :].numpy(), z[f"{name}::logits_tiny"]) <= 1e-3


def test_cspdarknet_shape_contract():
    import model
    m = model.SMRSELDWithCSPDarkNet().eval()
    assert sum(p.numel() for p in m.parameters()) == 8105806     # current reference model.py, use_small=True
    with torch.no_grad():
        assert tuple(m(torch.randn(1, 2, 4, 64)).shape) == (1, 2, 648, 14)


def test_loss_matches_reference_values(golden_dir):
    import loss
    z = np.load(golden_dir / "loss_golden.npz")
    logits, y, w = (torch.from_numpy(z[k]) for k in ("logits", "labels", "class_weights"))
    total, breakdown = loss.SMRSELDLoss("mse", 1.0, grid_size=(18, 36), class_weights=w)(logits, y)
    assert abs(total.item() - float(z["mse"])) <= 1e-6 and set(breakdown) == {"class_mse"}
    assert abs(breakdown["class_mse"] - float(z["mse_breakdown"])) <= 1e-6
    ce, bd = loss.SMRSELDLoss("ce", 1.0, grid_size=(18, 36), class_weights=w)(logits, y)
    assert abs(ce.item() - float(z["ce_weighted"])) <= 1e-5 and set(bd) == {"class_ce"}
    ce_u, _ = loss.SMRSELDLoss("ce", 1.0, grid_size=(18, 36))(logits, y)
    assert abs(ce_u.item() - float(z["ce_unweighted"])) <= 1e-5
    crit = loss.SMRSELDLoss("mse", grid_size=(18, 36))
    p = torch.softmax(logits, -1)
    assert abs(crit.aiur_loss(p, y).item() - float(z["aiur_on_probs"])) <= 1e-6
    assert abs(crit.converging_localization_loss(p, y).item() - float(z["cl_on_probs"])) <= 1e-7


def test_loss_accepts_compact_mask_labels():
    import loss
    g = torch.Generator().manual_seed(0)
    mask = (torch.rand(2, 3, 648, generator=g) < 0.02).to(torch.int32) * (1 << 5)
    mask[0, 0, 3] = (1 << 2) | (1 << 7)
    mask = mask.to(torch.uint16)
    dense = loss.mask_to_dense(mask, 14)
    assert dense.shape == (2, 3, 648, 14) and dense[0, 0, 3].tolist() == [0, 0, 1, 0, 0, 0, 0, 1] + [0] * 6
    assert (dense[..., 13] == (mask == 0)).all()
    logits = torch.randn(2, 3, 648, 14, generator=g)
    crit = loss.SMRSELDLoss("mse", grid_size=(18, 36))
    assert crit(logits, mask)[0].item() == crit(logits, dense)[0].item()


def test_run_head_on_cpu_is_the_stock_sequential():
    """seld_layernorm.run_head only rewrites the head on a ROCm device; on the CPU (BASELINE configs[0]) it must be
    the Sequential itself, also for a head that does not have the five-module shape."""
    import torch
    import torch.nn as nn
    import seld_layernorm
    from seld_linear import SeldLinear
    torch.manual_seed(0)
    head = nn.Sequential(SeldLinear(16, 512), nn.LayerNorm(512), nn.ReLU(), nn.Dropout(0.3), SeldLinear(512, 10)).eval()
    x = torch.randn(3, 7, 16)
    assert torch.equal(seld_layernorm.run_head(head, x), head(x))
    assert not seld_layernorm.applicable(head[1], head[0](x))
    odd = nn.Sequential(nn.Linear(16, 4), nn.Tanh())
    assert torch.equal(seld_layernorm.run_head(odd, x), odd(x))
