#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run ONCE in the build container).

Two kinds of fixtures are written:

1. REFERENCE-GENERATED (pin the oracle and the host mirror to the real reference):
   the torch/numpy-only reference modules (``utils.py``, ``loss.py``, ``model_crnn.py``,
   ``model_conformer.py``, ``resnet50_model.py``) are copied to a scratch dir under /tmp
   (never imported in place: ``config.Config()`` mkdirs next to its own file,
   config.py:99-102) and imported from there.  Only inputs/outputs (arrays) are stored --
   no reference source or bytecode.
2. ORACLE-GENERATED (value parity unpinned by the reference, see oracle/features.py):
   small log-mel vectors from the torch restatement, so the GPU box -- which has no
   /root/reference -- compares against committed numbers as well as the live oracle.

Usage:  python tests/golden/make_golden.py [--reference /root/reference]
"""
from __future__ import annotations

import argparse
import os
import shutil
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.dont_write_bytecode = True


def sd_to_npz(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def import_reference(ref_dir: Path):
    scratch = Path(tempfile.mkdtemp(prefix="seld_ref_"))
    for f in ("utils.py", "loss.py", "model_crnn.py", "model_conformer.py", "resnet50_model.py"):
        shutil.copy(ref_dir / f, scratch / f)
    sys.path.insert(0, str(scratch))
    import importlib
    mods = {m: importlib.import_module(m) for m in
            ("utils", "loss", "model_crnn", "model_conformer", "resnet50_model")}
    sys.path.remove(str(scratch))
    return mods, scratch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    args = ap.parse_args()
    mods, scratch = import_reference(Path(args.reference))
    torch.set_num_threads(4)

    # ---- 1a. polar_to_grid over every integer direction (utils.py:77-90)
    az = np.arange(-180, 181)
    el = np.arange(-90, 91)
    ii = np.zeros((az.size, el.size), dtype=np.int16)
    jj = np.zeros_like(ii)
    for a_i, a in enumerate(az):
        for e_i, e in enumerate(el):
            i, j = mods["utils"].polar_to_grid(int(a), int(e), I=18, J=36)
            ii[a_i, e_i], jj[a_i, e_i] = i, j
    np.savez_compressed(HERE / "polar_grid.npz", az=az, el=el, i=ii, j=jj)

    # ---- 1b. loss.py known answers on seeded tensors (loss.py:27-54,56-146,149-172)
    g = torch.Generator().manual_seed(7)
    B, T, I, J, M = 2, 6, 18, 36, 14
    logits = torch.randn(B, T, I * J, M, generator=g) * 2.0
    cls = torch.randint(0, M, (B, T, I * J), generator=g)
    sparse = torch.rand(B, T, I * J, generator=g) < 0.97
    cls[sparse] = M - 1
    cls[0, 0] = M - 1                                     # one frame with no events at all
    y = torch.nn.functional.one_hot(cls, M).float()
    y[1, 2, 5, 3] = 1.0                                   # a multi-hot cell (dataset.py:110)
    Loss = mods["loss"].SMRSELDLoss
    w = torch.ones(M)
    w[M - 1] = 0.05                                       # trainer.py:99-100
    out = {"logits": logits.numpy(), "labels": y.numpy(), "class_weights": w.numpy()}
    lg = logits.clone().requires_grad_(True)
    l_mse, bd = Loss(loss_type="mse", w_class=1.0, grid_size=(I, J), class_weights=w)(lg, y)
    l_mse.backward()
    out["mse"] = np.float64(l_mse.item())
    out["mse_breakdown"] = np.float64(bd["class_mse"])
    out["mse_grad"] = lg.grad.numpy().copy()
    lg = logits.clone().requires_grad_(True)
    l_ce, _ = Loss(loss_type="ce", w_class=1.0, grid_size=(I, J), class_weights=w)(lg, y)
    l_ce.backward()
    out["ce_weighted"] = np.float64(l_ce.item())
    out["ce_weighted_grad"] = lg.grad.numpy().copy()
    l_ce_u, _ = Loss(loss_type="ce", w_class=1.0, grid_size=(I, J))(logits, y)
    out["ce_unweighted"] = np.float64(l_ce_u.item())
    crit = Loss(loss_type="mse", grid_size=(I, J))
    probs = torch.softmax(logits, -1)
    out["aiur_on_probs"] = np.float64(crit.aiur_loss(probs, y).item())
    out["cl_on_probs"] = np.float64(crit.converging_localization_loss(probs, y).item())
    np.savez_compressed(HERE / "loss_golden.npz", **out)

    # ---- 1c. small-config models: full state_dict + input + eval-mode logits
    def run_model(name, ctor, x, seed):
        torch.manual_seed(seed)
        m = ctor()
        # perturb BatchNorm running stats so eval() exercises them
        with torch.no_grad():
            for mod in m.modules():
                if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                    mod.running_mean.uniform_(-0.2, 0.2)
                    mod.running_var.uniform_(0.5, 1.5)
        m.eval()
        with torch.no_grad():
            y_eval = m(x)
        blob = {"sd::" + k: v for k, v in sd_to_npz(m.state_dict()).items()}
        blob["x"] = x.numpy()
        blob["logits_eval"] = y_eval.numpy()
        blob["n_params"] = np.int64(sum(p.numel() for p in m.parameters()))
        np.savez_compressed(HERE / f"{name}.npz", **blob)
        return m

    gx = torch.Generator().manual_seed(11)
    x_small = torch.randn(2, 12, 4, 64, generator=gx) * 20.0 - 30.0   # dB-like range
    run_model("crnn_small",
              lambda: mods["model_crnn"].SELD_CRNN(n_channels=4, n_mels=64, grid_size=(3, 4), num_classes=14,
                                                    cnn_channels=[4, 8, 8, 16], rnn_hidden=8, rnn_layers=2,
                                                    dropout=0.3), x_small, 0)
    run_model("conformer_small",
              lambda: mods["model_conformer"].SELD_Conformer(n_channels=4, n_mels=64, grid_size=(3, 4),
                                                             num_classes=14, cnn_channels=[4, 8, 8, 16],
                                                             conf_d_model=16, conf_n_heads=4, conf_n_layers=2,
                                                             conf_kernel_size=7, dropout=0.3), x_small, 1)

    # ---- 1d. full-size models: param counts, key lists, seeded-init logits on a tiny input
    full = {}
    x_tiny = torch.randn(1, 4, 4, 64, generator=gx) * 20.0 - 30.0
    for name, ctor in (
        ("crnn", lambda: mods["model_crnn"].SELD_CRNN()),
        ("conformer", lambda: mods["model_conformer"].SELD_Conformer()),
        ("resnet_conformer", lambda: mods["resnet50_model"].SELD_ResNet50_Conformer()),
    ):
        torch.manual_seed(1234)
        m = ctor().eval()
        with torch.no_grad():
            y = m(x_tiny)
        sd = m.state_dict()
        full[f"{name}::n_params"] = np.int64(sum(p.numel() for p in m.parameters()))
        full[f"{name}::keys"] = np.array(list(sd.keys()))
        full[f"{name}::shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
        full[f"{name}::abs_sum"] = np.array([float(v.double().abs().sum()) for v in sd.values()])
        full[f"{name}::logits_tiny"] = y[:, :, ::37, :].numpy()       # subsample cells to stay small
    full["x_tiny"] = x_tiny.numpy()
    np.savez_compressed(HERE / "models_full.npz", **full)

    # ---- 2. oracle-generated log-mel vectors (value parity unpinned by the reference)
    from oracle import features as F
    L = 24000 + 123                                        # 1 s + ragged tail, 4 ch
    pcm = F.synth_pcm(0, 4, L, "noise")
    np.savez_compressed(HERE / "logmel_noise_1s.npz", pcm_i16=F.pcm_to_int16(pcm).numpy(),
                        logmel_from_i16=F.logmel_torch(F.int16_to_pcm(F.pcm_to_int16(pcm))).numpy())

    shutil.rmtree(scratch, ignore_errors=True)
    for f in sorted(HERE.glob("*.npz")):
        print(f"{f.name:28s} {f.stat().st_size / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
