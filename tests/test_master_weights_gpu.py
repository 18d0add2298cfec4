"""GPU test of the fp32-master / bf16-working-weight training mode (trainer.MasterWeightAdam): same numbers as the
autocast path it replaces (the bf16 working copies ARE the values autocast casts to every iteration), checkpoints
keep fp32 parameters under the reference's keys, and the model handed back is plain fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(master, device, steps=6):
    import trainer
    cfg = trainer.config
    saved = (cfg.MASTER_WEIGHTS, cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS)
    cfg.MASTER_WEIGHTS, cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS = master, "crnn", [16, 16, 32, 32]
    try:
        torch.manual_seed(0)
        model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), device).train()
        active = trainer.enable_master_weights(model, device)
        assert active == master
        crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
        opt = trainer.make_optimizer(model, 1e-3, device)
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(8, 250, 4, 64, generator=g) * 20 - 30).to(device)
        mask = (torch.rand(8, 250, 648, generator=g) < 0.01).to(torch.int32).mul(1 << 4).to(torch.uint16).to(device)
        losses = []
        for _ in range(steps):
            total, _ = trainer.train_step(model, crit, opt, x, mask, device)
            losses.append(total.item())
        sd = trainer.model_state_dict(model)
        trainer.disable_master_weights(model)
        return losses, sd, model
    finally:
        cfg.MASTER_WEIGHTS, cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS = saved


def test_master_weights_track_the_autocast_path(gpu_device):
    from seld_rnn import SeldGRU
    la, sda, ma = _run(False, gpu_device)
    lb, sdb, mb = _run(True, gpu_device)
    assert np.allclose(la, lb, rtol=2e-3), (la, lb)             # same arithmetic, different kernel order / rounding
    assert la[-1] < la[0] and lb[-1] < lb[0]
    assert list(sda.keys()) == list(sdb.keys())
    for k in sda:
        assert sdb[k].dtype == sda[k].dtype, k                   # checkpoints hold fp32 parameters
        if sda[k].dtype == torch.float32 and sda[k].numel() > 1:
            # six Adam steps move every parameter by at most ~6 * lr; the two runs may disagree on the sign of a
            # near-zero gradient component, so compare on that scale
            assert (sda[k] - sdb[k]).abs().max().item() <= 4e-3 + 1e-2 * sda[k].abs().max().item(), k
    assert all(p.dtype == torch.float32 for p in mb.parameters())   # handed back as a plain fp32 model
    assert not hasattr(mb, "_seld_master_weights")
    gru = [m for m in mb.modules() if isinstance(m, SeldGRU)][0]
    assert gru.weight_ih_l0.dtype == torch.float32


def test_multi_cast_matches_torch(gpu_device):
    import seld_native
    g = torch.Generator().manual_seed(0)
    shapes = [(3,), (64, 4, 3, 3), (1536, 2048), (7, 5), (9072, 512), (1,)] * 20          # 120 tensors: two launches
    src = [torch.randn(*s, generator=g).to(gpu_device) for s in shapes]
    src[1] = src[1].contiguous(memory_format=torch.channels_last)
    low = [torch.empty_like(s, dtype=torch.bfloat16) for s in src]
    assert seld_native.multi_cast(src, low)
    for a, b in zip(src, low):
        assert torch.equal(a.to(torch.bfloat16), b)
    back = [torch.empty_like(s) for s in src]
    assert seld_native.multi_cast(low, back)
    for a, b in zip(low, back):
        assert torch.equal(a.float(), b)
    # layout mismatch: refused, nothing written
    bad = torch.zeros(64, 4, 3, 3, device=gpu_device, dtype=torch.bfloat16)             # NCHW vs channels-last source
    assert not seld_native.multi_cast([src[1]], [bad]) and bad.abs().sum().item() == 0


def test_multi_cast_descriptor_cache_follows_the_addresses(gpu_device):
    import seld_native
    g = torch.Generator().manual_seed(1)
    src = [torch.randn(33, 7, generator=g).to(gpu_device), torch.randn(5, generator=g).to(gpu_device)]
    low = [torch.empty_like(s, dtype=torch.bfloat16) for s in src]
    cache = {}
    for _ in range(3):                                                   # miss, then hits
        src[0].add_(1.0)
        assert seld_native.multi_cast(src, low, cache)
        assert all(torch.equal(a.to(torch.bfloat16), b) for a, b in zip(src, low))
    first_key = cache["key"]
    src[1] = torch.randn(5, generator=g).to(gpu_device)                  # a new tensor at a new address: re-validated
    assert seld_native.multi_cast(src, low, cache) and cache["key"] != first_key
    assert all(torch.equal(a.to(torch.bfloat16), b) for a, b in zip(src, low))
    src[1] = torch.randn(5, generator=g).to(gpu_device).double()        # wrong dtype: refused even with a cache
    assert not seld_native.multi_cast(src, low, cache)


def test_own_adam_kernel_matches_the_framework_path(gpu_device):
    """trainer.MasterWeightAdam with the one-launch update of csrc/adam.hip (bf16 gradients read directly, bf16 working
    copies written in the same pass) against the same optimiser running cast + torch's fused Adam + cast, and against a
    plain float64 statement of Adam with L2 weight decay (trainer.py:112-116 upstream): 8 steps on tensors of ragged and
    channels-last shapes with a learning-rate change in between; state_dict in the framework's format."""
    import trainer
    torch.manual_seed(3)
    shapes = [(9072, 512), (768, 256), (37,), (5, 3, 3, 3), (64, 4, 3, 3), (1,)]

    def build(own):
        g = torch.Generator().manual_seed(5)
        low, masters, others = [], [], []
        for i, shape in enumerate(shapes):
            t = torch.randn(*shape, generator=g).to(gpu_device)
            if len(shape) == 4:
                t = t.contiguous(memory_format=torch.channels_last)
            if i % 2 == 0:                                  # a bf16 working weight with its fp32 master
                p = torch.nn.Parameter(t.to(torch.bfloat16))
                low.append(p)
                masters.append(t.clone())
            else:                                           # a plain fp32 parameter
                others.append(torch.nn.Parameter(t.clone()))
        opt = trainer.MasterWeightAdam(low, masters, others, lr=torch.tensor(1e-2, device=gpu_device), weight_decay=1e-4,
                                       fused=True, capturable=True)
        opt.own_kernel = own
        return opt

    a, b = build(True), build(False)
    ref = [t.detach().double().clone() for t in a._masters + [p.data for p in a._others]]
    ref_m = [torch.zeros_like(t) for t in ref]
    ref_v = [torch.zeros_like(t) for t in ref]
    g = torch.Generator().manual_seed(9)
    lr = 1e-2
    for step in range(1, 9):
        if step == 5:
            lr = 2.5e-3
            for opt in (a, b):
                opt.param_groups[0]["lr"].fill_(lr)
        grads = [torch.randn(*p.shape, generator=g) * 0.1 for p in a._low + a._others]
        for opt in (a, b):
            for p, gr in zip(opt._low + opt._others, grads):
                gr = gr.to(gpu_device).to(p.dtype)
                if p.dim() == 4:
                    gr = gr.contiguous(memory_format=torch.channels_last)
                p.grad = gr
            opt.step()
        for i, (p, gr) in enumerate(zip(a._low + a._others, grads)):
            gq = gr.to(gpu_device).to(p.dtype).double() + 1e-4 * ref[i]
            ref_m[i] = 0.9 * ref_m[i] + 0.1 * gq
            ref_v[i] = 0.999 * ref_v[i] + 0.001 * gq * gq
            ref[i] = ref[i] - (lr / (1 - 0.9 ** step)) * ref_m[i] / (ref_v[i].sqrt() / (1 - 0.999 ** step) ** 0.5 + 1e-8)
    assert a.own_steps == 8 and b.own_steps == 0
    mine = a._masters + [p.data for p in a._others]
    theirs = b._masters + [p.data for p in b._others]
    for i, (x, y, r) in enumerate(zip(mine, theirs, ref)):
        scale = r.abs().max().item() + 1e-6
        assert (x.double() - r).abs().max().item() <= 2e-6 * scale, i
        assert (x - y).abs().max().item() <= 2e-6 * scale, i
    for pa, m in zip(a._low, a._masters):                   # working copies = the masters rounded to bf16
        assert torch.equal(pa.data, m.to(torch.bfloat16))
    sa, sb = a.state_dict(), b.state_dict()
    assert sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert set(sa["state"][k]) == {"step", "exp_avg", "exp_avg_sq"} and float(sa["state"][k]["step"]) == 8.0
        assert (sa["state"][k]["exp_avg"] - sb["state"][k]["exp_avg"]).abs().max().item() <= 1e-6
