"""Side-stream weight gradients (seld_overlap.py): same numbers as the single-stream order, iteration after
iteration (a missed stream dependency shows up as a stale or half-written gradient), and the identity node really
runs after GRU layer 0's backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads(model, x, overlap, steps=1):
    import seld_overlap
    seld_overlap.enabled = overlap
    out = []
    for _ in range(steps):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(7)                                   # same dropout masks
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = model(x)
        (y.float() ** 2).mean().backward()
        out.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
    torch.cuda.synchronize()
    return out


@pytest.fixture()
def crnn():
    import trainer
    cfg = trainer.config
    was = (cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS, cfg.CRNN_RNN_HIDDEN, cfg.FUSED_GRU)
    cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS, cfg.CRNN_RNN_HIDDEN, cfg.FUSED_GRU = "crnn", [64, 128, 256, 512], 256, True
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), dev).train()
    yield model
    cfg.MODEL_TYPE, cfg.CRNN_CNN_CHANNELS, cfg.CRNN_RNN_HIDDEN, cfg.FUSED_GRU = was
    import seld_overlap
    seld_overlap.enabled = True


def test_overlapped_gradients_are_bit_identical(crnn):
    x = torch.randn(8, 250, 4, 64, device="cuda:0") * 20 - 30
    base, again = _grads(crnn, x, overlap=False, steps=2)
    # the library's convolution weight gradients accumulate with atomics: only parameters whose single-stream gradient
    # reproduces run to run can be held to bit equality (the head and the GRU -- everything the side stream touches)
    exact = [k for k in base if torch.equal(base[k], again[k])]
    assert all(k in exact for k in base if k.startswith(("fnn.", "rnn."))), exact
    runs = _grads(crnn, x, overlap=True, steps=12)
    for i, g in enumerate(runs):
        for k in base:
            if k in exact:
                assert torch.equal(g[k], base[k]), f"iteration {i}: {k} differs with the side stream"
            else:
                scale = base[k].float().abs().max().item() + 1e-12
                assert (g[k].float() - base[k].float()).abs().max().item() <= 2e-2 * scale, (i, k)


def test_side_stream_is_used_and_joined_late(crnn, monkeypatch):
    import seld_overlap
    events = []
    real_backward = seld_overlap._Deferred.backward
    real_launch = seld_overlap.launch_pending

    def spy_backward(ctx, *grads):
        events.append("join")
        return real_backward(ctx, *grads)

    def spy_launch(device):
        n = real_launch(device)
        if n:
            events.append(f"launch {n}")
        return n

    import seld_gru
    real_gru_backward = seld_gru.seld_native.gru_backward

    def spy_gru_backward(*args, **kwargs):
        events.append("recurrence")
        return real_gru_backward(*args, **kwargs)

    monkeypatch.setattr(seld_gru.seld_native, "gru_backward", spy_gru_backward)
    monkeypatch.setattr(seld_overlap._Deferred, "backward", staticmethod(spy_backward))
    monkeypatch.setattr(seld_overlap, "launch_pending", spy_launch)
    x = torch.randn(4, 250, 4, 64, device="cuda:0") * 20 - 30
    _grads(crnn, x, overlap=True)
    # the head's two Linears queue their weight gradients, which start beside layer 1's recurrence; layer 1's start
    # beside layer 0's recurrence; only then do the three identity nodes (2 Linears, GRU layer 1) join
    assert events == ["launch 2", "recurrence", "launch 1", "recurrence", "join", "join", "join"], events
    assert not seld_overlap._pending


def test_eval_and_no_grad_leave_no_alias_behind(crnn):
    x = torch.randn(2, 250, 4, 64, device="cuda:0")
    crnn.eval()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        crnn(x)
    assert "_deferred" not in crnn.fnn[0].__dict__ and "_deferred" not in crnn.fnn[4].__dict__
    crnn.train()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        crnn(x)
    assert "_deferred" not in crnn.fnn[0].__dict__ and "_deferred" not in crnn.fnn[4].__dict__


def test_stream_delay_holds_its_stream_and_only_that():
    import seld_native
    dev = torch.device("cuda:0")
    seld_native.ensure_init(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for ns in (25_000, 400_000):
        torch.cuda.synchronize()
        e0.record()
        seld_native.stream_delay(dev, ns)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3
        assert ns / 1e3 <= us + 1 and us < ns / 1e3 + 200, (ns, us)
    with pytest.raises(RuntimeError):
        seld_native.stream_delay(dev, 2_000_000)              # bounded: nothing can park a stream for long
