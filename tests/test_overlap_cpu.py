"""Control flow of seld_overlap.py without a GPU: the identity node created early in the forward pass runs late in
the backward pass, queued jobs are launched by the next ``launch_pending`` (or by the join), nothing is left behind.
Streams are replaced by recorders; the GPU behaviour itself is covered by tests/test_overlap_gpu.py."""
import contextlib

import pytest
import torch

import seld_overlap


class _FakeStream:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def wait_stream(self, other):
        self.log.append(f"{self.name} waits for {other.name}")


@pytest.fixture()
def streams(monkeypatch):
    log = []
    main, side = _FakeStream("main", log), _FakeStream("side", log)
    monkeypatch.setattr(torch.cuda, "current_stream", lambda device=None: main)
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())
    monkeypatch.setattr(seld_overlap, "side_stream", lambda device, which=0: side)
    monkeypatch.setattr(seld_overlap, "head_start_ns", 0)           # the delay kernel needs the HIP library
    del seld_overlap._pending[:]
    del seld_overlap._carried[:]
    seld_overlap._dirty.clear()
    yield log
    del seld_overlap._pending[:]
    del seld_overlap._carried[:]
    seld_overlap._dirty.clear()


class _Consumer(torch.autograd.Function):
    """Stands for a SeldLinear: its weight gradient is produced by a queued job into a preallocated tensor."""

    @staticmethod
    def forward(ctx, x, w, log):
        ctx.save_for_backward(x, w)
        ctx.log = log
        return x @ w.t()

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        ctx.log.append("consumer backward")
        dw = torch.empty_like(w)

        def job():
            ctx.log.append("job runs")
            dw.copy_(g.t() @ x)

        seld_overlap.submit(g.device, [], job)
        return g @ w, dw, None


class _Recurrence(torch.autograd.Function):
    """Stands for a BiGRU layer: launches what is pending right before its own kernel."""

    @staticmethod
    def forward(ctx, x, log):
        ctx.log = log
        return x * 2

    @staticmethod
    def backward(ctx, g):
        seld_overlap.launch_pending(g.device)
        ctx.log.append("recurrence backward")
        return g * 2, None


def test_identity_node_joins_after_the_layers_created_later(streams):
    log = streams
    torch.manual_seed(0)
    w = torch.randn(5, 3, requires_grad=True)
    x = torch.randn(4, 3, requires_grad=True)
    (alias,) = seld_overlap.defer(w)                      # created BEFORE the recurrence node, like GRU layer 0's
    h = _Recurrence.apply(x, log)
    y = _Consumer.apply(h, alias, log)
    y.sum().backward()
    assert log == ["consumer backward",                   # queues its weight gradient ...
                   "side waits for main", "job runs",     # ... which the recurrence launches beside itself
                   "recurrence backward",
                   "main waits for side"]                 # the identity node: last, right before AccumulateGrad
    ref_w = torch.ones(4, 5).t() @ (x.detach() * 2)
    assert torch.allclose(w.grad, ref_w) and not seld_overlap._pending


def test_join_launches_what_no_recurrence_picked_up(streams):
    log = streams
    w = torch.randn(5, 3, requires_grad=True)
    x = torch.randn(4, 3)
    (alias,) = seld_overlap.defer(w)
    _Consumer.apply(x, alias, log).sum().backward()       # no recurrence below the consumer
    assert log == ["consumer backward", "side waits for main", "job runs", "main waits for side"]
    assert torch.allclose(w.grad, torch.ones(4, 5).t() @ x) and not seld_overlap._pending


def test_jobs_of_an_abandoned_backward_are_dropped_by_the_next_forward(streams):
    seld_overlap.submit(torch.device("cpu"), [], lambda: (_ for _ in ()).throw(AssertionError("stale job ran")))
    w = torch.randn(2, 2, requires_grad=True)
    seld_overlap.defer(w)
    assert not seld_overlap._pending


def test_linear_alias_lives_for_one_forward_pass_only(streams):
    from seld_linear import SeldLinear
    lin = SeldLinear(3, 5)
    seld_overlap.defer_linear(lin)
    assert "_deferred" in lin.__dict__
    y = lin(torch.randn(2, 3))                             # CPU tensor: stock nn.Linear, but the alias is consumed
    assert "_deferred" not in lin.__dict__ and tuple(y.shape) == (2, 5)
    assert set(lin.state_dict()) == {"weight", "bias"}     # the alias never shows up as a parameter or buffer


def test_carried_jobs_wait_for_the_next_stage(streams):
    """A backward pass cut into stages (seld_cut.py): while ``carry`` is set a ``launch_now`` job is kept, its tensors
    are reported as unfinished, and ``launch_carried`` starts it on side stream 1 at the head of the next stage."""
    log = streams
    out = torch.zeros(3)
    seld_overlap.carry = True
    try:
        seld_overlap.launch_now(torch.device("cpu"), [out], lambda: (log.append("carried job runs"), out.fill_(1.0)),
                                last_of_stage=True, outputs=[out])
    finally:
        seld_overlap.carry = False
    assert log == [] and out.untyped_storage().data_ptr() in seld_overlap.carried_storages()
    assert seld_overlap.carried_outputs() == {out.untyped_storage().data_ptr()}
    seld_overlap.join(torch.device("cpu"))                  # the stage's own join has nothing to wait for
    assert log == []
    out.record_stream = lambda stream: None                 # CPU tensor standing in for a device allocation
    assert seld_overlap.launch_carried(torch.device("cpu")) == 1
    assert log == ["side waits for main", "carried job runs"] and not seld_overlap.carried_storages()
    seld_overlap.join(torch.device("cpu"))
    assert log[-1] == "main waits for side" and out.sum() == 3
