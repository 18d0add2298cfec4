"""The staged (cut) training step of seld_graph.py on the CPU: a backward pass cut at ``seld_cut.boundary`` points
gives the SAME numbers as the uncut one (same kernels, same order), single process and with two gloo ranks whose
gradient buckets are all-reduced asynchronously between the stages.  The GPU / HIP-graph side of the same code is
covered by tests/test_graph_gpu.py and tests/test_ddp_gpu.py."""
import os
import sys
from pathlib import Path

import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "sound-event-localization-detection_amd"


class _Net(torch.nn.Module):
    """Three blocks with two cut points between them (what model_crnn.run_cnn_blocks / resnet50_model mark)."""

    def __init__(self):
        super().__init__()
        import seld_cut
        self.cut = seld_cut.boundary
        self.a = torch.nn.Linear(12, 64)
        self.b = torch.nn.Linear(64, 48)
        self.norm = torch.nn.LayerNorm(48)
        self.c = torch.nn.Linear(48, 6)
        self.unused = torch.nn.Parameter(torch.zeros(5))          # a parameter the loss never reaches: zero gradient

    def forward(self, x):
        y = self.cut(torch.tanh(self.a(x)))
        y = self.cut(self.norm(torch.relu(self.b(y))))
        return self.c(y)


class _Criterion:
    @staticmethod
    def loss_tensor(pred, target):
        loss = torch.nn.functional.mse_loss(pred, target)
        return loss, loss


def _run(world=1, rank=0, staged=True, reduce_dtype="param", iters=12, late_bytes=None):
    import seld_graph
    torch.manual_seed(0)
    model = _Net()
    if late_bytes is not None:
        seld_graph.FlatGradients.LATE_FP32_BYTES = late_bytes
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    step = seld_graph.GraphedTrainStep(model, _Criterion(), opt, torch.device("cpu"), world=world, use_graphs=False,
                                       overlap_allreduce=staged, reduce_dtype=reduce_dtype, split=world == 1)
    g = torch.Generator().manual_seed(100 + rank)
    losses = []
    for _ in range(iters):
        x, y = torch.randn(16, 12, generator=g), torch.randn(16, 6, generator=g)
        total, _ = step(x, y)
        losses.append(float(total))
    flat = torch.cat([p.detach().flatten() for p in model.parameters()])
    return losses, flat, step.stats()


def test_cut_backward_is_bit_identical_single_process():
    sys.path[:0] = [p for p in (str(ROOT), str(PKG)) if p not in sys.path]
    torch.manual_seed(0)
    ref_model = _Net()
    opt = torch.optim.Adam(ref_model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(100)
    ref_losses = []
    for _ in range(12):                                             # the plain loop (trainer.py:165-179 upstream)
        x, y = torch.randn(16, 12, generator=g), torch.randn(16, 6, generator=g)
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(ref_model(x), y)
        loss.backward()
        ref_model.unused.grad = torch.zeros(5)                      # the flat buffer hands Adam zeros for it
        opt.step()
        ref_losses.append(float(loss))
    ref = torch.cat([p.detach().flatten() for p in ref_model.parameters()])
    for staged in (True, False):
        for wire in ("param", "fp32"):
            for late in (1 << 20, 0):                               # fp32 gradients deferred to the last bucket / not
                losses, flat, stats = _run(staged=staged, reduce_dtype=wire, late_bytes=late)
                assert losses == ref_losses and torch.equal(flat, ref), (staged, wire, late)
                assert stats["backward_stages"] == (3 if staged else 1)
                assert sum(b["bytes"] for b in stats["gradient_buckets"]) >= 4 * flat.numel()
    import seld_graph
    seld_graph.FlatGradients.LATE_FP32_BYTES = 1 << 20
    _, _, stats = _run(staged=True, late_bytes=0)
    sizes = [b["bytes"] for b in stats["gradient_buckets"]]
    assert sizes[0] == 4 * (48 * 6 + 6 + 2) and sizes[1] >= 4 * (64 * 48 + 48 + 48 + 48) and sizes[2] >= 4 * (12 * 64 + 64)


def _rank(rank, world, port, queue):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path[:0] = [p for p in (str(ROOT), str(PKG)) if p not in sys.path]
    torch.set_num_threads(1)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = {}
    for staged in (True, False):
        for wire in ("param", "fp32"):
            losses, flat, stats = _run(world, rank, staged, wire, late_bytes=0 if staged else None)
            out[(staged, wire)] = (losses, flat.tolist(), stats["allreduce_overlap"], len(stats["gradient_buckets"]))
    queue.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_gloo_ranks_overlapped_exchange_equals_the_blocking_one():
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7
    procs = [ctx.Process(target=_rank, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    (r0, a), (r1, b) = sorted((queue.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    base = a[(False, "param")]
    assert base[2] is False and base[3] == 1
    for key in a:
        assert a[key][1] == b[key][1], key                            # replicas agree
        assert a[key][1] == base[1], key                              # staged / wire dtype change nothing (fp32 model)
        assert a[key][0] == base[0]
        assert a[key][0] != b[key][0]                                 # the ranks really saw different batches
    assert a[(True, "param")][2] is True and a[(True, "param")][3] == 3


def test_four_gloo_ranks_stay_in_sync_under_the_overlapped_exchange():
    """Four ranks: more than one hop in the collective.  Replicas agree bit for bit in every mode; staged and blocking
    exchanges agree to rounding (their buffers are cut differently, so a ring may add the four contributions in another
    order -- with two ranks the sum is commutative and the test above demands equality)."""
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 311
    procs = [ctx.Process(target=_rank, args=(r, 4, port, queue)) for r in range(4)]
    for p in procs:
        p.start()
    results = [out for _, out in sorted((queue.get(timeout=300) for _ in procs), key=lambda t: t[0])]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    base = results[0][(False, "param")]
    for key in results[0]:
        for other in results[1:]:
            assert results[0][key][1] == other[key][1], key               # replicas agree
        assert torch.allclose(torch.tensor(results[0][key][1]), torch.tensor(base[1]), rtol=1e-5, atol=1e-6), key
        assert all(abs(x - y) <= 1e-5 * abs(y) for x, y in zip(results[0][key][0], base[0])), key
    assert len({tuple(r[(True, "param")][0]) for r in results}) == 4      # four different batches
    assert results[0][(True, "param")][2] is True and results[0][(True, "param")][3] == 3
