"""Weight gradients on side HIP streams, hidden under the BiGRU backward recurrences and the convolution backward.

The backward recurrence of a BiGRU layer (csrc/gru.hip) is a persistent kernel on 16 of the 256 CUs (one per
direction and 4-sequence tile at batch 32) for ~365 us, and the chain head -> layer 1 -> layer 0 is serial: the
other 240 CUs idle.  The WEIGHT gradients of the layers just before it on that chain (the head's Linears, then
layer 1's dW_ih / dW_hh) are not on the chain: nothing needs them before the optimiser (or DDP's bucket of that
parameter).  ``SeldLinear`` / ``seld_gru`` hand them to ``submit`` as jobs with preallocated outputs; the next
recurrence launch (``launch_pending``, called by ``seld_gru`` right before it enqueues the kernel) puts them on a side
stream that waits for everything the main stream had enqueued up to that point -- so they start WITH the recurrence
and fill the idle CUs instead of competing with the data-gradient GEMMs before it (measured: forking at the Linear's
own backward gained 2 %, forking at the recurrence launch 4 %).

Ordering without touching the autograd engine: ``defer(...)`` routes the parameters through an identity node that
is created EARLY in the forward pass (before GRU layer 0), so the engine -- highest sequence number first -- runs
its backward LATE: after layer 0's recurrence, before the convolution stack.  That node launches whatever is still
pending and makes the main stream wait for the side stream, so everything downstream of it (AccumulateGrad, DDP's
reducer hooks and their bucket copies / all-reduce, which then overlap the convolution backward) sees finished
gradients on the stream it expects.  If the engine ever ordered it differently the result would still be correct --
only the overlap would be lost.
"""
import os

import torch

import seld_native

enabled = os.environ.get("SELD_OVERLAP", "1") != "0"      # Config.OVERLAP_WEIGHT_GRADS via trainer.prepare_model_for_device
# Two side streams per device.  0: the ``submit`` jobs (head / GRU layer 1 weight gradients), which the ``_Deferred`` node
# joins right after GRU layer 0's recurrence.  1: the ``launch_now`` jobs (GRU layer 0's and the convolutions' weight
# gradients), joined by the stepper at the end of the backward pass.  With ONE stream (round 2) the _Deferred join also
# waited for layer 0's weight gradients, enqueued on the same stream a moment earlier: the convolution backward started
# ~140 us late (profiles/r02_iteration_timeline.txt, t = 1053 .. 1207 us).  SELD_SIDE_STREAMS=1 restores that for A/B runs.
two_streams = os.environ.get("SELD_SIDE_STREAMS", "2") != "1"
_streams = {}
_dirty = set()      # (device index, which): work enqueued on that side stream since the main stream last waited for it


def _index(device):
    if device.index is not None:
        return device.index
    return torch.cuda.current_device() if device.type == "cuda" else -1


def side_stream(device, which=0):
    key = (_index(device), which if two_streams else 0)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream(device=key[0])
    return _streams[key]


def _mark(device, which):
    _dirty.add((_index(device), which if two_streams else 0))


def _wait(device, which):
    """The main stream waits for side stream ``which`` if anything was enqueued there since the last wait (a wait on an
    idle stream would record an event outside a graph capture in flight)."""
    key = (_index(device), which if two_streams else 0)
    if key in _dirty:
        _dirty.discard(key)
        torch.cuda.current_stream(device).wait_stream(side_stream(device, which))


def active(x):
    return enabled and x.is_cuda and torch.is_grad_enabled()


# The recurrence's 16 workgroups each need a whole CU's LDS: if the GEMMs get to the CUs first the recurrence waits for
# a CU to drain (tools/bench_overlap.py: recurrence 383 us alone, 473 us beside a 139 us job, 440 us when the job starts
# 20 us late).  The main stream still has the dy layout converter (~12 us) to run before the recurrence launches.
head_start_ns = 25000

_pending = []       # (device, tensors, job): weight-gradient jobs waiting for the next recurrence launch (or the join)


def submit(device, tensors, job):
    """Queue ``job()`` (kernels reading / writing ``tensors``, all allocated on the main stream) for the side stream."""
    _pending.append((device, tensors, job))


def launch_pending(device):
    """Enqueue the queued jobs on the side stream, behind everything the main stream holds right now."""
    if not _pending:
        return 0
    jobs = list(_pending)
    del _pending[:]
    main = torch.cuda.current_stream(device)
    side = side_stream(device, 0)
    side.wait_stream(main)
    _mark(device, 0)
    with torch.cuda.stream(side):
        if head_start_ns:
            seld_native.stream_delay(device, head_start_ns)
        for _, _, job in jobs:
            job()
    for _, tensors, _ in jobs:
        for t in tensors:
            t.record_stream(side)               # main-stream allocations in use on the side stream
    return len(jobs)


# ---- convolution weight gradients beside the data-gradient chain --------------------------------------------------------
# Below GRU layer 0 the backward pass is the chain  tail(k) -> dgrad(k) -> tail(k-1) -> ...  with the weight gradient
# of block k hanging off tail(k): nothing needs it before the optimiser.  The tails are HBM-bound, the convolutions
# MFMA-bound, so a weight-gradient convolution on the side stream runs beside the next tail instead of after it.  Only the
# captured training step turns this on (``conv_wgrad_side``): it joins the side stream itself right after backward();
# under DistributedDataParallel the reducer's hooks read gradients when autograd delivers them, so the eager path keeps
# everything on one stream.
conv_wgrad_side = False
# A backward pass cut into stages (seld_cut.py, seld_graph.py): every stage is its own HIP graph, so a job launched in
# stage k has to be joined before stage k ends -- a weight gradient that becomes computable at the very end of a stage
# (GRU layer 0's, the last convolution's of the stage) would then run alone instead of beside the next stage.  While
# ``carry`` is set such jobs (``last_of_stage``) are kept and ``launch_carried`` starts them at the head of the NEXT stage; the gradient
# tensors they fill are not final until that stage's join (``carried_storages``: the stepper keeps those parameters out
# of the earlier stage's all-reduce bucket).
carry = False
_carried = []
# Developer A/B switch: "late" holds a convolution's weight gradient back until the NEXT convolution's
# backward, so that it runs beside that block's data-gradient convolution (both matrix-core bound) and the HBM-bound
# conv-tail kernels in between run alone; default: started as soon as it is computable (beside the next tail).
wgrad_order = os.environ.get("SELD_WGRAD_ORDER", "early")
_held = []


def release_held(device):
    """Start the weight-gradient jobs held back by ``wgrad_order = 'late'`` (called right before a data-gradient
    convolution is enqueued, and by ``join``)."""
    jobs = list(_held)
    del _held[:]
    for dev, tensors, job in jobs:
        _run_on_side(dev, tensors, job)
    return len(jobs)


def _run_on_side(device, tensors, job):
    main = torch.cuda.current_stream(device)
    side = side_stream(device, 1)
    side.wait_stream(main)
    _mark(device, 1)
    with torch.cuda.stream(side):
        job()
    for t in tensors:
        t.record_stream(side)


def launch_now(device, tensors, job, last_of_stage=False, outputs=(), hold=False):
    """Run ``job()`` on side stream 1 behind what the main stream holds now; ``join`` must follow before the results
    are read on the main stream.  ``last_of_stage``: the caller is the last node of a backward stage when a cut follows
    it (GRU layer 0; the convolution that consumes a ``seld_cut.boundary`` leaf) -- carried over while ``carry`` is set.
    ``outputs``: the gradient tensors the job fills (a subset of ``tensors``).  The caller must hand autograd VIEWS of
    them, not the tensors themselves: a carried job keeps its tensors referenced, and AccumulateGrad clones a gradient
    that is referenced elsewhere -- on the main stream, before the job has run (``carried_outputs`` lets the stepper
    check that every parameter gradient still aliases its job's output)."""
    if carry and last_of_stage:
        _carried.append((device, tensors, job, tuple(outputs)))
        return
    if wgrad_order == "late" and hold:
        _held.append((device, tensors, job))
        return
    _run_on_side(device, tensors, job)


def launch_carried(device):
    """Start the jobs the previous stage carried over (head of a stage, before its backward pass is enqueued)."""
    global carry
    jobs = list(_carried)
    del _carried[:]
    was, carry = carry, False
    try:
        for dev, tensors, job, _ in jobs:
            launch_now(dev, tensors, job)
    finally:
        carry = was
    return len(jobs)


def carried_storages():
    """Storage addresses of the tensors queued jobs will still write (their gradients are not final yet)."""
    return {t.untyped_storage().data_ptr() for _, tensors, _, _ in _carried for t in tensors}


def carried_outputs():
    """Storage addresses of the gradient tensors the queued jobs will fill."""
    return {t.untyped_storage().data_ptr() for _, _, _, outputs in _carried for t in outputs}


def join(device):
    """The main stream waits for everything queued on the side streams (pending jobs are launched first)."""
    launch_pending(device)
    release_held(device)
    _wait(device, 0)
    _wait(device, 1)


class _Deferred(torch.autograd.Function):
    @staticmethod
    def forward(ctx, *params):
        ctx.device = params[0].device
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        launch_pending(ctx.device)
        _wait(ctx.device, 0)
        return grads


def defer(*params):
    """Aliases of ``params`` whose gradients may be produced by ``submit``-ted jobs."""
    del _pending[:]           # jobs of a backward pass that was abandoned by an exception: their graph is gone
    del _carried[:]
    del _held[:]
    return _Deferred.apply(*params)


def defer_linear(*modules):
    """Give each ``SeldLinear`` a deferred alias of its parameters for the forward pass in flight."""
    for m in modules:
        ps = (m.weight,) if m.bias is None else (m.weight, m.bias)
        m.__dict__["_deferred"] = defer(*ps)
