"""One optimiser iteration as a replayed HIP graph (forward, loss, backward incl. the side-stream fork / join of
``seld_overlap``, gradient casts, fused Adam with a device-side step count, working-copy refresh).

Why: after the kernel work of round 1 an iteration of the CRNN is ~4.4 ms of GPU kernels but also ~3.5 ms of host time
to enqueue its ~300 launches (trainer.py:165-179 upstream is the same loop, one Python call per op) -- the step is
host-bound and GPU-bound at once, the Conformer models are plainly host-bound (55 % GPU-busy).  Every kernel of the
path takes its stream from the caller and nothing synchronises with the host, so the iteration is captured ONCE per
input shape and replayed: the host then enqueues two window gathers and one graph launch per iteration.

Data parallel (world > 1): the iteration is cut at its one exchange step,

    graph A  forward + loss + backward  ->  gradients gathered into ONE flat buffer per dtype
    eager    all-reduce of the flat buffers (RCCL over xGMI; gloo in the rehearsal mode)
    graph B  1/world scaling, gradient casts, fused Adam, working-copy refresh

so the collective itself is never captured (no dependence on a communicator's capture support) and the code path under
gloo on one GPU is the one RCCL runs on eight.  With one rank A and B are one graph.

Semantics are the eager loop's: the first ``WARMUP`` calls for a shape run eagerly (real training steps -- MIOpen /
hipBLASLt pick their kernels and allocate their workspaces there), the next call captures (capture executes nothing)
and replays.  Ragged last batches get their own graph.  The learning rate is a device scalar the schedulers' changes are
copied into, so a ReduceLROnPlateau step needs no re-capture."""
import logging
from contextlib import nullcontext

import torch
import torch.distributed as dist

logger = logging.getLogger("SMR_SELD")

WARMUP = 3


def _unwrap(model):
    return model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model


class FlatGradients:
    """One contiguous buffer per gradient dtype (bf16 working weights, fp32 everything else) for the data-parallel
    exchange: after the backward pass the gradients autograd produced are gathered into it by one multi-tensor copy per
    dtype, the buffers are all-reduced (one collective each) and ``p.grad`` is pointed at the buffer's views, which the
    optimiser then reads.  Only built for world > 1: a single rank hands autograd's tensors to the optimiser as they
    are (no accumulation, no copy)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        by_dtype = {}
        for p in self.params:
            by_dtype.setdefault(p.dtype, []).append(p)
        self.buffers, self.groups = [], []
        for dtype, group in by_dtype.items():
            total = sum((p.numel() + 7) // 8 * 8 for p in group)          # 16-byte aligned pieces (vector kernels)
            flat = torch.zeros(total, dtype=dtype, device=group[0].device)
            views, offset = [], 0
            for p in group:
                n = p.numel()
                view = flat[offset:offset + n]
                if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
                    # same memory order as the parameter (the optimiser's multi-tensor casts require it)
                    view = view.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
                else:
                    view = view.view(p.shape)
                views.append(view)
                offset += (n + 7) // 8 * 8
            self.buffers.append(flat)
            self.groups.append((group, views))

    def gather(self):
        """autograd's gradients -> the flat buffers; afterwards ``p.grad`` IS the buffer's view."""
        for group, views in self.groups:
            grads = []
            for p, v in zip(group, views):
                if p.grad is None:                    # a parameter the loss does not reach: contributes zeros
                    v.zero_()
                    grads.append(v)
                else:
                    grads.append(p.grad)
            torch._foreach_copy_(views, grads)
            for p, v in zip(group, views):
                p.grad = v

    def scale(self, factor):
        for flat in self.buffers:
            flat.mul_(factor)

    def all_reduce(self):
        for flat in self.buffers:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)


class GraphedTrainStep:
    """Callable ``(spectrograms, labels) -> (total, class_term)`` (detached device scalars, valid until the next call)
    with the semantics of ``trainer.train_step``."""

    def __init__(self, model, criterion, optimizer, device, world=1, autocast=None, use_graphs=True):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.device, self.world = device, world
        self.autocast = autocast if autocast is not None else nullcontext
        self.use_graphs = bool(use_graphs) and device.type == "cuda"
        import os
        self.wgrad_side = os.environ.get("SELD_WGRAD_SIDE", "1") != "0"     # developer switch for A/B runs
        self.params = [p for p in _unwrap(model).parameters() if p.requires_grad]
        self.flat = FlatGradients(self.params) if world > 1 else None
        self.shapes = {}             # key -> dict(calls, spec, labels, graph_a, graph_b, out)
        self.pool = None
        self.captured = self.replays = self.eager_calls = 0
        self.capture_error = None
        self._lr_tensors = []
        if self.use_graphs or any(isinstance(g["lr"], torch.Tensor) for g in optimizer.param_groups):
            for group in optimizer.param_groups:       # the learning rate as a device scalar (read by a capturable Adam)
                lr = group["lr"]
                t = lr if isinstance(lr, torch.Tensor) else torch.tensor(float(lr), dtype=torch.float32, device=device)
                group["lr"] = t
                self._lr_tensors.append(t)

    # ---- the two halves of an iteration (eager and captured runs execute exactly this code) ----------------------
    def _forward_backward(self, spec, labels):
        for p in self.params:                # autograd hands its gradient tensors over (no accumulate kernels); under
            p.grad = None                    # capture they land in the graph's pool and are rewritten by every replay
        with self.autocast():
            predictions = self.model(spec)
        total, term = self.criterion.loss_tensor(predictions, labels)
        side = self.wgrad_side and self.device.type == "cuda"
        if side:
            import seld_overlap
            seld_overlap.conv_wgrad_side = True
        try:
            total.backward()
        finally:
            if side:
                seld_overlap.conv_wgrad_side = False
                seld_overlap.join(self.device)           # weight gradients produced on the side stream are complete
        if self.flat is not None:
            self.flat.gather()
        return total.detach(), term.detach()

    def _update(self):
        if self.world > 1:
            self.flat.scale(1.0 / self.world)
        self.optimizer.step()

    def _sync_lr(self):
        """A scheduler replaced the device scalar by a Python float (ReduceLROnPlateau assigns param_group['lr']):
        copy the value into the scalar the captured Adam reads."""
        for group, t in zip(self.optimizer.param_groups, self._lr_tensors):
            if group["lr"] is not t:
                t.fill_(float(group["lr"]))
                group["lr"] = t

    # ---- call ---------------------------------------------------------------------------------------------------
    def __call__(self, spec, labels):
        self._sync_lr()
        if not self.use_graphs:
            return self._eager(spec, labels)
        key = (tuple(spec.shape), spec.dtype, tuple(labels.shape), labels.dtype)
        st = self.shapes.get(key)
        if st is None:
            st = self.shapes[key] = {"calls": 0, "graph_a": None}
        st["calls"] += 1
        if st["graph_a"] is None:
            if st["calls"] <= WARMUP or self.capture_error is not None:
                return self._eager(spec, labels)
            try:
                self._capture(st, spec, labels)
            except Exception as exc:                 # noqa: BLE001  -- a library call refused capture: stay eager, say so
                self.capture_error = f"{type(exc).__name__}: {exc}"
                logger.warning(f"HIP graph capture of the training step failed, running eagerly: {self.capture_error}")
                torch.cuda.synchronize(self.device)
                return self._eager(spec, labels)
        if spec.data_ptr() != st["spec"].data_ptr():           # a feed that gathers straight into the static
            st["spec"].copy_(spec, non_blocking=True)           # buffers (static_inputs) skips these two copies
        if labels.data_ptr() != st["labels"].data_ptr():
            st["labels"].copy_(labels, non_blocking=True)
        st["graph_a"].replay()
        if st["graph_b"] is not None:
            self.flat.all_reduce()
            st["graph_b"].replay()
        self.replays += 1
        return st["out"]

    def _eager(self, spec, labels):
        self.eager_calls += 1
        out = self._forward_backward(spec, labels)
        if self.world > 1:
            self.flat.all_reduce()
        self._update()
        return out

    def _capture(self, st, spec, labels):
        st["spec"], st["labels"] = spec.clone(), labels.clone()
        torch.cuda.synchronize(self.device)
        graph_a = torch.cuda.CUDAGraph()
        kwargs = {} if self.pool is None else {"pool": self.pool}
        with torch.cuda.graph(graph_a, **kwargs):
            out = self._forward_backward(st["spec"], st["labels"])
            if self.world <= 1:
                self._update()
        if self.pool is None:
            self.pool = graph_a.pool()
        graph_b = None
        if self.world > 1:
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b, pool=self.pool):
                self._update()
        st["graph_a"], st["graph_b"], st["out"] = graph_a, graph_b, out
        self.captured += 1

    def static_inputs(self, spec_shape, spec_dtype, labels_shape, labels_dtype):
        """The captured graph's input buffers for a batch of this shape, or None while that shape still runs eagerly:
        a feed may gather its batch directly into them (``seld_native.gather_windows(..., out=)``)."""
        st = self.shapes.get((tuple(spec_shape), spec_dtype, tuple(labels_shape), labels_dtype))
        if st is None or st.get("graph_a") is None:
            return None
        return st["spec"], st["labels"]

    def stats(self):
        return {"graphs": self.captured, "replays": self.replays, "eager_iterations": self.eager_calls,
                "capture_error": self.capture_error}

    def close(self):
        """Drop the graphs and hand the parameters ordinary (absent) gradients again."""
        self.shapes.clear()
        for p in self.params:
            p.grad = None
