"""One optimiser iteration as replayed HIP graphs (forward, loss, backward incl. the side-stream fork / join of
``seld_overlap``, gradient casts, fused Adam with a device-side step count, working-copy refresh).

Why: after the kernel work of round 1 an iteration of the CRNN is ~4.4 ms of GPU kernels but also ~3.5 ms of host time
to enqueue its ~300 launches (trainer.py:165-179 upstream is the same loop, one Python call per op) -- the step is
host-bound and GPU-bound at once, the Conformer models are plainly host-bound (55 % GPU-busy).  Every kernel of the
path takes its stream from the caller and nothing synchronises with the host, so the iteration is captured ONCE per
input shape and replayed: the host then enqueues two window gathers and one graph launch per iteration.

Data parallel (world > 1): the gradient all-reduce is OVERLAPPED WITH THE BACKWARD PASS.  The models mark cut points
(``seld_cut.boundary``: the CRNN / Conformer between the recurrent / attention part and the convolution stack, the CRNN
also before the last convolution block; the ResNet50-Conformer after its encoder and before ``layer4``), the backward pass is
captured as one graph per STAGE, and the gradients that are final at the end of a stage travel while the next stages run:

    graph A0   forward + loss + backward down to the last cut   -> bucket 0 gathered into its flat buffer
    eager      all-reduce(bucket 0) launched ASYNCHRONOUSLY (RCCL's own stream; gloo's worker thread in the rehearsal)
    graph A1   backward of the next stage (+ the weight gradients the previous stage carried over) -> bucket 1
    eager      all-reduce(bucket 1) async ... ; after the last stage: wait for every collective (a stream dependency
               under RCCL, no host block)
    graph B    1/world scaling, gradient casts, fused Adam, working-copy refresh

CRNN at batch 32: bucket 0 = head + GRU layer 1 + GRU biases (12.2 MB bf16) under the whole convolution backward
(~1.2 ms), bucket 1 = GRU layer 0's weights (7.1 MB, carried over: they are computed beside the last block's data
gradient) under blocks 2..0, bucket 2 = the convolution weights + every fp32 parameter (3.1 MB) exposed.  The
collectives themselves are never captured (no dependence on a communicator's capture support), so the code path under
gloo on one GPU is the one RCCL runs on eight.  With one rank everything is one graph and nothing is cut.

Wire dtype (``reduce_dtype``): "param" sums the bf16 working-weight gradients in bf16 (half the xGMI bytes; the
summation error of an 8-rank ring is bounded in tests/test_ddp_gpu.py::test_bf16_gradient_sum_error_of_eight_ranks),
"fp32" casts them into fp32 flat buffers that ARE the fp32 masters' gradients (one cast either way), as an autocast port
of the reference would reduce them.

Semantics are the eager loop's: the first ``WARMUP`` calls for a shape run eagerly (real training steps -- MIOpen /
hipBLASLt pick their kernels and allocate their workspaces there), the next call captures (capture executes nothing)
and replays.  Ragged last batches get their own graphs.  The learning rate is a device scalar the schedulers' changes are
copied into, so a ReduceLROnPlateau step needs no re-capture."""
import logging
import os
import time
from contextlib import nullcontext

import torch
import torch.distributed as dist

import seld_cut

logger = logging.getLogger("SMR_SELD")

WARMUP = 3
_retired = []        # the graphs of the last two closed steppers (see GraphedTrainStep.close)


def _unwrap(model):
    return model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model


def _flat_views(flat, group):
    """Views of ``flat`` shaped (and, for channels-last 4-D parameters, laid out) like the tensors of ``group``; every
    piece starts on a 16-byte boundary (vector kernels)."""
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
    return views


class FlatGradients:
    """Stage-ordered flat gradient buffers for the data-parallel exchange.

    ``add_bucket(params)`` is called once per backward stage (first iteration, eager) with the parameters whose
    gradients are final at the end of that stage; ``gather(k)`` copies autograd's gradients of bucket k into its flat
    buffer(s) with one multi-tensor copy per buffer and points ``p.grad`` (or, with an fp32 wire, the fp32 master's
    ``.grad``) at the buffer's views, which the optimiser then reads; ``all_reduce(k)`` is one collective per buffer.

    Wire dtype "param": one buffer per bucket and parameter dtype; under bf16 training the few fp32 gradients of a stage
    (norm parameters, GRU / convolution biases: ~40 KB in all, less than ``LATE_FP32_BYTES`` per stage) are not worth a
    collective of their own: those of ALL buckets share one buffer that travels with the last bucket, so an iteration
    costs (stages + 1) collectives.  Wire dtype "fp32": one fp32 buffer per bucket."""

    LATE_FP32_BYTES = 1 << 20

    def __init__(self, params, reduce_dtype="param", optimizer=None):
        self.params = [p for p in params if p.requires_grad]
        self.fp32_wire = reduce_dtype == "fp32"
        self.optimizer = optimizer
        self._master_of = {}
        if self.fp32_wire and optimizer is not None and hasattr(optimizer, "_low"):
            self._master_of = {id(p): m for p, m in zip(optimizer._low, optimizer._masters)}
        self.buckets = []            # per stage: list of groups {flat, params, views, cast}
        self._late_fp32 = []         # fp32 parameters of earlier buckets (wire "param"): travel with the last bucket
        self._cast_caches = {}
        self.final = False

    # ---- construction (first iteration) --------------------------------------------------------------------------
    def add_bucket(self, params, last):
        assert not self.final
        params = list(params)
        groups = []
        if self.fp32_wire:
            if params:
                groups.append(self._group(params, torch.float32))
        else:
            low = [p for p in params if p.dtype != torch.float32]
            full = [p for p in params if p.dtype == torch.float32]
            if low:
                by_dtype = {}
                for p in low:
                    by_dtype.setdefault(p.dtype, []).append(p)
                groups += [self._group(g, dt) for dt, g in by_dtype.items()]
            if last:
                full = self._late_fp32 + full
                self._late_fp32 = []
            elif sum(p.numel() * 4 for p in full) < self.LATE_FP32_BYTES:
                self._late_fp32 += full
                full = []
            if full:
                groups.append(self._group(full, torch.float32))
        self.buckets.append(groups)
        if last:
            self.final = True
            if self.fp32_wire and self._master_of and self.optimizer is not None:
                self.optimizer.external_master_grads = True        # the flat views ARE the masters' gradients

    def _group(self, params, dtype):
        total = sum((p.numel() + 7) // 8 * 8 for p in params)
        flat = torch.zeros(total, dtype=dtype, device=params[0].device)
        like = [self._master_of.get(id(p), p) for p in params] if self.fp32_wire else params
        views = _flat_views(flat, like)
        if self.fp32_wire:
            for p, v in zip(params, views):
                m = self._master_of.get(id(p))
                if m is not None:
                    m.grad = v
        return {"flat": flat, "params": params, "views": views,
                "cast": self.fp32_wire and any(p.dtype != torch.float32 for p in params)}

    # ---- per iteration ------------------------------------------------------------------------------------------
    def gather(self, k):
        """autograd's gradients of bucket k -> its flat buffers."""
        for gi, group in enumerate(self.buckets[k]):
            params, views = group["params"], group["views"]
            same_src, same_dst, cast_src, cast_dst = [], [], [], []
            for p, v in zip(params, views):
                if p.grad is None:                    # a parameter the loss does not reach: contributes zeros
                    v.zero_()
                elif p.grad.dtype == v.dtype:
                    if p.grad.data_ptr() != v.data_ptr():
                        same_src.append(p.grad)
                        same_dst.append(v)
                else:
                    cast_src.append(p.grad)
                    cast_dst.append(v)
            if same_src:
                torch._foreach_copy_(same_dst, same_src)
            if cast_src:
                import seld_native
                cache = self._cast_caches.setdefault((k, gi), {})
                if not (cast_src[0].is_cuda and seld_native.multi_cast(cast_src, cast_dst, cache)):
                    for s, d in zip(cast_src, cast_dst):
                        d.copy_(s)
            for p, v in zip(params, views):
                if v.dtype == p.dtype:
                    p.grad = v
                elif id(p) not in self._master_of:    # no master to hand the fp32 gradient to: cast it back in place
                    p.grad = v.to(p.dtype)

    def scale(self, factor):
        for groups in self.buckets:
            for group in groups:
                group["flat"].mul_(factor)

    def all_reduce(self, k, async_op=False):
        works = []
        for group in self.buckets[k]:
            w = dist.all_reduce(group["flat"], op=dist.ReduceOp.SUM, async_op=async_op)
            if async_op and w is not None:
                works.append(w)
        return works

    def describe(self):
        return [{"stage": k, "bytes": sum(g["flat"].numel() * g["flat"].element_size() for g in groups),
                 "buffers": [{"dtype": str(g["flat"].dtype).replace("torch.", ""),
                              "bytes": g["flat"].numel() * g["flat"].element_size(), "tensors": len(g["params"])}
                             for g in groups]}
                for k, groups in enumerate(self.buckets)]


class GraphedTrainStep:
    """Callable ``(spectrograms, labels) -> (total, class_term)`` (detached device scalars, valid until the next call)
    with the semantics of ``trainer.train_step``.

    ``overlap_allreduce``: cut the backward pass at the model's ``seld_cut.boundary`` points (world > 1; or ``split``
    to force the staged capture on one rank, which is how bench.py measures the windows that hide the collectives).
    ``reduce_dtype``: "param" | "fp32" (see the module docstring)."""

    def __init__(self, model, criterion, optimizer, device, world=1, autocast=None, use_graphs=True,
                 overlap_allreduce=True, reduce_dtype="param", split=False):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.device, self.world = device, world
        self.autocast = autocast if autocast is not None else nullcontext
        self.use_graphs = bool(use_graphs) and device.type == "cuda"
        self.wgrad_side = os.environ.get("SELD_WGRAD_SIDE", "1") != "0"     # developer switch for A/B runs
        self.batch_reductions = os.environ.get("SELD_BATCH_REDUCTIONS", "1") != "0"      # developer switch for A/B runs
        self.params = [p for p in _unwrap(model).parameters() if p.requires_grad]
        self.exchange = world > 1 or split                       # flat buffers and separate graphs
        self.staged = self.exchange and bool(overlap_allreduce) and os.environ.get("SELD_OVERLAP_ALLREDUCE", "1") != "0"
        self.flat = FlatGradients(self.params, reduce_dtype, optimizer) if self.exchange else None
        self.shapes = {}             # key -> dict(calls, spec, labels, graphs, graph_b, out)
        self.pool = None
        self.captured = self.replays = self.eager_calls = 0
        self.capture_error = None
        self.timing = False          # record HIP events around every graph replay (bench.py: the overlap windows)
        self._segments = []
        self._cuts = []
        self._lr_tensors = []
        if self.use_graphs or any(isinstance(g["lr"], torch.Tensor) for g in optimizer.param_groups):
            for group in optimizer.param_groups:       # the learning rate as a device scalar (read by a capturable Adam)
                lr = group["lr"]
                t = lr if isinstance(lr, torch.Tensor) else torch.tensor(float(lr), dtype=torch.float32, device=device)
                group["lr"] = t
                self._lr_tensors.append(t)

    # ---- the pieces of an iteration (eager and captured runs execute exactly this code) ---------------------------
    def _stage(self, k, spec, labels):
        """Stage 0: forward + loss + backward down to the last cut; stage k > 0: the backward pass between two cuts.
        Ends with every side stream joined and the stage's gradients in their flat buffer.  Returns True after the
        last stage."""
        side = self.wgrad_side and self.device.type == "cuda"
        import seld_overlap
        if k == 0:
            for p in self.params:            # autograd hands its gradient tensors over (no accumulate kernels); under
                p.grad = None                # capture they land in the graph's pool and are rewritten by every replay
            with seld_cut.recording() if self.staged else nullcontext() as rec:
                with self.autocast():
                    predictions = self.model(spec)
            self._cuts = list(rec.cuts) if self.staged else []
            total, term = self.criterion.loss_tensor(predictions, labels)
            self._out = (total.detach(), term.detach())
            root, grad = total, None
            if total.is_cuda and total.dim() == 0 and total.dtype == torch.float32:
                import seld_native
                grad = seld_native.unit_gradient(total.device)     # no fill kernel, and the fused loss skips its x 1
        else:
            outer, leaf = self._cuts[len(self._cuts) - k]
            root, grad = outer, leaf.grad
            leaf.grad = None
        last = k == len(self._cuts)
        if side:
            seld_overlap.conv_wgrad_side = True
            seld_overlap.carry = not last
        batching = self.device.type == "cuda" and self.batch_reductions
        if batching:
            import seld_linear
            seld_linear.begin_batch()        # the Linears' chunk / column sums: one multi-tensor launch each, below
        try:
            if k > 0 and side:
                seld_overlap.launch_carried(self.device)      # weight gradients the previous stage handed over
            torch.autograd.backward(root, grad)
        except BaseException:
            if batching:
                seld_linear.abandon_batch()
            raise
        finally:
            if side:
                seld_overlap.conv_wgrad_side = False
                seld_overlap.carry = False
                seld_overlap.join(self.device)           # weight gradients produced on the side streams are complete
        if batching:
            seld_linear.flush_batch()
        if last:
            self._cuts = []
        if self.flat is not None:
            if not self.flat.final:
                self._discover_bucket(last)
            self.flat.gather(k)
        return last

    def _discover_bucket(self, last):
        """First iteration: the parameters whose gradients this stage completed (everything left, after the last)."""
        import seld_overlap
        taken = {id(p) for groups in self.flat.buckets for g in groups for p in g["params"]} | \
            {id(p) for p in self.flat._late_fp32}
        rest = [p for p in self.params if id(p) not in taken]
        if last:
            ready = rest
        else:
            unfinished = seld_overlap.carried_storages()
            ready = [p for p in rest if p.grad is not None
                     and p.grad.untyped_storage().data_ptr() not in unfinished]
            # every gradient a carried job will fill must still be aliased by a parameter's .grad: had autograd cloned
            # it (it does when the tensor is referenced elsewhere) the clone would hold whatever was in memory
            aliased = {p.grad.untyped_storage().data_ptr() for p in rest if p.grad is not None}
            lost = seld_overlap.carried_outputs() - aliased
            if lost:
                raise RuntimeError(f"{len(lost)} weight gradient(s) handed over to the next backward stage are no longer "
                                   f"aliased by a parameter's .grad (cloned by autograd before the side-stream job ran)")
        self.flat.add_bucket(ready, last)

    def _update(self):
        if self.world > 1:
            if hasattr(self.optimizer, "seld_grad_scale"):          # trainer.MasterWeightAdam: 1 / world inside its update
                self.optimizer.seld_grad_scale = 1.0 / self.world
            else:
                self.flat.scale(1.0 / self.world)
        self.optimizer.step()

    def _sync_lr(self):
        """A scheduler replaced the device scalar by a Python float (ReduceLROnPlateau assigns param_group['lr']):
        copy the value into the scalar the captured Adam reads."""
        for group, t in zip(self.optimizer.param_groups, self._lr_tensors):
            if group["lr"] is not t:
                t.fill_(float(group["lr"]))
                group["lr"] = t

    def _exchange(self, k, works):
        if self.world > 1:
            works += self.flat.all_reduce(k, async_op=True)

    @staticmethod
    def _finish(works):
        for w in works:          # RCCL: the current stream waits for the collective's stream; gloo: the host waits
            w.wait()

    # ---- call ---------------------------------------------------------------------------------------------------
    def __call__(self, spec, labels):
        self._sync_lr()
        if not self.use_graphs:
            return self._eager(spec, labels)
        key = (tuple(spec.shape), spec.dtype, tuple(labels.shape), labels.dtype)
        st = self.shapes.get(key)
        if st is None:
            st = self.shapes[key] = {"calls": 0, "graphs": None}
        st["calls"] += 1
        if st["graphs"] is None:
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
        events = [self._event()] if self.timing else None
        works = []
        for k, graph in enumerate(st["graphs"]):
            graph.replay()
            if events is not None:
                events.append(self._event())
            if st["graph_b"] is not None:
                self._exchange(k, works)
        if st["graph_b"] is not None:
            self._finish(works)
            st["graph_b"].replay()
            if events is not None:
                events.append(self._event())
        if events is not None:
            self._segments.append(events)
        self.replays += 1
        return st["out"]

    def _event(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _eager(self, spec, labels):
        self.eager_calls += 1
        works, k = [], 0
        while True:
            last = self._stage(k, spec, labels)
            self._exchange(k, works)
            if last:
                break
            k += 1
        self._finish(works)
        self._update()
        return self._out

    def _capture(self, st, spec, labels):
        st["spec"], st["labels"] = spec.clone(), labels.clone()
        torch.cuda.synchronize(self.device)
        if self.world > 1 and dist.is_initialized() and dist.get_backend() == "nccl":
            # the communicator's watchdog thread polls the events of the warm-up iterations' collectives (every 100 ms);
            # an event query from another thread while a global-mode capture is open can invalidate the capture.  All
            # of them are complete after the synchronisation above: give the watchdog one period to retire them.
            time.sleep(0.3)
        graphs, k = [], 0
        while True:
            graph = torch.cuda.CUDAGraph()
            kwargs = {} if self.pool is None else {"pool": self.pool}
            with torch.cuda.graph(graph, **kwargs):
                last = self._stage(k, st["spec"], st["labels"])
                if last and not self.exchange:
                    self._update()
            if self.pool is None:
                self.pool = graph.pool()
            graphs.append(graph)
            if last:
                break
            k += 1
        graph_b = None
        if self.exchange:
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b, pool=self.pool):
                self._update()
        st["graphs"], st["graph_b"], st["out"] = graphs, graph_b, self._out
        self.captured += 1

    def static_inputs(self, spec_shape, spec_dtype, labels_shape, labels_dtype):
        """The captured graph's input buffers for a batch of this shape, or None while that shape still runs eagerly:
        a feed may gather its batch directly into them (``seld_native.gather_windows(..., out=)``)."""
        st = self.shapes.get((tuple(spec_shape), spec_dtype, tuple(labels_shape), labels_dtype))
        if st is None or st.get("graphs") is None:
            return None
        return st["spec"], st["labels"]

    def segment_ms(self):
        """Average duration of every replayed graph of an iteration ([stage 0, stage 1, ..., update]; needs
        ``timing``; synchronises)."""
        if not self._segments:
            return None
        torch.cuda.synchronize(self.device)
        n = min(len(e) for e in self._segments)
        runs = [e for e in self._segments if len(e) == n]
        return [sum(e[i].elapsed_time(e[i + 1]) for e in runs) / len(runs) for i in range(n - 1)]

    def stats(self):
        out = {"graphs": self.captured, "replays": self.replays, "eager_iterations": self.eager_calls,
               "capture_error": self.capture_error}
        if self.flat is not None:
            buckets = self.flat.describe()
            out["allreduce_overlap"] = bool(self.staged and len(buckets) > 1)
            out["backward_stages"] = len(buckets)
            out["gradient_buckets"] = buckets
            out["reduce_dtype"] = "fp32" if self.flat.fp32_wire else "param (bf16 working weights, fp32 the rest)"
        return out

    def close(self):
        """Drop the graphs and hand the parameters ordinary (absent) gradients again."""
        if self.device.type == "cuda" and self.shapes:
            # Nothing of a replay may be in flight when its graph is destroyed -- and the HIP runtime releases a finished
            # launch's host-side command memory on its own callback thread a little AFTER the device reports idle (two
            # test processes of round 3 aborted inside that thread's free() right after a stepper had been closed, DESIGN.md
            # section 9): the graphs are parked and destroyed when the NEXT stepper closes, long after their last replay.
            torch.cuda.synchronize(self.device)
            _retired.append(dict(self.shapes))
            del _retired[:-2]
        self.shapes.clear()
        self._cuts = []
        for p in self.params:
            p.grad = None
