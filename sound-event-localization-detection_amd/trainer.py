"""Training / evaluation loops for SELD on MI355X (drop-in for the reference's ``trainer.py``).

``train_model`` / ``test_model`` keep the reference's signatures, return values, checkpoint format
and epoch semantics (trainer.py:23-392, :394-711):
  Adam(lr, weight_decay as L2-in-grad) + ReduceLROnPlateau(min, factor, patience) on the TEST loss,
  early stopping on the TRAIN loss (PATIENCE, MIN_DELTA), best checkpoint on the TEST loss,
  periodic checkpoints (keep last N), history dict, loss-curve PNG, reload of the best weights.

What is different underneath:
  * one process per GPU: under ``torchrun`` the model is wrapped in DistributedDataParallel
    (RCCL over xGMI, bucketed all-reduce overlapped with backward); window indices -- not files --
    are sharded across ranks so the window set is exactly the single-process one; the per-epoch
    loss sums are all-reduced so LR-plateau / early-stop / best-checkpoint decisions agree on
    every rank; rank 0 alone writes checkpoints, history and plots;
  * bf16 autocast + channels-last convolutions on ROCm devices (conv / linear / attention
    contractions on the matrix cores);
  * device feed: when the loader wraps our ``SELDDataset`` the batches are gathered on the GPU
    from the device-resident feature timeline and the compact uint16 label mask
    (seld_window_gather), so the 290 MB/step dense-label upload and the worker pipes disappear;
    the sampler's order (shuffle / sequential), batch size and drop_last are honoured;
  * fused softmax+MSE loss kernel; no per-step ``.item()`` host syncs (losses accumulate on
    the device and are read once per epoch).
"""
import gc
import logging
import os
import random
from contextlib import nullcontext
from datetime import datetime
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, RandomSampler, SequentialSampler  # noqa: F401
from tqdm import tqdm

from config import Config
from dataset import SELDDataset
from loss import SMRSELDLoss
from model import SMRSELDWithCSPDarkNet
from model_conformer import SELD_Conformer
from model_crnn import SELD_CRNN
from resnet50_model import SELD_ResNet50_Conformer
from seld_rnn import SeldGRU
from utils import get_local_rank, get_rank, get_world_size, safe_torch_load
from visualization import plot_loss_curves, visualize_grid_predictions, visualize_loss_components  # noqa: F401

logger = logging.getLogger("SMR_SELD")
config = Config()


# ------------------------------------------------------------------------------------------------
# model factory (trainer.py:50-95 / :432-473)
# ------------------------------------------------------------------------------------------------

def build_model(grid_size, use_small_model=True, model_type=None, n_channels=None):
    kind = config.MODEL_TYPE if model_type is None else model_type
    common = dict(n_channels=config.N_CHANNELS if n_channels is None else n_channels, grid_size=grid_size,
                  num_classes=config.NUM_CLASSES)
    if kind == "crnn":
        logger.info("Initializing CRNN model...")
        return SELD_CRNN(n_mels=config.N_MELS, cnn_channels=config.CRNN_CNN_CHANNELS,
                         rnn_hidden=config.CRNN_RNN_HIDDEN, rnn_layers=config.CRNN_RNN_LAYERS,
                         dropout=config.CRNN_DROPOUT, **common)
    if kind == "conformer":
        logger.info("Initializing Conformer model...")
        return SELD_Conformer(n_mels=config.N_MELS, cnn_channels=config.CRNN_CNN_CHANNELS,
                              conf_d_model=config.CONF_D_MODEL, conf_n_heads=config.CONF_N_HEADS,
                              conf_n_layers=config.CONF_N_LAYERS, conf_kernel_size=config.CONF_KERNEL_SIZE,
                              dropout=config.CONF_DROPOUT, **common)
    if kind == "resnet_conformer":
        logger.info("Initializing ResNet50-Conformer model...")
        return SELD_ResNet50_Conformer(n_mels=config.N_MELS, conf_d_model=config.RESNET_CONF_D_MODEL,
                                       conf_n_heads=config.RESNET_CONF_N_HEADS,
                                       conf_n_layers=config.RESNET_CONF_N_LAYERS,
                                       dropout=config.RESNET_DROPOUT, **common)
    logger.info("Initializing CSPDarkNet (CNN) model...")
    return SMRSELDWithCSPDarkNet(use_small=use_small_model, **common)


def prepare_model_for_device(model, device):
    """Move to the device; on a GPU use channels-last weights for the convolutions."""
    model = model.to(device)
    if device.type == "cuda" and getattr(config, "CHANNELS_LAST", True):
        model = model.to(memory_format=torch.channels_last)
    SeldGRU.fused_enabled = bool(getattr(config, "FUSED_GRU", True))
    if device.type == "cuda":
        import seld_convtail
        import seld_gru
        seld_convtail.enabled = bool(getattr(config, "FUSED_CONV_TAIL", True))
        import model_crnn
        model_crnn._Conv3x3.enabled = bool(getattr(config, "CONV_DGRAD_AS_FORWARD", True)) and \
            os.environ.get("SELD_CONV_AS_FWD", "1") != "0"          # developer switch for A/B runs
        if os.environ.get("SELD_DWCONV") is None:
            import seld_dwconv
            seld_dwconv.enabled = getattr(config, "FUSED_DWCONV", "auto")
            # "auto" weighed fewer GPU microseconds against more host work (round 1: +3 % at d_model 512, -7 % on the
            # host-bound d_model-256 Conformer); a captured iteration has no host work to weigh
            if seld_dwconv.enabled == "auto" and graph_step_enabled(device):
                seld_dwconv.enabled = True
        import seld_layernorm
        seld_layernorm.enabled = bool(getattr(config, "FUSED_LAYERNORM", True)) and \
            os.environ.get("SELD_LAYERNORM", "1") != "0"            # developer switch for A/B runs
        if os.environ.get("SELD_OVERLAP") is None:
            import seld_overlap
            seld_overlap.enabled = bool(getattr(config, "OVERLAP_WEIGHT_GRADS", True))
        import seld_linear
        seld_linear.conv1x1_as_gemm = bool(getattr(config, "CONV1X1_AS_GEMM", True)) and \
            os.environ.get("SELD_CONV1X1_GEMM", "1") != "0"         # developer switch for A/B runs
        import model_conformer
        model_conformer.MultiHeadSelfAttention.fused_qkv = bool(getattr(config, "FUSED_QKV", True)) and \
            os.environ.get("SELD_FUSED_QKV", "1") != "0"            # developer switch for A/B runs
        for module in model.modules():
            if isinstance(module, SeldGRU) and SeldGRU.fused_enabled:
                seld_gru.pack_parameters(module)
            if isinstance(module, model_conformer.MultiHeadSelfAttention):
                module.pack_parameters()
        if os.environ.get("SELD_CUT_LEVELS") is None:
            import seld_cut
            seld_cut.levels = int(getattr(config, "ALLREDUCE_CUT_LEVELS", 2))
        import seld_tuned
        seld_tuned.enable(device, bool(getattr(config, "TUNED_GEMMS", True)))
    SMRSELDLoss.fused_enabled = bool(getattr(config, "FUSED_LOSS", True))
    return model


def autocast_context(device):
    if device.type == "cuda" and getattr(config, "AMP_DTYPE", "bf16") == "bf16":
        return torch.autocast(device_type="cuda", dtype=torch.bfloat16)
    return nullcontext()


# ------------------------------------------------------------------------------------------------
# distributed plumbing
# ------------------------------------------------------------------------------------------------

def ensure_process_group(device):
    """Initialise torch.distributed when launched by torchrun (WORLD_SIZE > 1).  RCCL ('nccl') on
    GPUs, gloo on CPUs.  Returns (rank, world_size)."""
    world = get_world_size()
    if world <= 1:
        return 0, 1
    if not dist.is_initialized():
        # SELD_DIST_BACKEND=gloo on a GPU: the rehearsal mode in which several ranks share one device (RCCL refuses
        # that); gradients then travel through the host -- functional checks only
        backend = os.environ.get("SELD_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if device.type == "cuda" and backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group(backend=backend, device_id=device)
        else:
            dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def wrap_ddp(model, device, world):
    if world <= 1:
        return model
    if getattr(config, "SYNC_BATCHNORM", False) and device.type == "cuda":
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    kwargs = dict(bucket_cap_mb=getattr(config, "DDP_BUCKET_MB", 25), gradient_as_bucket_view=True,
                  broadcast_buffers=False)
    if device.type == "cuda":
        kwargs.update(device_ids=[device.index], output_device=device.index)
    ddp = torch.nn.parallel.DistributedDataParallel(model, **kwargs)
    # DDP has just broadcast rank 0's parameters -- the bf16 working copies included.  The fp32 masters behind them
    # are not module parameters: broadcast them too (the reference never seeds, so every rank initialised its own)
    # and re-derive the working copies, or the ranks would part ways at the first optimiser step.
    _broadcast_masters(model)
    return ddp


def _broadcast_masters(model):
    state = getattr(model, "_seld_master_weights", None)
    if state is not None:
        with torch.no_grad():
            for master in state[1]:
                dist.broadcast(master, src=0)
            for p, master in zip(state[0], state[1]):
                p.data.copy_(master)


def broadcast_replica_state(model, world):
    """What constructing DistributedDataParallel does to the replicas, for the captured-step path that drives the
    gradient exchange itself (seld_graph.py): every rank takes rank 0's parameters, buffers and fp32 masters."""
    if world <= 1:
        return model
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)
    _broadcast_masters(model)
    return model


def all_reduce_sums(values, device):
    """Sum a small list of Python floats over all ranks (one collective per epoch phase)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(values)
    t = torch.tensor(values, dtype=torch.float64, device=device if device.type == "cuda" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.tolist()


def shard_indices(order, rank, world, pad=True):
    """DistributedSampler semantics on an explicit index order: pad (by wrapping) to a multiple of the
    world size so every rank runs the same number of steps, then take every world-th index."""
    order = list(order)
    if world <= 1:
        return order
    if pad and len(order) % world:
        order += order[: world - len(order) % world]
    return order[rank::world]


# ------------------------------------------------------------------------------------------------
# batch sources
# ------------------------------------------------------------------------------------------------

def epoch_order(n, shuffle, seed, epoch):
    """The index order of one epoch: a seeded permutation (the same on every rank) or sequential."""
    if shuffle:
        g = torch.Generator().manual_seed(seed * 100003 + epoch)
        return torch.randperm(n, generator=g).tolist()
    return list(range(n))


def batched(order, batch_size, drop_last):
    out = [order[lo:lo + batch_size] for lo in range(0, len(order), batch_size)]
    return out[:-1] if drop_last and out and len(out[-1]) < batch_size else out


class LoaderFeed:
    """The stock path: iterate the DataLoader the caller built (CPU tensors -> device).  With more than one rank the
    caller's loader would hand EVERY rank the full epoch (the reference is single-process: main.py:60-74 builds a plain
    shuffling DataLoader), so the epoch's index order is sharded like DeviceFeed's and a loader with the caller's
    batch size, workers, collate function and pinning is driven by an explicit batch sampler."""

    def __init__(self, loader, device, rank=0, world=1, seed=0):
        self.loader, self.device = loader, device
        self.rank, self.world, self.seed = rank, world, seed
        sampler = getattr(loader, "sampler", None)
        self.shuffle = isinstance(sampler, RandomSampler)
        if world > 1 and getattr(loader, "batch_size", None) is None:
            raise ValueError("data-parallel training needs a DataLoader with a batch_size (got a batch_sampler-only loader)")
        if world > 1 and not isinstance(sampler, (RandomSampler, SequentialSampler)):
            # a Weighted / SubsetRandom / custom sampler cannot be sharded by re-deriving its order: refuse instead of
            # silently training on a different (sequential) order
            raise ValueError(f"data-parallel training shards the epoch's index order itself and supports the stock "
                             f"Random / Sequential samplers only (got {type(sampler).__name__})")
        if world > 1:
            n = len(loader.dataset)
            if n % world:
                logger.info(f"  data parallel: {world - n % world} window(s) of every epoch are seen twice (the index "
                            f"order is padded by wrapping to a multiple of {world} ranks, DistributedSampler semantics)")

    def _batches_of_rank(self, epoch):
        order = shard_indices(epoch_order(len(self.loader.dataset), self.shuffle, self.seed, epoch), self.rank, self.world)
        return batched(order, self.loader.batch_size, bool(self.loader.drop_last))

    def __len__(self):
        return len(self.loader) if self.world <= 1 else len(self._batches_of_rank(0))

    def batches(self, epoch):
        loader = self.loader
        if self.world > 1:
            # the caller's loader with this rank's batches: workers, collate function, pinning, worker initialisation,
            # generator and prefetch depth are carried over
            extra = {}
            if loader.num_workers > 0:
                extra = dict(prefetch_factor=loader.prefetch_factor, worker_init_fn=loader.worker_init_fn,
                             multiprocessing_context=loader.multiprocessing_context)
            loader = DataLoader(loader.dataset, batch_sampler=self._batches_of_rank(epoch), num_workers=loader.num_workers,
                                collate_fn=loader.collate_fn, pin_memory=loader.pin_memory, generator=loader.generator,
                                timeout=loader.timeout, **extra)
        for spectrograms, labels in loader:
            yield (spectrograms.to(self.device, non_blocking=True), labels.to(self.device, non_blocking=True))


class DeviceFeed:
    """Batches gathered on the GPU from the dataset's device timeline.  Honours the DataLoader's
    batch size, drop_last and sampler kind (RandomSampler -> a fresh permutation every epoch,
    otherwise sequential); shards the index order over the ranks."""

    def __init__(self, loader, device, rank, world, seed=0):
        self.dataset = loader.dataset
        self.batch_size = loader.batch_size
        self.drop_last = bool(loader.drop_last)
        self.shuffle = isinstance(loader.sampler, RandomSampler)
        self.device, self.rank, self.world, self.seed = device, rank, world, seed

    def _order(self, epoch):
        return shard_indices(epoch_order(len(self.dataset), self.shuffle, self.seed, epoch), self.rank, self.world)

    def __len__(self):
        n = len(shard_indices(range(len(self.dataset)), self.rank, self.world))
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def batches(self, epoch, out=None):
        """``out``: see SELDDataset.device_batch (gather straight into a captured step's input buffers)."""
        for idx in batched(self._order(epoch), self.batch_size, self.drop_last):
            yield self.dataset.device_batch(idx, out=out)


def make_feed(loader, device, rank, world):
    ds = loader.dataset
    use_device = (getattr(config, "DEVICE_FEED", True) and device.type == "cuda" and isinstance(ds, SELDDataset)
                  and getattr(ds, "spec_tm", None) is not None and loader.batch_size is not None)
    seed = config.SEED if getattr(config, "SEED", None) is not None else random.randrange(1 << 30)
    if world > 1:                          # every rank must draw the same permutation
        seed = int(all_reduce_sums([float(seed) if rank == 0 else 0.0], device)[0])
    if use_device:
        return DeviceFeed(loader, device, rank, world, seed)
    return LoaderFeed(loader, device, rank, world, seed)


# ------------------------------------------------------------------------------------------------
# one optimisation step (shared with bench.py)
# ------------------------------------------------------------------------------------------------

def train_step(model, criterion, optimizer, spectrograms, labels, device):
    """forward (autocast) -> loss -> backward (DDP all-reduce overlaps here) -> Adam.  Returns the
    detached (total, class) loss tensors; nothing is synchronised with the host."""
    optimizer.zero_grad(set_to_none=True)
    with autocast_context(device):
        predictions = model(spectrograms)
    total, term = criterion.loss_tensor(predictions, labels)
    total.backward()
    optimizer.step()
    return total.detach(), term.detach()


def eval_step(model, criterion, spectrograms, labels, device):
    with autocast_context(device):
        predictions = model(spectrograms)
    total, term = criterion.loss_tensor(predictions, labels)
    return predictions, total.detach(), term.detach()


# ------------------------------------------------------------------------------------------------
# fp32 master weights behind bf16 working copies
# ------------------------------------------------------------------------------------------------
# Under autocast every conv / Linear / GRU weight is cast fp32 -> bf16 once per iteration and every weight gradient
# bf16 -> fp32: one tiny kernel each (36 per CRNN iteration, 336 per ResNet50-Conformer iteration, ~5 us apiece on an
# otherwise GPU-bound step).  With MASTER_WEIGHTS the model holds the bf16 values themselves -- bit-identical to what
# autocast would produce from the fp32 masters -- and the optimiser owns the fp32 masters: per iteration ONE
# multi-tensor copy of the bf16 gradients into the masters' fp32 gradients, fused Adam on the masters, ONE
# multi-tensor copy back.  Under DDP the weight gradients are then all-reduced in bf16 (half the xGMI bytes).
# BatchNorm / LayerNorm parameters and the GRU / convolution biases stay fp32 (autocast keeps those ops in fp32 as well);
# nn.Linear biases are working copies too (the GEMM epilogue adds them in the compute dtype either way).

def enable_master_weights(model, device):
    """Convert in place; returns True when active.  Call before wrap_ddp / make_optimizer."""
    if not (device.type == "cuda" and getattr(config, "MASTER_WEIGHTS", True)
            and getattr(config, "AMP_DTYPE", "bf16") == "bf16"):
        return False
    import seld_gru
    target = unwrap(model)
    low, masters, names = [], [], {}
    for mod_name, module in target.named_modules():
        if not isinstance(module, (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Linear, torch.nn.GRU)):
            continue
        for pname, p in module.named_parameters(recurse=False):
            # weight matrices, and the biases of Linear layers (they enter the GEMM epilogue in the compute dtype: a
            # per-iteration cast otherwise); GRU / conv / norm biases stay fp32 (kernels that read them in fp32)
            working = p.ndim >= 2 or (isinstance(module, torch.nn.Linear) and pname == "bias")
            if working and p.dtype == torch.float32 and p.requires_grad:
                master = p.detach().clone()
                p.data = p.data.to(torch.bfloat16)
                low.append(p)
                masters.append(master)
                names[f"{mod_name}.{pname}" if mod_name else pname] = master
        if isinstance(module, SeldGRU) and SeldGRU.fused_enabled:
            seld_gru.pack_parameters(module)          # keep (forward, reverse) pairs adjacent in their new dtype
    for module in target.modules():                   # q / k / v of the attention layers likewise (seld_pack.py)
        if type(module).__name__ == "MultiHeadSelfAttention" and hasattr(module, "pack_parameters"):
            module.pack_parameters()
    target._seld_master_weights = (low, masters, names)
    return True


def disable_master_weights(model):
    """Give the model its fp32 weights back (what ``train_model`` returns and what checkpoints hold)."""
    target = unwrap(model)
    state = getattr(target, "_seld_master_weights", None)
    if state is None:
        return
    for p, master in zip(state[0], state[1]):
        p.data = master.detach().clone()
        p.grad = None
    del target._seld_master_weights


def model_state_dict(model):
    """``state_dict`` with the fp32 masters in place of the bf16 working copies (checkpoint format unchanged)."""
    target = unwrap(model)
    sd = target.state_dict()
    state = getattr(target, "_seld_master_weights", None)
    if state is not None:
        for name, master in state[2].items():
            sd[name] = master.detach().clone()
    return sd


class MasterWeightAdam(torch.optim.Adam):
    """Adam over (fp32 masters of the bf16 working weights) + (the remaining fp32 parameters)."""

    def __init__(self, low, masters, others, **kwargs):
        super().__init__(list(masters) + list(others), **kwargs)
        self._low, self._masters, self._others = list(low), list(masters), list(others)
        self.fused_casts = self.fallback_casts = 0        # how often the one-launch casts applied (diagnostics)
        self.own_steps = 0                                # updates done by csrc/adam.hip
        self._uniform_steps = True                        # every parameter has been updated in every step so far
        self._adam_cache = {}
        if os.environ.get("SELD_OWN_ADAM") == "0":
            self.own_kernel = False
        self._grad_cast_cache, self._weight_cast_cache = {}, {}
        for m in self._masters:
            m.grad = torch.zeros_like(m)

    own_kernel = True         # Config.FUSED_ADAM_KERNEL via make_optimizer; SELD_OWN_ADAM=0 switches it off (developer A/B)
    seld_grad_scale = 1.0          # every gradient is multiplied by this on its way into the update (data parallel: 1 / world,
                              # set by seld_graph.GraphedTrainStep -- folded into the kernel instead of a pass over the buffers)

    def _own_step(self):
        """The whole update as ONE multi-tensor launch per 48 tensors (csrc/adam.hip): the bf16 gradients are read as they
        are and the bf16 working copies written from the new masters in the same pass -- 28 B per parameter instead of the
        40 B of cast + fused Adam + cast.  Same arithmetic as the framework's fused Adam; the state keeps its format
        (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq``), so ``state_dict`` is unchanged.  Returns False when the
        call does not qualify (then the three-launch path below runs)."""
        import seld_native
        group = self.param_groups[0]
        lr = group["lr"]
        if not (self.own_kernel and self._uniform_steps and len(self.param_groups) == 1 and isinstance(lr, torch.Tensor)
                and lr.is_cuda and not group.get("amsgrad") and not group.get("maximize")
                and not group.get("decoupled_weight_decay", False)):
            return False
        external = getattr(self, "external_master_grads", False)
        params = self._masters + self._others
        grads = [(m.grad if external else p.grad) for p, m in zip(self._low, self._masters)] + [p.grad for p in self._others]
        if any(g is None for g in grads):
            return False
        steps = []
        for p in params:
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps.append(st["step"])
        lows = [p.data for p in self._low] + [None] * len(self._others)
        exp_avgs = [self.state[p]["exp_avg"] for p in params]
        exp_avg_sqs = [self.state[p]["exp_avg_sq"] for p in params]
        # descriptor checks first (nothing may have run when this returns False): a dry call with count 0 is not
        # available, so the eligibility test is multi_adam's own and the step counts are advanced only after it passed
        beta1, beta2 = group["betas"]
        if not self._adam_checked(grads, params, exp_avgs, exp_avg_sqs, lows):
            return False
        torch._foreach_add_(steps, 1)                 # every parameter's own counter, as the framework keeps them
        ok = seld_native.multi_adam(grads, params, exp_avgs, exp_avg_sqs, lows, lr, steps[0], beta1, beta2, group["eps"],
                                    group["weight_decay"], float(self.seld_grad_scale), self._adam_cache)
        assert ok
        self.fused_casts += 1
        self.own_steps += 1
        return True

    def _adam_checked(self, grads, params, exp_avgs, exp_avg_sqs, lows):
        import seld_native
        for g, p, m, v, lo in zip(grads, params, exp_avgs, exp_avg_sqs, lows):
            if not (p.is_cuda and p.dtype == torch.float32 and g.dtype in (torch.float32, torch.bfloat16)
                    and seld_native._dense_like(g, p) and seld_native._dense_like(m, p) and seld_native._dense_like(v, p)
                    and (lo is None or (lo.dtype == torch.bfloat16 and seld_native._dense_like(lo, p)))):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        import seld_native
        if self._own_step():
            return None
        grads = [p.grad for p in self._low]
        if any(p.grad is None for p in self._low + self._others):
            self._uniform_steps = False           # the framework skips such parameters: their step counts fall behind
        if self.seld_grad_scale != 1.0:
            external = getattr(self, "external_master_grads", False)
            scaled = [g for g in ([m.grad for m in self._masters] if external else grads) + [p.grad for p in self._others]
                      if g is not None]
            torch._foreach_mul_(scaled, float(self.seld_grad_scale))
        master_grads = [m.grad for m in self._masters]
        if getattr(self, "external_master_grads", False):
            # the data-parallel exchange reduced in fp32: the masters' gradients ARE the all-reduced flat buffer's views
            # (seld_graph.FlatGradients, wire dtype "fp32"), nothing to cast
            self.fused_casts += 1
        elif all(g is not None for g in grads) and seld_native.multi_cast(grads, master_grads, self._grad_cast_cache):
            self.fused_casts += 1
        else:
            self.fallback_casts += 1
            for m, g in zip(self._masters, grads):       # a parameter without a gradient, or a layout mismatch
                m.grad.zero_() if g is None else m.grad.copy_(g)
        out = super().step(closure)
        working = [p.data for p in self._low]
        if not seld_native.multi_cast(self._masters, working, self._weight_cast_cache):
            torch._foreach_copy_(working, self._masters)
        return out

    def zero_grad(self, set_to_none=True):
        for p in self._low + self._others:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()


def graph_step_enabled(device, world=1):
    """Captured training iterations (seld_graph.py): ROCm device, Config.GRAPH_STEP, and no SyncBatchNorm (its
    collectives sit inside the forward pass, which the captured path never interrupts)."""
    return (device.type == "cuda" and bool(getattr(config, "GRAPH_STEP", True))
            and os.environ.get("SELD_GRAPH_STEP", "1") != "0"
            and not (world > 1 and getattr(config, "SYNC_BATCHNORM", False)))


def make_optimizer(model, learning_rate, device, capturable=False):
    kwargs = dict(lr=learning_rate, weight_decay=config.WEIGHT_DECAY)
    if device.type == "cuda":
        kwargs["fused"] = True            # one multi-tensor kernel instead of ~4 launches per parameter
        if capturable:                    # step count and learning rate live on the device (graph replay)
            kwargs["capturable"] = True
            kwargs["lr"] = torch.tensor(float(learning_rate), dtype=torch.float32, device=device)
    state = getattr(unwrap(model), "_seld_master_weights", None)
    if state is not None:
        low_ids = {id(p) for p in state[0]}
        others = [p for p in model.parameters() if id(p) not in low_ids]
        opt = MasterWeightAdam(state[0], state[1], others, **kwargs)
        if not getattr(config, "FUSED_ADAM_KERNEL", True):
            opt.own_kernel = False
        return opt
    return torch.optim.Adam(model.parameters(), **kwargs)


def make_stepper(model, criterion, optimizer, device, world=1):
    """The per-iteration callable of the epoch loop: ``(spectrograms, labels) -> (total, term)``.  On a ROCm device a
    ``seld_graph.GraphedTrainStep`` (captured iteration, flat-buffer gradient exchange); otherwise ``train_step``."""
    if graph_step_enabled(device, world) and not isinstance(model, torch.nn.parallel.DistributedDataParallel):
        import seld_graph
        return seld_graph.GraphedTrainStep(model, criterion, optimizer, device, world,
                                           autocast=lambda: autocast_context(device),
                                           overlap_allreduce=bool(getattr(config, "OVERLAP_ALLREDUCE", True)),
                                           reduce_dtype=getattr(config, "GRAD_REDUCE_DTYPE", "param"))
    return lambda spectrograms, labels: train_step(model, criterion, optimizer, spectrograms, labels, device)


def checkpoint_payload(epoch, model, optimizer, train_loss, test_loss):
    """The dict format of trainer.py:278-285 (the Config INSTANCE is pickled, as upstream).  A capturable Adam keeps its
    learning rate (and step counts) in device tensors; the file holds what the reference's plain Adam would have written --
    a float learning rate, host-side step counts, no ``capturable`` flag -- so that upstream code can format, compare or
    resume from it on any box."""
    opt_state = optimizer.state_dict()
    for group in opt_state.get("param_groups", []):
        if isinstance(group.get("lr"), torch.Tensor):
            group["lr"] = float(group["lr"])
        if group.get("capturable"):
            group["capturable"] = False
    # (state_dict() hands out the LIVE per-parameter state dicts: copy before touching them -- a host-side step count
    # in the running capturable Adam would make its fused kernel dereference a host pointer)
    opt_state["state"] = {k: dict(v) for k, v in opt_state.get("state", {}).items()}
    for entry in opt_state["state"].values():
        if isinstance(entry.get("step"), torch.Tensor):
            entry["step"] = entry["step"].detach().float().cpu()
    return {"epoch": epoch, "model_state_dict": model_state_dict(model),
            "optimizer_state_dict": opt_state, "train_loss": train_loss, "test_loss": test_loss,
            "config": config}


def unwrap(model):
    return model.module if isinstance(model, torch.nn.parallel.DistributedDataParallel) else model


# ------------------------------------------------------------------------------------------------
# train_model
# ------------------------------------------------------------------------------------------------

def train_model(train_loader, test_loader, num_epochs=None, batch_size=None, learning_rate=None,
                device=None, use_small_model=True):
    """Complete training run.  Like the reference (trainer.py:36-38) the epoch count, batch size and
    learning rate come from ``config``; the arguments are accepted for signature compatibility."""
    num_epochs, batch_size, learning_rate = config.NUM_EPOCHS, config.BATCH_SIZE, config.LEARNING_RATE
    device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", get_local_rank() if get_world_size() > 1 else torch.cuda.current_device())
    if getattr(config, "SEED", None) is not None:
        torch.manual_seed(config.SEED)
    rank, world = ensure_process_group(device)
    is_main = rank == 0

    train_dataset, test_dataset = train_loader.dataset, test_loader.dataset
    grid = (train_dataset.I, train_dataset.J)
    logger.info(f"Train dataset: {len(train_dataset)} windows ({len(train_loader)} batches)")
    logger.info(f"Test dataset: {len(test_dataset)} windows ({len(test_loader)} batches)")
    logger.info(f"Grid dimensions: {grid[0]}x{grid[1]} = {train_dataset.total_cells} cells; world size {world}")

    # input channels follow the dataset's feature set (4 log-mel; 7 with intensity vectors; 36 for 8-ch GCC-PHAT)
    n_channels = getattr(train_loader.dataset, "n_channels", None)
    model = prepare_model_for_device(build_model(grid, use_small_model, n_channels=n_channels), device)
    n_params = sum(p.numel() for p in model.parameters())
    enable_master_weights(model, device)
    graphed = graph_step_enabled(device, world)
    # captured iterations exchange gradients themselves (bucketed all-reduce between the stage graphs, seld_graph.py); the eager
    # path keeps DistributedDataParallel's bucketed, overlapped reducer
    model = broadcast_replica_state(model, world) if graphed else wrap_ddp(model, device, world)

    class_weights = torch.ones(config.NUM_CLASSES, device=device)
    class_weights[config.NUM_CLASSES - 1] = 0.05            # trainer.py:99-100
    criterion = SMRSELDLoss(loss_type=config.LOSS_TYPE, w_class=config.W_CLASS, w_aiur=config.W_AIUR,
                            w_cl=config.W_CL, grid_size=grid, class_weights=class_weights,
                            three_term=getattr(config, "THREE_TERM_LOSS", False))
    optimizer = make_optimizer(model, learning_rate, device, capturable=graphed)
    stepper = make_stepper(model, criterion, optimizer, device, world)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode="min", factor=config.LR_DECAY_FACTOR,
                                                           patience=config.LR_DECAY_PATIENCE)
    logger.info(f"Model parameters: {n_params:,}")
    logger.info(f"Optimizer: Adam (lr={learning_rate}, weight_decay={config.WEIGHT_DECAY}); "
                f"ReduceLROnPlateau (factor={config.LR_DECAY_FACTOR}, patience={config.LR_DECAY_PATIENCE})")

    train_feed = make_feed(train_loader, device, rank, world)
    test_feed = make_feed(test_loader, device, rank, world)
    logger.info(f"Batch source: {type(train_feed).__name__}; autocast: {getattr(config, 'AMP_DTYPE', 'fp32')}")

    train_losses, test_losses = [], []
    best_train_loss = best_test_loss = float("inf")
    best_epoch = 0
    stale_epochs = 0
    kept_checkpoints = []
    epoch = 0

    for epoch in range(1, num_epochs + 1):
        started = datetime.now()
        # ---- train -----------------------------------------------------------------------------
        model.train()
        loss_sum = torch.zeros((), dtype=torch.float64, device=device)
        term_sum = torch.zeros((), dtype=torch.float64, device=device)
        steps = 0
        static = getattr(stepper, "static_inputs", None) if isinstance(train_feed, DeviceFeed) else None
        source = train_feed.batches(epoch, out=static) if static is not None else train_feed.batches(epoch)
        bar = tqdm(source, total=len(train_feed), desc=f"Epoch {epoch}/{num_epochs} [Train]",
                   leave=False, disable=not is_main)
        for spectrograms, labels in bar:
            total, term = stepper(spectrograms, labels)
            loss_sum += total.double()
            term_sum += term.double()
            steps += 1
            if is_main and steps % 25 == 0:
                bar.set_postfix({"loss": f"{total.item():.4f}", "lr": f"{float(optimizer.param_groups[0]['lr']):.6f}"})
        tr_loss, tr_term, tr_steps = all_reduce_sums([loss_sum.item(), term_sum.item(), float(steps)], device)
        avg_train_loss, avg_train_term = tr_loss / max(tr_steps, 1.0), tr_term / max(tr_steps, 1.0)

        # ---- evaluate --------------------------------------------------------------------------
        model.eval()
        loss_sum.zero_()
        term_sum.zero_()
        steps = 0
        with torch.no_grad():
            for spectrograms, labels in tqdm(test_feed.batches(0), total=len(test_feed), leave=False,
                                             desc=f"Epoch {epoch}/{num_epochs} [Test]", disable=not is_main):
                _, total, term = eval_step(model, criterion, spectrograms, labels, device)
                loss_sum += total.double()
                term_sum += term.double()
                steps += 1
        te_loss, te_term, te_steps = all_reduce_sums([loss_sum.item(), term_sum.item(), float(steps)], device)
        avg_test_loss, avg_test_term = te_loss / max(te_steps, 1.0), te_term / max(te_steps, 1.0)

        train_losses.append(avg_train_loss)
        test_losses.append(avg_test_loss)

        old_lr = float(optimizer.param_groups[0]["lr"])
        scheduler.step(avg_test_loss)
        new_lr = float(optimizer.param_groups[0]["lr"])
        if new_lr != old_lr:
            logger.info(f"  Learning rate reduced: {old_lr:.6f} -> {new_lr:.6f}")

        seconds = (datetime.now() - started).total_seconds()
        logger.info(f"\nEpoch {epoch}/{num_epochs} - Duration: {seconds:.1f}s")
        logger.info(f"  Train Loss: {avg_train_loss:.6f} ({config.LOSS_TYPE.upper()}: {avg_train_term:.6f})")
        logger.info(f"  Test Loss:  {avg_test_loss:.6f} ({config.LOSS_TYPE.upper()}: {avg_test_term:.6f})")
        logger.info(f"  Learning Rate: {new_lr:.6f}")

        # early stopping watches the TRAIN loss (trainer.py:262-270)
        if avg_train_loss < best_train_loss - config.MIN_DELTA:
            logger.info(f"  Train loss improved by {best_train_loss - avg_train_loss:.6f}")
            best_train_loss, best_epoch, stale_epochs = avg_train_loss, epoch, 0
        else:
            stale_epochs += 1
            logger.info(f"  No train loss improvement for {stale_epochs} epoch(s)")

        # best checkpoint watches the TEST loss (trainer.py:273-287)
        if avg_test_loss < best_test_loss - config.MIN_DELTA:
            gained = best_test_loss - avg_test_loss
            best_test_loss = avg_test_loss
            if is_main:
                torch.save(checkpoint_payload(epoch, model, optimizer, avg_train_loss, avg_test_loss),
                           config.CHECKPOINT_PATH / "best_model.pth")
            logger.info(f"  New best model saved! (test loss improvement: {gained:.6f})")

        if epoch % config.SAVE_EVERY_N_EPOCHS == 0 and is_main:
            path = config.CHECKPOINT_PATH / f"checkpoint_epoch_{epoch}.pth"
            torch.save(checkpoint_payload(epoch, model, optimizer, avg_train_loss, avg_test_loss), path)
            kept_checkpoints.append(path)
            logger.info(f"  Checkpoint saved: {path.name}")
            if len(kept_checkpoints) > config.KEEP_LAST_N_CHECKPOINTS:
                old = kept_checkpoints.pop(0)
                if old.exists():
                    old.unlink()
                    logger.info(f"  Removed old checkpoint: {old.name}")

        if device.type == "cuda" and epoch % 5 == 0:
            torch.cuda.empty_cache()
            gc.collect()

        if stale_epochs >= config.PATIENCE:
            logger.info(f"\nEARLY STOPPING at epoch {epoch}: no train loss improvement for {config.PATIENCE} epochs "
                        f"(best train {best_train_loss:.6f} at epoch {best_epoch}, best test {best_test_loss:.6f})")
            break

    logger.info(f"\nTRAINING COMPLETE: {epoch} epochs, best train loss {best_train_loss:.6f} at epoch {best_epoch}, "
                f"best test loss {best_test_loss:.6f}")

    stamp = datetime.now().strftime("%Y%m%d_%H%M%S")
    if is_main and train_losses:
        plot_loss_curves(train_losses, test_losses, save_path=config.OUTPUT_PATH / f"loss_curves_{stamp}.png")
    if world > 1:
        dist.barrier()
    if hasattr(stepper, "close"):
        logger.info(f"Captured training step: {stepper.stats()}")
        stepper.close()
    disable_master_weights(model)                 # the returned model holds plain fp32 parameters again
    best_path = config.CHECKPOINT_PATH / "best_model.pth"
    if best_path.exists():
        best = safe_torch_load(best_path, map_location=device)
        unwrap(model).load_state_dict(best["model_state_dict"])
        logger.info(f"Best model loaded from epoch {best['epoch']}")

    history = {
        "train_losses": train_losses, "test_losses": test_losses,
        "best_train_loss": best_train_loss, "best_test_loss": best_test_loss,
        "best_epoch": best_epoch, "total_epochs": epoch,
        "config": {"num_epochs": num_epochs, "batch_size": batch_size, "learning_rate": learning_rate,
                   "grid_size": grid,
                   # not upstream: how the epoch was sharded (one process: world 1, every batch on this rank)
                   "world_size": world, "batches_per_rank": len(train_feed), "batch_source": type(train_feed).__name__},
    }
    if is_main:
        history_path = config.OUTPUT_PATH / f"training_history_{stamp}.pth"
        torch.save(history, history_path)
        logger.info(f"Training history saved to {history_path}")
    return unwrap(model), history


# ------------------------------------------------------------------------------------------------
# test_model
# ------------------------------------------------------------------------------------------------

def test_model(test_loader, model_path=None, batch_size=None, device=None, num_visualizations=5,
               save_visualizations=True):
    """Evaluate a checkpoint (trainer.py:394-711): loss, argmax accuracies, frames with events and a few
    ground-truth-vs-prediction plots.  Accuracies and event counts are reduced on the device batch by
    batch instead of collecting every logit on the host (N x 9 MB x 2 upstream); only the frames that
    are actually plotted are copied back.  Runs on the calling rank only (rank 0 under torchrun)."""
    batch_size = batch_size or config.BATCH_SIZE
    model_path = Path(model_path or (config.CHECKPOINT_PATH / "best_model.pth"))
    device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    test_dataset = test_loader.dataset
    grid = (test_dataset.I, test_dataset.J)
    logger.info(f"Test dataset: {len(test_dataset)} windows ({len(test_loader)} batches)")
    if not model_path.exists():
        raise FileNotFoundError(f"Model checkpoint not found: {model_path}")
    if get_rank() != 0:
        logger.info("test_model runs on rank 0 only")
        return {}

    checkpoint = safe_torch_load(model_path, map_location=device)
    model = prepare_model_for_device(build_model(grid, True, n_channels=getattr(test_dataset, "n_channels", None)),
                                     device)
    model.load_state_dict(checkpoint["model_state_dict"])
    model.eval()
    logger.info(f"Model loaded (epoch {checkpoint['epoch']}, test loss {checkpoint['test_loss']:.6f})")
    criterion = SMRSELDLoss(loss_type=config.LOSS_TYPE, w_class=config.W_CLASS, w_aiur=config.W_AIUR,
                            w_cl=config.W_CL, grid_size=grid,          # un-weighted, trainer.py:483-489
                            three_term=getattr(config, "THREE_TERM_LOSS", False))

    feed = make_feed(test_loader, device, 0, 1)
    bg = config.NUM_CLASSES - 1
    loss_sum = torch.zeros((), dtype=torch.float64, device=device)
    term_sum = torch.zeros((), dtype=torch.float64, device=device)
    correct = torch.zeros((), dtype=torch.int64, device=device)
    correct_events = torch.zeros((), dtype=torch.int64, device=device)
    n_cells = torch.zeros((), dtype=torch.int64, device=device)
    n_events = torch.zeros((), dtype=torch.int64, device=device)
    active_per_frame = []                                     # [B, T] int32 per batch (small)
    steps = 0
    with torch.no_grad():
        for spectrograms, labels in tqdm(feed.batches(0), total=len(feed), desc="Testing"):
            predictions, total, term = eval_step(model, criterion, spectrograms, labels, device)
            loss_sum += total.double()
            term_sum += term.double()
            steps += 1
            pred_cls = predictions.argmax(dim=-1)
            true_cls = _label_classes(labels, config.NUM_CLASSES)
            events = true_cls != bg
            hit = pred_cls == true_cls
            correct += hit.sum()
            correct_events += (hit & events).sum()
            n_cells += hit.numel()
            n_events += events.sum()
            active_per_frame.append(events.sum(dim=-1).to(torch.int32).cpu())
    avg_test_loss = loss_sum.item() / max(steps, 1)
    avg_test_term = term_sum.item() / max(steps, 1)
    overall_accuracy = 100.0 * correct.item() / max(n_cells.item(), 1)
    non_bg_accuracy = 100.0 * correct_events.item() / n_events.item() if n_events.item() > 0 else 0.0
    logger.info(f"Total Loss: {avg_test_loss:.6f}  (Class {config.LOSS_TYPE.upper()}: {avg_test_term:.6f})")
    logger.info(f"Overall Accuracy: {overall_accuracy:.2f}%  Non-Background Accuracy: {non_bg_accuracy:.2f}%")
    logger.info(f"Active Events: {n_events.item()} / {n_cells.item()}")

    active = torch.cat(active_per_frame, dim=0) if active_per_frame else torch.zeros((0, 0), dtype=torch.int32)
    frames_with_events = [(int(w), int(t), int(active[w, t])) for w, t in (active > 0).nonzero().tolist()]
    logger.info(f"Found {len(frames_with_events)} frames with active events")
    results = {
        "test_loss": avg_test_loss, f"class_{config.LOSS_TYPE}": avg_test_term,
        "overall_accuracy": overall_accuracy, "non_bg_accuracy": non_bg_accuracy,
        "num_frames_with_events": len(frames_with_events), "visualizations": [],
        "checkpoint_epoch": checkpoint["epoch"],
    }
    if not frames_with_events:
        logger.warning("No frames with active events found! Cannot create visualizations.")
        return results

    chosen = random.sample(frames_with_events, min(num_visualizations, len(frames_with_events)))
    chosen.sort(key=lambda f: f[2], reverse=True)
    if save_visualizations:
        (config.OUTPUT_PATH / "test_visualizations").mkdir(exist_ok=True)
    for viz, (window_idx, time_idx, num_active) in enumerate(chosen, start=1):
        spec, labels = test_dataset[window_idx]
        with torch.no_grad(), autocast_context(device):
            logits = model(spec.unsqueeze(0).to(device))[0, time_idx].float().cpu()
        save_path = None
        if save_visualizations:
            save_path = config.OUTPUT_PATH / "test_visualizations" / f"test_viz_{viz}_window{window_idx}_frame{time_idx}.png"
        fig = visualize_grid_predictions(ground_truth=labels[time_idx], predictions=logits, time_frame=time_idx,
                                         grid_size=grid, title_prefix=f"Window {window_idx}, ", save_path=save_path)
        results["visualizations"].append({"window_idx": window_idx, "time_idx": time_idx, "num_active": num_active,
                                          "figure": fig, "save_path": save_path})
    logger.info("TESTING COMPLETE")
    return results


def _label_classes(labels, num_classes):
    """argmax over classes of the labels, for dense float labels or the compact uint16 mask
    (lowest set bit = first maximal class, as torch.argmax picks the first maximum; mask 0 = background)."""
    if labels.dtype != torch.uint16:
        return labels.argmax(dim=-1)
    m = labels.to(torch.int32)
    lowest = (m & -m).float().log2().to(torch.int64)
    return torch.where(m == 0, torch.full_like(lowest, num_classes - 1), lowest)
