"""Two data-parallel ranks through the whole GPU hot path (bench.py: features, labels, windows, CRNN with the HIP
autograd functions, DistributedDataParallel, master-weight Adam) on ONE GPU.  RCCL refuses two ranks on one device,
so the rehearsal uses the gloo backend (SELD_DIST_BACKEND=gloo; gradients travel through the host) -- it checks the
plumbing, never performance: the ranks start from DIFFERENT seeds and must hold identical weights afterwards."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


_bench = {}


def _bench_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SELD_DIST_BACKEND")}


def _rehearsal_line():
    """ONE two-rank rehearsal of bench.py per test session (a launch costs most of a minute on a fresh box and the driver
    gives the whole GPU suite 900 s): ``python bench.py --gpus 2 --rehearse-gloo`` starts its own ranks with exactly the
    command the driver uses for N > 1 (``python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 --master-port P bench.py --gpus 2 ...``, bench.py::launch_ranks) and relays rank 0's line."""
    if "line" not in _bench:
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                              "--rehearse-gloo", "--rehearsal-clips", "8"], env=_bench_env(), capture_output=True, text=True,
                             timeout=900, cwd=str(ROOT))
        assert out.returncode == 0, out.stderr[-2000:]
        _bench["line"] = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    return _bench["line"]


def test_two_ranks_share_one_gpu_and_stay_in_sync(gpu_device):
    line = _rehearsal_line()
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["replicas_in_sync"] is True
    assert line["config"]["master_weights"]["per_tensor_fallbacks"] == 0
    assert line["value"] > 0 and 0.0 < line["config"]["final_loss"] < 1.0


def test_bench_launches_its_own_ranks_or_refuses(gpu_device):
    """``python bench.py --gpus 2`` without torchrun: refuses on a one-GPU box (exit code 2, nothing launched) and, with
    --rehearse-gloo, starts its own two ranks and prints a 2-rank line marked as a rehearsal."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer GPUs than ranks")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=_bench_env(), capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert out.returncode == 2 and "--rehearse-gloo" in out.stderr and not out.stdout.strip()
    line = _rehearsal_line()
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["config"]["backend"] == "gloo"
    assert line["config"]["replicas_in_sync"] is True and len(line["config"]["per_rank_clips_per_s"]) == 2
    assert line["allreduce_overlap"] is True and len(line["gradient_buckets"]) == 3


def test_train_model_shards_the_device_feed(gpu_device, tmp_path):
    """trainer.train_model under two ranks with the DEVICE feed (tests/ddp_train_worker.py): the epoch's window order is
    sharded inside the epoch loop (disjoint halves), DDP + the epoch-sum all-reduce keep weights and history identical
    although the ranks initialised different weights, and rank 0 alone writes the checkpoint."""
    env = dict(os.environ, SELD_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "ddp_train_worker.py"), str(tmp_path)]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = sorted((json.loads(l.split("RANKLINE ", 1)[1]) for l in out.stdout.splitlines() if "RANKLINE " in l),
                   key=lambda d: d["rank"])
    assert [d["rank"] for d in lines] == [0, 1]
    a, b = lines
    n = a["windows"]
    assert n == b["windows"] and n >= 8
    assert not set(a["shard_epoch1"]) & set(b["shard_epoch1"]) or n % 2 == 1          # padded by wrapping when odd
    assert sorted(set(a["shard_epoch1"]) | set(b["shard_epoch1"])) == list(range(n))
    assert len(a["shard_epoch1"]) == len(b["shard_epoch1"]) == (n + 1) // 2
    for d in lines:
        assert d["config"]["world_size"] == 2 and d["config"]["batch_source"] == "DeviceFeed"
        assert d["config"]["batches_per_rank"] == ((n + 1) // 2 + 2) // 3
        assert d["wrote_checkpoint"]
    assert a["param_sum"] == b["param_sum"] and a["param_abs"] == b["param_abs"]
    assert a["train_losses"] == b["train_losses"] and a["test_losses"] == b["test_losses"]
    assert len(a["train_losses"]) == 2 and all(0 < v < 1 for v in a["train_losses"])


_runs = {}


def _exchange_run(mode, port_base, kind="crnn"):
    """One two-rank run of tests/ddp_step_worker.py per mode, model and test session (the runs are deterministic)."""
    if (mode, kind) not in _runs:
        _runs[(mode, kind)] = _exchange_run_uncached(mode, port_base, kind)
    return _runs[(mode, kind)]


def _exchange_run_uncached(mode, port_base, kind="crnn"):
    env = dict(os.environ, SELD_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = port_base + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "tests" / "ddp_step_worker.py"), mode, kind]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = sorted((json.loads(l.split("RANKLINE ", 1)[1]) for l in out.stdout.splitlines() if "RANKLINE " in l),
                   key=lambda d: d["rank"])
    assert [d["rank"] for d in lines] == [0, 1]
    return lines


def test_overlapped_gradient_exchange_is_bit_identical_to_the_blocking_one(gpu_device):
    """The north-star step (trainer.py:165-179 sharded over ranks, gradient all-reduce overlapped with backward): the
    captured iteration cut into backward stages whose gradient buckets are all-reduced asynchronously while the next
    stage replays must give the SAME losses and weights, bit for bit, as the uncut iteration followed by one blocking
    exchange -- 16 iterations (3 eager warm-ups, capture, 13 replays) on per-rank batches, two ranks over gloo."""
    staged = _exchange_run("staged", 30100)
    blocking = _exchange_run("blocking", 30300)
    for lines in (staged, blocking):
        assert lines[0]["digest"] == lines[1]["digest"]                         # replicas agree
        assert lines[0]["losses"] != lines[1]["losses"]                         # on different batches
        assert all(0 < v < 1 for d in lines for v in d["losses"])
    for a, b in zip(staged, blocking):
        assert a["losses"] == b["losses"] and a["digest"] == b["digest"]
    st = staged[0]["stats"]
    assert st["capture_error"] is None and st["graphs"] == 1 and st["replays"] == 13 and st["eager_iterations"] == 3
    assert st["allreduce_overlap"] is True and st["backward_stages"] == 3
    sizes = [b["bytes"] for b in st["gradient_buckets"]]
    # bucket 0: head + GRU layer 1 (+ nothing fp32: those travel last); bucket 1: GRU layer 0's four weight matrices
    # (carried over the first cut); bucket 2: the convolutions + every fp32 parameter
    assert sizes[0] > sizes[1] > sizes[2] > 0
    assert sizes[1] == 2 * (2 * 768 * 128 + 2 * 768 * 256)
    assert blocking[0]["stats"]["allreduce_overlap"] is False and blocking[0]["stats"]["backward_stages"] == 1
    assert sum(sizes) == sum(b["bytes"] for b in blocking[0]["stats"]["gradient_buckets"])


def test_fp32_wire_and_eager_ddp_paths_track_the_default(gpu_device):
    """Config.GRAD_REDUCE_DTYPE = 'fp32' (gradients cast up before the exchange; the buffers are the masters' gradients)
    and Config.GRAPH_STEP = False (eager loop under DistributedDataParallel's bucketed reducer): replicas stay in sync
    and the loss curve follows the default path's (not bit for bit: the sum is rounded at a different point / the
    reducer averages before the cast)."""
    staged = _exchange_run("staged", 30500)
    for mode, port in (("fp32wire", 30700), ("ddp", 30900)):
        lines = _exchange_run(mode, port)
        assert lines[0]["digest"] == lines[1]["digest"], mode
        for a, b in zip(staged, lines):
            assert all(abs(x - y) <= 2e-3 * abs(x) for x, y in zip(a["losses"], b["losses"])), (mode, a["losses"], b["losses"])
    assert lines[0]["stats"] is None                                            # the last run was the eager DDP loop


def test_bf16_gradient_sum_error_of_eight_ranks(gpu_device):
    """The default exchange sums the bf16 working-weight gradients IN bf16 (seld_graph.FlatGradients, wire dtype 'param':
    half the xGMI bytes of the fp32 exchange an autocast port of trainer.py:165-179 would do).  What that costs, measured on
    the full-size CRNN's own gradients: eight per-rank gradient sets (eight different batches through the same weights)
    are added the way a ring all-reduce adds them -- one rank after the other, every partial sum rounded to bf16 -- then
    scaled by 1/8 (exact), and compared with the fp32 sum of the same bf16 gradients (wire dtype 'fp32').  Bar: relative L2
    error of every tensor <= 6e-3 (the bf16 rounding of the gradient itself is 1.1e-3 .. 2.3e-3), whole buffer <= 4e-3."""
    import torch
    import seld_graph
    import trainer
    cfg = trainer.config
    saved = cfg.MODEL_TYPE
    cfg.MODEL_TYPE = "crnn"
    try:
        torch.manual_seed(0)
        model = trainer.prepare_model_for_device(trainer.build_model((18, 36)), gpu_device).train()
        trainer.enable_master_weights(model, gpu_device)
        crit = trainer.SMRSELDLoss("mse", 1.0, grid_size=(18, 36))
        low = [(n, p) for n, p in model.named_parameters() if p.dtype == torch.bfloat16]
        assert len(low) >= 14
        g = torch.Generator().manual_seed(11)
        per_rank = []
        for r in range(8):
            x = (torch.randn(8, 250, 4, 64, generator=g) * 20 - 30).to(gpu_device)
            m = ((torch.rand(8, 250, 648, generator=g) < 0.02).to(torch.int32) << 3).to(torch.uint16).to(gpu_device)
            for p in model.parameters():
                p.grad = None
            with trainer.autocast_context(gpu_device):
                out = model(x)
            total, _ = crit.loss_tensor(out, m)
            total.backward()
            import seld_overlap
            seld_overlap.join(gpu_device)
            torch.cuda.synchronize()
            per_rank.append([p.grad.detach().clone() for _, p in low])
    finally:
        cfg.MODEL_TYPE = saved
    worst, num, den, rows = 0.0, 0.0, 0.0, []
    for i, (name, p) in enumerate(low):
        grads = [per_rank[r][i] for r in range(8)]
        assert all(gr.dtype == torch.bfloat16 for gr in grads)
        acc = grads[0]
        for gr in grads[1:]:
            acc = (acc.float() + gr.float()).to(torch.bfloat16)          # one hop of the ring: add, round to the wire dtype
        ring = (acc * 0.125).float()                                      # x 1/world in bf16: exact
        exact = torch.stack([gr.double() for gr in grads]).sum(0) / 8.0
        fp32_wire = (torch.stack([gr.float() for gr in grads]).sum(0) * 0.125).double()
        e_ring = ((ring.double() - exact).norm() / exact.norm()).item()
        e_fp32 = ((fp32_wire - exact).norm() / exact.norm()).item()
        e_once = ((exact.float().to(torch.bfloat16).double() - exact).norm() / exact.norm()).item()
        rows.append((name, e_ring, e_fp32, e_once))
        worst = max(worst, e_ring)
        num += (ring.double() - exact).pow(2).sum().item()
        den += exact.pow(2).sum().item()
        assert e_fp32 <= 1e-6, (name, e_fp32)
    whole = (num / den) ** 0.5
    print("bf16 ring-sum error of 8 ranks, relative L2 per tensor (ring bf16 / fp32 wire / one bf16 rounding):")
    for name, a, b, c in rows:
        print(f"  {name:34s} {a:.2e} {b:.1e} {c:.2e}")
    print(f"  whole buffer {whole:.2e}, worst tensor {worst:.2e}")
    assert worst <= 6e-3 and whole <= 4e-3, (worst, whole)


@pytest.mark.parametrize("kind", ["conformer", "resnet_conformer"])
def test_overlapped_exchange_of_the_other_models(gpu_device, kind):
    """The cut points of the Conformer (shared encoder) and of the ResNet50-Conformer (after the encoder, before layer4)
    under two ranks: the staged capture succeeds, two / three gradient buckets travel, replicas stay in sync and the losses
    stay finite and bounded.  (That the staged iteration tracks the uncut one for these models is checked on one rank,
    tests/test_graph_gpu.py::test_other_models_staged_capture; a second, blocking, two-rank launch per model for the same
    comparison was dropped to keep the suite inside the driver's 900 s -- the CRNN keeps its bit-for-bit comparison above.)"""
    staged = _exchange_run("staged", 31100 if kind == "conformer" else 31300, kind)
    assert staged[0]["digest"] == staged[1]["digest"]
    assert staged[0]["losses"] != staged[1]["losses"]                            # on different batches
    assert all(0 < v < 1.5 for d in staged for v in d["losses"])
    st = staged[0]["stats"]
    assert st["capture_error"] is None and st["allreduce_overlap"] is True
    assert st["backward_stages"] == (2 if kind == "conformer" else 3)
    assert all(b["bytes"] > 0 for b in st["gradient_buckets"])
