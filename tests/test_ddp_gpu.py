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


def test_two_ranks_share_one_gpu_and_stay_in_sync(gpu_device):
    env = dict(os.environ, SELD_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    port = 29600 + os.getpid() % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
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
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SELD_DIST_BACKEND")}
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300, cwd=str(ROOT))
    assert out.returncode == 2 and "--rehearse-gloo" in out.stderr and not out.stdout.strip()
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--rehearse-gloo"], env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["rehearsal"] is True and line["config"]["backend"] == "gloo"
    assert line["config"]["replicas_in_sync"] is True and len(line["config"]["per_rank_clips_per_s"]) == 2


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
