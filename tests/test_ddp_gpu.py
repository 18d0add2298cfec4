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
