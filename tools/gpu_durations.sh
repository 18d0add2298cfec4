#!/bin/bash
# Slowest GPU tests of the suite (on a GPU box): tools/gpu_durations.sh [pytest args...]  ->  gpurun_out/gpu_durations.log
# The whole suite takes ~10 min on a fresh box (the driver's limit for it is 900 s); the two-rank rehearsals of
# tests/test_ddp_gpu.py (eight torch.distributed.run launches) are about half of that.
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --durations=40 "$@" > gpurun_out/gpu_durations.log 2>&1
grep -A45 "slowest" gpurun_out/gpu_durations.log | cut -c1-150
tail -2 gpurun_out/gpu_durations.log
