#!/bin/bash
# rocprofv3 kernel-trace of the benchmark (run on the GPU box): summaries land in gpurun_out/prof_<tag>/
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
ls -R $OUT | head -20
