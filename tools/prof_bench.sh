#!/bin/bash
# rocprofv3 kernel-trace of the benchmark (run on the GPU box).  Only summaries are kept (the raw trace of ~100k
# dispatches is far over gpurun's 64 MiB return limit): whole-process stats, statistics of the timed region, the
# timeline of one optimiser iteration and the dispatches of the hand-written feature kernels.
#   usage: bash tools/prof_bench.sh TAG [bench.py arguments...]
set -e
TAG=${1:-r02}
shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
RAW=/tmp/prof_raw_$TAG
rm -rf $RAW && mkdir -p $OUT $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-rooflines --no-other-workloads "$@" > $OUT/bench_stdout.log 2>&1
find $RAW -name "*kernel_stats.csv" -exec cp {} $OUT/ \;
# statistics of the timed region only (the whole-process summary above includes MIOpen's find-mode trials)
find $RAW -name "*kernel_trace.csv" -exec python3 $GRAFT_REPO_ROOT/tools/trace_timed_region.py {} 2 $OUT/bench_timed_region_stats.csv \; > $OUT/timed_region.log 2>&1
find $RAW -name "*kernel_trace.csv" -exec python3 $GRAFT_REPO_ROOT/tools/iter_timeline.py {} 40 \; > $OUT/iteration_timeline.txt 2>&1
find $RAW -name "*kernel_trace.csv" -exec sh -c 'head -1 "$1" > '$OUT'/kernel_trace_logmel_gru.csv; grep -E "logmel|gru_|softmax_mse|rasterise|gather_rows|expand_kernel" "$1" | head -400 >> '$OUT'/kernel_trace_logmel_gru.csv' _ {} \;
ls -la $OUT
