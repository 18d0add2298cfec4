#!/bin/bash
# clean steady-state per-kernel breakdown of one optimiser iteration (MIOpen find cache warmed by a first run)
set -e
TAG=${1:-iter}; MODEL=${2:-crnn}
python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 5 $MODEL
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG; RAW=/tmp/prof_raw_$TAG
rm -rf $RAW && mkdir -p $OUT $RAW
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW -o it -- python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 40 $MODEL > $OUT/stdout.log 2>&1
find $RAW -name "*kernel_stats.csv" -exec cp {} $OUT/iter_kernel_stats.csv \;
tail -1 $OUT/stdout.log
