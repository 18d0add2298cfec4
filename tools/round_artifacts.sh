#!/bin/bash
# Everything the round's profiles/ directory is built from, on a GPU box (two calls: PART = a | b):
#   tools/round_artifacts.sh TAG a|b      (then, back in the container: python tools/collect_profiles.py TAG ...)
TAG=${1:-r02}
PART=${2:-a}
cd $GRAFT_REPO_ROOT
if [ "$PART" = "a" ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; tail -2 gpurun_out/pytest_$TAG.log
  timeout -k 10 300 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err && cut -c1-160 gpurun_out/bench_$TAG.json
  SELD_GRAPH_STEP=0 timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_${TAG}_eager.json 2> gpurun_out/bench_${TAG}_eager.err && cut -c1-160 gpurun_out/bench_${TAG}_eager.json
  bash tools/prof_bench.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; cat gpurun_out/prof_$TAG/timed_region.log
else
  bash tools/pmc_logmel.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; tail -3 gpurun_out/pmc_$TAG.log
  timeout -k 10 300 python bench.py --model conformer --no-cpu-baseline > gpurun_out/bench_${TAG}_conformer.json 2> gpurun_out/bench_${TAG}_conformer.err; cut -c1-160 gpurun_out/bench_${TAG}_conformer.json
  bash tools/prof_bench.sh ${TAG}_conformer --model conformer > gpurun_out/prof_${TAG}_conformer.log 2>&1; cat gpurun_out/prof_${TAG}_conformer/timed_region.log
  timeout -k 10 300 python bench.py --model resnet_conformer --no-cpu-baseline --steps 2 > gpurun_out/bench_${TAG}_resnet_conformer.json 2> gpurun_out/bench_${TAG}_resnet_conformer.err; cut -c1-160 gpurun_out/bench_${TAG}_resnet_conformer.json
  timeout -k 10 300 python bench.py --features logmel_gcc --channels 8 --no-cpu-baseline > gpurun_out/bench_${TAG}_mic8_gcc.json 2> gpurun_out/bench_${TAG}_mic8_gcc.err; cut -c1-160 gpurun_out/bench_${TAG}_mic8_gcc.json
  bash tools/prof_bench.sh ${TAG}_mic8_gcc --features logmel_gcc --channels 8 > gpurun_out/prof_${TAG}_mic8_gcc.log 2>&1; cat gpurun_out/prof_${TAG}_mic8_gcc/timed_region.log
  timeout -k 10 300 python bench.py --gpus 2 --rehearse-gloo --steps 1 --warmup 1 > gpurun_out/bench_${TAG}_rehearsal_2ranks.json 2> gpurun_out/bench_${TAG}_rehearsal_2ranks.err; cut -c1-160 gpurun_out/bench_${TAG}_rehearsal_2ranks.json
fi
