#!/bin/bash
# Everything the round's profiles/ directory is built from, in one GPU-box call:
#   tools/round_artifacts.sh TAG      (then, back in the container: python tools/collect_profiles.py TAG ...)
TAG=${1:-r01}
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_$TAG.log 2>&1; tail -2 gpurun_out/pytest_$TAG.log
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err && cut -c1-160 gpurun_out/bench_$TAG.json
bash tools/prof_bench.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; cat gpurun_out/prof_$TAG/timed_region.log
bash tools/pmc_logmel.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1
python bench.py --model conformer --no-cpu-baseline > gpurun_out/bench_${TAG}_conformer.json 2> gpurun_out/bench_${TAG}_conformer.err
bash tools/prof_bench.sh ${TAG}_conformer --model conformer > gpurun_out/prof_${TAG}_conformer.log 2>&1; cat gpurun_out/prof_${TAG}_conformer/timed_region.log
python bench.py --model resnet_conformer --no-cpu-baseline --steps 2 > gpurun_out/bench_${TAG}_resnet_conformer.json 2> gpurun_out/bench_${TAG}_resnet_conformer.err
