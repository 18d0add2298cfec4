#!/bin/bash
# Everything the round's profiles/ directory is built from, on a GPU box (two calls: PART = a | b):
#   tools/round_artifacts.sh TAG a|b      (then, back in the container: python tools/collect_profiles.py TAG ...)
TAG=${1:-r03}
PART=${2:-a}
cd $GRAFT_REPO_ROOT
if [ "$PART" = "a" ]; then
  # the driver's command (default bench: headline + kernels[] + allreduce_overlap windows + other_workloads + cpu_baseline)
  timeout -k 10 500 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err && cut -c1-160 gpurun_out/bench_$TAG.json
  bash tools/prof_bench.sh $TAG > gpurun_out/prof_$TAG.log 2>&1; cat gpurun_out/prof_$TAG/timed_region.log
  timeout -k 10 300 python bench.py --gpus 2 --rehearse-gloo --steps 1 --warmup 1 > gpurun_out/bench_${TAG}_rehearsal_2ranks.json 2> gpurun_out/bench_${TAG}_rehearsal_2ranks.err; cut -c1-160 gpurun_out/bench_${TAG}_rehearsal_2ranks.json
  bash tools/pmc_logmel.sh $TAG > gpurun_out/pmc_$TAG.log 2>&1; tail -3 gpurun_out/pmc_$TAG.log
else
  bash tools/prof_bench.sh ${TAG}_conformer --model conformer > gpurun_out/prof_${TAG}_conformer.log 2>&1; cat gpurun_out/prof_${TAG}_conformer/timed_region.log
  bash tools/prof_bench.sh ${TAG}_resnet_conformer --model resnet_conformer --loss three_term --gaussian-augment --steps 2 > gpurun_out/prof_${TAG}_resnet_conformer.log 2>&1; cat gpurun_out/prof_${TAG}_resnet_conformer/timed_region.log
  bash tools/prof_bench.sh ${TAG}_mic8_gcc --features logmel_gcc --channels 8 > gpurun_out/prof_${TAG}_mic8_gcc.log 2>&1; cat gpurun_out/prof_${TAG}_mic8_gcc/timed_region.log
  PMC_KEEP_ROWS=24 bash tools/pmc_kernel.sh ${TAG}_spatial "gcc_q15|gcc_mfma|logmel_main" python3 $GRAFT_REPO_ROOT/tools/bench_spatial.py logmel_gcc 8 8 3 > gpurun_out/pmc_${TAG}_spatial.log 2>&1; tail -3 gpurun_out/pmc_${TAG}_spatial.log
  PMC_KEEP_ROWS=24 bash tools/pmc_kernel.sh ${TAG}_foa "logmel_iv|foa_iv|logmel_main" python3 $GRAFT_REPO_ROOT/tools/bench_spatial.py logmel_iv 4 8 3 > gpurun_out/pmc_${TAG}_foa.log 2>&1; tail -3 gpurun_out/pmc_${TAG}_foa.log
fi
