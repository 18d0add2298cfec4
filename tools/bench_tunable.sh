#!/bin/bash
# Developer tool (GPU box): time every hipBLASLt / rocBLAS solution for the GEMM shapes of the three models
# (PyTorch TunableOp) and write the selections to gpurun_out/gemm_gfx950.csv; copy it to
# sound-event-localization-detection_amd/tuned/ to ship it (seld_tuned.py).
set -o pipefail
export SELD_TUNED_GEMMS=tune SELD_TUNED_GEMMS_OUT=gpurun_out/gemm_tuned.csv
mkdir -p gpurun_out
rm -f gpurun_out/gemm_tuned*.csv
for args in "--model crnn" "--model conformer" "--model resnet_conformer" "--model crnn --features logmel_gcc --channels 8"; do
  timeout -k 10 500 python bench.py $args --steps 2 --warmup 2 --no-cpu-baseline 2>gpurun_out/tune.err | cut -c1-110 || exit 1
  wc -l gpurun_out/gemm_tuned.csv
done
cp gpurun_out/gemm_tuned.csv gpurun_out/gemm_gfx950.csv
