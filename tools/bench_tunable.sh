#!/bin/bash
# Developer tool (GPU box): time every hipBLASLt / rocBLAS solution for the GEMM shapes of the bench workloads
# (PyTorch TunableOp) and write the selections to gpurun_out/gemm_gfx950.csv; copy it to
# sound-event-localization-detection_amd/tuned/ to ship it (seld_tuned.py).  Starts from the shipped table: shapes it already
# holds are not re-tuned.
set -o pipefail
export SELD_TUNED_GEMMS=tune SELD_TUNED_GEMMS_OUT=gpurun_out/gemm_tuned.csv
mkdir -p gpurun_out
rm -f gpurun_out/gemm_tuned*.csv
cp sound-event-localization-detection_amd/tuned/gemm_gfx950.csv gpurun_out/gemm_tuned.csv 2>/dev/null
for args in "--model crnn" "--model conformer" "--model resnet_conformer --loss three_term --gaussian-augment" "--model crnn --features logmel_gcc --channels 8"; do
  timeout -k 10 900 python bench.py $args --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-rooflines --no-other-workloads 2>gpurun_out/tune.err | cut -c1-110 || { tail -5 gpurun_out/tune.err; exit 1; }
  wc -l gpurun_out/gemm_tuned.csv
done
cp gpurun_out/gemm_tuned.csv gpurun_out/gemm_gfx950.csv
