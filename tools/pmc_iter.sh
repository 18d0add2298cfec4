#!/bin/bash
# MFMA utilisation of the model phase: rocprofv3 PMC passes (own runs, kernel-trace only) over a few eager optimiser
# iterations (tools/iter_profile.py), keeping the rows of the convolution / GEMM / recurrence / attention kernels.
#   usage: bash tools/pmc_iter.sh TAG [model]       -> gpurun_out/pmc_TAG/pass*.csv, then tools/mfma_util.py TAG
TAG=${1:-r02_iter}
MODEL=${2:-crnn}
cd $GRAFT_REPO_ROOT
PMC_KEEP_ROWS=600 SELD_GRAPH_STEP=0 bash tools/pmc_kernel.sh $TAG "conv|Cijk|igemm|gru_|attn|bwd_kernel|gemm|Gemm" python3 $GRAFT_REPO_ROOT/tools/iter_profile.py 4 $MODEL
