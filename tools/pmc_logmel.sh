#!/bin/bash
# rocprofv3 PMC passes over the log-mel micro-benchmark (counters in their own runs, kernel-trace only).
# usage: tools/pmc_logmel.sh TAG   -> gpurun_out/pmc_TAG/pass*.csv (logmel rows only)
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "WRITE_SIZE SQ_BUSY_CYCLES SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  RAW=/tmp/pmc_raw_${TAG}_$i
  rm -rf $RAW
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $RAW -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_logmel.py 32 3 > $OUT/pass$i.stdout 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.stdout; continue; }
  F=$(find $RAW -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then head -1 $F > $OUT/pass$i.csv; grep logmel $F | tail -24 >> $OUT/pass$i.csv; fi
done
ls -la $OUT
