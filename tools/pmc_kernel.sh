#!/bin/bash
# rocprofv3 PMC passes over a micro-benchmark; keeps the rows of kernels matching FILTER.
# usage: tools/pmc_kernel.sh TAG FILTER python3 tools/bench_xxx.py args...
set -e
TAG=$1; FILTER=$2; shift 2
ROOTDIR=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$ROOTDIR/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for CTRS in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "WRITE_SIZE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  RAW=/tmp/pmc_raw_${TAG}_$i
  rm -rf $RAW
  rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $RAW -o p -- "$@" > $OUT/pass$i.stdout 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.stdout; continue; }
  F=$(find $RAW -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then head -1 $F > $OUT/pass$i.csv; grep -E "$FILTER" $F | tail -${PMC_KEEP_ROWS:-64} >> $OUT/pass$i.csv; fi
done
ls -la $OUT
