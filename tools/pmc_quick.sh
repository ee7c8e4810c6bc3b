#!/bin/bash
# tools/pmc_quick.sh <tag> <python script> : kernel trace + three counter passes of a short script, summarised per kernel
set -e
TAG=$1; SCRIPT=$2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python3 $ROOT/$SCRIPT > /dev/null 2> $OUT/trace.err || true
i=0
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o w -- python3 $ROOT/$SCRIPT > /dev/null 2> $OUT/p$i.err || true
done
cd $ROOT
python tools/pmc_summary.py $OUT/summary.json $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 > $OUT/summary.txt || true
cat $OUT/summary.txt
