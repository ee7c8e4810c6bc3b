#!/bin/bash
# Profiles of one round, run on the GPU box from the repo root: tools/profile_round.sh r03
#   1. rocprofv3 --kernel-trace --stats of the training-only bench command (--no-extras: 25 training steps with dropout on
#      plus one direct launch of each roofline kernel, so the % column of the summary is the step)
#   2. three separate --pmc passes (FETCH_SIZE / WRITE_SIZE / MFMA-busy + clock) of a short training-only bench run
# Outputs under gpurun_out/prof_<tag>/ ; tools/pmc_summary.py and profiles/summarize.py condense them.
set -e
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o w -- python3 $ROOT/bench.py --steps 20 --warmup 5 --kernel-reps 1 --no-cpu-baseline --no-extras --profile > $OUT/bench_profiled.json 2> $OUT/bench_profiled.err
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  D=$OUT/pmc_$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --output-format csv -d $D -o w -- python3 $ROOT/bench.py --steps 3 --warmup 2 --kernel-reps 1 --no-cpu-baseline --no-extras --profile > /dev/null 2> $D.err || true
done
cd $ROOT
python profiles/summarize.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) 25 > $OUT/kernel_summary.txt
python tools/pmc_summary.py $OUT/pmc_summary.json $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $OUT/pmc_summary.txt
head -40 $OUT/kernel_summary.txt
head -30 $OUT/pmc_summary.txt
