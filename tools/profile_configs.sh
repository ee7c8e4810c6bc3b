#!/bin/bash
# Kernel summaries and bench lines of BASELINE configs[2] (mpnet-base, 32 x 256) and configs[4] (bert-base dims, 128 x 384), run on
# the GPU box from the repo root: tools/profile_configs.sh r04   -> gpurun_out/cfg_<tag>/{c3,c5}_kernel_summary.txt, bench_*.json
set -e
TAG=${1:-r04}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/cfg_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -o w -- python3 $ROOT/bench.py --model all-mpnet-base-v2 --batch 32 --seq-len 256 --steps 10 --warmup 3 --kernel-reps 1 --no-cpu-baseline --no-extras --profile > $OUT/c3_profiled.json 2> $OUT/c3_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -o w -- python3 $ROOT/bench.py --model bert-base-uncased --batch 128 --seq-len 384 --steps 5 --warmup 2 --kernel-reps 1 --no-cpu-baseline --no-extras --profile > $OUT/c5_profiled.json 2> $OUT/c5_profiled.err
cd $ROOT
python profiles/summarize.py $(find $OUT/c3 -name "*kernel_stats.csv" | head -1) 14 > $OUT/c3_kernel_summary.txt
python profiles/summarize.py $(find $OUT/c5 -name "*kernel_stats.csv" | head -1) 7 > $OUT/c5_kernel_summary.txt
python bench.py --model all-mpnet-base-v2 --batch 32 --seq-len 256 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_c3_mpnet.json 2> $OUT/bench_c3.err
python bench.py --model bert-base-uncased --batch 128 --seq-len 384 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_c5_bertbase.json 2> $OUT/bench_c5.err
python bench.py --model bert-base-uncased --batch 128 --seq-len 384 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --train-precision fp8 > $OUT/bench_c5_bertbase_fp8train.json 2> $OUT/bench_c5_fp8.err
head -14 $OUT/c3_kernel_summary.txt
head -14 $OUT/c5_kernel_summary.txt
for f in bench_c3_mpnet bench_c5_bertbase bench_c5_bertbase_fp8train; do cut -c1-200 $OUT/$f.json; done
