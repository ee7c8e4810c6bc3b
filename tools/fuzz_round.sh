#!/bin/bash
# A second pass of tools/fuzz_shapes.py per family with seeds of the caller's (run on the GPU box from the repo root):
#   tools/fuzz_round.sh <tag> <seed0>        logs: gpurun_out/<tag>_fuzz_<family>.log; stops at the first family that fails
set -e
TAG=${1:-fz}
S=${2:-211}
mkdir -p gpurun_out
run() { fam=$1; n=$2; shift 2; timeout -k 10 300 python tools/fuzz_shapes.py $n $S 0 "$@" > gpurun_out/${TAG}_fuzz_$fam.log 2>&1; echo "$fam seed=$S: $(grep -c '^ok' gpurun_out/${TAG}_fuzz_$fam.log) ok"; S=$((S + 2)); }
run enc 30
run f16 20 f16
run f16w 12 f16w
run fused 10 fused
run gemm 30 gemm
run wgrad 20 wgrad
run attn 30 attn
run x3 12 x3
run fp8train 12 fp8train
run ln768 8 ln768
