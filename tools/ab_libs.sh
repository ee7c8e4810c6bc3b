#!/bin/bash
# Process-alternating A/B of whole libraries on the default training step (run on the GPU box from the repo root; works on the
# box's copy of the tree): tools/ab_libs.sh tools/libqst_base.so tools/libqst_x.so ...   -> ms per step per library and round
set -e
P=quadruplet-sentence-transformer_amd/libqst.so
cp $P /tmp/libqst_keep.so
for r in 1 2 3; do
  for L in "$@"; do
    cp $L $P
    python bench.py --no-extras --no-cpu-baseline --steps 50 --warmup 10 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $r', '$L', b['ms_per_step'], b['value'])"
  done
done
cp /tmp/libqst_keep.so $P
