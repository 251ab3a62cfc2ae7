#!/bin/bash
# GPU visit r05h: conv2 tile height A/B in the two-stream mode
TAG=${1:-r05h}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for t in x6_conv_bm=0 x6_conv_bm=128 x6_conv_bm=256 x6_conv_bm=0 x6_conv_bm=128 x6_conv_bm=256; do
timeout 300 python bench.py --no-cpu-baseline --no-f32-mfma-leg --no-plain-leg --no-clock-sample --tune $t > $OUT/b_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_$t.json')); print('$t', d['value'], d['ms_per_step'], d['verified'], d['roofline']['avg_launch_us'])"
done
