#!/bin/bash
# GPU visit: spread of the timed rounds (default bench vs longer warm-up)
TAG=${1:-r02ar}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in 3 20 3; do
timeout 300 python bench.py --warmup $w --no-cpu-baseline --no-f32-mfma-leg > $OUT/b_w$w.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_w$w.json')); print('warmup $w', d['value'], d['ms_per_step'], d['rounds']['ms_per_step_each'])"
done
