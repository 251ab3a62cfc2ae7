#!/bin/bash
# GPU visit: decodes in flight A/B (2 vs 3 vs 4), interleaved to see box drift
TAG=${1:-r02ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for t in 2 3 2 3 4; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mfma-leg --streams $t > $OUT/b_s$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_s$t.json')); r=d['rounds']; print('streams $t', d['value'], d['ms_per_step'], r['ms_per_step_min'], d['verified'])"
done
