#!/bin/bash
# GPU visit: spread of the timed rounds with and without the host's cyclic GC
TAG=${1:-r02ax}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for g in 0 2 1 0; do
WN_BENCH_GC=$g timeout 300 python bench.py --no-cpu-baseline --no-f32-mfma-leg > $OUT/b_gc$g.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_gc$g.json')); print('gc $g', d['value'], d['ms_per_step'], d['rounds']['ms_per_step_each'])"
done
