#!/bin/bash
# GPU visit r05l: threshold of linear()'s six-product route at the d = 512 configs
TAG=${1:-r05l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for w in config4 config3; do
for t in x6_linear_min=60 x6_linear_min=40 x6_linear_min=20 x6_linear_min=60 x6_linear_min=40 x6_linear_min=20; do
timeout 400 python bench.py --workload $w --no-cpu-baseline --no-f32-mfma-leg --no-plain-leg --min-seconds 1 --tune $t > $OUT/b_${w}_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_${w}_$t.json')); print('$w $t', d['value'], d['ms_per_step'], d['verified'])"
done
done
