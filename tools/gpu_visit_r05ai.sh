#!/bin/bash
# GPU visit r05ai: the round's last seconds -- smoke() and a short default bench line on the
# final library (shipped kernels ISA-identical to the r05ae commit, new host dispatch code)
TAG=${1:-r05ai}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 40 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 60 python bench.py --no-cpu-baseline --no-plain-leg --no-f32-mfma-leg --min-seconds 1.0 > $OUT/bench_config2_short.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config2_short.json')); r=d['roofline']; print('config2', d['value'], d['ms_per_step'], r['frac'], d['verified'])"
tail -n 2 $OUT/b.err | cut -c1-200
