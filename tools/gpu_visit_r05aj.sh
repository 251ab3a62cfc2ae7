#!/bin/bash
# GPU visit r05aj: headline (two decodes in flight) with the four config-2 prepared forms on
# against the default, alternating, on one box
TAG=${1:-r05aj}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
B="python bench.py --no-cpu-baseline --no-plain-leg --no-f32-mfma-leg --no-clock-sample --min-seconds 1.5"
T="x6r_pro=2,attn_gload=1,ctc_wave=2,dwconv_tiled=1"
for i in 1 2; do
  timeout 30 $B > $OUT/bench_default_$i.json 2>> $OUT/b.err
  timeout 30 $B --tune $T > $OUT/bench_prepared_$i.json 2>> $OUT/b.err
done
python - <<PY
import json
for n in ('default_1', 'prepared_1', 'default_2', 'prepared_2'):
    d = json.load(open('$OUT/bench_%s.json' % n))
    print(n, d['value'], d['ms_per_step'], d['verified'])
PY
