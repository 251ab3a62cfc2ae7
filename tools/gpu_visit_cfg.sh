#!/bin/bash
# GPU visit: decoder output layers on the six-product GEMM -- full suite + config 3
TAG=${1:-r02az}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest.log | cut -c1-300
for t in x6_linear=1 x6_linear=0; do
timeout 400 python bench.py --workload config3 --no-cpu-baseline --no-f32-mfma-leg --tune $t > $OUT/bench_config3_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config3_$t.json')); print('config3 $t', d['value'], d['ms_per_step'], d['verified'])"
done
