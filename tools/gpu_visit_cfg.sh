#!/bin/bash
# GPU visit: large fp32 linear() GEMMs on the six-product kernel -- parity + configs 3 / 4
TAG=${1:-r02at}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_parity.py tests/test_gpu_x6.py -q -s > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -E "^\[config|passed|failed" $OUT/pytest.log | cut -c1-420
for w in config3 config4; do
for t in x6_linear=1 x6_linear=0; do
timeout 400 python bench.py --workload $w --no-cpu-baseline --no-f32-mfma-leg --tune $t > $OUT/bench_${w}_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_${w}_$t.json')); print('$w $t', d['value'], d['ms_per_step'], d['verified'])"
done; done
