#!/bin/bash
# GPU visit: prefix-beam priority + wave-per-row CTC top-k -- parity tests, bench A/B
TAG=${1:-r02l}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_parity.py tests/test_gpu_ops.py -q > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -5 $OUT/pytest.log | cut -c1-220
for tune in ""; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune "$tune" > $OUT/b.json 2> $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/b.json')); r=d['roofline']; print('tune[$tune]', d['value'], d['ms_per_step'], r['achieved'], r['avg_launch_us'], d['verified'])"
done
true
true
find $OUT -size +20M -delete
