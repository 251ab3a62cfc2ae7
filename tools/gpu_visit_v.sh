#!/bin/bash
# GPU visit: per-stage slab descriptors (plane images over 2 GB): tests + config 3 / 2 bench
TAG=${1:-r02ah}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_x6.py tests/test_gpu_bench_parity.py -q > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -2 $OUT/pytest.log | cut -c1-200
for w in config3 config2; do
timeout 400 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['verified'], d.get('f32_mfma_only',{}).get('value'))"
done
