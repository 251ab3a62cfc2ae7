#!/bin/bash
# GPU visit: fused six-product FFN (csrc/ffn_x6f.hip) -- its tests, the micro-benchmark rows,
# a default bench.py run and the same with the fused kernel off
TAG=${1:-r03a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_x6.py -q -s -x -k "ffn or on_chip" > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -E "max \|err\||passed|failed|Error|error|assert" $OUT/pytest.log | cut -c1-200 | head -40
timeout 300 python tools/bench_x6.py --only ffn > $OUT/bench_x6.txt 2>&1
cat $OUT/bench_x6.txt | cut -c1-250
timeout 400 python bench.py --no-cpu-baseline > $OUT/bench_config2.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config2.json')); r=d['roofline']; print('fused', d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['verified'], d.get('f32_mfma_only',{}).get('value'))"
timeout 400 python bench.py --no-cpu-baseline --no-f32-mfma-leg --tune ffn_x6f=0 > $OUT/bench_config2_pair.json 2> $OUT/b2.err
python -c "
import json; d=json.load(open('$OUT/bench_config2_pair.json')); r=d['roofline']; print('pair', d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['verified'])"
tail -3 $OUT/b.err $OUT/b2.err | cut -c1-300
