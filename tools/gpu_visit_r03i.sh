#!/bin/bash
TAG=${1:-r03i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_cabi.py -q -x > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -4 $OUT/pytest.log | cut -c1-300
bash tools/gpu_pmc.sh $TAG/pmc 2>&1 | grep -E "^p[123] |^kt |ffn_x6f|gemm_x6|conv1_x3|ffn_reduce|gemm_rowln|attention|gemm_f32" | cut -c1-220
timeout 600 python bench.py > $OUT/bench_config2.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config2.json')); r=d['roofline']; print('config2', d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['verified'], d.get('f32_mfma_only',{}).get('value'), d.get('plain_decode',{}).get('value'), d.get('cpu_baseline',{}).get('value'), r.get('traffic'), r.get('algorithmic_bytes'))"
find $OUT -name "*.db" -size +20M -delete
find $OUT -name "*.csv" -size +8M -delete
