#!/bin/bash
# GPU visit: folded rel-pos attention -- tests + A/B + kernel stats
TAG=${1:-r02bb}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_ffn_fused.py tests/test_gpu_bench_parity.py tests/test_gpu_parity.py -q -s > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -E "folded rel-pos|passed|failed" $OUT/pytest.log | cut -c1-200
for t in attn_fold=1 attn_fold=2 attn_fold=1; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mfma-leg --tune $t > $OUT/b_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_$t.json')); print('$t', d['value'], d['ms_per_step'], d['verified'])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md > /dev/null; head -1 $OUT/kernel_stats.md; grep -E "attention|relpos" $OUT/kernel_stats.md | cut -c1-170
find $OUT -size +20M -delete
