#!/bin/bash
# GPU visit r05o: conv2's last partial round as K slices -- parity tests, A/B, kernel stats
TAG=${1:-r05o}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_parity.py tests/test_gpu_x6.py -q -x > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest.log | cut -c1-300
for t in x6_conv_tail=0 x6_conv_tail=1 x6_conv_tail=0 x6_conv_tail=1; do
timeout 300 python bench.py --no-cpu-baseline --no-f32-mfma-leg --no-clock-sample --tune $t > $OUT/b_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_$t.json')); print('$t', d['value'], d['ms_per_step'], d['verified'], d['plain_decode']['value'])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof1 -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --no-plain-leg --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof_s1.json 2> $OUT/prof1.err
python tools/rocpd_stats.py $OUT/prof1/prof_results.db $OUT/kernel_stats_streams1.md | grep -E "total|gemm_x6_kernel|conv_tail" | cut -c1-160
find $OUT -name "*.db" -size +20M -delete
