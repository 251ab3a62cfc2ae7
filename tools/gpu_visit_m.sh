#!/bin/bash
# GPU visit: 64-row fused FFN (two blocks per CU) -- tests + A/B bench + kernel stats
TAG=${1:-r02s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_ffn_fused.py tests/test_gpu_bench_parity.py -q > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest.log | cut -c1-200
for v in 1 0; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune ffn_bm64=$v > $OUT/b_bm64_$v.json 2> $OUT/b_$v.err
python -c "
import json; d=json.load(open('$OUT/b_bm64_$v.json')); r=d['roofline']; print('bm64=$v', d['value'], d['ms_per_step'], r['achieved'], r['avg_launch_us'], d['verified'])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -7 | cut -c1-170
find $OUT -size +20M -delete
