#!/bin/bash
# GPU visit: conv2 K order (channel block outside, taps inside) -- tests + A/B + kernel stats
TAG=${1:-r02av}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_x6.py tests/test_gpu_bench_parity.py -q > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -2 $OUT/pytest.log | cut -c1-300
for t in x6_conv_order=1 x6_conv_order=0; do
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$t -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --streams 1 --min-seconds 0.2 --tune $t > $OUT/bench_under_rocprof_$t.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof_$t/prof_results.db $OUT/kernel_stats_$t.md > /dev/null; echo $t; grep -E "true, false" $OUT/kernel_stats_$t.md | cut -c1-170
done
find $OUT -size +20M -delete
