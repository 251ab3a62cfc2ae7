#!/bin/bash
# GPU visit r05g: x6_sub A/B in both stream modes, kernel stats of x6_sub = 2
TAG=${1:-r05g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for st in 1 2; do
for t in x6_sub=0 x6_sub=1 x6_sub=2 x6_sub=0 x6_sub=1 x6_sub=2; do
timeout 300 python bench.py --no-cpu-baseline --no-f32-mfma-leg --no-plain-leg --no-clock-sample --streams $st --tune $t > $OUT/b_s${st}_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_s${st}_$t.json')); print('streams $st $t', d['value'], d['ms_per_step'], d['verified'], d['roofline']['avg_launch_us'])"
done
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof1 -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --no-plain-leg --streams 1 --min-seconds 0.2 --tune x6_sub=2 > $OUT/bench_under_rocprof_s1.json 2> $OUT/prof1.err
python tools/rocpd_stats.py $OUT/prof1/prof_results.db $OUT/kernel_stats_streams1_x6sub2.md | grep -E "total|gemm_x6_kernel|ffn_reduce" | cut -c1-160
find $OUT -name "*.db" -size +20M -delete
