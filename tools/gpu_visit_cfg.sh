#!/bin/bash
# GPU visit: FFN w_2 in 8 K slices (two blocks per CU) vs 4 -- kernel stats
TAG=${1:-r02aw}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for t in x6_ffn_s=8 x6_ffn_s=0; do
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$t -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --streams 1 --min-seconds 0.2 --tune $t > $OUT/bench_under_rocprof_$t.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof_$t/prof_results.db $OUT/kernel_stats_$t.md > /dev/null; echo $t; head -1 $OUT/kernel_stats_$t.md; grep -E "x6_kernel<128, 1|x6_kernel<256, 1|ffn_reduce" $OUT/kernel_stats_$t.md | cut -c1-170
done
find $OUT -size +20M -delete
