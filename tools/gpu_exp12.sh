#!/bin/bash
# kernel stats of the bf16 mode (config 5 and config 2)
OUT=gpurun_out/exp12
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof5 -o prof -- python bench.py --workload config5 --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_config5_bf16.json 2> $OUT/bench5.err
python tools/rocpd_stats.py $OUT/prof5/prof_results.db $OUT/kernel_stats_config5_bf16.md | head -14 | cut -c1-210
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof2 -o prof -- python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > $OUT/bench_config2_bf16.json 2> $OUT/bench2.err
python tools/rocpd_stats.py $OUT/prof2/prof_results.db $OUT/kernel_stats_config2_bf16.md | head -26 | cut -c1-210
cat $OUT/bench_config2_bf16.json | cut -c1-300
find $OUT -size +20M -delete
