#!/bin/bash
TAG=${1:-exp10}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 3 --warmup 1 --workload config5 > $OUT/bench_config5.json 2> $OUT/err.log
cat $OUT/bench_config5.json | cut -c1-900; tail -3 $OUT/err.log | grep -v amdgpu
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats_config5.md | head -16 | cut -c1-200
