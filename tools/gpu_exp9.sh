#!/bin/bash
TAG=${1:-exp9}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for w in config3 config4; do
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload $w > $OUT/bench_$w.json 2> $OUT/err_$w.log
python -c "import json;d=json.load(open('$OUT/bench_$w.json'));print('$w', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['config']['encoder_frames_per_gpu'])" || tail -5 $OUT/err_$w.log
python tools/rocpd_stats.py $OUT/prof_$w/prof_results.db $OUT/kernel_stats_$w.md | head -26 | cut -c1-190
done
