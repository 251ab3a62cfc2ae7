#!/bin/bash
# GPU visit: kernel stats of the default decode step (1 and 2 decodes in flight) + streams-1 bench
TAG=${1:-r03f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
EXTRA="$2"
for st in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof$st -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --streams $st --min-seconds 0.2 $EXTRA > $OUT/bench_under_rocprof_s$st.json 2> $OUT/prof$st.err
python tools/rocpd_stats.py $OUT/prof$st/prof_results.db $OUT/kernel_stats_streams$st.md | head -30 | cut -c1-200
done
for st in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-f32-mfma-leg --streams $st $EXTRA > $OUT/bench_s$st.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_s$st.json')); r=d['roofline']; print('streams $st', d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['verified'])"
done
find $OUT -name "*.db" -size +20M -delete
find $OUT -name "*.csv" -size +8M -delete
