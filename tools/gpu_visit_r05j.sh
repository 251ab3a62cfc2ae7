#!/bin/bash
# GPU visit r05j: bench lines of configs 3 / 4 / 5 on the current code + kernel stats of configs 3 / 4
TAG=${1:-r05j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for w in config3 config4; do
timeout 400 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['verified'], d.get('plain_decode',{}).get('value'), d.get('f32_mfma_only',{}).get('value'))"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$w -o prof -- python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-f32-mfma-leg --no-plain-leg --streams 1 --min-seconds 0.1 > $OUT/bench_under_rocprof_$w.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof_$w/prof_results.db $OUT/kernel_stats_${w}_streams1.md | head -22 | cut -c1-170
done
for dt in fp32 bf16 fp8; do
timeout 400 python bench.py --workload config5 --dtype $dt --no-cpu-baseline > $OUT/bench_config5_$dt.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config5_$dt.json')); print('config5 $dt', d['value'], d['ms_per_step'], d['verified'], d['verify'].get('identical'), d['roofline']['frac'], d['roofline'].get('whole_decode_frac'))"
done
find $OUT -name "*.db" -size +20M -delete
