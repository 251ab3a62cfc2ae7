#!/bin/bash
# GPU visit: fp8 mode (operator + model tests), the whole bf16 suite with the pipelined
# kernel in the dispatch, config-5 bench lines bf16 / fp8, config-2 fp32 regression.
TAG=${1:-r02f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_fp8.py -q -s > $OUT/pytest_fp8.log 2>&1
echo "fp8 tests exit $?"; grep -E "fp8 encoder|passed|failed|Error" $OUT/pytest_fp8.log | tail -8 | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_bf16.py -q > $OUT/pytest_bf16.log 2>&1
echo "bf16 tests exit $?"; tail -4 $OUT/pytest_bf16.log | cut -c1-200
for dt in bf16 fp8; do
  timeout 300 python bench.py --workload config5 --dtype $dt --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_$dt.json 2> $OUT/bench_config5_$dt.err
  echo "config5 $dt exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_config5_$dt.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['whole_decode_frac'], d['verified'])"
done
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cat $OUT/bench.json | cut -c1-2500
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof5 -o prof -- python bench.py --workload config5 --dtype fp8 --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --min-seconds 0.1 > $OUT/bench5_under_rocprof.json 2> $OUT/prof5.err
python tools/rocpd_stats.py $OUT/prof5/prof_results.db $OUT/kernel_stats_config5_fp8.md | head -20 | cut -c1-200
find $OUT -size +20M -delete
