#!/bin/bash
# GPU visit r05ab: transpose-read attention as the default -- bf16 / fp8 tests, config 5 parity + lines + kernel stats
TAG=${1:-r05ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fp8.py -q -x > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest.log | cut -c1-300
timeout 600 python -m pytest tests/test_gpu_bench_parity.py -q -x -k config5 > $OUT/pytest_c5.log 2>&1
echo "config5 parity exit $?"; tail -2 $OUT/pytest_c5.log | cut -c1-300
for dt in bf16 fp8; do
timeout 400 python bench.py --workload config5 --dtype $dt --no-cpu-baseline > $OUT/bench_config5_$dt.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config5_$dt.json')); r=d['roofline']; print('config5 $dt', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r.get('whole_decode_frac'))"
done
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --workload config5 --dtype bf16 --steps 4 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-plain-leg --streams 1 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats_config5_bf16.md > /dev/null; head -10 $OUT/kernel_stats_config5_bf16.md | cut -c1-150
find $OUT -size +20M -delete
