#!/bin/bash
# GPU visit: row-LN fused GEMM -- tests, BASELINE-shaped parity, bench A/B
TAG=${1:-r02n}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_ffn_fused.py tests/test_gpu_bench_parity.py -q -s > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -E "^\[|passed|failed|Error" $OUT/pytest.log | tail -14 | cut -c1-300
for tune in "" "gemm_rowln=0"; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune "$tune" > $OUT/b.json 2> $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/b.json')); r=d['roofline']; print('tune[$tune]', d['value'], d['ms_per_step'], r['achieved'], r['avg_launch_us'], d['verified'])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -22 | cut -c1-170
find $OUT -size +20M -delete
