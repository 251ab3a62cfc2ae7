#!/bin/bash
# GPU visit: fused fp32 FFN -- operator tests, model A/B, BASELINE-shaped parity, bench A/B
TAG=${1:-r02g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_ffn_fused.py -x -q -s > $OUT/pytest_ffn.log 2>&1
echo "ffn tests exit $?"; grep -E "^\[|passed|failed|Error|assert" $OUT/pytest_ffn.log | tail -12 | cut -c1-220
timeout 600 python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_fp8.py -x -q -s > $OUT/pytest_bench_parity.log 2>&1
echo "bench-parity exit $?"; grep -E "^\[config|passed|failed|Error" $OUT/pytest_bench_parity.log | tail -8 | cut -c1-330
for ff in 1 0; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --tune ffn_fused=$ff > $OUT/bench_ff$ff.json 2> $OUT/bench_ff$ff.err
  echo "bench ffn_fused=$ff exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_ff$ff.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['avg_launch_us'], r['whole_decode_frac'], d['verified'])"
done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --streams 1 > $OUT/bench_streams1.json 2>/dev/null
python -c "
import json; d=json.load(open('$OUT/bench_streams1.json')); r=d['roofline']; print('streams1', d['value'], d['ms_per_step'], r['achieved'], r['avg_launch_us'])"
for wl in config3 config4; do
  timeout 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$wl.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/bench_$wl.json')); r=d['roofline']; print('$wl', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['whole_decode_frac'], d['verified'])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -12 | cut -c1-200
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof2 -o prof -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --streams 2 --min-seconds 0.2 > $OUT/bench_under_rocprof_s2.json 2> $OUT/prof2.err
python tools/rocpd_stats.py $OUT/prof2/prof_results.db $OUT/kernel_stats_streams2.md | head -8 | cut -c1-200
find $OUT -size +20M -delete
