#!/bin/bash
# GPU visit: x6 FFN in the model -- tests + bench + kernel stats
TAG=${1:-r02u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_x6.py tests/test_gpu_ffn_fused.py tests/test_gpu_bench_parity.py -q -s > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -E "x6 FFN vs|passed|failed|Error" $OUT/pytest.log | cut -c1-200 | head
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_us'], d['verified'], d.get('f32_mfma_only'))"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -12 | cut -c1-170
find $OUT -size +20M -delete
