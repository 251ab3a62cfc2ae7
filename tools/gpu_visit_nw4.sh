#!/bin/bash
# GPU visit: 128-row x6 tiles on four waves as the default -- tests + bench + kernel stats
TAG=${1:-r02aq}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_x6.py tests/test_gpu_bench_parity.py tests/test_gpu_ffn_fused.py -q > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest.log | cut -c1-200
timeout 200 python tools/bench_x6.py --only w1,w2,qkv,big,ffn 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee $OUT/bench_x6.txt
for t in x6_nw4=0 x6_nw4=3 x6_nw4=0; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mfma-leg --tune $t > $OUT/b_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_$t.json')); r=d['roofline']; print('$t', d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_us'], d['verified'])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -14 | cut -c1-170
find $OUT -size +20M -delete
