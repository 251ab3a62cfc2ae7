#!/bin/bash
# GPU visit r05k: QKV prologue fold -- tests, A/B
TAG=${1:-r05k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_ffn_fused.py tests/test_gpu_bench_parity.py -q -x > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -4 $OUT/pytest.log | cut -c1-300
for st in 2 1; do
for t in x6r_pro=0 x6r_pro=1 x6r_pro=0 x6r_pro=1; do
timeout 300 python bench.py --no-cpu-baseline --no-f32-mfma-leg --no-plain-leg --no-clock-sample --streams $st --tune $t > $OUT/b_s${st}_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_s${st}_$t.json')); print('streams $st $t', d['value'], d['ms_per_step'], d['verified'])"
done
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof1 -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --no-plain-leg --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof_s1.json 2> $OUT/prof1.err
python tools/rocpd_stats.py $OUT/prof1/prof_results.db $OUT/kernel_stats_streams1.md | grep -E "total|x6r_kernel|ffn_reduce" | cut -c1-160
find $OUT -name "*.db" -size +20M -delete
