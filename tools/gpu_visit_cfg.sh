#!/bin/bash
# GPU visit: CTC head on the six-product GEMM -- full suite + bench + kernel stats
TAG=${1:-r02au}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest.log | cut -c1-300
for t in x6_linear=1 x6_linear=0 x6_linear=1; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-f32-mfma-leg --tune $t > $OUT/b_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_$t.json')); print('$t', d['value'], d['ms_per_step'], d['verified'])"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md > /dev/null; grep -E "ctc|x6_kernel<128, 0, 0|split" $OUT/kernel_stats.md | cut -c1-170; head -1 $OUT/kernel_stats.md
find $OUT -size +20M -delete
