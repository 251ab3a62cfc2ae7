#!/bin/bash
TAG=${1:-exp3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 300 python tools/bench_gemm.py --only w1,ctc --tiles 0,5,6 2>&1 | grep -v "^{" | tee $OUT/gemm.log
timeout 300 python tools/bench_gemm.py --only w2,out --tiles 0,4,7 2>&1 | grep -v "^{" | tee -a $OUT/gemm.log
timeout 300 python tools/bench_gemm.py --only qkv --tiles 0,1,5,6,8 2>&1 | grep -v "^{" | tee -a $OUT/gemm.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
cat $OUT/prof_bench.json
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -12
