#!/bin/bash
TAG=${1:-exp2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 300 python tools/bench_gemm.py --only w1,w2,qkv,out,ctc --variants 0,1 2>&1 | grep -v "^{" | tee $OUT/gemm.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
cat $OUT/prof_bench.json
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -24
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc --output-format csv -- python tools/bench_gemm.py --only w1,w2 --reps 3 > $OUT/pmc1.log 2>&1
f=$(find $OUT/pmc1 -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python tools/pmc_summary.py "$f" gemm
