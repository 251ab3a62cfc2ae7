#!/bin/bash
TAG=${1:-exp4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/bench_gemm.py --only w1 --tiles 0,5 --variants 0,4,8,12 2>&1 | grep -v "^{" | tee $OUT/gemm.log
timeout 300 python tools/bench_gemm.py --only w2 --tiles 0,8 --variants 0,4,8,12 2>&1 | grep -v "^{" | tee -a $OUT/gemm.log
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU -d $OUT/pmc1 -o pmc --output-format csv -- python tools/bench_gemm.py --only w1 --tiles 5 --reps 3 > $OUT/pmc1.log 2>&1
f=$(find $OUT/pmc1 -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python tools/pmc_summary.py "$f" gemm
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES -d $OUT/pmc2 -o pmc --output-format csv -- python tools/bench_gemm.py --only w1 --tiles 5 --reps 3 > $OUT/pmc2.log 2>&1
f=$(find $OUT/pmc2 -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python tools/pmc_summary.py "$f" gemm
tail -3 $OUT/pmc2.log
