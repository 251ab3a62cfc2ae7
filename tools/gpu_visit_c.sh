#!/bin/bash
# GPU visit: ablation / schedule variants of the pipelined bf16 GEMM (gemm_variant bits:
# 1 no stagger, 2 no DMA, 4 no fragment reads, 8 no MFMA, 16 deeper issue order)
TAG=${1:-r02c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 800 python tools/bench_gemm.py --lowp bf16 --tiles 8 --variants 0,1,16,17,2,4,6,8,10 --only wh_qkv,wh_w2n,sq8k --reps 10 > $OUT/gemm_variants.txt 2>&1
echo "bench exit $?"; grep -v "^{" $OUT/gemm_variants.txt | cut -c1-150
