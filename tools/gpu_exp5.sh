#!/bin/bash
TAG=${1:-exp5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/bench_gemm.py --only w1 --tiles 0,5 --variants 0,16,32,48 2>&1 | grep -v "^{" | tee $OUT/gemm.log
