#!/bin/bash
# GPU visit: MXFP8 quantiser + GEMM correctness, bf16 pipelined regression, micro-bench
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_fp8.py -x -q > $OUT/pytest_fp8.log 2>&1
echo "fp8 tests exit $?"; tail -25 $OUT/pytest_fp8.log | cut -c1-200
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -k "pipelined" > $OUT/pytest_pipelined.log 2>&1
echo "pipelined tests exit $?"; tail -3 $OUT/pytest_pipelined.log
timeout 600 python tools/bench_gemm.py --lowp fp8 --tiles 8 --variants 0 --only wh_w1,wh_w2 --reps 20 > $OUT/gemm_fp8.txt 2>&1
echo "bench exit $?"; grep -v "^{" $OUT/gemm_fp8.txt | cut -c1-150
timeout 600 python tools/bench_gemm.py --lowp fp8 --c-bf16 --tiles 8 --only wh_w1 --reps 20 > $OUT/gemm_fp8_mxc.txt 2>&1
echo "bench exit $?"; grep -v "^{" $OUT/gemm_fp8_mxc.txt | cut -c1-150
