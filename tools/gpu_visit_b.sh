#!/bin/bash
# GPU visit: pipelined bf16 GEMM -- correctness + race screen, then the micro-benchmark
# against the register-staged kernels on the Whisper-large shapes.
TAG=${1:-r02b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_bf16.py -x -q -k "pipelined" > $OUT/pytest_pipelined.log 2>&1
echo "pipelined tests exit $?"; tail -15 $OUT/pytest_pipelined.log
timeout 600 python tools/bench_gemm.py --lowp bf16 --tiles 1,7,8 --only wh_w1,wh_w2,wh_qkv,wh_out --reps 20 > $OUT/gemm_lowp_f32c.txt 2>&1
echo "bench exit $?"; grep -v "^{" $OUT/gemm_lowp_f32c.txt | cut -c1-150
timeout 600 python tools/bench_gemm.py --lowp bf16 --c-bf16 --tiles 7,8 --only wh_w1,wh_qkv --reps 20 > $OUT/gemm_lowp_bf16c.txt 2>&1
echo "bench exit $?"; grep -v "^{" $OUT/gemm_lowp_bf16c.txt | cut -c1-150
