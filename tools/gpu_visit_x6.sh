#!/bin/bash
# GPU visit: six-product fp32 GEMM -- tests + micro-benchmark
TAG=${1:-x6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_x6.py -q -s > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -E "max \|err\||passed|failed|Error|error" $OUT/pytest.log | cut -c1-200 | head -40
timeout 300 python tools/bench_x6.py > $OUT/bench_x6.txt 2>&1
cat $OUT/bench_x6.txt | cut -c1-250
