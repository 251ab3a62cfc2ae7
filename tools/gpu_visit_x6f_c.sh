#!/bin/bash
# GPU visit: fused six-product FFN -- tests + ablation rows
TAG=${1:-r03c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_x6.py -q -s -x -k "ffn or on_chip" > $OUT/pytest.log 2>&1
echo "tests exit $?"; grep -E "passed|failed|Error|error|assert" $OUT/pytest.log | cut -c1-200 | head -20
timeout 300 python tools/bench_x6.py --only ffn > $OUT/bench_x6.txt 2>&1
grep -v amdgpu.ids $OUT/bench_x6.txt | cut -c1-250
