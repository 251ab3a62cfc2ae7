#!/bin/bash
TAG=${1:-r02am}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_bench_parity.py -q -s 2>&1 | grep -E "^\[config|passed|failed|Error" | cut -c1-600
