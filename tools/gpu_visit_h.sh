#!/bin/bash
# GPU visit: streaming tests (single + batched sessions)
TAG=${1:-r02j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "chunk or stream" > $OUT/pytest_stream.log 2>&1
echo "pytest exit $?"; tail -25 $OUT/pytest_stream.log | cut -c1-220
