#!/bin/bash
# GPU visit: the whole -m gpu suite (device-side attention beam search, 2-rank tests, ...)
TAG=${1:-r02i}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -12 $OUT/pytest_gpu.log | cut -c1-250
