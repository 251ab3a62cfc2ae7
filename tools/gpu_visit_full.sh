#!/bin/bash
# GPU visit: the whole -m gpu suite + smoke()
TAG=${1:-r02z}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1
echo "tests exit $?"; tail -4 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log | cut -c1-200
