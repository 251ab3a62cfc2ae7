#!/bin/bash
# GPU visit r05al: the experimental legs of the two encoder bit-identity tests (WN_EXPERIMENTAL=1)
TAG=${1:-r05al}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
WN_EXPERIMENTAL=1 timeout 14 python -m pytest -q -x tests/test_gpu_ffn_fused.py -k "qkv_prologue or folded_relpos" > $OUT/pytest_experimental.log 2>&1
echo "exit $?"; tail -4 $OUT/pytest_experimental.log | cut -c1-200
