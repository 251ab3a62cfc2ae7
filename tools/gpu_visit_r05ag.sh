#!/bin/bash
# GPU visit r05ag: tools/check_prepared2.py (interleaved timing per key, bit-identity sweep)
TAG=${1:-r05ag}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 100 python tools/check_prepared2.py > $OUT/check_prepared2.txt 2>&1
echo "exit $?"; tail -32 $OUT/check_prepared2.txt | cut -c1-220
