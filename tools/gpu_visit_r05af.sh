#!/bin/bash
# GPU visit r05af: the last 2.7 GPU-minutes of round 3 -- tools/check_prepared.py
TAG=${1:-r05af}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 140 python tools/check_prepared.py > $OUT/check_prepared.txt 2>&1
echo "exit $?"; tail -20 $OUT/check_prepared.txt | cut -c1-250
