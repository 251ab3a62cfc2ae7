#!/bin/bash
# GPU visit r05ah: tools/check_prepared3.py (config 5: attn_bf16_dma 5 against 4)
TAG=${1:-r05ah}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 80 python tools/check_prepared3.py > $OUT/check_prepared3.txt 2>&1
echo "exit $?"; tail -14 $OUT/check_prepared3.txt | cut -c1-220
