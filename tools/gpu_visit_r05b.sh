#!/bin/bash
# GPU visit r05b: clock stamps of the pipelined GEMM at the Whisper-large shapes; the corrected
# DMA-attention test
TAG=${1:-r05b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python tools/lp_clocks.py --lowp bf16 > $OUT/lp_clocks_bf16.txt 2>&1; echo "clocks bf16 $?"
timeout 300 python tools/lp_clocks.py --lowp fp8 > $OUT/lp_clocks_fp8.txt 2>&1; echo "clocks fp8 $?"
cat $OUT/lp_clocks_bf16.txt $OUT/lp_clocks_fp8.txt
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -x -k "dma_staging" > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest.log | cut -c1-300
