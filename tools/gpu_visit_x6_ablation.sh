#!/bin/bash
# GPU visit: x6 GEMM ablation (DMA ceiling / MFMA ceiling)
TAG=${1:-x6abl}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for pr in 0 1 2 3; do
echo "probe $pr"
timeout 200 python tools/bench_x6.py --only big,w1 --probe $pr 2>&1 | grep -v amdgpu.ids | cut -c1-200 | tee -a $OUT/probe.txt
done
