#!/bin/bash
# prefix beam search: per-frame phase cycles of workgroup 0 (WN_PB_CYCLES), config 2's batch
TAG=$1; shift; OUT=gpurun_out/$TAG; mkdir -p $OUT
WN_PB_CYCLES=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-f32-mfma-leg --no-clock-sample --no-plain-leg --no-nbest-leg --no-e2e-leg --min-seconds 0.05 --streams 1 "$@" 2>&1 >/dev/null | grep "prefix beam wg0" | tail -3 | tee -a $OUT/pb_cycles.txt
