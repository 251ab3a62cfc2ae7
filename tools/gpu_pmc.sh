#!/bin/bash
# PMC evidence for the bench kernels (single stream so kernels do not overlap):
# pass 1 MFMA / wave-cycle counters, pass 2 FETCH_SIZE, pass 3 WRITE_SIZE.
TAG=${1:-pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clock-sample --no-f32-mfma-leg --streams 1 --min-seconds 0.1"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/p1 -o pmc --output-format csv -- $CMD > $OUT/p1.log 2>&1; echo "p1 $?"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/p2 -o pmc --output-format csv -- $CMD > $OUT/p2.log 2>&1; echo "p2 $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/p3 -o pmc --output-format csv -- $CMD > $OUT/p3.log 2>&1; echo "p3 $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o prof -- $CMD > $OUT/kt.log 2>&1; echo "kt $?"
python tools/pmc_table.py $OUT | tee $OUT/pmc_table.md
rm -f $OUT/p1/*counter_collection.csv.bak
