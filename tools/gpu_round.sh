#!/bin/bash
# One GPU-box visit: parity tests, bench line, rocprofv3 kernel-trace stats.
# Usage (from repo root on the GPU box): bash tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest_gpu.log
tail -5 $OUT/pytest_gpu.log
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
echo "== rocprofv3 kernel-trace stats"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
echo "rocprof exit $?"; cat $OUT/prof_bench.json
find $OUT/prof -name '*kernel_stats*' | head; 
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -40 "$f"
# keep only the small summaries (traces are big)
find $OUT/prof -name '*kernel_trace*' -size +20M -delete
