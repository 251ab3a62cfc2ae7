#!/bin/bash
# Round-end visit A: the whole -m gpu suite, smoke(), the default bench line (with cpu_baseline)
TAG=${1:-r02fa}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $OUT/smoke.log | cut -c1-200
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"
python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['avg_launch_us'], d['verified'], d.get('f32_mfma_only'), d['cpu_baseline'])"
