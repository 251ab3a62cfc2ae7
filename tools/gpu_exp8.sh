#!/bin/bash
TAG=${1:-exp8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "pipeline" 2>&1 | tail -5
for st in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams $st 2>$OUT/err$st.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams',$st, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])"
tail -2 $OUT/err$st.log | grep -v amdgpu.ids
done
