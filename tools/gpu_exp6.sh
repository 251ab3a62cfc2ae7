#!/bin/bash
TAG=${1:-exp6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for sl in 1 2 3 4; do
python - <<PY 2>&1 | grep -v "^{" | tee -a $OUT/gemm.log
import sys; sys.path.insert(0,'.')
from wenet_amd import _lib
L=_lib.lib(); L.wn_tune_set(b'gemm_sleep', $sl)
sys.argv=['x','--only','w1','--tiles','0,5','--variants','0,2']
print('sleep', $sl)
exec(open('tools/bench_gemm.py').read())
PY
done
