#!/bin/bash
# GPU visit: the whole -m gpu suite, smoke(), then the default bench.py
TAG=${1:-r03e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
# WN_XDIST=N: N pytest workers sharing the GPU, one test FILE per worker at a time (the tune
# knobs are per-process state and tests/test_gpu_dist.py uses fixed rendezvous ports); the
# driver's own round-end run is the plain serial command
XD=${WN_XDIST:+-n $WN_XDIST --dist loadfile}
timeout 1500 python -m pytest tests -m gpu -q -x --durations=40 $XD > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -5 $OUT/pytest.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > $OUT/bench_config2.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config2.json')); r=d['roofline']; print('config2', d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['verified'], d.get('f32_mfma_only',{}).get('value'), d.get('cpu_baseline',{}).get('value'))"
tail -n 3 $OUT/b.err | cut -c1-300
