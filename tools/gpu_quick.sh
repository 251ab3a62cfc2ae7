#!/bin/bash
# quick visit: (optional pytest -k expr) + bench under rocprofv3 kernel stats
TAG=${1:-q}; KEXPR=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$KEXPR" ]; then
timeout 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
else
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
fi
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
cat $OUT/prof_bench.json
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -${3:-14}
