#!/bin/bash
TAG=${1:-exp7}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
for tune in "" "gemm_tile_conv=2,gemm_tile_glu=4" "gemm_tile_conv=6,gemm_tile_glu=2"; do
echo "== tune [$tune]"
rm -rf $OUT/prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --tune "$tune" > $OUT/prof_bench.json 2> $OUT/prof.err
python -c "import json;d=json.load(open('$OUT/prof_bench.json'));print(d['value'], d['ms_per_step'], d['roofline'])"
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -22 | cut -c1-170
done
