#!/bin/bash
TAG=${1:-exp1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q -k "prefix or beam or golden or decode" > $OUT/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $OUT/pytest.log
timeout 300 python tools/bench_gemm.py --only w1 --variants 0,1,2,3 2>&1 | tee $OUT/gemm_w1.log
for t in 0 1 2 4; do
python - <<PY 2>&1 | tee -a $OUT/gemm_tiles.log
import sys; sys.path.insert(0,'.')
from wenet_amd import _lib
L=_lib.lib(); L.wn_tune_set(b'gemm_tile', $t)
sys.argv=['x','--only','w2,out,qkv','--variants','0']
print('tile', $t)
exec(open('tools/bench_gemm.py').read())
PY
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.json 2> $OUT/prof.err
cat $OUT/prof_bench.json
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -12
# PMC pass on the w1 GEMM
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/pmc1 -o pmc --output-format csv -- python tools/bench_gemm.py --only w1 --reps 3 > $OUT/pmc1.log 2>&1
echo "pmc exit $?"; find $OUT/pmc1 -name '*.csv' | head
f=$(find $OUT/pmc1 -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'gemm' in r['Kernel_Name']:
        agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in v.items():
        print('   ',c, sum(vals)/len(vals), len(vals))
PY
rm -rf $OUT/prof/*.db.tmp
