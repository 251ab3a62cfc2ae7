#!/bin/bash
# GPU visit r05ae: V-tile swizzle for the transpose reads -- the whole suite, smoke, bench lines, LDS conflict counter
TAG=${1:-r05ae}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
bash tools/gpu_visit_final_a.sh $TAG
for dt in fp8 bf16; do
timeout 400 python bench.py --workload config5 --dtype $dt --no-cpu-baseline --no-plain-leg > $OUT/bench_config5_$dt.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config5_$dt.json')); print('config5 $dt', d['value'], d['ms_per_step'])"
done
CMD="python bench.py --workload config5 --dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-plain-leg --min-seconds 0.05"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d $OUT/p4 -o pmc --output-format csv -- $CMD > $OUT/p4.log 2>&1; echo "p4 $?"
python - <<PY
import csv, glob, collections, re
f = glob.glob('$OUT/p4/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n = r['Kernel_Name']
    if 'attention_bf16' in n:
        agg['attention'][r['Counter_Name']].append(float(r['Counter_Value']))
for k, c in agg.items():
    print(k, {n: round(sum(v)/len(v)) for n, v in c.items()})
PY
find $OUT -name "*.csv" -size +8M -delete
