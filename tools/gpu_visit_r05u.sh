#!/bin/bash
# GPU visit r05u: L2 / HBM traffic of the config 5 kernels (FETCH_SIZE, TCC hit / miss)
TAG=${1:-r05u}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
CMD="python bench.py --workload config5 --dtype fp8 --steps 1 --warmup 1 --no-cpu-baseline --no-plain-leg --min-seconds 0.05 --streams 1"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/p2 -o pmc --output-format csv -- $CMD > $OUT/p2.log 2>&1; echo "p2 $?"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/p5 -o pmc --output-format csv -- $CMD > $OUT/p5.log 2>&1; echo "p5 $?"
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/p3 -o pmc --output-format csv -- $CMD > $OUT/p3.log 2>&1; echo "p3 $?"
python - <<PY
import csv, glob, collections
for d in ('p2','p5','p3'):
    f = glob.glob('$OUT/'+d+'/**/*counter_collection.csv', recursive=True)
    if not f: print(d,'no csv'); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r['Kernel_Name']
        if 'attention_bf16' in n or 'gemm_lp' in n or 'vt_pack' in n or 'layernorm' in n:
            agg[n.split('::')[-1][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        print(d, k, {n: round(sum(v)/len(v)) for n, v in c.items()}, len(next(iter(c.values()))))
PY
find $OUT -name "*.csv" -size +8M -delete
