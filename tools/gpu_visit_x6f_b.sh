#!/bin/bash
# GPU visit: fused six-product FFN -- ablation rows + PMC of the micro-benchmark
TAG=${1:-r03b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 300 python tools/bench_x6.py --only ffn > $OUT/bench_x6.txt 2>&1
grep -v amdgpu.ids $OUT/bench_x6.txt | cut -c1-250
CMD="python tools/bench_x6.py --only ffn --reps 5"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/p1 -o pmc --output-format csv -- $CMD > $OUT/p1.log 2>&1; echo "p1 $?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM -d $OUT/p4 -o pmc --output-format csv -- $CMD > $OUT/p4.log 2>&1; echo "p4 $?"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/p2 -o pmc --output-format csv -- $CMD > $OUT/p2.log 2>&1; echo "p2 $?"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT/p3 -o pmc --output-format csv -- $CMD > $OUT/p3.log 2>&1; echo "p3 $?"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o prof -- $CMD > $OUT/kt.log 2>&1; echo "kt $?"
python tools/pmc_table.py $OUT | grep -E "kernel|ffn_x6f|gemm_x6|x6_split" | cut -c1-260 | tee $OUT/pmc_table.md
python - <<PY
import csv, glob, collections
for d in ('p1','p4'):
    f = glob.glob('$OUT/'+d+'/**/*counter_collection.csv', recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if 'ffn_x6f' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        print(k, {n: round(sum(v)/len(v)) for n, v in c.items()}, len(next(iter(c.values()))))
PY
find $OUT -name "*.db" -size +20M -delete
find $OUT -name "*.csv" -size +8M -delete
