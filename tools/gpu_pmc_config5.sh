#!/bin/bash
# PMC passes for the config5 (Whisper-large-v3 encoder) kernels, bf16 mode
TAG=${1:-r03s}
DT=${2:-bf16}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --workload config5 --dtype $DT --steps 1 --warmup 1 --no-cpu-baseline --no-plain-leg --min-seconds 0.05"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/p1 -o pmc --output-format csv -- $CMD > $OUT/p1.log 2>&1; echo "p1 $?"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d $OUT/p4 -o pmc --output-format csv -- $CMD > $OUT/p4.log 2>&1; echo "p4 $?"
python - <<PY
import csv, glob, collections, re
for d in ('p1','p4'):
    f = glob.glob('$OUT/'+d+'/**/*counter_collection.csv', recursive=True)
    if not f: continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        n = r['Kernel_Name']
        if 'attention_bf16' in n or 'gemm_lp' in n or 'layernorm' in n:
            m = re.search(r'(\\w+_kernel(<[^>]*>)?)', n)
            agg[m.group(1) if m else n[-70:]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        print(k, {n: round(sum(v)/len(v)) for n, v in c.items()}, len(next(iter(c.values()))))
PY
find $OUT -name "*.csv" -size +8M -delete
