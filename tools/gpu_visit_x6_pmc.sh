#!/bin/bash
# GPU visit: PMC counters of the x6 GEMM micro-benchmark (MFMA busy, effective clock)
TAG=${1:-x6pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
CMD="python tools/bench_x6.py --only big,w1 --reps 3"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/p1 -o pmc --output-format csv -- $CMD > $OUT/p1.log 2>&1; echo "p1 $?"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/p2 -o pmc --output-format csv -- $CMD > $OUT/p2.log 2>&1; echo "p2 $?"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o prof -- $CMD > $OUT/kt.log 2>&1; echo "kt $?"
python - <<'PY'
import csv, glob, collections, sqlite3, os
out = os.environ.get('OUT', 'gpurun_out/x6pmc')
def load(d):
    f = glob.glob(os.path.join(out, d, '**', '*counter_collection.csv'), recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f[0])):
            agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return agg
p1, p2 = load('p1'), load('p2')
db = glob.glob(os.path.join(out, 'kt', '**', '*.db'), recursive=True)
dur = {}
if db:
    cur = sqlite3.connect(db[0]).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    try:
        for name, n, avg in cur.execute('select name, count(*), avg(end-start)/1e3 from kernels group by name'):
            dur[name] = (n, avg)
    except Exception as e:
        print('kernels table?', e, tabs[:20])
for k in p1:
    if 'x6' not in k and 'gemm_f32' not in k: continue
    c = p1[k]; m = lambda v: sum(v)/len(v) if v else float('nan')
    gui = m(c.get('GRBM_GUI_ACTIVE', [])) / 8
    d = dur.get(k, (0, float('nan')))
    print(k[:90], 'n', len(c.get('GRBM_GUI_ACTIVE', [])), 'avg_us', round(d[1],1), 'gui_cycles', round(gui),
          'clk_GHz', round(gui / (d[1]*1e3), 3) if d[1]==d[1] else None,
          'mfma_busy', round(m(c.get('SQ_VALU_MFMA_BUSY_CYCLES', [])) / (gui*1024), 3),
          {n: round(m(v) / max(m(c.get('SQ_WAVE_CYCLES', [1])), 1), 3) for n, v in c.items() if n.startswith('SQ_WAIT') or n.startswith('SQ_ACTIVE')},
          {n: round(m(v)) for n, v in p2.get(k, {}).items()})
PY
find $OUT -size +20M -delete
