#!/bin/bash
# Round-end visit B: kernel stats (1 and 2 decodes in flight), the other BASELINE configs, PMC
TAG=${1:-r02fb}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
for st in 1 2; do
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof$st -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --streams $st --min-seconds 0.2 > $OUT/bench_under_rocprof_s$st.json 2> $OUT/prof$st.err
python tools/rocpd_stats.py $OUT/prof$st/prof_results.db $OUT/kernel_stats_streams$st.md | head -4 | cut -c1-170
done
for w in config3 config4; do
timeout 400 python bench.py --workload $w --no-cpu-baseline > $OUT/bench_$w.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_$w.json')); r=d['roofline']; print('$w', d['value'], d['ms_per_step'], r['achieved'], r['frac'], d['verified'], d.get('f32_mfma_only',{}).get('value'))"
done
for dt in bf16 fp8; do
timeout 400 python bench.py --workload config5 --dtype $dt --no-cpu-baseline > $OUT/bench_config5_$dt.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config5_$dt.json')); r=d['roofline']; print('config5 $dt', d['value'], d['ms_per_step'], r['achieved'], r['frac'])"
done
bash tools/gpu_pmc.sh $TAG/pmc 2>&1 | grep -E "^p[123] |^kt |gemm_x6|conv1_x3|ffn_reduce|x6_split" | cut -c1-200
find $OUT -name "*.db" -size +20M -delete
find $OUT -name "*.csv" -size +8M -delete
