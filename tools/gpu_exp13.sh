#!/bin/bash
# bf16-storage form of the bf16 mode: parity (both forms), A/B bench lines, kernel stats
OUT=gpurun_out/exp13
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 400 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x > $OUT/pytest_bf16.log 2>&1; echo "pytest exit $?"; tail -12 $OUT/pytest_bf16.log | cut -c1-200
for st in 0 1; do
  timeout 100 python bench.py --workload config5 --dtype bf16 --tune bf16_store=$st --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c5_$st.json 2>/dev/null
  timeout 100 python bench.py --dtype bf16 --tune bf16_store=$st --steps 10 --warmup 3 --no-cpu-baseline > $OUT/c2_$st.json 2>/dev/null
  timeout 100 python bench.py --workload config4 --dtype bf16 --tune bf16_store=$st --steps 10 --warmup 3 --no-cpu-baseline > $OUT/c4_$st.json 2>/dev/null
done
for f in $OUT/c*.json; do echo "$f $(python -c "import json; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'])")"; done
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof5 -o prof -- python bench.py --workload config5 --dtype bf16 --tune bf16_store=1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof5/prof_results.db $OUT/kernel_stats_config5_bf16_stored.md | head -12 | cut -c1-190
find $OUT -size +20M -delete
