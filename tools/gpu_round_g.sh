#!/bin/bash
# Round-end visit r01g: all GPU parity tests, the headline bench line (with the CPU
# baseline), rocprofv3 kernel stats of the same command, and one bench line per
# other BASELINE config (fp32 and the bf16-operand mode).
TAG=${1:-r01g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?"; tail -4 $OUT/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cat $OUT/bench.json
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
echo "rocprof exit $?"
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -12 | cut -c1-200
for wl in config3 config4; do
  timeout 200 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_$wl.json 2>/dev/null
  timeout 200 python bench.py --workload $wl --dtype bf16 --steps 10 --warmup 2 --no-cpu-baseline > $OUT/bench_${wl}_bf16.json 2>/dev/null
done
timeout 200 python bench.py --workload config5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_config5.json 2>/dev/null
timeout 200 python bench.py --workload config5 --dtype bf16 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_config5_bf16.json 2>/dev/null
timeout 100 python bench.py --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_config2_bf16.json 2>/dev/null
for f in $OUT/bench_config*.json; do echo "$f: $(cut -c1-40 $f | head -1) $(python -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['dtype'][:4])")"; done
find $OUT -size +20M -delete
