#!/bin/bash
# GPU visit: bf16 attention changes (bf16 Q/K/V, 2 sub-tiles per barrier) -- bf16 / fp8 suites,
# config-5 lines with A/B knobs, kernel stats
TAG=${1:-r02k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fp8.py -q > $OUT/pytest_bf16.log 2>&1
echo "bf16+fp8 tests exit $?"; tail -6 $OUT/pytest_bf16.log | cut -c1-220
for tune in "" "qkv_bf16=0" "qkv_bf16=0,attn_bf16_sub=1"; do
  timeout 300 python bench.py --workload config5 --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline --tune "$tune" > $OUT/b5.json 2> $OUT/b5.err
  python -c "
import json; d=json.load(open('$OUT/b5.json')); print('fp8 tune[$tune]', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
cp $OUT/b5.json $OUT/bench_config5_fp8_old_attention.json
timeout 300 python bench.py --workload config5 --dtype fp8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_fp8.json 2>/dev/null
timeout 300 python bench.py --workload config5 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_config5_bf16.json 2>/dev/null
python -c "
import json
for n in ('fp8','bf16'):
    d=json.load(open('$OUT/bench_config5_%s.json' % n)); r=d['roofline']; print(n, d['value'], d['ms_per_step'], r['achieved'], r['frac'], r['whole_decode_frac'])"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof5 -o prof -- python bench.py --workload config5 --dtype fp8 --steps 3 --warmup 1 --no-cpu-baseline --streams 1 --min-seconds 0.1 > $OUT/bench5_under_rocprof.json 2> $OUT/prof5.err
python tools/rocpd_stats.py $OUT/prof5/prof_results.db $OUT/kernel_stats_config5_fp8.md | head -12 | cut -c1-200
find $OUT -size +20M -delete
