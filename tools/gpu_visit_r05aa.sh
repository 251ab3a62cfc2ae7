#!/bin/bash
# GPU visit r05aa: V by DMA + ds_read_b64_tr_b16 (no V^T image) -- tests, A/B, kernel stats
TAG=${1:-r05aa}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -x -k "dma_staging" > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -3 $OUT/pytest.log | cut -c1-300
for t in attn_bf16_dma=2 attn_bf16_dma=4 attn_bf16_dma=2 attn_bf16_dma=4; do
timeout 400 python bench.py --workload config5 --dtype fp8 --steps 10 --warmup 2 --min-seconds 1 --no-cpu-baseline --no-plain-leg --tune $t > $OUT/b_fp8_$t.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/b_fp8_$t.json')); print('fp8 $t', d['value'], d['ms_per_step'], d['verified'], d['verify'].get('identical'))"
done
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --workload config5 --dtype fp8 --steps 4 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-plain-leg --streams 1 --tune attn_bf16_dma=4 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats_config5_fp8_dma4.md > /dev/null; head -12 $OUT/kernel_stats_config5_fp8_dma4.md | cut -c1-170
find $OUT -size +20M -delete
