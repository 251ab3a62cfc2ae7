#!/bin/bash
# kernel time of the attention forms in a config-2 decode: rocprofv3 stats of a short single-stream bench
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --no-clock-sample --no-plain-leg --no-nbest-leg --min-seconds 0.2 --streams 1 --tune attn_x6=$v > $OUT/b_$v.json 2> $OUT/b_$v.err
  python tools/rocpd_stats.py $OUT/kt_$v/prof_results.db $OUT/stats_attn_x6_$v.md > /dev/null
  echo "attn_x6=$v:"; grep -i "attention\|attn_x6" $OUT/stats_attn_x6_$v.md | cut -c1-150
  find $OUT -name "*.db" -size +20M -delete
done
