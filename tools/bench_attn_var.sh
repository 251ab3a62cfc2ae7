#!/bin/bash
# attention_x6_kernel with parts left out (WN_ABLATION build): kernel time per variant, encoder
# only (tools/prof_encoder.py: the outputs of these variants are wrong by design)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for v in "$@"; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o prof -- python tools/prof_encoder.py --tune attn_x6_var=$v > $OUT/b_$v.txt 2> $OUT/b_$v.err
  python tools/rocpd_stats.py $OUT/kt_$v/prof_results.db $OUT/stats_attn_var_$v.md > /dev/null
  echo "attn_x6_var=$v:"; grep -i "attention_x6" $OUT/stats_attn_var_$v.md | cut -c1-120
  find $OUT -name "*.db" -delete
done
