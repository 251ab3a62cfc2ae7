#!/bin/bash
# kernel stats of an encoder-only loop (tools/prof_encoder.py) per tune setting:
#   tools/enc_stats.sh TAG GREP "key=value,..." ["key=value,..." ...]
TAG=$1; PAT=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
n=0
for v in "$@"; do
  n=$((n + 1))
  timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/kt_$n -o prof -- python tools/prof_encoder.py --tune "$v" > $OUT/enc_$n.txt 2> $OUT/enc_$n.err
  python tools/rocpd_stats.py $OUT/kt_$n/prof_results.db $OUT/enc_stats_$n.md > /dev/null
  echo "[$v] $(tail -1 $OUT/enc_$n.txt)"; grep -E "$PAT" $OUT/enc_stats_$n.md | cut -c1-130
  find $OUT -name "*.db" -delete
done
