#!/bin/bash
# A/B of an environment switch on ONE box, alternating:  tools/ab_env.sh TAG VAR A B [bench args]
# -> gpurun_out/TAG/ab_<VAR>.txt (value, ms_per_step, plain per run)
TAG=$1; VAR=$2; A=$3; B=$4; shift 4
OUT=gpurun_out/$TAG; mkdir -p $OUT
ARGS="--no-cpu-baseline --no-f32-mfma-leg --no-nbest-leg --no-e2e-leg --no-two-stream-leg --no-clock-sample $*"
for rep in 1 2 3; do
  for v in $A $B; do
    env $VAR=$v timeout 600 python bench.py $ARGS > $OUT/ab_${VAR}_${v}_$rep.json 2>> $OUT/ab.err
    python - "$OUT/ab_${VAR}_${v}_$rep.json" "$VAR=$v" <<'PY' | tee -a $OUT/ab_$VAR.txt
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(sys.argv[2], 'value', d['value'], 'ms', d['ms_per_step'], 'plain', d.get('plain_decode', {}).get('value'), 'verified', d['verified'], 'ffn_us', d['roofline']['avg_launch_us'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
  done
done
tail -3 $OUT/ab.err 2>/dev/null | cut -c1-300
