#!/bin/bash
# GPU visit r06a (first visit of the next round): the five kernel forms of DESIGN.md section 7
# (bit-identical and +2.1 % on the headline in the last seconds of round 3, profiles/r05af-r05al,
# but never under the WHOLE suite): the WN_EXPERIMENTAL legs, each key A/B against the default
# on THIS box.  Then: flip the defaults (g_x6r_pro = 2, g_attn_gload = 1, g_ctc_wave = 2,
# g_dwconv_tiled = 1, g_attn_bf16_dma = 5) and run tools/gpu_visit_suite.sh.  ~9 GPU-minutes.
TAG=${1:-r06a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
WN_EXPERIMENTAL=1 timeout 900 python -m pytest -q -x \
  "tests/test_gpu_bf16.py::test_bf16_attention_dma_staging_is_bit_identical" \
  "tests/test_gpu_ffn_fused.py::test_qkv_prologue_fold_is_bit_identical_to_the_reduce_launch" \
  "tests/test_gpu_ffn_fused.py::test_encoder_with_folded_relpos_attention_matches_the_two_contraction_form" \
  "tests/test_gpu_parity.py::test_end_to_end_vs_oracle_ragged_batch" > $OUT/pytest_experimental.log 2>&1
echo "experimental legs exit $?"; tail -4 $OUT/pytest_experimental.log | cut -c1-300
B="python bench.py --no-cpu-baseline --no-plain-leg --no-f32-mfma-leg --no-clock-sample"
for t in "" "x6r_pro=2" "attn_gload=1" "ctc_wave=2" "dwconv_tiled=1" "x6r_pro=2,attn_gload=1,ctc_wave=2,dwconv_tiled=1" ""; do
  n=$(echo "${t:-default}" | tr ',=' '__')
  timeout 300 $B ${t:+--tune $t} > $OUT/bench_config2_$n.json 2>> $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/bench_config2_$n.json')); print('config2 ${t:-default}', d['value'], d['ms_per_step'], d['verified'])"
done
for t in "beam_cu_mask=8001" "beam_cu_mask=8032" "beam_cu_mask=8032,x6_conv_cus=248" "beam_cu_mask=8001,x6_conv_cus=248" "beam_cu_mask=16016,x6_conv_cus=240" ""; do
  n=$(echo "${t:-default_b}" | tr ',=' '__')
  timeout 300 $B ${t:+--tune $t} > $OUT/bench_config2_$n.json 2>> $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/bench_config2_$n.json')); print('config2 ${t:-default}', d['value'], d['ms_per_step'], d['verified'])"
done
for dt in fp8 bf16; do
  for t in "" "attn_bf16_dma=5" ""; do
    n=$(echo "${t:-default}" | tr ',=' '__')
    timeout 300 python bench.py --workload config5 --dtype $dt --no-cpu-baseline --no-plain-leg ${t:+--tune $t} > $OUT/bench_config5_${dt}_$n.json 2>> $OUT/b.err
    python -c "
import json; d=json.load(open('$OUT/bench_config5_${dt}_$n.json')); print('config5 $dt ${t:-default}', d['value'], d['ms_per_step'])"
  done
done
tail -n 3 $OUT/b.err | cut -c1-300
