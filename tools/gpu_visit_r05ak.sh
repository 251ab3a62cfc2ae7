#!/bin/bash
# GPU visit r05ak: the round's last seconds -- real-reference golden cases and the attention-mode
# tests with every prepared form switched on for the session (tests/conftest.py WN_TUNE)
TAG=${1:-r05ak}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
WN_TUNE="x6r_pro=2,attn_gload=1,ctc_wave=2,dwconv_tiled=1,attn_bf16_dma=5" timeout 30 python -m pytest -q -x --durations=5 \
  tests/test_gpu_parity.py -k "golden_case or attention_mode_vs_reference or bench" > $OUT/pytest_wn_tune.log 2>&1
echo "exit $?"; tail -12 $OUT/pytest_wn_tune.log | cut -c1-200
