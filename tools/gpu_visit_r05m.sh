#!/bin/bash
# GPU visit r05m: the real data path (wav -> text) and batched streaming on the current code
TAG=${1:-r05m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python tools/bench_datapath.py --hours 20 --workers 4,8 --out $OUT/datapath_wav_to_text.json > $OUT/datapath.log 2>&1; echo "datapath $?"; tail -6 $OUT/datapath.log | cut -c1-250
timeout 600 python tools/bench_streaming.py --sessions 1,16,64,128 --left 4 --out $OUT/streaming_batched_left4.json > $OUT/streaming.log 2>&1; echo "streaming $?"; tail -6 $OUT/streaming.log | cut -c1-250
