#!/bin/bash
# GPU visit r02a: BASELINE-shaped parity tests (printing what they compared), the
# touched parity tests, the headline bench line, the 2-rank self-test of the N > 1
# path on one GPU, and a kernel-stats profile of the unchanged round-1 kernels.
TAG=${1:-r02a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_bench_parity.py -x -q -s > $OUT/pytest_bench_parity.log 2>&1
echo "bench-parity exit $?"; grep -E "^\[config|passed|failed|Error|assert" $OUT/pytest_bench_parity.log | head -30
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "golden_case or end_to_end or rescoring_vs_oracle or stream" > $OUT/pytest_parity.log 2>&1
echo "parity exit $?"; grep -E "^\[|passed|failed" $OUT/pytest_parity.log | tail -25
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench exit $?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
WN_BENCH_SHARE_GPU=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --min-seconds 0.5 > $OUT/bench_share2.json 2> $OUT/bench_share2.err
echo "share2 exit $?"; cat $OUT/bench_share2.json; tail -3 $OUT/bench_share2.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 --min-seconds 0.2 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
echo "rocprof exit $?"
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats.md | head -24 | cut -c1-220
find $OUT -size +20M -delete
