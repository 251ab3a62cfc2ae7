#!/bin/bash
# GPU visit r05c: LDS-staged epilogue of the pipelined GEMM -- tests, clock stamps, config 5 lines,
# PMC passes of the config 5 kernels
TAG=${1:-r05c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_fp8.py -q -x > $OUT/pytest.log 2>&1
echo "tests exit $?"; tail -4 $OUT/pytest.log | cut -c1-300
timeout 300 python tools/lp_clocks.py --lowp bf16 > $OUT/lp_clocks_bf16.txt 2>&1; echo "clocks bf16 $?"
grep -E "^wh|middle block wave 0" $OUT/lp_clocks_bf16.txt
timeout 300 python tools/lp_clocks.py --lowp fp8 > $OUT/lp_clocks_fp8.txt 2>&1; echo "clocks fp8 $?"
grep -E "^wh|middle block wave 0" $OUT/lp_clocks_fp8.txt
timeout 600 python -m pytest tests/test_gpu_bench_parity.py -q -x -k config5 > $OUT/pytest_c5.log 2>&1
echo "config5 parity exit $?"; tail -2 $OUT/pytest_c5.log | cut -c1-300
for dt in bf16 fp8; do
timeout 400 python bench.py --workload config5 --dtype $dt --steps 10 --warmup 2 --min-seconds 1 --no-cpu-baseline --no-plain-leg > $OUT/bench_config5_$dt.json 2> $OUT/b.err
python -c "
import json; d=json.load(open('$OUT/bench_config5_$dt.json')); print('$dt', d['value'], d['ms_per_step'], d['verified'], d['verify'].get('identical'), d['roofline']['frac'])"
done
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o prof -- python bench.py --workload config5 --dtype fp8 --steps 4 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-plain-leg --streams 1 > $OUT/bench_under_rocprof.json 2> $OUT/prof.err
python tools/rocpd_stats.py $OUT/prof/prof_results.db $OUT/kernel_stats_config5_fp8.md > /dev/null; head -12 $OUT/kernel_stats_config5_fp8.md | cut -c1-200
bash tools/gpu_pmc_config5.sh $TAG/pmc bf16 2>&1 | tail -12
find $OUT -size +20M -delete
