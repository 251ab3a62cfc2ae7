#!/bin/bash
# bf16-operand GEMM: parity tests, tile micro-bench, config-5 / config-2 bench lines
OUT=gpurun_out/exp11
mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_bf16.py -q -m gpu > $OUT/pytest_bf16.log 2>&1; echo "pytest exit $?"; tail -25 $OUT/pytest_bf16.log
timeout 200 python tools/bench_gemm.py --bf16 --tiles 0,1,7 --only wh_w1,wh_w2,wh_qkv,wh_out,w1_512,w2_512 --reps 10 > $OUT/gemm_bf16_big.log 2>&1; grep -v "^{" $OUT/gemm_bf16_big.log
timeout 200 python tools/bench_gemm.py --bf16 --tiles 0,1,5,7 --only w1,w2,qkv,out,ctc,sub_out --reps 20 > $OUT/gemm_bf16_small.log 2>&1; grep -v "^{" $OUT/gemm_bf16_small.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof5 -o prof -- python bench.py --workload config5 --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_config5_bf16.json 2> $OUT/bench5.err; echo "bench5 exit $?"; cat $OUT/bench_config5_bf16.json; tail -3 $OUT/bench5.err
python tools/rocpd_stats.py $OUT/prof5/prof_results.db $OUT/kernel_stats_config5_bf16.md | head -12 | cut -c1-200
find $OUT/prof5 -size +20M -delete
timeout 200 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_config2_bf16.json 2> $OUT/bench2b.err; echo "bench2 bf16 exit $?"; cat $OUT/bench_config2_bf16.json
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_config2_fp32.json 2> $OUT/bench2.err; echo "bench2 fp32 exit $?"; cat $OUT/bench_config2_fp32.json
