#!/bin/bash
# ONE parameterised GPU visit (replaces the per-visit scripts of rounds 1-3):
#
#   gpurun --timeout S -- 'tools/gpu_visit.sh TAG STEP [STEP ...]'
#
# every STEP is one word, fields separated by ':' (use '+' for a blank inside a field):
#   suite                       the whole `pytest -m gpu` suite (serial, like the driver's run)
#   tests:<pytest args>         e.g. tests:tests/test_gpu_rescore.py+-k+bit
#   smoke                       __graft_entry__.smoke()
#   bench:<workload>[:<args>]   bench.py --workload <workload> <args> -> bench_<workload>[_n].json
#   stats:<workload>[:<args>]   rocprofv3 --kernel-trace --stats of a short bench run, one decode in
#                               flight (stats2: two) -> kernel_stats_<workload>_streams<n>.md
#   pmc:<workload>[:<args>]     the three PMC passes + kernel trace (tools/pmc_table.py)
#                               -> pmc_table_<workload>.md
#   probe                       launch-cost probe (one-wave-per-SIMD kernels on slow boxes)
#   py:<script>[:<args>]        python <script> <args> -> <script>.txt
# env WN_TUNE / WN_EXPERIMENTAL pass through.  Everything lands in gpurun_out/TAG/.
TAG=${1:?tag}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ulimit -c 0
SHORT="--steps 5 --warmup 2 --no-cpu-baseline --no-f32-mfma-leg --no-clock-sample --no-plain-leg --no-nbest-leg --no-e2e-leg --no-two-stream-leg --min-seconds 0.2"
n=0
for step in "$@"; do
  n=$((n + 1))
  IFS=':' read -r kind a1 a2 <<< "$step"
  a1=${a1//+/ }; a2=${a2//+/ }
  case $kind in
    suite)
      timeout 2400 python -m pytest tests -m gpu -q -x --durations=25 > $OUT/pytest_suite.log 2>&1
      echo "[$n] suite exit $?"; tail -4 $OUT/pytest_suite.log | cut -c1-300 ;;
    tests)
      timeout 1500 python -m pytest -q -x $a1 > $OUT/pytest_$n.log 2>&1
      echo "[$n] tests ($a1) exit $?"; tail -4 $OUT/pytest_$n.log | cut -c1-300 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)
      f=$OUT/bench_${a1}_$n.json
      timeout 900 python bench.py --workload $a1 $a2 > $f 2>> $OUT/bench.err
      python - "$f" "$a1 $a2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print('bench', sys.argv[2], 'FAILED', e); sys.exit(0)
r = d['roofline']
print('bench', sys.argv[2], '| value', d['value'], 'ms', d['ms_per_step'], 'plain',
      d.get('plain_decode', {}).get('value'), '| roofline', r['achieved'], r['frac'],
      r.get('avg_launch_us'), '| verified', d['verified'], '| cpu',
      d.get('cpu_baseline', {}).get('value'))
PY
      ;;
    stats|stats2)
      st=1; [ $kind = stats2 ] && st=2      # decodes in flight (stats2: the headline's pipeline)
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt${st}_$a1 -o prof -- \
        python bench.py --workload $a1 $SHORT --streams $st $a2 > $OUT/stats${st}_$a1.json 2> $OUT/stats${st}_$a1.err
      python tools/rocpd_stats.py $OUT/kt${st}_$a1/prof_results.db $OUT/kernel_stats_${a1}_streams$st.md | head -34 | cut -c1-220
      find $OUT -name "*.db" -size +20M -delete ;;
    pmc)
      CMD="python bench.py --workload $a1 --steps 2 --warmup 1 --no-cpu-baseline --no-clock-sample --no-f32-mfma-leg --no-plain-leg --no-nbest-leg --no-e2e-leg --no-two-stream-leg --streams 1 --min-seconds 0.1 $a2"
      P=$OUT/pmc_$a1; mkdir -p $P
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $P/p1 -o pmc --output-format csv -- $CMD > $P/p1.log 2>&1; echo "p1 $?"
      timeout 600 rocprofv3 --pmc FETCH_SIZE -d $P/p2 -o pmc --output-format csv -- $CMD > $P/p2.log 2>&1; echo "p2 $?"
      timeout 600 rocprofv3 --pmc WRITE_SIZE -d $P/p3 -o pmc --output-format csv -- $CMD > $P/p3.log 2>&1; echo "p3 $?"
      timeout 600 rocprofv3 --kernel-trace --stats -d $P/kt -o prof -- $CMD > $P/kt.log 2>&1; echo "kt $?"
      # the roofline kernel of the workload (bench.py's roofline.kernel) and its record key
      dt=fp32; case "$a2" in *bf16*) dt=bf16 ;; *fp8*) dt=fp8 ;; esac
      case $a1:$dt in
        config2:fp32) re='ffn_x6f_kernel' ;;
        config5:fp32|config3:fp32|config4:fp32) re='gemm_x6_kernel<(128|256), 2, [13]' ;;
        *:bf16) re='gemm_lp_kernel<0, 3' ;;
        *) re='gemm_lp_kernel<1, 3' ;;
      esac
      python tools/pmc_table.py $P "$re" $a1:$dt > $OUT/pmc_table_${a1}_$dt.md; head -30 $OUT/pmc_table_${a1}_$dt.md | cut -c1-220
      cp $P/pmc_roofline_kernel.json $OUT/pmc_roofline_kernel_${a1}_$dt.json 2>/dev/null
      find $OUT -name "*.db" -size +20M -delete; find $OUT -name "*.csv" -size +8M -delete ;;
    probe)
      # launch-cost probe (tools/probes/launch_probe.hip): which resource shape pays on a slow box
      hipcc --offload-arch=gfx950 -O2 tools/probes/launch_probe.hip -o /tmp/launch_probe 2> /dev/null
      timeout 120 /tmp/launch_probe > $OUT/launch_probe_$n.txt 2>&1; tail -10 $OUT/launch_probe_$n.txt
      # what the box says about itself (slow boxes vs normal ones: clocks, power cap, partitions)
      { rocm-smi --showperflevel --showclocks --showpower --showmaxpower --showcomputepartition \
          --showmemorypartition --showfwinfo 2>&1; rocminfo 2>/dev/null | grep -i -E "name:|compute unit|max clock|wavefront|xnack" | head -40; } > $OUT/box_info_$n.txt 2>&1 ;;
    py)
      timeout 900 python $a1 $a2 > $OUT/$(basename $a1 .py)_$n.txt 2>&1
      echo "[$n] $a1 exit $?"; tail -12 $OUT/$(basename $a1 .py)_$n.txt | cut -c1-300 ;;
    *) echo "unknown step $step" ;;
  esac
done
tail -n 3 $OUT/bench.err 2>/dev/null | cut -c1-300
