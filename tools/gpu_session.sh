#!/bin/bash
# ONE GPU call, any mix of steps, logs under gpurun_out/<tag>/.   usage (through gpurun):
#   bash tools/gpu_session.sh <tag> <step> [<step> ...]
# steps:
#   tests[:<pytest -k expression or file>]   pytest -m gpu (everything, or a selection)
#   ab[:<workload>[:steps[:warmup]]]         every build_variants/*.so on a bench workload, with output checksums
#   bench[:<bench.py args, comma separated>] one bench line (default: the driver's command, 20 steps after 5)
#   others                                   the bench lines of BASELINE configs 2, 4, 5 and the cache-less headline
#   profile[:<workload>]                     tools/profile_gpu.sh: kernel trace + PMC passes, steady state
#   mix[:<workload>]                         tools/profile_mix.sh: SQ_INSTS_VALU_* classes
#   hostreg                                  tools/ubench/hostreg: page-locking rates
#   probe:<python file>[:args]               any probe script under tools/
cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
  echo "=== $step"
  case $name in
    tests)
      if [ -z "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tee $out/pytest_gpu_full.txt | grep -v "^$" | tail -60
      elif [ -f "$arg" ]; then timeout 900 python -m pytest "$arg" -m gpu -x -q -s 2>&1 | tee -a $out/pytest_sel.txt | tail -40
      else timeout 900 python -m pytest tests -m gpu -x -q -s -k "$arg" 2>&1 | tee -a $out/pytest_sel.txt | tail -40; fi ;;
    ab)
      IFS=: read wl steps warm <<< "$arg"
      STEPS=${steps:-5} WARMUP=${warm:-3} BENCH_TIMEOUT=300 bash tools/gpu_ab2.sh $tag ${wl:-superover_grid} ;;
    bench)
      a=${arg//,/ }; [ -z "$a" ] && a="--steps 20 --warmup 5"
      timeout 600 python bench.py $a > $out/bench_$(echo "$a" | tr -c 'a-zA-Z0-9\n' '_').json 2> $out/bench.err
      tail -1 $out/bench_$(echo "$a" | tr -c 'a-zA-Z0-9\n' '_').json | cut -c1-600 ;;
    others) bash tools/gpu_final_benches.sh $tag ;;
    profile) bash tools/profile_gpu.sh ${tag}_${arg:-superover_grid} ${arg:+--workload $arg} 2>&1 | tail -30 ;;      # (STEPS / WARMUP from the environment: 3 / 3)
    mix) bash tools/profile_mix.sh ${tag}_${arg:-superover_grid} ${arg:+--workload $arg} --steps 2 --warmup 1 2>&1 | tail -20 ;;
    hostreg) timeout 300 tools/ubench/hostreg 2>&1 | tee $out/hostreg.txt ;;
    probe) IFS=: read f a <<< "$arg"; timeout 900 python tools/$f ${a//,/ } 2>&1 | tee $out/probe_${f%.py}.txt | tail -40 ;;
    *) echo "unknown step $step" ;;
  esac
done
