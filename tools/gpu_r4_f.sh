#!/bin/bash
cd $GRAFT_REPO_ROOT
for smp in 1840 5520 44100; do
  st=$((44100*5/smp)); wu=$((44100*3/smp))
  echo "== samples $smp steps $st warmup $wu"
  timeout 120 python bench.py --no-cpu-baseline --no-host-path --samples $smp --steps $st --warmup $wu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
