#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/memspill
export PYTHONPATH=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in memspill base; do
  if [ $lib = memspill ]; then export ACME_HIP_LIB=$GRAFT_REPO_ROOT/build_variants/libacme_hip_memspill.so; else unset ACME_HIP_LIB; fi
  timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM -d $GRAFT_REPO_ROOT/gpurun_out/memspill/$lib -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --samples 4410 --steps 2 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/memspill/$lib.log 2>&1
  python - $GRAFT_REPO_ROOT/gpurun_out/memspill/$lib $lib <<'PY'
import glob, sys, sqlite3
print("==", sys.argv[2])
for f in sorted(glob.glob(sys.argv[1] + "/*.db")) + sorted(glob.glob(sys.argv[1] + "/*/*.db")):
    con = sqlite3.connect(f)
    for r in con.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%acme_run_kernel%' group by counter_name"):
        print("%-28s dispatches=%d per_dispatch=%.6g" % r)
PY
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/memspill/$lib.log | cut -c1-200
done
find $GRAFT_REPO_ROOT/gpurun_out/memspill -name "*.db" -delete
