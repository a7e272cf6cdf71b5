#!/bin/bash
# quick SQ counter pass for one bench configuration (run from repo root via gpurun)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
export PYTHONPATH=$ROOT; cd /tmp; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 1 --samples 2205 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS -d $OUT/a -o b -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 -d $OUT/b -o b -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH SQ_LDS_MEM_VIOLATIONS -d $OUT/c -o b -- $BENCH > $OUT/c.log 2>&1
python - $OUT <<'PY'
import sqlite3, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/b_results.db")):
    con = sqlite3.connect(f)
    for r in con.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%acme%' group by counter_name"):
        print(f.split('/')[-2], r[0], r[1], "%.6g" % r[2])
PY
tail -3 $OUT/b.log $OUT/c.log | grep -i -E "error|invalid|not" | head
