#!/bin/bash
# quick SQ counter passes for one bench configuration (run from repo root via gpurun)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/pmcq; rm -rf "$OUT"; mkdir -p "$OUT"
export PYTHONPATH=$ROOT; cd /tmp; export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 1 --samples 2205 $*"
i=0
for grp in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" \
  "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64" \
  "SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_INSTS_BRANCH SQ_LDS_DATA_FIFO_FULL" \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_CYCLES SQ_BUSY_CU_CYCLES" \
  "SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp -d $OUT/g$i -o b -- $BENCH > $OUT/g$i.log 2>&1
done
python - $OUT <<'PY'
import sqlite3, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/b_results.db")):
    con = sqlite3.connect(f)
    for r in con.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%acme%' group by counter_name"):
        print(f.split('/')[-2], r[0], r[1], "%.6g" % r[2])
PY
grep -l -i -E "error|invalid" $OUT/*.log | head
