#!/bin/bash
# developer: instruction-fetch and issue counters of the cooperative kernel on the 20-unknown clipper chain (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out/${1:-coop_pmc}; rm -rf "$OUT"; mkdir -p "$OUT"
export PYTHONPATH=$ROOT; cd /tmp; export TMPDIR=/tmp
CMD="python $ROOT/tools/generic_shape_probe.py 8192 551 nn_20"
i=0
for grp in \
  "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
  "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp -d $OUT/g$i -o b -- $CMD > $OUT/g$i.log 2>&1 < /dev/null
done
python - $OUT <<'PY'
import sqlite3, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/*/b_results.db") + glob.glob(sys.argv[1] + "/*/b_results.db")):
    con = sqlite3.connect(f)
    try:
        tabs = [r[0] for r in con.execute("select name from sqlite_master where type='view' or type='table'")]
        q = "select counter_name, count(*), avg(value), kernel_name from counters_collection where kernel_name like '%coop%' group by counter_name"
        for r in con.execute(q):
            print(f.split('/')[-3] if f.count('/') > 6 else f.split('/')[-2], r[0], r[1], "%.6g" % r[2])
    except Exception as e:
        print("query failed", e, tabs[:10])
PY
grep -l -i -E "error|invalid" $OUT/*.log | head
