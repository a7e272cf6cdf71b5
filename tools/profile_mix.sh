#!/bin/bash
# VALU instruction mix of a workload's run kernel (fp64 arithmetic against everything else).
# usage: tools/profile_mix.sh <tag> [bench args...]
set -u
TAG=${1:-mix}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/mix_$TAG
mkdir -p "$OUT"
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-host-path --no-other-workloads $*"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
    -d "$OUT/p1" -o bench -- $BENCH > "$OUT/p1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_EXP_GDS SQ_INSTS_FLAT SQ_INSTS_VSKIPPED \
    -d "$OUT/p2" -o bench -- $BENCH > "$OUT/p2.log" 2>&1
python - "$OUT" "$TAG" <<'PY' | tee "$OUT/summary.txt"
import glob, sys, sqlite3
out = sys.argv[1]
print("==", sys.argv[2])
for grp in ("p1", "p2"):
    for f in sorted(glob.glob(out + f"/{grp}/*.db")):
        con = sqlite3.connect(f)
        for r in con.execute("select counter_name, count(*), avg(value) from counters_collection "
                             "where kernel_name like '%acme_run_kernel%' or kernel_name like '%acme_lane_kernel%' group by counter_name"):
            print("%-28s dispatches=%d per_dispatch=%.6g" % r)
PY
tail -3 "$OUT/p1.log" "$OUT/p2.log" | grep -i "error\|invalid\|not" | head
find "$OUT" -name "*.db" -delete
