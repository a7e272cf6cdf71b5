#!/bin/bash
# The two SQ counter passes of tools/profile_gpu.sh only (instruction mix, waits), for quick looks at a workload.
# usage: tools/profile_sq.sh <tag> [bench args...]
set -u
TAG=${1:-sq}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p "$OUT"
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_sq2" -o bench -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
python - "$OUT" "$TAG" <<'PY' | tee "$OUT/summary.txt"
import glob, sys, sqlite3
out = sys.argv[1]
print("==", sys.argv[2])
for grp in ("pmc_sq", "pmc_sq2"):
    for f in sorted(glob.glob(out + f"/{grp}/*.db")):
        con = sqlite3.connect(f)
        for r in con.execute("select counter_name, count(*), avg(value) from counters_collection "
                             "where kernel_name like '%acme%' group by counter_name"):
            print("%-28s dispatches=%d per_dispatch=%.6g" % r)
PY
rm -rf "$OUT"/pmc_sq "$OUT"/pmc_sq2
