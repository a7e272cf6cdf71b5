#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box.
# usage: tools/profile_gpu.sh <tag> [bench args...]   (run from the repo root via gpurun)
# Kernel trace/stats and every PMC group are separate runs (gpurun refuses --pmc combined
# with trace domains other than --kernel-trace/--stats).
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --no-cpu-baseline $*"
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
echo "== pmc: HBM read"
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "== pmc: HBM write"
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "== pmc: SQ (LDS conflicts, VALU, waits)"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq.log" 2>&1
echo "== pmc: SQ2 (instruction mix)"
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_sq2" -o bench -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
find "$OUT" -name "*.csv" | head -40
python - "$OUT" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)):
    print("--", f)
    print(open(f).read()[:3000])
for grp in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in sorted(glob.glob(out + f"/{grp}/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for row in csv.DictReader(open(f)):
            k = (row.get("Kernel_Name", "")[:60], row.get("Counter_Name", ""))
            acc[k][0] += float(row.get("Counter_Value", 0) or 0)
            acc[k][1] += 1
        print("--", f)
        for (kn, cn), (v, n) in sorted(acc.items()):
            if "acme" in kn:
                print(f"{kn:60s} {cn:24s} sum={v:.6g} dispatches={n} per_dispatch={v / max(n, 1):.6g}")
PY
