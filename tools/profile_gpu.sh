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
export BENCH_ARGS="$*"
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
python - "$OUT" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
import sqlite3
# rocprofv3 (ROCm 7.2) writes rocpd sqlite databases; summarise them as text
for f in sorted(glob.glob(out + "/trace/*.db")):
    con = sqlite3.connect(f)
    print("== kernel trace / stats (", os.path.basename(os.path.dirname(f)), ")")
    print("%-70s %6s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 6"):
        print("%-70s %6d %14.1f %14.1f %8.3f" % (r[0][:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))
    for r in con.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, max(vgpr_count), "
                         "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
                         "from kernels where name like '%acme%' group by name"):
        print("dispatches=%d avg_us=%.1f min_us=%.1f max_us=%.1f vgpr=%s agpr=%s sgpr=%s lds_bytes=%s scratch=%s grid=%s wg=%s" % r[1:])
vals = {}
for grp in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in sorted(glob.glob(out + f"/{grp}/*.db")):
        con = sqlite3.connect(f)
        print("== counters", grp)
        for r in con.execute("select counter_name, count(*), avg(value) from counters_collection "
                             "where kernel_name like '%acme_run_kernel%' or kernel_name like '%acme_lane_kernel%' group by counter_name"):   # (not the one-off solve / Jacobian kernels)
            print("%-28s dispatches=%d per_dispatch=%.6g" % r)
            vals[r[0]] = r[2]
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    import json, shlex
    a = shlex.split(os.environ.get("BENCH_ARGS", ""))
    def opt(name, default):
        return a[a.index(name) + 1] if name in a else default
    wl = opt("--workload", "superover_grid")
    n_def = {"diodeclipper_sweep": 4096, "birdie_grid": 2048}.get(wl, 8192)      # bench.py's defaults
    t_def = 176400 if wl == "birdie_grid" else 44100
    rec = {"runs": [{"workload": wl, "instances": int(opt("--instances", n_def)),
                     "samples": int(opt("--samples", t_def)),
                     "fetch_size_kb_per_launch": vals["FETCH_SIZE"], "write_size_kb_per_launch": vals["WRITE_SIZE"],
                     "sq_insts_valu_per_launch": vals.get("SQ_INSTS_VALU"),
                     "grbm_gui_active_per_launch": vals.get("GRBM_GUI_ACTIVE"), "sq_waves": vals.get("SQ_WAVES"),
                     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), mean over the launches of "
                               "python bench.py --no-cpu-baseline " + " ".join(a)}]}
    with open(out + "/pmc_traffic.json", "w") as fh:
        json.dump(rec, fh, indent=1)
PY
# the rocpd databases (tens of MB per pass) have served their purpose: gpurun merges at most 64 MiB back
find "$OUT" -name "*.db" -delete
