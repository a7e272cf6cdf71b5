#!/bin/bash
# Collect the rocprofv3 evidence for one bench configuration on the GPU box -- STEADY STATE ONLY: every figure
# is a mean over the dispatches of the TIMED steps (the warm-up dispatches, the first of which is the cold one,
# are dropped), so that the summary's average duration is directly comparable with the bench line's kernel_ms.
# usage: tools/profile_gpu.sh <tag> [bench args...]   (run from the repo root via gpurun; STEPS / WARMUP env)
# Kernel trace/stats and every PMC group are separate runs (gpurun refuses --pmc combined with trace domains
# other than --kernel-trace/--stats).
set -u
TAG=${1:-r3}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export PYTHONPATH=$ROOT
cd /tmp && export TMPDIR=/tmp
export WARM=${WARMUP:-3} NSTEPS=${STEPS:-3}
BENCH="python $ROOT/bench.py --no-cpu-baseline --no-host-path --no-other-workloads --steps $NSTEPS --warmup $WARM $*"
export BENCH_ARGS="$*"
echo "== kernel trace + stats"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o bench -- $BENCH > "$OUT/trace.log" 2>&1
grep "^{\"metric\"" "$OUT/trace.log" | tail -1 > "$OUT/bench_line.json"      # (the log ends with the profiler's own lines)
echo "== pmc: HBM read"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "== pmc: HBM write"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "== pmc: SQ (LDS conflicts, VALU, waits)"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE \
    -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq.log" 2>&1
echo "== pmc: executed fp64 arithmetic"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 \
    -d "$OUT/pmc_f64" -o bench -- $BENCH > "$OUT/pmc_f64.log" 2>&1
echo "== pmc: SQ2 (waits, branches, fetch)"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_BRANCH SQ_IFETCH SQ_WAVES \
    -d "$OUT/pmc_sq2" -o bench -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
echo "== pmc: lanes (thread-cycles of the VALU instructions against their wave-cycles: EXEC-active lanes)"
timeout 300 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU \
    -d "$OUT/pmc_lanes" -o bench -- $BENCH > "$OUT/pmc_lanes.log" 2>&1
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import collections, glob, json, os, shlex, sqlite3, sys
out = sys.argv[1]
warm = int(os.environ["WARM"])
sel = "(name like '%acme_run_kernel%' or name like '%acme_lane_kernel%' or name like '%acme_coop_kernel%')"
print("command: python bench.py --no-cpu-baseline --steps %s --warmup %s %s" % (os.environ["NSTEPS"], warm, os.environ.get("BENCH_ARGS", "")))
print("(steady state: the %d warm-up dispatches of the run kernel are dropped from every figure below)" % warm)
kern = {}
for f in sorted(glob.glob(out + "/trace/**/*.db", recursive=True)):
    con = sqlite3.connect(f)
    print("== rocprofv3 --kernel-trace --stats: top kernels (all dispatches)")
    print("%-86s %6s %14s %14s %8s" % ("kernel", "calls", "total_ms", "avg_ms", "pct"))
    for r in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels limit 5"):
        print("%-86s %6d %14.3f %14.3f %8.3f" % (r[0][:86], r[1], r[2] / 1e3, r[3] / 1e3, r[4]))      # (the table holds microseconds)
    rows = con.execute("select name, start, end, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x "
                       "from kernels where " + sel + " order by start").fetchall()
    steady = rows[warm:]
    if steady:
        d = [(r[2] - r[1]) / 1e3 for r in steady]
        r = steady[0]
        kern = dict(name=r[0], dispatches=len(d), avg_us=sum(d) / len(d), min_us=min(d), max_us=max(d))
        print("== run kernel, timed steps only: dispatches=%d avg_us=%.1f min_us=%.1f max_us=%.1f vgpr=%s agpr=%s sgpr=%s lds_bytes=%s scratch=%s grid=%s wg=%s"
              % (len(d), kern["avg_us"], kern["min_us"], kern["max_us"], r[3], r[4], r[5], r[6], r[7], r[8], r[9]))
        print("   kernel: " + r[0])
vals = {}
for grp in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_f64", "pmc_sq2", "pmc_lanes"):
    for f in sorted(glob.glob(out + f"/{grp}/**/*.db", recursive=True)):
        con = sqlite3.connect(f)
        print("== counters", grp, "(per dispatch, timed steps only)")
        per = collections.defaultdict(dict)
        for n, disp, v in con.execute("select counter_name, dispatch_id, value from counters_collection where "
                                      + sel.replace("name", "kernel_name") + " order by dispatch_id"):
            per[n][disp] = per[n].get(disp, 0.0) + v
        for n, dv in sorted(per.items()):
            ids = sorted(dv)[warm:]
            if ids:
                vals[n] = sum(dv[i] for i in ids) / len(ids)
                print("%-28s dispatches=%d per_dispatch=%.6g" % (n, len(ids), vals[n]))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    a = shlex.split(os.environ.get("BENCH_ARGS", ""))
    opt = lambda name, default: a[a.index(name) + 1] if name in a else default      # noqa: E731
    wl = opt("--workload", "superover_grid")
    n_def = {"diodeclipper_sweep": 4096, "birdie_grid": 2048}.get(wl, 8192)      # bench.py's defaults
    t_def = 176400 if wl == "birdie_grid" else 4410 if wl.startswith("clipper_chain") else 44100
    try:        # what exactly ran: the traced run's own bench line (solver stack, kernel variant, its kernel_ms)
        line = json.loads(open(out + "/bench_line.json").read().strip().splitlines()[-1])
    except Exception:
        line = {}
    rec = {"workload": wl, "instances": int(opt("--instances", n_def)), "samples": int(opt("--samples", t_def)),
           "solver": line.get("config", {}).get("solver"), "kernel": line.get("roofline", {}).get("kernel"),
           "bench_kernel_ms": line.get("roofline", {}).get("kernel_ms"),
           "sq_insts_valu_fma_f64_per_launch": vals.get("SQ_INSTS_VALU_FMA_F64"), "sq_insts_valu_mul_f64_per_launch": vals.get("SQ_INSTS_VALU_MUL_F64"),
           "sq_insts_valu_add_f64_per_launch": vals.get("SQ_INSTS_VALU_ADD_F64"), "sq_insts_valu_trans_f64_per_launch": vals.get("SQ_INSTS_VALU_TRANS_F64"),
           "fetch_size_kb_per_launch": vals["FETCH_SIZE"], "write_size_kb_per_launch": vals["WRITE_SIZE"],
           "sq_insts_valu_per_launch": vals.get("SQ_INSTS_VALU"), "sq_insts_salu_per_launch": vals.get("SQ_INSTS_SALU"),
           "sq_insts_lds_per_launch": vals.get("SQ_INSTS_LDS"), "sq_insts_branch_per_launch": vals.get("SQ_INSTS_BRANCH"),
           "sq_lds_bank_conflict_per_launch": vals.get("SQ_LDS_BANK_CONFLICT"), "sq_lds_idx_active_per_launch": vals.get("SQ_LDS_IDX_ACTIVE"),
           "grbm_gui_active_per_launch": vals.get("GRBM_GUI_ACTIVE"), "sq_waves": vals.get("SQ_WAVES"),
           "sq_thread_cycles_valu_per_launch": vals.get("SQ_THREAD_CYCLES_VALU"), "sq_active_inst_valu_per_launch": vals.get("SQ_ACTIVE_INST_VALU"),
           "sq_wave_cycles_per_launch": vals.get("SQ_WAVE_CYCLES"), "sq_wait_any_per_launch": vals.get("SQ_WAIT_ANY"),
           "sq_wait_inst_any_per_launch": vals.get("SQ_WAIT_INST_ANY"), "sq_active_inst_any_per_launch": vals.get("SQ_ACTIVE_INST_ANY"),
           "kernel_avg_ms_profiled": kern.get("avg_us", 0.0) / 1e3 if kern else None,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ groups (separate passes) and --kernel-trace --stats; "
                     "means over the dispatches of the TIMED steps of python bench.py --no-cpu-baseline --steps %s --warmup %s %s"
                     % (os.environ["NSTEPS"], warm, " ".join(a))}
    with open(out + "/pmc_record.json", "w") as fh:
        json.dump(rec, fh, indent=1)
    if vals.get("SQ_LDS_IDX_ACTIVE"):
        print("lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.4f" % (vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"]))
    if vals.get("GRBM_GUI_ACTIVE") and vals.get("SQ_INSTS_VALU"):
        print("valu_issue_frac = SQ_INSTS_VALU / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs / 4) = %.4f" % (vals["SQ_INSTS_VALU"] / (1024 * vals["GRBM_GUI_ACTIVE"] / 8 / 4)))
    if vals.get("SQ_THREAD_CYCLES_VALU") and vals.get("SQ_ACTIVE_INST_VALU"):
        print("exec_active_lane_frac = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) = %.4f" % (vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_ACTIVE_INST_VALU"])))
    if vals.get("SQ_WAVE_CYCLES") and vals.get("SQ_WAIT_ANY") is not None:
        print("a wave's life: issuing SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = %.3f, waiting SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f, in s_waitcnt SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f"
              % (vals.get("SQ_ACTIVE_INST_ANY", 0) / vals["SQ_WAVE_CYCLES"], vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"], vals.get("SQ_WAIT_INST_ANY", 0) / vals["SQ_WAVE_CYCLES"]))
    if vals.get("SQ_INSTS_VALU_FMA_F64") is not None and kern:
        fl = 64.0 * (2 * vals["SQ_INSTS_VALU_FMA_F64"] + vals.get("SQ_INSTS_VALU_MUL_F64", 0) + vals.get("SQ_INSTS_VALU_ADD_F64", 0) + vals.get("SQ_INSTS_VALU_TRANS_F64", 0))
        print("executed fp64: 64 lanes x (2 FMA + MUL + ADD + TRANS) = %.4g flop per launch = %.2f TFLOP/s over %.1f ms (masked lanes counted: an upper bound of useful work)"
              % (fl, fl / (kern["avg_us"] * 1e-6) / 1e12, kern["avg_us"] / 1e3))
PY
# the rocpd databases (tens of MB per pass) have served their purpose: gpurun merges at most 64 MiB back
find "$OUT" -name "*.db" -delete
