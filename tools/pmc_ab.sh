#!/bin/bash
# SQ counters of every library under build_variants/ on one bench workload, steady state only (the
# dispatches of the warm-up steps are dropped), in one GPU call.   tools/pmc_ab.sh <tag> [bench args]
# Counter groups are separate rocprofv3 passes (--pmc only; no trace domains).
ROOT=${GRAFT_REPO_ROOT:-$PWD}; tag=$1; shift
OUT=$ROOT/gpurun_out/$tag; mkdir -p "$OUT"
export PYTHONPATH=$ROOT; cd /tmp; export TMPDIR=/tmp
WARM=${WARMUP:-3}; STEPS=${STEPS:-3}
GROUPS_=("SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
         "SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SMEM SQ_INST_CYCLES_SALU")
for so in $ROOT/build_variants/*.so; do
  name=$(basename $so .so); i=0
  for grp in "${GROUPS_[@]}"; do
    i=$((i+1)); [ -n "$ONLY_GROUP" ] && [ "$ONLY_GROUP" != "$i" ] && continue
    rm -rf $OUT/${name}_g$i
    ACME_HIP_LIB=$so timeout 300 rocprofv3 --pmc $grp -d $OUT/${name}_g$i -o b -- python $ROOT/bench.py --no-cpu-baseline --steps $STEPS --warmup $WARM "$@" > $OUT/${name}_g$i.log 2>&1
  done
done
python - $OUT $WARM <<'PY' | tee $OUT/pmc_ab.txt
import sqlite3, glob, sys, os, collections
out, warm = sys.argv[1], int(sys.argv[2])
for d in sorted(glob.glob(out + "/*_g*/")):
    for f in glob.glob(d + "/**/*.db", recursive=True):
        con = sqlite3.connect(f)
        rows = con.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like '%acme_run_kernel%' or kernel_name like '%acme_lane_kernel%' order by dispatch_id").fetchall()
        per = collections.defaultdict(dict)
        for n, disp, v in rows:
            per[n][disp] = per[n].get(disp, 0.0) + v
        for n, dv in sorted(per.items()):
            ids = sorted(dv)[warm:]               # drop the warm-up steps (the first of them is the cold one)
            if ids:
                print(os.path.basename(d.rstrip('/')), n, len(ids), "%.6g" % (sum(dv[i] for i in ids) / len(ids)))
PY
find "$OUT" -name "*.db" -delete
