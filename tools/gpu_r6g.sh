#!/bin/bash
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6g; mkdir -p $out
export TMPDIR=/tmp
echo "=== throughput"
timeout 600 python tools/generic_shape_probe.py 8192 1102 "clipper chain, 1" < /dev/null 2>&1 | grep -v amdgpu.ids | tee $out/shape_probe.txt
echo "=== pmc sq only"
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d $GRAFT_REPO_ROOT/$out/pmc -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --no-other-workloads --steps 2 --warmup 2 --workload clipper_chain_20 > $GRAFT_REPO_ROOT/$out/pmc.log 2>&1
python - <<'PY'
import glob, sqlite3, collections, os
for f in glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r6g/pmc/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    per = collections.defaultdict(dict)
    for n, d, v in con.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like '%acme_coop_kernel%'"):
        per[n][d] = per[n].get(d, 0.0) + v
    vals = {n: sum(dv[i] for i in sorted(dv)[2:]) / max(1, len(sorted(dv)[2:])) for n, dv in per.items()}
    print(vals, "conflict frac", vals.get("SQ_LDS_BANK_CONFLICT", 0) / max(vals.get("SQ_LDS_IDX_ACTIVE", 1), 1))
PY
find $GRAFT_REPO_ROOT/$out -name "*.db" -delete
