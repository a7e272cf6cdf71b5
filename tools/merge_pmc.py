#!/usr/bin/env python3
"""Merge the pmc_record.json files tools/profile_gpu.sh wrote (gpurun_out/prof_<tag>/) into
profiles/pmc_traffic.json, replacing the record of the same (workload, instances, samples), and copy each
summary.txt / bench_line.json next to it under profiles/ with the given round prefix.
usage: tools/merge_pmc.py r3 gpurun_out/prof_r3_headline [gpurun_out/prof_r3_birdie ...]"""
import json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix, dirs = sys.argv[1], sys.argv[2:]
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
doc = json.load(open(path))
for d in dirs:
    rec = json.load(open(os.path.join(d, "pmc_record.json")))
    key = (rec["workload"], rec["instances"], rec["samples"])
    # waves per launch from the dispatch's grid (the SQ_WAVES mean can include a neighbouring dispatch's waves)
    m = re.search(r"grid=(\d+)", open(os.path.join(d, "summary.txt")).read())
    if m:
        rec["sq_waves"] = int(m.group(1)) / 64.0
    doc["runs"] = [r for r in doc["runs"] if (r.get("workload"), r.get("instances"), r.get("samples")) != key] + [rec]
    tag = "%s_%dx%d" % key
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", f"{prefix}_rocprofv3_summary_{tag}.txt"))
    shutil.copy(os.path.join(d, "bench_line.json"), os.path.join(ROOT, "profiles", f"{prefix}_profiled_bench_line_{tag}.json"))
    print("merged", key)
json.dump(doc, open(path, "w"), indent=1)
