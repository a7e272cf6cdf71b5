#!/usr/bin/env python3
"""Merge the pmc_record.json files tools/profile_gpu.sh wrote (gpurun_out/prof_<tag>/) into
profiles/pmc_traffic.json, replacing the record of the same (workload, instances, samples, solver, kernel), and copy each
summary.txt / bench_line.json next to it under profiles/ with the given round prefix.
usage: tools/merge_pmc.py r3 gpurun_out/prof_r3_headline [gpurun_out/prof_r3_birdie ...]"""
import json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prefix, dirs = sys.argv[1], sys.argv[2:]
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
doc = json.load(open(path))
for d in dirs:
    rec = json.load(open(os.path.join(d, "pmc_record.json")))
    key = (rec["workload"], rec["instances"], rec["samples"], rec.get("solver"), rec.get("kernel"))
    # waves per launch from the dispatch's grid (the SQ_WAVES mean can include a neighbouring dispatch's waves)
    m = re.search(r"grid=(\d+)", open(os.path.join(d, "summary.txt")).read())
    if m:
        rec["sq_waves"] = int(m.group(1)) / 64.0
    # (records of earlier rounds carry no solver / kernel: a new record of the same workload and size replaces them)
    doc["runs"] = [r for r in doc["runs"] if (r.get("workload"), r.get("instances"), r.get("samples")) != key[:3]
                   or (r.get("solver") is not None and (r.get("solver"), r.get("kernel")) != key[3:])] + [rec]
    tag = "%s_%dx%d" % key[:3]
    if rec.get("solver") and "Caching" not in rec["solver"] and rec["workload"].startswith("superover"):
        tag += "_cacheless"
    shutil.copy(os.path.join(d, "summary.txt"), os.path.join(ROOT, "profiles", f"{prefix}_rocprofv3_summary_{tag}.txt"))
    shutil.copy(os.path.join(d, "bench_line.json"), os.path.join(ROOT, "profiles", f"{prefix}_profiled_bench_line_{tag}.json"))
    print("merged", key)
json.dump(doc, open(path, "w"), indent=1)
