#!/usr/bin/env python3
"""profiles/<tag>_* from gpurun_out/final (see tools/collect_profiles.sh)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = "gpurun_out/final"
shutil.copy(max(glob.glob(f"{src}/trace/*/*kernel_stats.csv"), key=os.path.getmtime), f"profiles/{tag}_kernel_stats.csv")
line = [l for l in open(f"{src}/bench.log") if l.startswith("{")][-1]
open(f"profiles/{tag}_bench.json", "w").write(line)
bench = json.loads(line)
out = {}
for name, d in (("FETCH_SIZE_KB", "fetch"), ("WRITE_SIZE_KB", "write")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(max(glob.glob(f"{src}/{d}/*/*counter_collection.csv"), key=os.path.getmtime))):
        if "amos::" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[name] = {"launches": len(v), "avg": round(sum(v) / len(v), 1)}
for v in out.values():
    v["hbm_bytes_per_launch_corrected"] = int((2 * v["FETCH_SIZE_KB"]["avg"] + v["WRITE_SIZE_KB"]["avg"]) * 1024)
frames = bench["config"]["frames_per_launch"]
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0",
           "frames_per_launch": frames,
           "correction": "FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section), re-calibrated for 4 B/lane and 16 B/lane loads with "
                         "tools/fetch_calib.hip (1 GiB stream -> 524 299 KB either way); WRITE_SIZE exact",
           "kernels": out}, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
stage = {"import": [("amos::k_pyramid_level0_wide", 1)], "pyramid": [("amos::k_pyramid_level<true>", bench["config"]["n_levels"] - 1)], "fast": [("amos::k_fast_cells", 1)],
         "octree": [("amos::k_octree", 1)], "orient": [("amos::k_orient", 1)], "blur": [("amos::k_blur", 1)],
         "describe": [("amos::k_describe", 1)], "match": [("amos::k_bf_best2<true>", 1)]}
traffic = {"c2": {}}
for st, ks in stage.items():
    tot = sum(out[k]["hbm_bytes_per_launch_corrected"] * n for k, n in ks if k in out)
    traffic["c2"][st] = {"batch": frames, "hbm_bytes_per_launch": tot, "source": f"profiles/{tag}_pmc_summary.json (all launches of the stage in one pass)"}
    print(st, round(tot / 1e6, 1), "MB per launch of", frames, "frames")
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)
