#!/usr/bin/env python3
"""profiles/<tag>_* from gpurun_out/prof_r5 (tools/collect_profiles_r5.sh; round 4: tools/make_profile_summary_r4.py): the bench line, rocprofv3's kernel stats of the
default command (MIOpen / rocBLAS kernels of the mask network included), and per-kernel counters of the HIP kernels:
HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), vector / scalar / LDS
instruction counts and the VALU-busy fraction.  Also rewrites profiles/traffic.json, which bench.py quotes as
roofline.traffic (labelled with its source)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/prof_r5"
shutil.copy(max(glob.glob(f"{src}/trace/*/*kernel_stats.csv"), key=os.path.getmtime), f"profiles/{tag}_kernel_stats.csv")
line = [l for l in open(f"{src}/bench.json") if l.startswith("{")][-1]
open(f"profiles/{tag}_bench.json", "w").write(line)
bench = json.loads(line)
leg = bench.get("extract_match_leg", bench)
frames = leg.get("frames_per_launch", bench["config"]["frames_per_launch"])
per_kernel = collections.defaultdict(dict)
for d in sorted(glob.glob(f"{src}/pmc*/")):
    files = glob.glob(d + "*/*counter_collection.csv")
    if not files:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    rows_all = [r for r in csv.DictReader(open(max(files, key=os.path.getmtime))) if "amos::" in r["Kernel_Name"]]
    # the command also runs its one-frame latency block (bench.py drop_in_latency): only the launches of the batch size count here --
    # per kernel the rows within 12 x of its largest grid
    biggest = collections.defaultdict(int)
    for r in rows_all:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        biggest[k] = max(biggest[k], int(r["Grid_Size"]))
    for r in rows_all:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if int(r["Grid_Size"]) * 12 >= biggest[k]:   # (the pyramid levels differ by 9 x in grid size; a one-frame launch is 16 - 128 x smaller than a 128-frame one)
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, x in v.items():
            per_kernel[k][c] = {"launches": len(x), "avg_per_launch": round(sum(x) / len(x), 1)}
for k, v in per_kernel.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_per_launch_corrected"] = int((2 * v["FETCH_SIZE"]["avg_per_launch"] + v["WRITE_SIZE"]["avg_per_launch"]) * 1024)
    if "SQ_INSTS_VALU" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]["avg_per_launch"] > 0:
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = the launch's duration in shader clocks.  The packed 16-bit / byte-permute /
        # three-operand instructions these kernels are made of issue at 0.9 - 1.0 wave-instructions per clock and CU (4 SIMDs x one per 4 cycles;
        # 32-bit add / logic / fp32 FMA at 1.6 - 1.75: profiles/r04_valu_issue.md): instructions / (256 CUs x cycles) = utilisation of THAT ceiling
        cycles = v["GRBM_GUI_ACTIVE"]["avg_per_launch"] / 8.0
        v["duration_shader_cycles"] = round(cycles, 1)
        v["valu_issue_utilisation"] = round(v["SQ_INSTS_VALU"]["avg_per_launch"] / (256.0 * cycles), 4)
json.dump({"command": "rocprofv3 --pmc <set> (one pass per set) -- python3 bench.py --gpus 1 --config c2 --steps 3 --warmup 1 --cpu-frames 0",
           "sets": ["FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS", "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"],
           "frames_per_launch": frames,
           "note": "counter collection serialises kernels: these are per-launch figures of each kernel ALONE on the chip",
           "correction": "HBM bytes = FETCH_SIZE x 2 + WRITE_SIZE (KB units; the x 2 is the gfx950 correction of MI355X_MICROARCH.md, HBM section, "
                         "re-calibrated in round 1 with tools/fetch_calib.hip)",
           "kernels": per_kernel}, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
stage = {"import": ["amos::k_pyramid_level0_wide"], "pyramid": ["amos::k_pyramid_level<true>", "amos::k_pyramid_tail"], "fast": ["amos::k_fast_cells<16>", "amos::k_fast_cells<20>"],
         "octree": ["amos::k_octree"], "orient": ["amos::k_orient"], "blur": ["amos::k_blur"], "describe": ["amos::k_describe"],
         "match": ["amos::k_bf_best2_mfma<true", "amos::k_bf_best2<true>"]}
traffic = {"source": f"profiles/{tag}_pmc_summary.json", "c2": {}}
for st, prefixes in stage.items():
    ks = [k for k in per_kernel if any(k.startswith(p) for p in prefixes)]  # template arguments of the instance that ran
    tot = 0
    for k in ks:
        v = per_kernel.get(k)
        if v and "hbm_bytes_per_launch_corrected" in v:
            per_pass = v["FETCH_SIZE"]["launches"] / max(per_kernel["amos::k_octree"]["FETCH_SIZE"]["launches"], 1)  # launches of this kernel per pass
            tot += v["hbm_bytes_per_launch_corrected"] * per_pass
    traffic["c2"][st] = {"batch": frames, "hbm_bytes_per_launch": int(tot)}
    valu = sum(per_kernel[k]["SQ_INSTS_VALU"]["avg_per_launch"] * per_kernel[k]["SQ_INSTS_VALU"]["launches"] for k in ks if k in per_kernel and "SQ_INSTS_VALU" in per_kernel[k])
    n_pass = max(per_kernel["amos::k_octree"]["SQ_INSTS_VALU"]["launches"], 1) if "SQ_INSTS_VALU" in per_kernel.get("amos::k_octree", {}) else 1
    print(f"{st:9s} {tot / 1e6:8.1f} MB HBM, {valu / n_pass / 1e6:7.2f} M VALU wave-instructions per pass of {frames} frames")
json.dump(traffic, open("profiles/traffic.json", "w"), indent=1)

# ---- steady-state per-kernel table of the traced command (rocprofv3's own kernel_stats.csv covers the whole process,
# including MIOpen's solver search on the first call of every convolution shape, whose naive reference kernels dwarf
# everything else): the timed steps of the mask-on region and of the mask-off leg, from kernel_trace.csv
trace = max(glob.glob(f"{src}/trace/*/*kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))


def window(first_kernel, n_last):
    idx = [i for i, r in enumerate(rows) if first_kernel in r["Kernel_Name"]]
    if len(idx) < n_last + 1:
        return []
    return rows[idx[-n_last - 1]:idx[-1]]


def table(rs, path, what):
    acc = collections.defaultdict(list)
    for r in rs:
        acc[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    total = sum(sum(v) for v in acc.values())
    span = int(rs[-1]["End_Timestamp"]) - int(rs[0]["Start_Timestamp"]) if rs else 0
    with open(path, "w") as f:
        f.write(f"# {what}; window {span / 1e6:.3f} ms wall, {total / 1e6:.3f} ms of kernel time (kernels of different lanes overlap)\n")
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage"\n')
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            f.write(f'"{k[:160]}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / max(total, 1):.2f}\n')


lanes = bench["config"]["lanes_per_gpu"]
on = window("k_import_color_mask", 3 * lanes)  # the last three timed steps of the mask-on region
if on:
    table(on, f"profiles/{tag}_steady_mask_on_kernel_stats.csv", "last 3 steps of the default command's timed region (BASELINE configs[2], mask on)")
off = window("k_pyramid_level0_wide", 8 * leg.get("lanes_per_gpu", 4))  # eight steps of the mask-off leg
if off:
    table(off, f"profiles/{tag}_steady_mask_off_kernel_stats.csv", "8 steps of the extract+match leg (BASELINE configs[1], mask off)")
