#!/usr/bin/env python3
"""Experiment (GPU): does it help to give the lanes unequal batch sizes so that they do not run in phase?
usage: stagger_exp.py "128,128,128,128" "160,128,128,96" ..."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
synth = importlib.import_module("amos_slam_amd.synth")
H, W = 480, 640
frames_np = synth.frames(0, 0, 256, H, W)
base = torch.from_numpy(frames_np).cuda()
dummies = []
DUMMY = os.environ.get("STAGGER_DUMMY", "")   # e.g. "0,1,0,1": dummy HIP streams created before each handle (shifts the stream -> hardware queue mapping)
for spec in sys.argv[1:]:
    sizes = [int(v) for v in spec.split(",")]
    lanes = []
    for li, n in enumerate(sizes):
        if DUMMY:
            for _ in range(int(DUMMY.split(",")[li % len(DUMMY.split(","))])):
                dummies.append(pkg.OrbMatcher(device=0))  # owns one new HIP stream
        ext = pkg.OrbExtractor(max_width=W, max_height=H, max_batch=n, device=0)
        mt = pkg.OrbMatcher(device=0, stream=ext.stream)
        _, d_desc, d_counts, cap = ext.batch_results_device()
        fr = base[:n] if n <= 256 else base.repeat((n + 255) // 256, 1, 1)[:n].contiguous()
        pq = torch.arange(n, dtype=torch.int32, device="cuda")
        pt = (pq - 1) % n
        dm = torch.zeros((n, cap, 4), dtype=torch.int32, device="cuda")
        lanes.append((ext, mt, fr, d_desc, d_counts, cap, pq, pt, dm, n))
    def step():
        for ext, mt, fr, d_desc, d_counts, cap, pq, pt, dm, n in lanes:
            ext.extract_batch_device(fr.data_ptr(), H * W, W, W, H, n)
            mt.bruteforce_best2_batch_device(d_desc, cap * 32, d_counts, pq.data_ptr(), pt.data_ptr(), n, cap, 256, dm.data_ptr())
    def sync():
        for l in lanes:
            l[0].sync()
        torch.cuda.synchronize()
    for _ in range(5):
        step()
    sync()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100):
            step()
        sync()
        dt = time.perf_counter() - t0
        res.append(sum(sizes) * 100 / dt)
    print(spec, [round(r) for r in res], flush=True)
    del lanes
