#!/usr/bin/env python3
"""Single-frame latency of the mask pass (what `yolact::evalImage` costs per frame when Tracking.cc calls it frame by frame):
MaskEngine.eval_bgr_batch on ONE 640 x 480 BGR frame resident on the GPU, synchronised per call; then 4 and 32 frames per call."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
mask_mod = importlib.import_module("amos_slam_amd.mask")
eng = mask_mod.MaskEngine(device="cuda:0").prepare()
rng = np.random.default_rng(0)
for n in (1, 4, 32):
    frames = torch.as_tensor(rng.integers(0, 256, (n, 480, 640, 3), dtype=np.uint8), device="cuda:0")
    for _ in range(5):
        eng.eval_bgr_batch(frames, chunk=n)
    torch.cuda.synchronize()
    reps = 30 if n == 1 else 10
    t = time.perf_counter()
    for _ in range(reps):
        eng.eval_bgr_batch(frames, chunk=n)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print("mask pass, %2d frame(s) per call: %.2f ms per call, %.2f ms per frame" % (n, ms, ms / n), flush=True)

# the same through one captured HIP graph per batch size
for n in (1, 4):
    frames = torch.as_tensor(rng.integers(0, 256, (n, 480, 640, 3), dtype=np.uint8), device="cuda:0")
    want = eng.eval_bgr_batch(frames, chunk=n).clone()
    eng.capture_graph(batch=n)
    for _ in range(5):
        masks, found = eng.eval_bgr_graph(frames)
    torch.cuda.synchronize()
    same = bool(torch.equal(torch.where(found[:, None, None], masks, torch.zeros_like(masks)), want))
    reps = 50
    t = time.perf_counter()
    for _ in range(reps):
        eng.eval_bgr_graph(frames)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / reps * 1e3
    print("mask pass as one HIP graph, %d frame(s) per call: %.2f ms per call, %.2f ms per frame; same masks as the eager pass: %s" % (n, ms, ms / n, same), flush=True)
