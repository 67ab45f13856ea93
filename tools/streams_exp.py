#!/usr/bin/env python3
"""Experiment: S independent handles (own streams), each with B/S frames per step."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
synth = importlib.import_module("amos_slam_amd.synth")
B, W, H = 256, 640, 480
frames = torch.from_numpy(synth.frames(0, 0, B, H, W)).cuda()
for S in (1, 2, 4):
    b = B // S
    lanes = []
    for s in range(S):
        ext = pkg.OrbExtractor(max_batch=b)
        m = pkg.OrbMatcher(stream=ext.stream)
        _, d_desc, d_counts, cap = ext.batch_results_device()
        pq = torch.arange(b, dtype=torch.int32, device="cuda")
        pt = (pq - 1) % b
        out = torch.empty((b, cap, 4), dtype=torch.int32, device="cuda")
        lanes.append((ext, m, frames[s * b:(s + 1) * b], d_desc, d_counts, cap, pq, pt, out))
    def step():
        for ext, m, fr, d_desc, d_counts, cap, pq, pt, out in lanes:
            ext.extract_batch_device(fr.data_ptr(), H * W, W, W, H, b)
            m.bruteforce_best2_batch_device(d_desc, cap * 32, d_counts, pq.data_ptr(), pt.data_ptr(), b, cap, 256, out.data_ptr())
    def sync():
        for l in lanes: l[0].sync()
        torch.cuda.synchronize()
    for _ in range(3): step()
    sync(); t0 = time.perf_counter()
    for _ in range(10): step()
    sync(); dt = time.perf_counter() - t0
    print(f"S={S}: {B * 10 / dt:.0f} frames/s, {dt / 10 * 1e3:.3f} ms/step")
    del lanes
