#!/usr/bin/env python3
"""Single-frame (host-buffer, PCIe-inclusive) latency of the drop-in API on one GPU."""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

amos = entry.load_package()
synth = importlib.import_module("amos_slam_amd.synth")
ext = amos.OrbExtractor()
m = amos.OrbMatcher()
frames = [synth.frame(0, k) for k in range(60)]
prev = None
for f in frames[:10]:
    k, d = ext.extract(f)
t0 = time.perf_counter()
for f in frames[10:]:
    k, d = ext.extract(f)
t1 = time.perf_counter()
for f in frames[10:]:
    k, d = ext.extract(f)
    if prev is not None:
        m.bruteforce_best2(d, prev)
    prev = d
t2 = time.perf_counter()
print(f"amos_orb_extract: {(t1 - t0) / 50 * 1e3:.3f} ms/frame ({50 / (t1 - t0):.0f} frames/s); "
      f"extract + bruteforce match: {(t2 - t1) / 50 * 1e3:.3f} ms/frame ({50 / (t2 - t1):.0f} frames/s)")
