#!/usr/bin/env python3
"""GPU: time of amos_corners_good_features_device + amos_corners_subpix_device + amos_lk_track_device on resident 640 x 480 frames
(Tracking.cc:894-896 as one chain), beside the CPU oracle on one host thread."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402

amos = entry.load_package()
synth = importlib.import_module("amos_slam_amd.synth")
import oracle_binding as ob  # noqa: E402

f0, f1 = synth.frame(9, 10), synth.frame(9, 11)
det = amos.CornerDetector()
lk = amos.LkTracker(640, 480, stream=det.stream)
st = torch.cuda.ExternalStream(det.stream)
d0, d1 = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
d_xy = torch.zeros((1000, 2), dtype=torch.float32, device="cuda")
d_n = torch.zeros(1, dtype=torch.int32, device="cuda")
d_next = torch.zeros((1000, 2), dtype=torch.float32, device="cuda")
d_st = torch.zeros(1000, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()


def chain(parts):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record(st)
    det.good_features_device(d0.data_ptr(), 640, 640, 480, d_xy.data_ptr(), 1000, d_n.data_ptr())
    ev[1].record(st)
    det.subpix_device(d0.data_ptr(), 640, 640, 480, d_xy.data_ptr(), count_ptr=d_n.data_ptr(), n=1000)
    ev[2].record(st)
    lk.track_device(d0.data_ptr(), 640, d1.data_ptr(), 640, d_xy.data_ptr(), 1000, d_next.data_ptr(), d_st.data_ptr(), None)
    ev[3].record(st)
    st.synchronize()
    parts.append([ev[k].elapsed_time(ev[k + 1]) for k in range(3)])


parts = []
for _ in range(12):
    chain(parts)
ms = np.mean(parts[2:], 0)
n = int(d_n.item())
t0 = time.perf_counter()
xy = ob.good_features_to_track(f0)
t1 = time.perf_counter()
xs = ob.corner_subpix(f0, xy)
t2 = time.perf_counter()
ob.lk_track(f0, f1, xs)
t3 = time.perf_counter()
print({"corners": n, "candidates": det.candidate_count(), "gpu_ms": {"good_features": round(float(ms[0]), 4), "subpix": round(float(ms[1]), 4), "lk": round(float(ms[2]), 4)},
       "cpu_oracle_ms": {"good_features": round((t1 - t0) * 1e3, 2), "subpix": round((t2 - t1) * 1e3, 2), "lk": round((t3 - t2) * 1e3, 2)}})
