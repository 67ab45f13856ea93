#!/usr/bin/env python3
"""GPU: where the per-frame C++ drop-in path spends its time (round 5): amos_host_frame_latency (tests/host/host_capi.cc) in its default
form, with the pyramid copied back every frame (round 4's default), with the eager mask pass (AMOS_MASK_GRAPH=0, round 4's default), and
the cost of one amos_match create + first search + destroy cycle (what a stack-constructed ORBmatcher cost per call site before the pool)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import importlib  # noqa: E402

import torch  # noqa: E402

synth = importlib.import_module("amos_slam_amd.synth")
import host_binding as hb  # noqa: E402

mask_mod = importlib.import_module("amos_slam_amd.mask")
out = {}
frames = np.stack([synth.frame(0, k) for k in range(16)])
bgr = np.repeat(frames[:, :, :, None], 3, axis=3)
out["orb_only_default"] = hb.host_frame_latency(frames, iters=200)
out["orb_only_pyramid_always"] = hb.host_frame_latency(frames, iters=200, pyramid_mode=1)

# matcher handle churn: create + one list-distance call + destroy, per cycle
d = np.random.default_rng(0).integers(0, 256, (1000, 32), dtype=np.uint8)
off = np.arange(0, 1001, dtype=np.int32) * 10
idx = np.random.default_rng(1).integers(0, 1000, 10000).astype(np.int32)
m = pkg.OrbMatcher(device=0)
m.list_distances(d, d, off, idx)
t0 = time.perf_counter()
for _ in range(200):
    m.list_distances(d, d, off, idx)
t1 = time.perf_counter()
for _ in range(200):
    mm = pkg.OrbMatcher(device=0)
    mm.list_distances(d, d, off, idx)
    mm.close()
t2 = time.perf_counter()
out["matcher_persistent_handle_search_ms"] = round((t1 - t0) / 200 * 1e3, 4)
out["matcher_create_search_destroy_ms"] = round((t2 - t1) / 200 * 1e3, 4)

# with the mask network: weights through a .pth, as the reference's class takes them
import tempfile  # noqa: E402

raw = mask_mod.MaskEngine(device="cpu", seed=0)
with torch.no_grad():
    b = raw.net.prediction_layers[0].conf_layer.bias
    bb = b.detach().view(3, 81).clone()
    bb[:, 1] += 5.0
    bb[1, 3] += 5.5
    b.copy_(bb.view(-1))
wpath = os.path.join(tempfile.mkdtemp(prefix="amos_r5_"), "w.pth")
torch.save(raw.net.state_dict(), wpath)
os.environ["AMOS_MASK_DEVICE"] = "cuda:0"
py_file = os.path.join(ROOT, "amos-slam_amd", "mask", "yolact_interface.py")
out["full_default"] = hb.host_frame_latency(frames, bgr, py_file, wpath, iters=100)
os.environ["AMOS_MASK_GRAPH"] = "0"
out["full_eager_mask_pass"] = hb.host_frame_latency(frames, bgr, py_file, wpath + "", iters=100)
print(json.dumps(out, indent=1))
