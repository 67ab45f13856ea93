#!/usr/bin/env python3
"""GPU: the one-frame mask pass (MaskEngine.frame_session: pre-processing + network + detection + mask assembly as ONE HIP graph, pinned host
buffers) replayed N times; meant to run under `rocprofv3 --kernel-trace` (tools/r5_one_frame_trace.sh).  AMOS_ONE_FRAME_EAGER=1 runs the
eager pass instead (the kernels then carry their launch order in the trace even where the profiler does not see graph nodes)."""
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
eng = mask_mod.MaskEngine(device="cuda:0", seed=0)
with torch.no_grad():
    b = eng.net.prediction_layers[0].conf_layer.bias
    bb = b.detach().cpu().view(3, 81).clone()
    bb[:, 1] += 5.0
    bb[1, 3] += 5.5
    b.copy_(bb.view(-1).to(b.device))
eng.prepare()
rng = np.random.default_rng(0)
frame = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if os.environ.get("AMOS_ONE_FRAME_EAGER") == "1":
    d = torch.as_tensor(frame, device="cuda:0")[None]
    for _ in range(5):
        eng.eval_bgr_batch(d, chunk=1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        eng.eval_bgr_batch(d, chunk=1)
        torch.cuda.synchronize()
    print("eager one-frame pass: %.3f ms" % ((time.perf_counter() - t) / n * 1e3))
else:
    s = eng.frame_session(480, 640)
    s.frame_in.numpy()[...] = frame
    for _ in range(5):
        s.run()
    t = time.perf_counter()
    for _ in range(n):
        s.run()
    print("graph one-frame pass (host frame in, host mask out): %.3f ms, found %s" % ((time.perf_counter() - t) / n * 1e3, s.run()))
