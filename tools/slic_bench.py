#!/usr/bin/env python3
"""SLIC (cluster::SLIC from the Lab image on): GPU ms per frame, resident batch, vs the CPU oracle."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
pkg = entry.load_package()
import oracle_binding as ob
rng = np.random.default_rng(0)
n, w, h = 32, 640, 480
base = np.kron(rng.integers(0, 256, (n, (h + 23) // 24, (w + 23) // 24, 3)), np.ones((1, 24, 24, 1)))[:, :h, :w]
lab = np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)
depth = rng.integers(0, 20000, (n, h, w)).astype(np.uint16)
d_lab, d_depth = torch.from_numpy(lab).cuda(), torch.from_numpy(depth.view(np.int16)).cuda()
nc = pkg.Slic.center_count(w, h, 5)[0]
d_labels = torch.zeros((n, h, w), dtype=torch.float64, device="cuda")
d_cent = torch.zeros((n, nc, 8), dtype=torch.int32, device="cuda")
sl = pkg.Slic(max_width=w, max_height=h, max_batch=n)
torch.cuda.synchronize()
for _ in range(3):
    sl.run_batch_device(d_lab.data_ptr(), d_depth.data_ptr(), w, h, n, d_labels.data_ptr(), d_cent.data_ptr())
sl.sync()
t0 = time.perf_counter()
for _ in range(10):
    sl.run_batch_device(d_lab.data_ptr(), d_depth.data_ptr(), w, h, n, d_labels.data_ptr(), d_cent.data_ptr())
sl.sync()
gpu_ms = (time.perf_counter() - t0) / 10 / n * 1e3
t0 = time.perf_counter()
for f in range(3):
    lo, co = ob.slic(lab[f], depth[f])
cpu_ms = (time.perf_counter() - t0) / 3 * 1e3
ok = np.array_equal(d_labels[2].cpu().numpy(), lo)
# per iteration and pixel: lab 3 B read x 3 passes, distance word 8 B fill + atomics, index 4 B, label 8 B write + read
alg = (5 * (3 * 3 + 8 * 2 + 4 * 2 + 8 * 2) + 8) * w * h
print(f"SLIC 640x480, {nc} centres, 5 iterations: GPU {gpu_ms:.3f} ms/frame ({1e3/gpu_ms:.0f} frames/s, batch {n}, ~{alg/gpu_ms/1e6:.0f} GB/s of "
      f"~{alg/1e6:.0f} MB/frame), CPU oracle {cpu_ms:.1f} ms/frame ({1e3/cpu_ms:.1f} frames/s, 1 thread); bit-exact: {ok}")
