"""GPU box: the brute-force best-2 match, xor + popcount kernel against the i8 MFMA kernel, same resident descriptors
(128 frames of ~1000 real descriptors, frame k vs k-1).  Prints ms per launch and checks that the results are identical."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
pkg = entry.load_package()
import importlib
synth = importlib.import_module("amos_slam_amd.synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ext = pkg.OrbExtractor(max_batch=B)
frames = torch.from_numpy(synth.frames(0, 0, B)).cuda()
ext.extract_batch_device(frames.data_ptr(), 480 * 640, 640, 640, 480, B)
ext.sync()
_, d_desc, d_counts, cap = ext.batch_results_device()
m = pkg.OrbMatcher(stream=ext.stream)
pq = torch.arange(B, dtype=torch.int32, device="cuda")
pt = (pq - 1) % B
stream = torch.cuda.ExternalStream(ext.stream)
res = {}
for kern in ("popcount", "mfma"):
    m.set_bruteforce_kernel(kern)
    out = torch.zeros((B, cap, 4), dtype=torch.int32, device="cuda")
    for it in range(3):
        m.bruteforce_best2_batch_device(d_desc, cap * 32, d_counts, pq.data_ptr(), pt.data_ptr(), B, cap, 256, out.data_ptr())
    m.sync()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for it in range(20):
        m.bruteforce_best2_batch_device(d_desc, cap * 32, d_counts, pq.data_ptr(), pt.data_ptr(), B, cap, 256, out.data_ptr())
    e1.record(stream)
    m.sync()
    torch.cuda.synchronize()
    res[kern] = out.cpu().numpy()
    print(f"{kern}: {e0.elapsed_time(e1) / 20:.4f} ms per launch of {B} pairs (capacity {cap})")
n_kp = [len(ext.batch_fetch(f)[0]) for f in range(min(B, 16))]
print("identical:", all(np.array_equal(res["popcount"][f, :n_kp[f]], res["mfma"][f, :n_kp[f]]) for f in range(min(B, 16))))
