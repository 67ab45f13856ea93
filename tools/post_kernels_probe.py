"""GPU box: the post-processing kernels whose time depends on the data, at both ends of what the data can be (32 frames):
top-k over rows that are all -1 / hold 2 000 real scores / are fully random, and the person mask with 0 / 3 / 15 flagged detections.
  python tools/post_kernels_probe.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
amos = __import__("amos-slam_amd")
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


rows, n, k = 32 * 80, 19248, 200
v = torch.empty(rows, k, device=dev)
i = torch.empty(rows, k, dtype=torch.int64, device=dev)
for name, x in (("all -1", torch.full((rows, n), -1.0, device=dev)),
                ("2 000 real scores per row", torch.where(torch.rand(rows, n, device=dev) < 2000 / n, torch.rand(rows, n, device=dev), torch.full((rows, n), -1.0, device=dev))),
                ("random rows", torch.randn(rows, n, device=dev))):
    ours = timed(lambda: amos.mask_topk_rows(st, x.data_ptr(), v.data_ptr(), i.data_ptr(), rows, n, k))
    ref = timed(lambda: x.topk(k, dim=1))
    print("top-k %-28s kernel %.3f ms, torch.topk %.3f ms" % (name, ours, ref), flush=True)
B, nd, ph, pw, H, W = 32, 15, 138, 138, 480, 640
masks = torch.rand(B, nd, ph, pw, device=dev)
out = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
for flagged in (0, 3, 15):
    flags = torch.zeros(B, nd, dtype=torch.uint8, device=dev)
    flags[:, :flagged] = 1
    ms = timed(lambda: amos.mask_person_mask(st, masks.data_ptr(), flags.data_ptr(), out.data_ptr(), B, nd, ph, pw, H, W))
    print("person mask, %2d of 15 detections flagged: %.3f ms" % (flagged, ms), flush=True)
