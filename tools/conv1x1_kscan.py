"""GPU box: amos_mask_conv1x1_device at M = 32 x 69 x 69 rows, N = 256 / 512 over K: separates the steady-state rate of the main loop
(large K) from prologue / epilogue / tile-quantisation costs (small K).   python tools/conv1x1_kscan.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
amos = __import__("amos-slam_amd")
dev = torch.device("cuda:0")
cl = torch.channels_last
st = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for B, H in ((32, 69), (32, 35), (43, 69)):  # 43 x 69 x 69 rows = 1600 row tiles x 2 = 3200 groups = 6.25 rounds of 512
    for N in (256, 512):
        for K in (64, 256, 1024, 4096):
            x = torch.randn(B, K, H, H, device=dev).contiguous(memory_format=cl)
            w = torch.randn(N, K, 1, 1, device=dev).contiguous(memory_format=cl)
            y = torch.empty(B, N, H, H, device=dev).contiguous(memory_format=cl)
            ms = timed(lambda: amos.mask_conv1x1(st, x.data_ptr(), w.data_ptr(), None, None, y.data_ptr(), B, H, H, K, N, 1, False))
            groups = ((B * H * H + 127) // 128) * (N // 128)
            print("B %d H %d N %4d K %5d: %7.3f ms %6.1f TF  (%d groups = %.2f rounds of 512)" % (B, H, N, K, ms, 2.0 * B * H * H * N * K / ms / 1e9, groups, groups / 512), flush=True)
