"""HBM copy / fill bandwidth reachable from torch on this GPU (reference point for the streaming kernels)."""
import torch
dev = "cuda:0"
for mb in (80, 160, 320, 1280):
    n = mb * 1024 * 1024
    a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
    b = torch.empty_like(a)
    for name, fn, bytes_ in (("copy", lambda: b.copy_(a), 2 * n), ("fill", lambda: b.zero_(), n), ("sum", lambda: a.view(torch.int32).sum(), n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{mb:5d} MB {name:5s} {ms*1e3:8.1f} us  {bytes_/ms/1e9:7.2f} TB/s", flush=True)
