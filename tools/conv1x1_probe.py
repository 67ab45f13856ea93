"""GPU box: amos_mask_conv1x1_device against channels-last F.conv2d + the bias/ReLU pass, per 1 x 1 shape of the mask network:
max error vs a float64 reference, ms per call and TFLOP/s of both.   python tools/conv1x1_probe.py [batch]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
amos = __import__("amos-slam_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
# (name, cin, cout, stride, H, residual)
SHAPES = [("l1 c1 64-64", 64, 64, 1, 138, 0), ("l1 c1 256-64", 256, 64, 1, 138, 0), ("l1 c3 64-256 +res", 64, 256, 1, 138, 1),
          ("l1 ds 64-256", 64, 256, 1, 138, 0), ("l2 c1 256-128 @138", 256, 128, 1, 138, 0), ("l2 c1 512-128", 512, 128, 1, 69, 0),
          ("l2 c3 128-512 +res", 128, 512, 1, 69, 1), ("l2 ds 256-512 /2", 256, 512, 2, 138, 0), ("l3 c1 512-256 @69", 512, 256, 1, 69, 0),
          ("l3 c1 1024-256", 1024, 256, 1, 35, 0), ("l3 c3 256-1024 +res", 256, 1024, 1, 35, 1), ("l3 ds 512-1024 /2", 512, 1024, 2, 69, 0),
          ("l4 c1 1024-512 @35", 1024, 512, 1, 35, 0), ("l4 c1 2048-512", 2048, 512, 1, 18, 0), ("l4 c3 512-2048 +res", 512, 2048, 1, 18, 1),
          ("l4 ds 1024-2048 /2", 1024, 2048, 2, 35, 0), ("fpn lat 2048-256", 2048, 256, 1, 18, 0), ("fpn lat 1024-256", 1024, 256, 1, 35, 0),
          ("fpn lat 512-256", 512, 256, 1, 69, 0)]


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


cl = torch.channels_last
tot = [0.0, 0.0]
for name, ci, co, s, H, res in SHAPES:
    x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 1, 1, device=dev) * (1.0 / ci ** 0.5)).contiguous(memory_format=cl)
    b = torch.randn(co, device=dev)
    Ho = (H - 1) // s + 1
    r = torch.randn(B, co, Ho, Ho, device=dev).contiguous(memory_format=cl) if res else None
    y = torch.empty(B, co, Ho, Ho, device=dev).contiguous(memory_format=cl)
    flops = 2.0 * B * co * Ho * Ho * ci
    st = torch.cuda.current_stream(dev).cuda_stream

    def ours():
        amos.mask_conv1x1(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), B, H, H, ci, co, s, True)
        return y

    def lib():
        z = F.conv2d(x, w, None, s, 0)
        amos.mask_bias_act(st, z.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, z.numel(), co, True)
        return z

    yo = ours().clone()
    yl = lib()
    nb = min(B, 2)  # float64 reference on a slice
    ref = F.conv2d(x[:nb].double(), w.double(), b.double(), s, 0)
    if res:
        ref = ref + r[:nb].double()
    ref = ref.relu()
    eo, el = (yo[:nb].double() - ref).abs().max().item(), (yl[:nb].double() - ref).abs().max().item()
    to, tl = timed(ours), timed(lib)
    tot[0] += to
    tot[1] += tl
    print("%-22s ours %7.3f ms %6.1f TF err %.2e | miopen+epilogue %7.3f ms %6.1f TF err %.2e | x%.2f" %
          (name, to, flops / to / 1e9, eo, tl, flops / tl / 1e9, el, tl / to), flush=True)
print("sum: ours %.3f ms, miopen+epilogue %.3f ms" % tuple(tot))
