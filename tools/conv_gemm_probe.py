"""GPU box: amos_mask_conv_device (fp32 MFMA implicit GEMM, fused epilogue) against channels-last F.conv2d + the bias/ReLU pass on the
3 x 3 shapes of the mask network: max error vs a float64 reference, ms per call and TFLOP/s of both.   python tools/conv_gemm_probe.py [batch]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
amos = __import__("amos-slam_amd")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
# (name, cin, cout, k, stride, H)
SHAPES = [("l1 3x3 64", 64, 64, 3, 1, 138), ("l2 3x3 128 /2", 128, 128, 3, 2, 138), ("l2 3x3 128", 128, 128, 3, 1, 69), ("l3 3x3 256 /2", 256, 256, 3, 2, 69),
          ("l3 3x3 256", 256, 256, 3, 1, 35), ("l4 3x3 512 /2", 512, 512, 3, 2, 35), ("l4 3x3 512", 512, 512, 3, 1, 18), ("fpn/proto/head 3x3 256 @69", 256, 256, 3, 1, 69),
          ("fpn/head 3x3 256 @35", 256, 256, 3, 1, 35), ("fpn/head 3x3 256 @18", 256, 256, 3, 1, 18), ("fpn down 256 /2 @18", 256, 256, 3, 2, 18),
          ("proto 3x3 256 @138", 256, 256, 3, 1, 138), ("heads merged 256-384 @69", 256, 384, 3, 1, 69), ("heads merged 256-384 @35", 256, 384, 3, 1, 35),
          ("heads merged 256-384 @18", 256, 384, 3, 1, 18), ("heads merged 256-384 @9", 256, 384, 3, 1, 9)]


def timed(fn, n=8):
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
for name, ci, co, k, s, H in SHAPES:
    p = k // 2
    x = torch.randn(B, ci, H, H, device=dev).contiguous(memory_format=cl)
    w = (torch.randn(co, ci, k, k, device=dev) * (1.0 / (ci * k * k) ** 0.5)).contiguous(memory_format=cl)
    b = torch.randn(co, device=dev)
    Ho = (H + 2 * p - k) // s + 1
    y = torch.empty(B, co, Ho, Ho, device=dev).contiguous(memory_format=cl)
    flops = 2.0 * B * co * Ho * Ho * ci * k * k
    st = torch.cuda.current_stream(dev).cuda_stream

    def ours():
        amos.mask_conv(st, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), B, H, H, ci, co, k, k, s, p, True)
        return y

    def lib():
        z = F.conv2d(x, w, None, s, p)
        amos.mask_bias_act(st, z.data_ptr(), b.data_ptr(), None, z.numel(), co, True)
        return z

    yo = ours().clone()
    yl = lib()
    ref = F.conv2d(x[:1].double(), w.double(), b.double(), s, p).relu()
    eo, el = (yo[:1].double() - ref).abs().max().item(), (yl[:1].double() - ref).abs().max().item()
    to, tl = timed(ours), timed(lib)
    tot[0] += to
    tot[1] += tl
    print("%-28s ours %7.3f ms %6.1f TF err %.2e | miopen+epilogue %7.3f ms %6.1f TF err %.2e | x%.2f" %
          (name, to, flops / to / 1e9, eo, tl, flops / tl / 1e9, el, tl / to), flush=True)
print("sum: ours %.3f ms, miopen+epilogue %.3f ms" % tuple(tot))
