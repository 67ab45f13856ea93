"""GPU box: per-layer timing of the mask network's convolution shapes, three ways:
  nhwc   channels-last F.conv2d + this project's in-place bias/ReLU kernel (what mask/net.py runs)
  nchw   contiguous F.conv2d + the same epilogue written with torch ops (bias add + relu_)
  fused  torch.ops.aten.miopen_convolution_relu on contiguous fp32 (MIOpen's fusion plan, when it applies)
Prints ms per call and TFLOP/s; the answer decides whether any layer should leave the channels-last path.
  python tools/conv_fuse_probe.py [batch]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
amos = __import__("amos-slam_amd")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
# (name, cin, cout, k, stride, pad, H) of YOLACT-R50 at 550 x 550
SHAPES = [("conv1 7x7/2", 3, 64, 7, 2, 3, 550), ("l1 1x1 64-64", 64, 64, 1, 1, 0, 138), ("l1 3x3 64", 64, 64, 3, 1, 1, 138),
          ("l1 1x1 64-256", 64, 256, 1, 1, 0, 138), ("l1 1x1 256-64", 256, 64, 1, 1, 0, 138), ("l2 3x3 128 /2", 128, 128, 3, 2, 1, 138),
          ("l2 3x3 128", 128, 128, 3, 1, 1, 69), ("l2 1x1 128-512", 128, 512, 1, 1, 0, 69), ("l2 1x1 512-128", 512, 128, 1, 1, 0, 69),
          ("l3 3x3 256", 256, 256, 3, 1, 1, 35), ("l3 1x1 256-1024", 256, 1024, 1, 1, 0, 35), ("l3 1x1 1024-256", 1024, 256, 1, 1, 0, 35),
          ("l4 3x3 512", 512, 512, 3, 1, 1, 18), ("l4 1x1 512-2048", 512, 2048, 1, 1, 0, 18), ("l4 1x1 2048-512", 2048, 512, 1, 1, 0, 18),
          ("fpn 3x3 256 @69", 256, 256, 3, 1, 1, 69), ("fpn 3x3 256 @35", 256, 256, 3, 1, 1, 35), ("proto 3x3 256 @138", 256, 256, 3, 1, 1, 138),
          ("head 3x3 256-243 @69", 256, 243, 3, 1, 1, 69), ("head 3x3 256-96 @69", 256, 96, 3, 1, 1, 69)]


def timed(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


tot = {"nhwc": 0.0, "nchw": 0.0, "fused": 0.0}
for name, ci, co, k, s, p, H in SHAPES:
    x = torch.randn(B, ci, H, H, device=dev)
    w = torch.randn(co, ci, k, k, device=dev) * 0.05
    b = torch.randn(co, device=dev)
    xl, wl = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    Ho = (H + 2 * p - k) // s + 1
    flops = 2.0 * B * co * Ho * Ho * ci * k * k

    def nhwc():
        y = F.conv2d(xl, wl, None, s, p)
        amos.mask_bias_act(torch.cuda.current_stream(dev).cuda_stream, y.data_ptr(), b.data_ptr(), None, y.numel(), co, True)
        return y

    def nchw():
        y = F.conv2d(x, w, None, s, p)
        y += b.view(1, -1, 1, 1)
        return y.relu_()

    def fused():
        return torch.ops.aten.miopen_convolution_relu(x, w, b, [s, s], [p, p], [1, 1], 1)

    row = []
    for key, fn in (("nhwc", nhwc), ("nchw", nchw), ("fused", fused)):
        try:
            ms = timed(fn)
            tot[key] += ms
            row.append("%s %7.3f ms %6.1f TF" % (key, ms, flops / ms / 1e9))
        except Exception as e:  # noqa: BLE001 - report and go on with the next layer
            row.append("%s failed: %s" % (key, str(e)[:60]))
    if key == "fused":
        try:
            d = (fused() - nchw()).abs().max().item()
            row.append("max|fused-nchw| %.2e" % d)
        except Exception:  # noqa: BLE001
            pass
    print("%-22s %s" % (name, " | ".join(row)), flush=True)
print("sum over the listed shapes (one call each):", {k: round(v, 3) for k, v in tot.items()})
