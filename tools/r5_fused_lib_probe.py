#!/usr/bin/env python3
"""GPU: the one-frame convolutions the library keeps: library convolution + this project's bias / ReLU pass (what the pass runs) against the library's own
fused convolution + bias + ReLU (torch.ops.aten.miopen_convolution_relu), microseconds per call inside a HIP graph of 20 calls."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

cl = torch.channels_last
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
stream = torch.cuda.Stream()
SHAPES = [("l3 c2 3x3 256 @35", 256, 256, 3, 1, 35), ("l3 c1 1024-256 @35", 1024, 256, 1, 1, 35), ("l4 c1 2048-512 @18", 2048, 512, 1, 1, 18), ("l4 c2 3x3 512 @18", 512, 512, 3, 1, 18),
          ("l4 c3 512-2048 @18", 512, 2048, 1, 1, 18), ("l3 c2 3x3 256 /2 @69", 256, 256, 3, 2, 69), ("head 3x3 256 @18", 256, 256, 3, 1, 18), ("head 3x3 256 @9", 256, 256, 3, 1, 9)]


def graph_us(fn, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
    stream.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        for _ in range(n):
            fn()
    stream.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            g.replay()
            e1.record(stream)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for name, ci, co, k, s, H in SHAPES:
    pad = k // 2
    x = torch.randn(1, ci, H, H, device="cuda").contiguous(memory_format=cl)
    w = (torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5).contiguous(memory_format=cl)
    b = torch.randn(co, device="cuda")

    def lib():
        z = F.conv2d(x, w, None, s, pad)
        pkg.mask_bias_act(stream.cuda_stream, z.data_ptr(), b.data_ptr(), None, z.numel(), co, True)
        return z

    def fused():
        return torch.ops.aten.miopen_convolution_relu(x, w, b, [s, s], [pad, pad], [1, 1], 1)

    try:
        with torch.cuda.stream(stream):
            a, f = lib(), fused()
        stream.synchronize()
        err = float((a - f).abs().max())
        tl, tf = graph_us(lib), graph_us(fused)
        print("%-24s library + bias pass %6.1f us | miopen_convolution_relu %6.1f us | max diff %.1e | layout kept %s" % (name, tl, tf, err, f.is_contiguous(memory_format=cl)), flush=True)
    except Exception as e:  # noqa: BLE001
        print("%-24s fused form failed: %s" % (name, str(e)[:120]), flush=True)
