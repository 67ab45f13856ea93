#!/usr/bin/env python3
"""GPU: the stem (conv 7 x 7 / 2 + bias + ReLU + max-pool) at 1 and 64 frames: the project's one-kernel stem (amos_mask_stem_device) against the
chain it replaces (layout copy + library convolution + amos_mask_bias_relu_maxpool_device); microseconds per call inside a HIP graph of n calls.
python tools/r5_stem_probe.py"""
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


def graph_us(fn, n):
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
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(stream):
            e0.record(stream)
            g.replay()
            e1.record(stream)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


w = (torch.randn(64, 3, 7, 7, device="cuda") / 12).contiguous(memory_format=cl)
bias = torch.randn(64, device="cuda")
packed = torch.empty(pkg.mask_stem_weight_floats(), device="cuda")
pkg.mask_stem_weights(torch.cuda.current_stream().cuda_stream, w.data_ptr(), w.stride(), packed.data_ptr())
torch.cuda.synchronize()
for B in (1, 2, 8, 64):
    x = torch.randn(B, 3, 550, 550, device="cuda")
    y = torch.empty(B, 64, 138, 138, device="cuda").contiguous(memory_format=cl)
    y2 = torch.empty_like(y)

    def ours():
        pkg.mask_stem(stream.cuda_stream, x.data_ptr(), x.stride(), packed.data_ptr(), bias.data_ptr(), y.data_ptr(), B, 550, 550)

    def lib():
        raw = F.conv2d(x.contiguous(memory_format=cl), w, None, 2, 3)
        pkg.mask_bias_relu_maxpool(stream.cuda_stream, raw.data_ptr(), bias.data_ptr(), y2.data_ptr(), B, 275, 275, 64)

    n = 20 if B <= 8 else 4
    to, tl = graph_us(ours, n), graph_us(lib, n)
    stream.synchronize()
    d = float((y - y2).abs().max())
    flops = 2 * 147 * 64 * 275 * 275 * B
    print("%2d frame(s): one kernel %8.1f us (%5.1f TFLOP/s of the convolution) | copy + library convolution + pool kernel %8.1f us | x%.2f | max |diff| %.2e"
          % (B, to, flops / to / 1e6, tl, tl / to, d), flush=True)
