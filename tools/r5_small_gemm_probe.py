#!/usr/bin/env python3
"""GPU: every convolution shape of the mask network at ONE frame per launch (and at 4): amos_mask_conv_ws_device (the project's MFMA GEMM with
split-K for small launches, bias + residual + ReLU in its epilogue) against the library convolution + the project's bias / ReLU pass:
microseconds per call inside a HIP graph of 20 calls (launch gaps excluded the way the one-frame session excludes them), max error of both
against a float64 convolution.   python tools/r5_small_gemm_probe.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cl = torch.channels_last
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
# (name, cin, cout, k, stride, H, residual)
SHAPES = [("l1 c1 64-64", 64, 64, 1, 1, 138, 0), ("l1 c2 3x3 64", 64, 64, 3, 1, 138, 0), ("l1 c3 64-256 +res", 64, 256, 1, 1, 138, 1), ("l1 ds 64-256", 64, 256, 1, 1, 138, 0),
          ("l1 c1 256-64", 256, 64, 1, 1, 138, 0), ("l2 c1 256-128 @138", 256, 128, 1, 1, 138, 0), ("l2 c2 3x3 128 /2", 128, 128, 3, 2, 138, 0),
          ("l2 c3 128-512 +res", 128, 512, 1, 1, 69, 1), ("l2 ds 256-512 /2", 256, 512, 1, 2, 138, 0), ("l2 c1 512-128", 512, 128, 1, 1, 69, 0),
          ("l2 c2 3x3 128", 128, 128, 3, 1, 69, 0), ("l3 c1 512-256 @69", 512, 256, 1, 1, 69, 0), ("l3 c2 3x3 256 /2", 256, 256, 3, 2, 69, 0),
          ("l3 c3 256-1024 +res", 256, 1024, 1, 1, 35, 1), ("l3 ds 512-1024 /2", 512, 1024, 1, 2, 69, 0), ("l3 c1 1024-256", 1024, 256, 1, 1, 35, 0),
          ("l3 c2 3x3 256", 256, 256, 3, 1, 35, 0), ("l4 c1 1024-512 @35", 1024, 512, 1, 1, 35, 0), ("l4 c2 3x3 512 /2", 512, 512, 3, 2, 35, 0),
          ("l4 c3 512-2048 +res", 512, 2048, 1, 1, 18, 1), ("l4 ds 1024-2048 /2", 1024, 2048, 1, 2, 35, 0), ("l4 c1 2048-512", 2048, 512, 1, 1, 18, 0),
          ("l4 c2 3x3 512", 512, 512, 3, 1, 18, 0), ("fpn lat 2048-256", 2048, 256, 1, 1, 18, 0), ("fpn lat 1024-256 +res", 1024, 256, 1, 1, 35, 1),
          ("fpn lat 512-256 +res", 512, 256, 1, 1, 69, 1), ("fpn pred 3x3 @18", 256, 256, 3, 1, 18, 0), ("fpn pred 3x3 @35", 256, 256, 3, 1, 35, 0),
          ("fpn pred/proto 3x3 @69", 256, 256, 3, 1, 69, 0), ("fpn down 3x3 /2 @18", 256, 256, 3, 2, 18, 0), ("fpn down 3x3 /2 @9", 256, 256, 3, 2, 9, 0),
          ("head 3x3 256-384 @69", 256, 384, 3, 1, 69, 0), ("head 3x3 256-384 @35", 256, 384, 3, 1, 35, 0), ("head 3x3 256-384 @18", 256, 384, 3, 1, 18, 0),
          ("head 3x3 256-256 @9", 256, 256, 3, 1, 9, 0), ("head 3x3 256-256 @5", 256, 256, 3, 1, 5, 0), ("proto 3x3 @138", 256, 256, 3, 1, 138, 0)]
stream = torch.cuda.Stream()
ws = torch.zeros(128 << 20, dtype=torch.uint8, device="cuda")


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


tot = [0.0, 0.0]
for name, ci, co, k, s, H, res in SHAPES:
    pad = k // 2
    x = torch.randn(B, ci, H, H, device="cuda").contiguous(memory_format=cl)
    w = (torch.randn(co, ci, k, k, device="cuda") / (ci * k * k) ** 0.5).contiguous(memory_format=cl)
    b = torch.randn(co, device="cuda")
    Ho = (H + 2 * pad - k) // s + 1
    r = torch.randn(B, co, Ho, Ho, device="cuda").contiguous(memory_format=cl) if res else None
    y = torch.empty(B, co, Ho, Ho, device="cuda").contiguous(memory_format=cl)
    nb = pkg.mask_conv_workspace_bytes(B, H, H, ci, co, k, k, s, pad)

    def ours():
        pkg.mask_conv_ws(stream.cuda_stream, x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), B, H, H, ci, co, k, k, s, pad, True,
                         ws.data_ptr(), ws.numel())

    def lib():
        z = F.conv2d(x, w, None, s, pad)
        pkg.mask_bias_act(stream.cuda_stream, z.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, z.numel(), co, True)
        return z

    with torch.cuda.stream(stream):
        ours()
        yl = lib()
    stream.synchronize()
    ref = F.conv2d(x.double(), w.double(), b.double(), s, pad)
    if res:
        ref = ref + r.double()
    ref = ref.relu()
    bound = F.conv2d(x.double().abs(), w.double().abs(), None, s, pad) + 1
    eo, el = float(((y.double() - ref).abs() / bound).max()), float(((yl.double() - ref).abs() / bound).max())
    to, tl = graph_us(ours), graph_us(lib)
    tot[0] += to
    tot[1] += tl
    wino = ""
    if k == 3 and s == 1 and pkg.mask_winograd_supported(ci, co):
        # the same layer as Winograd F(2 x 4): 64 and 32 output channels per work-group
        u = torch.empty(24 * ci * co, device="cuda")
        pkg.mask_winograd24_weights(stream.cuda_stream, w.data_ptr(), u.data_ptr(), ci, co)
        yw = torch.empty_like(y)
        for mode in (0, 1):
            pkg.mask_winograd24_narrow_mode(mode)

            def wrun():
                pkg.mask_winograd24_conv(stream.cuda_stream, x.data_ptr(), u.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, yw.data_ptr(), B, H, H, ci, co, True)
            with torch.cuda.stream(stream):
                wrun()
            stream.synchronize()
            ew = float(((yw.double() - ref).abs() / bound).max())
            wino += " | winograd %d ch/group %6.1f us err %.1e" % (32 if mode else 64, graph_us(wrun), ew)
        pkg.mask_winograd24_narrow_mode(-1)
    print("%-26s split-K bytes %9d | ours %7.1f us err %.1e | library + bias pass %7.1f us err %.1e | x%.2f%s" % (name, nb, to, eo, tl, el, tl / to, wino), flush=True)
print("sum: ours %.1f us, library %.1f us (%d frame(s))" % (tot[0], tot[1], B))
