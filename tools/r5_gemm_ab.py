#!/usr/bin/env python3
"""GPU: the network's 1 x 1 convolutions through amos_mask_conv1x1_device at the bench's 64 frames per launch: ms per launch, bytes moved
per launch (x + residual + y + w, each once) and the rate they stand for.   python tools/r5_gemm_ab.py [frames]
Run once per library build (AMOS_FRONTEND_LIB) for an A/B."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import torch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cl = torch.channels_last
torch.manual_seed(0)
st = torch.cuda.current_stream()
SHAPES = [("l1 c1 64-64", 64, 64, 1, 138, 0), ("l1 c1 256-64", 256, 64, 1, 138, 0), ("l1 c3 64-256 +res", 64, 256, 1, 138, 1), ("l1 ds 64-256", 64, 256, 1, 138, 0),
          ("l2 c1 256-128 @138", 256, 128, 1, 138, 0), ("l2 c1 512-128", 512, 128, 1, 69, 0), ("l2 c3 128-512 +res", 128, 512, 1, 69, 1),
          ("l2 ds 256-512 /2", 256, 512, 2, 138, 0), ("l3 c1 512-256 @69", 512, 256, 1, 69, 0), ("l3 c1 1024-256", 1024, 256, 1, 35, 0),
          ("l3 c3 256-1024 +res", 256, 1024, 1, 35, 1), ("l3 ds 512-1024 /2", 512, 1024, 2, 69, 0), ("l4 c1 1024-512 @35", 1024, 512, 1, 35, 0),
          ("l4 c1 2048-512", 2048, 512, 1, 18, 0), ("l4 c3 512-2048 +res", 512, 2048, 1, 18, 1), ("l4 ds 1024-2048 /2", 1024, 2048, 2, 35, 0),
          ("fpn lat 1024-256 +res", 1024, 256, 1, 35, 1), ("fpn lat 512-256 +res", 512, 256, 1, 69, 1)]
tot = 0.0
rows = []
for name, ci, co, s, H, res in SHAPES:
    x = torch.randn(B, ci, H, H, device="cuda").contiguous(memory_format=cl)
    w = (torch.randn(co, ci, 1, 1, device="cuda") / ci ** 0.5).contiguous(memory_format=cl)
    b = torch.randn(co, device="cuda")
    Ho = (H - 1) // s + 1
    r = torch.randn(B, co, Ho, Ho, device="cuda").contiguous(memory_format=cl) if res else None
    y = torch.empty(B, co, Ho, Ho, device="cuda").contiguous(memory_format=cl)

    def launch():
        pkg.mask_conv1x1(st.cuda_stream, x.data_ptr(), w.data_ptr(), b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(), B, H, H, ci, co, s, True)
    best = 1e9
    for rep in range(3):
        for _ in range(3):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            launch()
        e1.record(st)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    bytes_ = 4.0 * (B * Ho * Ho * ci + (2 if res else 1) * B * Ho * Ho * co + ci * co)   # strided layers read a quarter of x's pixels
    flops = 2.0 * B * Ho * Ho * ci * co
    tot += best
    rows.append({"layer": name, "ms": round(best, 4), "GBps": round(bytes_ / best / 1e6, 0), "TFLOPs": round(flops / best / 1e9, 1), "crc": float(y.double().sum())})
    print("%-24s %8.4f ms  %6.0f GB/s  %6.1f TFLOP/s" % (name, best, bytes_ / best / 1e6, flops / best / 1e9), flush=True)
print("sum %.3f ms  (%s)" % (tot, os.environ.get("AMOS_FRONTEND_LIB", "default build")))
print(json.dumps(rows))
