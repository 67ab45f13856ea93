#!/usr/bin/env python3
"""GPU: amos_mask_winograd_conv_device against a float64 convolution (error relative to the sum of |terms|) and against the direct
implicit GEMM (amos_mask_conv_device) in time, on the mask network's stride-1 3 x 3 layers at `--frames` frames per launch."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

amos = entry.load_package()
F = torch.nn.functional
cl = torch.channels_last


def run(b, cin, cout, h, w, check=True, reps=10, relu=True, use_res=False):
    dev = "cuda"
    x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=cl)
    wgt = (torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5).contiguous(memory_format=cl)
    bias = torch.randn(cout, device=dev)
    res = torch.randn(b, cout, h, w, device=dev).contiguous(memory_format=cl) if use_res else None
    u = torch.empty(16 * cin * cout, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    amos.mask_winograd_weights(st, wgt.data_ptr(), u.data_ptr(), cin, cout)
    y = torch.full((b, cout, h, w), float("nan"), device=dev).contiguous(memory_format=cl)
    yd = torch.empty_like(y)

    def wino():
        amos.mask_winograd_conv(st, x.data_ptr(), u.data_ptr(), bias.data_ptr(), res.data_ptr() if use_res else None, y.data_ptr(), b, h, w, cin, cout, relu)

    has_direct = amos.mask_conv_supported(cin, cout, 3, 3, 1, 1)

    def direct():
        if not has_direct:
            yd.copy_(torch.relu(F.conv2d(x, wgt, bias, 1, 1) + (res if use_res else 0)) if relu else F.conv2d(x, wgt, bias, 1, 1) + (res if use_res else 0))
            return
        amos.mask_conv(st, x.data_ptr(), wgt.data_ptr(), bias.data_ptr(), res.data_ptr() if use_res else None, yd.data_ptr(), b, h, w, cin, cout, 3, 3, 1, 1, relu)

    wino()
    direct()
    torch.cuda.synchronize()
    out = {}
    if check:
        n = min(b, 2)
        exact = F.conv2d(x[:n].double(), wgt.double(), bias.double(), 1, 1)
        if use_res:
            exact = exact + res[:n].double()
        if relu:
            exact = exact.relu()
        bound = F.conv2d(x[:n].double().abs(), wgt.double().abs(), None, 1, 1)
        out["err_wino"] = float(((y[:n].double() - exact).abs() / bound).max())
        out["err_direct"] = float(((yd[:n].double() - exact).abs() / bound).max())
        out["finite"] = bool(torch.isfinite(y).all())
        out["max_abs_diff"] = float((y - yd).abs().max())
    for name, fn in (("wino", wino), ("direct", direct)):
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        e1.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name + "_ms"] = round(ms, 4)
        out[name + "_direct_equiv_TF"] = round(2.0 * b * h * w * cout * 9 * cin / (ms * 1e-3) / 1e12, 1)
    out["speedup"] = round(out["direct_ms"] / out["wino_ms"], 3)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=32)
    ap.add_argument("--small-only", action="store_true")
    ap.add_argument("--f24", action="store_true", help="F(2 x 4) against F(2 x 2): time and error on the network's layer shapes")
    ap.add_argument("--f24-time", action="store_true", help="time of the F(2 x 4) kernel on four layer shapes, no checks (experiment builds)")
    ap.add_argument("--big-only", action="store_true", help="timing of the three largest layer shapes only, no checks (experiment builds)")
    a = ap.parse_args()
    torch.manual_seed(0)
    if a.big_only:
        for cin, cout, hw in ((256, 256, 138), (256, 256, 69), (64, 64, 138)):
            r = run(a.frames, cin, cout, hw, hw, check=False)
            print((cin, cout, hw), r["wino_ms"], r["wino_direct_equiv_TF"], flush=True)
        sys.exit(0)
    if a.f24_time:
        for cin, cout, hw in ((256, 256, 138), (256, 256, 69), (64, 64, 138), (512, 512, 18)):
            b = a.frames
            x = torch.randn(b, cin, hw, hw, device="cuda").contiguous(memory_format=cl)
            wgt = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).contiguous(memory_format=cl)
            bias = torch.randn(cout, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            u = torch.empty(24 * cin * cout, device="cuda")
            amos.mask_winograd24_weights(st, wgt.data_ptr(), u.data_ptr(), cin, cout)
            y = torch.empty((b, cout, hw, hw), device="cuda").contiguous(memory_format=cl)
            fn = lambda: amos.mask_winograd24_conv(st, x.data_ptr(), u.data_ptr(), bias.data_ptr(), None, y.data_ptr(), b, hw, hw, cin, cout, True)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            e1.synchronize()
            print((cin, cout, hw), round(e0.elapsed_time(e1) / 10, 4), end="  ", flush=True)
        print()
        sys.exit(0)
    if a.f24:
        # F(2 x 4) against F(2 x 2) on every stride-1 3 x 3 layer shape of the network: time and error relative to the sum of |terms|
        for cin, cout, hw in ((256, 256, 138), (256, 256, 69), (256, 384, 69), (64, 64, 138), (128, 128, 69), (256, 256, 35), (256, 384, 35), (512, 512, 18)):
            b = a.frames
            x = torch.randn(b, cin, hw, hw, device="cuda").contiguous(memory_format=cl)
            wgt = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).contiguous(memory_format=cl)
            bias = torch.randn(cout, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            out = {}
            n = min(b, 2)
            exact = F.conv2d(x[:n].double(), wgt.double(), bias.double(), 1, 1).relu()
            bound = F.conv2d(x[:n].double().abs(), wgt.double().abs(), None, 1, 1)
            for fam, npos, mk, cv in (("22", 16, amos.mask_winograd_weights, amos.mask_winograd_conv), ("24", 24, amos.mask_winograd24_weights, amos.mask_winograd24_conv)):
                u = torch.empty(npos * cin * cout, device="cuda")
                mk(st, wgt.data_ptr(), u.data_ptr(), cin, cout)
                y = torch.full((b, cout, hw, hw), float("nan"), device="cuda").contiguous(memory_format=cl)
                fn = lambda: cv(st, x.data_ptr(), u.data_ptr(), bias.data_ptr(), None, y.data_ptr(), b, hw, hw, cin, cout, True)
                for _ in range(3):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                e1.synchronize()
                out[fam + "_ms"] = round(e0.elapsed_time(e1) / 10, 4)
                out[fam + "_err"] = float(((y[:n].double() - exact).abs() / bound).max())
            out["speedup_24_over_22"] = round(out["22_ms"] / out["24_ms"], 3)
            print((b, cin, cout, hw), out, flush=True)
        sys.exit(0)
    for shape in ((1, 32, 64, 6, 6), (2, 64, 64, 21, 17), (1, 32, 128, 9, 9), (2, 256, 64, 5, 5), (3, 48, 192, 12, 7), (2, 32, 64, 1, 3), (40, 32, 64, 2, 4)):
        print(shape, run(*shape, reps=2, relu=(shape[1] != 32), use_res=(shape[1] == 64)), flush=True)
    if not a.small_only:
        for cin, cout, hw in ((256, 256, 138), (256, 256, 69), (256, 384, 69), (64, 64, 138), (128, 128, 69), (256, 256, 35), (512, 512, 18)):
            print((a.frames, cin, cout, hw, hw), run(a.frames, cin, cout, hw, hw, check=True), flush=True)
