#!/usr/bin/env python3
"""GPU: k_winograd24_conv on the network's large 256-channel layers with the channel-blocked activation layout [b][c / 8][h][w][8] on the
input and / or the output against channels-last on both: same bits (checked), time per launch (HIP events, 10 launches)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import torch  # noqa: E402

cl = torch.channels_last
torch.manual_seed(0)
st = torch.cuda.current_stream()
out = []
for frames, cin, cout, hw in ((64, 256, 256, 138), (64, 256, 256, 69), (64, 256, 384, 69), (64, 64, 64, 138), (64, 128, 128, 69), (64, 256, 256, 35)):
    x = torch.randn(frames, cin, hw, hw, device="cuda").contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).contiguous(memory_format=cl)
    b = torch.randn(cout, device="cuda")
    u = torch.empty(24 * cin * cout, device="cuda")
    pkg.mask_winograd24_weights(st.cuda_stream, w.data_ptr(), u.data_ptr(), cin, cout)
    # NHWC memory [b][h][w][c] -> blocked [b][c/8][h][w][8]
    xb = x.permute(0, 2, 3, 1).reshape(frames, hw, hw, cin // 8, 8).permute(0, 3, 1, 2, 4).contiguous()
    ys = {}
    row = {"frames": frames, "cin": cin, "cout": cout, "hw": hw}
    flops = 2.0 * 24 * frames * ((hw + 1) // 2) * ((hw + 3) // 4) * cin * cout
    cfgs = (("nhwc_nhwc", x, 0, 0), ("blocked_nhwc", xb, 1, 0), ("blocked_blocked", xb, 1, 1), ("nhwc_blocked", x, 0, 1))
    best = {c[0]: 1e9 for c in cfgs}
    for rep in range(3):   # configurations interleaved, best of three rounds: no configuration pays for the clock ramp of the first launches
        for name, xin, ib, ob in cfgs:
            y = torch.full((frames * hw * hw * cout,), float("nan"), device="cuda")

            def launch():
                pkg.mask_winograd24_conv_layout(st.cuda_stream, xin.data_ptr(), u.data_ptr(), b.data_ptr(), None, y.data_ptr(), frames, hw, hw, cin, cout, True, ib, ob)
            for _ in range(3):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                launch()
            e1.record(st)
            e1.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / 10)
            yy = y.view(frames, cout // 8, hw, hw, 8).permute(0, 2, 3, 1, 4).reshape(frames, hw, hw, cout) if ob else y.view(frames, hw, hw, cout)
            ys[name] = yy
    for name in best:
        row[name + "_ms"] = round(best[name], 4)
        row[name + "_tflops"] = round(flops / best[name] / 1e9, 1)
    for k in ys:
        assert torch.equal(ys[k], ys["nhwc_nhwc"]), k
    out.append(row)
    print(json.dumps(row), flush=True)
    del x, xb, ys, y
