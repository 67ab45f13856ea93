#!/usr/bin/env python3
"""GPU: k_winograd24_conv one work-group per id against the persistent form (one work-group per CU walking the ids, the next id's first requests
under the current id's last stages and epilogue) on the network's layers at 64 frames: same bits (checked), ms per launch, best of three
interleaved rounds of 10 launches."""
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
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for cin, cout, hw in ((256, 256, 138), (256, 256, 69), (256, 384, 69), (64, 64, 138), (128, 128, 69), (256, 256, 35), (512, 512, 18), (256, 256, 18)):
    x = torch.randn(frames, cin, hw, hw, device="cuda").contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).contiguous(memory_format=cl)
    b = torch.randn(cout, device="cuda")
    u = torch.empty(24 * cin * cout, device="cuda")
    pkg.mask_winograd24_weights(st.cuda_stream, w.data_ptr(), u.data_ptr(), cin, cout)
    flops = 2.0 * 24 * frames * ((hw + 1) // 2) * ((hw + 3) // 4) * cin * cout
    best, ys = {0: 1e9, 1: 1e9}, {}
    try:
        for rep in range(3):
            for mode in (0, 1):
                pkg.mask_winograd24_persistent_mode(mode)
                y = torch.full((frames, cout, hw, hw), float("nan"), device="cuda").contiguous(memory_format=cl)

                def launch():
                    pkg.mask_winograd24_conv(st.cuda_stream, x.data_ptr(), u.data_ptr(), b.data_ptr(), None, y.data_ptr(), frames, hw, hw, cin, cout, True)
                for _ in range(3):
                    launch()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
                for _ in range(10):
                    launch()
                e1.record(st)
                e1.synchronize()
                best[mode] = min(best[mode], e0.elapsed_time(e1) / 10)
                ys[mode] = y
    finally:
        pkg.mask_winograd24_persistent_mode(-1)
    assert torch.equal(ys[0], ys[1]) and bool(torch.isfinite(ys[0]).all()), (cin, cout, hw)
    print(json.dumps({"frames": frames, "cin": cin, "cout": cout, "hw": hw, "per_id_ms": round(best[0], 4), "persistent_ms": round(best[1], 4),
                      "gain_percent": round(100 * (1 - best[1] / best[0]), 1), "persistent_tflops": round(flops / best[1] / 1e9, 1)}), flush=True)
    del x, ys, y
