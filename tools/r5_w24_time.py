#!/usr/bin/env python3
"""GPU: ms per launch of amos_mask_winograd24_conv on the network's layers at 64 frames (best of three rounds of 10); one line per
library build (AMOS_FRONTEND_LIB), for sweeps of build-time constants (tools/w24_variants.sh G<n>)."""
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
out = []
for cin, cout, hw in ((256, 256, 138), (256, 256, 69), (256, 384, 69), (64, 64, 138), (128, 128, 69), (256, 256, 35), (512, 512, 18)):
    x = torch.randn(frames, cin, hw, hw, device="cuda").contiguous(memory_format=cl)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (cin * 9) ** 0.5).contiguous(memory_format=cl)
    b = torch.randn(cout, device="cuda")
    u = torch.empty(24 * cin * cout, device="cuda")
    y = torch.empty(frames, cout, hw, hw, device="cuda").contiguous(memory_format=cl)
    pkg.mask_winograd24_weights(st.cuda_stream, w.data_ptr(), u.data_ptr(), cin, cout)
    best = 1e9
    for rep in range(3):
        for _ in range(3):
            pkg.mask_winograd24_conv(st.cuda_stream, x.data_ptr(), u.data_ptr(), b.data_ptr(), None, y.data_ptr(), frames, hw, hw, cin, cout, True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            pkg.mask_winograd24_conv(st.cuda_stream, x.data_ptr(), u.data_ptr(), b.data_ptr(), None, y.data_ptr(), frames, hw, hw, cin, cout, True)
        e1.record(st)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    out.append("%d/%d@%d %.4f" % (cin, cout, hw, best))
    del x, y
print(os.path.basename(os.environ.get("AMOS_FRONTEND_LIB", "base")), " | ".join(out), flush=True)
