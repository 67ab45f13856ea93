#!/usr/bin/env python3
"""Launches amos_mask_winograd_conv_device a few times on one layer shape (default: the network's largest, 3 x 3 256 -> 256 at
138 x 138, 32 frames) for profiler runs: tools/wino_pmc.sh."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

amos = entry.load_package()
b, cin, cout, hw = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 256, 256, 138)))
cl = torch.channels_last
x = torch.randn(b, cin, hw, hw, device="cuda").contiguous(memory_format=cl)
w = (torch.randn(cout, cin, 3, 3, device="cuda") / 48).contiguous(memory_format=cl)
f24 = os.environ.get("W_FAMILY", "24") == "24"   # F(2 x 4) (default) or F(2 x 2)
u = torch.empty((24 if f24 else 16) * cin * cout, device="cuda")
y = torch.empty(b, cout, hw, hw, device="cuda").contiguous(memory_format=cl)
bias = torch.zeros(cout, device="cuda")
st = torch.cuda.current_stream().cuda_stream
(amos.mask_winograd24_weights if f24 else amos.mask_winograd_weights)(st, w.data_ptr(), u.data_ptr(), cin, cout)
for _ in range(4):
    (amos.mask_winograd24_conv if f24 else amos.mask_winograd_conv)(st, x.data_ptr(), u.data_ptr(), bias.data_ptr(), None, y.data_ptr(), b, hw, hw, cin, cout, True)
torch.cuda.synchronize()
print("ok", float(y.sum()))
