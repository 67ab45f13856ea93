#!/usr/bin/env python3
"""GPU, experiment build (tools/wino_variants.sh TIMING; AMOS_FRONTEND_LIB=.../libamos_frontend_TIMING.so): where the waves of
k_winograd_conv spend their cycles -- per stage: first half (16 MFMAs + transform), wait + barrier, second half; prologue + loop; epilogue."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

amos = entry.load_package()
b, cin, cout, hw = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (32, 256, 256, 138)))
cl = torch.channels_last
x = torch.randn(b, cin, hw, hw, device="cuda").contiguous(memory_format=cl)
w = (torch.randn(cout, cin, 3, 3, device="cuda") / 48).contiguous(memory_format=cl)
u = torch.empty(16 * cin * cout, device="cuda")
y = torch.empty(b, cout, hw, hw, device="cuda").contiguous(memory_format=cl)
st = torch.cuda.current_stream().cuda_stream
amos.mask_winograd_weights(st, w.data_ptr(), u.data_ptr(), cin, cout)
buf = torch.zeros((4096, 8, 8), dtype=torch.int64, device="cuda")
lib = amos.lib()
for it in range(3):
    if it == 2:
        assert lib.amos_mask_winograd_timing_buffer(C.c_void_p(buf.data_ptr())) == 0
    amos.mask_winograd_conv(st, x.data_ptr(), u.data_ptr(), None, None, y.data_ptr(), b, hw, hw, cin, cout, True)
    torch.cuda.synchronize()
t = buf.cpu().numpy()
t = t[t[:, 0, 3] > 0]  # work-groups that ran
n = t[:, :, 3].astype(np.float64)
print("work-groups sampled", len(t), "stages", int(n[0, 0]), "(s_memtime counts shader cycles)")
for k, name in enumerate(("first half (16 MFMAs, V tile of the next stage)", "vmcnt / lgkmcnt wait + barrier", "second half (16 MFMAs, requests)")):
    per = t[:, :, k] / n
    print("%-52s mean %.1f cycles per stage (min wave %.1f, max wave %.1f)" % (name, per.mean(), per.min(), per.max()))
print("prologue + loop: mean %.0f cycles; epilogue: mean %.0f cycles" % (t[:, :, 4].mean(), t[:, :, 5].mean()))
first = t[:, 0, 6]
print("work-group starts span %.0f cycles" % (first.max() - first.min()))
