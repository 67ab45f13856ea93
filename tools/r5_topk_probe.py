#!/usr/bin/env python3
"""GPU: the row-wise top-k of the detector (200 of 19 248 per class row) at one frame (80 rows) and 64 frames (5 120 rows): dense rows (random
weights keep every prior above the threshold) and sparse rows (300 live scores, the rest -1), microseconds per call in a HIP graph.  Run once
as is and once with AMOS_TOPK_LDS=0 (the five-scan kernel).   python tools/r5_topk_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import torch  # noqa: E402

stream = torch.cuda.Stream()
torch.manual_seed(0)


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


for rows in (80, 5120):
    for kind in ("dense", "sparse"):
        if kind == "dense":
            x = torch.rand(rows, 19248, device="cuda")
        else:
            x = torch.full((rows, 19248), -1.0, device="cuda")
            pos = torch.stack([torch.randperm(19248, device="cuda")[:300] for _ in range(rows)])
            x.scatter_(1, pos, torch.rand(rows, 300, device="cuda") * 0.9 + 0.05)
        v = torch.empty(rows, 200, device="cuda")
        i = torch.empty(rows, 200, dtype=torch.int64, device="cuda")
        t = graph_us(lambda: pkg.mask_topk_rows_sparse(stream.cuda_stream, x.data_ptr(), v.data_ptr(), i.data_ptr(), rows, 19248, 200, -1.0), 10)
        wv, _ = x.topk(200, dim=1)
        print("%5d rows, %-6s: %8.1f us  (values == torch.topk: %s)  AMOS_TOPK_LDS=%s" % (rows, kind, t, bool(torch.equal(v, wv)), os.environ.get("AMOS_TOPK_LDS", "1")), flush=True)
