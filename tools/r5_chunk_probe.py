#!/usr/bin/env python3
"""GPU: does running the memory-bound first stages of the backbone over CHUNKS of the 64 frames (so that a layer's output is still in the
256 MB Infinity Cache when the next layer reads it) beat one launch per layer over all 64?  Stem + layer1 (+ layer2) of ResNet50Trunk,
folded weights, eager launches on one stream, milliseconds per pass.   python tools/r5_chunk_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import importlib  # noqa: E402

import torch  # noqa: E402

net_mod = importlib.import_module("amos_slam_amd.mask.net")
cl = torch.channels_last
torch.manual_seed(0)
torch.backends.cudnn.benchmark = True
trunk = net_mod.ResNet50Trunk().cuda().eval()
trunk.fold_batch_norms()
trunk.to(memory_format=cl)
net_mod.prepare_winograd_weights(trunk)
net_mod.prepare_stem_weight(trunk)
B = 64
x = torch.randn(B, 3, 550, 550, device="cuda")


def stem(xc):
    from amos_slam_amd import mask_stem
    b = xc.shape[0]
    y = torch.empty((b, 64, 138, 138), dtype=torch.float32, device="cuda", memory_format=cl)
    mask_stem(torch.cuda.current_stream().cuda_stream, xc.data_ptr(), xc.stride(), net_mod._stem_weight(trunk.conv1).data_ptr(), trunk.conv1.bias.data_ptr(), y.data_ptr(), b, 550, 550)
    return y


def run(chunk, upto):
    outs = []
    for i in range(0, B, chunk):
        y = stem(x[i:i + chunk])
        for li in range(upto):
            y = trunk.layers[li](y)
        outs.append(y)
    return outs[0] if len(outs) == 1 else torch.cat(outs, 0)


def time_ms(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


with torch.no_grad():
    for upto in (1, 2):
        ref = run(64, upto)
        for chunk in (64, 32, 16, 8, 4):
            out = run(chunk, upto)
            same = torch.equal(out, ref)
            t = time_ms(lambda: run(chunk, upto))
            print("stem + layer1%s, chunks of %2d frames: %7.3f ms (same bits as one launch: %s)" % (" + layer2" if upto == 2 else "", chunk, t, same), flush=True)
