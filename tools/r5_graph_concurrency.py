#!/usr/bin/env python3
"""GPU: do two chains of small kernels overlap (a) as two parallel branches of ONE HIP graph, (b) as two graphs replayed on two streams, against (c) one
chain after the other?  Each chain = 40 launches of a small convolution (this project's GEMM on a 35 x 35 x 256 tensor: ~20 work-groups, ~13 us)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
import torch  # noqa: E402

cl = torch.channels_last
dev = "cuda"
N = 40


def make():
    x = torch.randn(1, 256, 35, 35, device=dev).contiguous(memory_format=cl)
    w = torch.randn(256, 256, 1, 1, device=dev).contiguous(memory_format=cl) / 16
    b = torch.zeros(256, device=dev)
    y = torch.empty_like(x)
    return x, w, b, y


def chain(t, stream):
    x, w, b, y = t
    for _ in range(N // 2):
        pkg.mask_conv(stream.cuda_stream, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), 1, 35, 35, 256, 256, 1, 1, 1, 0, True)
        pkg.mask_conv(stream.cuda_stream, y.data_ptr(), w.data_ptr(), b.data_ptr(), None, x.data_ptr(), 1, 35, 35, 256, 256, 1, 1, 1, 0, True)


A, B = make(), make()
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
for t, s in ((A, s0), (B, s1)):
    chain(t, s)
torch.cuda.synchronize()


def capture(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        fn()
    return g


def both_branches():
    s1.wait_stream(s0)
    chain(A, s0)
    chain(B, s1)
    s0.wait_stream(s1)


def serial():
    chain(A, s0)
    chain(B, s0)


g_one = capture(lambda: chain(A, s0), s0)
g_two = capture(lambda: chain(B, s1), s1)
g_branches = capture(both_branches, s0)
g_serial = capture(serial, s0)
torch.cuda.synchronize()


def time_us(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e6


def two_graphs():
    with torch.cuda.stream(s0):
        g_one.replay()
    with torch.cuda.stream(s1):
        g_two.replay()


print("one chain (graph)                         : %7.1f us" % time_us(lambda: g_one.replay()))
print("two chains, one after the other (graph)   : %7.1f us" % time_us(lambda: g_serial.replay()))
print("two chains, two branches of one graph     : %7.1f us" % time_us(lambda: g_branches.replay()))
print("two chains, two graphs on two streams     : %7.1f us" % time_us(two_graphs))
