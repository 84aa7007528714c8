#!/usr/bin/env python3
"""Does a replayed HIP graph run two independent branches side by side?  Two chains of small dependent kernels, recorded
once on one stream and once forked onto a second stream and joined; prints the replay time of both graphs."""
import time
import torch

dev = "cuda"
a = torch.randn(64, 64, device=dev) * 0.1
b = torch.randn(64, 64, device=dev) * 0.1
N = 200


def chain(x):
    for _ in range(N):
        x = x * 1.0001 + 0.1
    return x


def record(two_streams):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if two_streams:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                y2 = chain(b)
            y1 = chain(a)
            main.wait_stream(side)
        else:
            y1 = chain(a)
            y2 = chain(b)
        out = y1 + y2
    return g, out


for two in (False, True, False, True):
    torch.cuda.synchronize()
    g, out = record(two)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    print(f"two_streams={two}: {(time.perf_counter() - t0) / 20 * 1e6:8.1f} us per replay of 2 x {N} small kernels")
