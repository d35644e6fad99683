#!/usr/bin/env python3
"""Decode-step projections (160 hypothesis rows) with the K loop split over workgroups + the separate reduce launch, replayed as 20
dependent launches in a graph (tools/bench_dec_linear.py's method): is a split worth its extra node for the K = 4096 projection?"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
dt = torch.bfloat16
M = int(os.environ.get("M", 160))


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for (N, Kd, what) in [(1024, 4096, "fc2"), (1024, 1024, "out/q proj"), (4096, 1024, "fc1"), (10000, 1024, "vocabulary")]:
    x = torch.randn(M, Kd, device="cuda").to(dt)
    W = (torch.randn(N, Kd, device="cuda") / Kd ** 0.5).to(dt)
    b = torch.randn(N, device="cuda").to(dt)
    r = torch.randn(M, N, device="cuda").to(dt)
    y = torch.empty(M, N, device="cuda", dtype=dt)
    row = []
    for sp in (1, 2, 4, 8):
        if Kd // 64 // sp < 4:
            continue
        fn = lambda: K.gemm(x, W, y, M, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, bias=b, resid=r, ld_resid=N, split_k=sp)
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        row.append("split %d: %.2f us" % (sp, timeit(g.replay) / 20))
    print("%-12s N=%5d K=%5d  %s" % (what, N, Kd, " | ".join(row)), flush=True)
