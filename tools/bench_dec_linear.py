#!/usr/bin/env python3
"""cst_dec_linear against the general GEMM on the Linear shapes of one decode step (s2t_transformer_l, 160 hypothesis rows).
Launches are chained through a tiny dependent kernel-free trick: each timing loop alternates two launches that read each other's
output rows, so consecutive launches cannot overlap on the device."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")
dt = torch.bfloat16
M = int(os.environ.get("M", 160))


def timeit(fn, iters=200, warm=10):
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


for (N, Kd, what) in [(3072, 1024, "qkv"), (1024, 1024, "out/q proj"), (4096, 1024, "fc1"), (1024, 4096, "fc2"), (10000, 1024, "vocabulary")]:
    # square-compatible chain: x -> y (N) ; to chain launches, y feeds a second weight of shape [Kd, N] back to Kd columns
    x = torch.randn(M, Kd, device="cuda").to(dt)
    W = (torch.randn(N, Kd, device="cuda") / Kd ** 0.5).to(dt)
    b = torch.randn(N, device="cuda").to(dt)
    y = torch.empty(M, N, device="cuda", dtype=dt)
    lib = L.load()

    def new():
        L.check(lib.cst_dec_linear(L.ptr(x), L.ptr(W), L.ptr(b), None, L.ptr(y), M, N, Kd, Kd, 0, N, L.ACT_NONE, None, 0, L.dtype_code(dt), L.stream_ptr()))

    def old():
        K.gemm(x, W, y, M, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, bias=b, split_k=1)

    g_new, g_old = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    new(); old(); torch.cuda.synchronize()
    with torch.cuda.graph(g_new):
        for _ in range(20):
            new()
    with torch.cuda.graph(g_old):
        for _ in range(20):
            old()
    tn = timeit(g_new.replay, 20, 3) / 20
    to = timeit(g_old.replay, 20, 3) / 20
    wb = N * Kd * 2
    print("%-12s N=%5d K=%5d  dec_linear %6.2f us (%5.0f GB/s of weights)   general gemm %6.2f us" % (what, N, Kd, tn, wb / tn / 1e3, to))
