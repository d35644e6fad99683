#!/usr/bin/env python3
"""GEMM shape sweep (bf16, all four operand layouts): separates the per-workgroup fixed cost from the per-K-tile cost.
Usage (GPU box): python tools/bench_gemm_sweep.py [layouts e.g. kk,km,mm]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
dt = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


lay = (sys.argv[1] if len(sys.argv) > 1 else "kk,km,mm").split(",")
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (47968, 3072, 768), (47968, 3072, 1536), (47968, 3072, 3072), (47968, 768, 3072),
          (47968, 768, 768), (47968, 2304, 768), (3072, 768, 47968), (768, 768, 47968)]
for (M, N, Kd) in shapes:
    for l in lay:
        ak, bk = l[0] == "k", l[1] == "k"
        A = (torch.rand(M, Kd, device="cuda") * 2 - 1).to(dt) if ak else (torch.rand(Kd, M, device="cuda") * 2 - 1).to(dt)
        B = (torch.rand(N, Kd, device="cuda") * 2 - 1).to(dt) if bk else (torch.rand(Kd, N, device="cuda") * 2 - 1).to(dt)
        C = torch.empty(M, N, device="cuda", dtype=dt)
        t = timeit(lambda: K.gemm(A, B, C, M, N, Kd, a_kmajor=int(ak), b_kmajor=int(bk), lda=A.shape[1], ldb=B.shape[1], ldc=N))
        ref = ""
        if l == "kk":
            tt = timeit(lambda: torch.matmul(A, B.t()))
            ref = "   hipBLASLt %7.3f ms %7.1f TF/s" % (tt, 2.0 * M * N * Kd / tt / 1e9)
        print("%6d x %5d x %6d  %s  %8.3f ms %7.1f TF/s%s" % (M, N, Kd, l, t, 2.0 * M * N * Kd / t / 1e9, ref), flush=True)
        del A, B, C
