#!/usr/bin/env python3
"""Split-K sweep of the weight-gradient GEMMs of the small layers (dW = dY^T X: both operands mn-major, M x N = one weight matrix, K =
the token rows of the batch): time per launch INCLUDING the split-K reduce, for forced split counts next to the library's own choice.
    python tools/bench_gemm_splitk.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")

SHAPES = [(1536, 512, 7901), (512, 2048, 7901), (2048, 512, 7901), (512, 512, 7901), (512, 512, 12000), (512, 512, 4064), (1536, 512, 4064),
          (512, 2048, 4064), (768, 768, 31760), (768, 512, 47968), (2304, 768, 31760), (3072, 768, 31760)]


def t(fn, iters=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for (m, n, k) in SHAPES:
    a = (torch.rand(k, m, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(k, n, device="cuda") * 2 - 1).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    row = []
    for sp in (-1, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
        if sp > 1 and k // 64 // sp < 4:
            continue
        ms = t(lambda: K.gemm(a, b, c, m, n, k, a_kmajor=0, b_kmajor=0, lda=m, ldb=n, ldc=n, split_k=sp))
        row.append("%s %.1f us (%.0f TF/s)" % ("auto" if sp < 0 else "s=%d" % sp, ms * 1e3, 2.0 * m * n * k / ms / 1e9))
    print("dW %5d x %5d x %6d: %s" % (m, n, k, " | ".join(row)), flush=True)
