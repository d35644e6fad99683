#!/usr/bin/env python3
"""Epilogue cost of the w2v FFN GEMMs (bf16, M=47968): plain vs bias+GELU+aux_out (fc1 fwd) vs act'(aux_in) (dX through GELU)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")
dt = torch.bfloat16


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
    return s.elapsed_time(e) / iters


M, D, F = 47968, 768, 3072
x = torch.randn(M, D, device="cuda").to(dt); w1 = (torch.randn(F, D, device="cuda") * 0.03).to(dt); b1 = torch.zeros(F, device="cuda", dtype=dt)
h = torch.empty(M, F, device="cuda", dtype=dt); z = torch.empty(M, F, device="cuda", dtype=dt)
w2 = (torch.randn(D, F, device="cuda") * 0.03).to(dt); dy = torch.randn(M, D, device="cuda").to(dt); dz = torch.empty(M, F, device="cuda", dtype=dt)
y = torch.empty(M, D, device="cuda", dtype=dt)
fl = 2.0 * M * D * F
for act in ("gelu", "relu"):
    A = L.ACT_GELU if act == "gelu" else L.ACT_RELU
    t0 = timeit(lambda: K.gemm(x, w1, h, M, F, D, a_kmajor=1, b_kmajor=1, lda=D, ldb=D, ldc=F))
    t1 = timeit(lambda: K.gemm(x, w1, h, M, F, D, a_kmajor=1, b_kmajor=1, lda=D, ldb=D, ldc=F, bias=b1, act=A, aux_out=z, ld_aux_out=F))
    t2 = timeit(lambda: K.gemm(dy, w2, dz, M, F, D, a_kmajor=1, b_kmajor=0, lda=D, ldb=F, ldc=F))
    t3 = timeit(lambda: K.gemm(dy, w2, dz, M, F, D, a_kmajor=1, b_kmajor=0, lda=D, ldb=F, ldc=F, dact=A, aux_in=z, ld_aux_in=F))
    t4 = timeit(lambda: K.gemm(h, w2, y, M, D, F, a_kmajor=1, b_kmajor=1, lda=F, ldb=F, ldc=D, bias=b1[:D], resid=x, ld_resid=D))
    print("%s: fc1 plain %.3f ms %.0f TF | +bias+act+aux_out %.3f ms %.0f TF | dX plain %.3f ms %.0f TF | dX*act'(aux_in) %.3f ms %.0f TF | fc2+bias+resid %.3f ms %.0f TF"
          % (act, t0, fl / t0 / 1e9, t1, fl / t1 / 1e9, t2, fl / t2 / 1e9, t3, fl / t3 / 1e9, t4, fl / t4 / 1e9))
