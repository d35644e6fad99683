#!/usr/bin/env python3
"""Host-side cost per launch of the Python -> ctypes -> C ABI path (no device sync inside the timed loops)."""
import importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")
CF = importlib.import_module("chimera-st_amd.functional")
dt = torch.bfloat16
x = torch.randn(256, 512, device="cuda").to(dt); w = torch.randn(512, 512, device="cuda").to(dt); b = torch.zeros(512, device="cuda", dtype=dt)
y = torch.empty(256, 512, device="cuda", dtype=dt)
def t(fn, n=2000):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
print("K.gemm (desc build + ctypes call)        host %.1f us/call, incl. device %.1f us" % t(lambda: K.gemm(x, w, y, 256, 512, 512, a_kmajor=1, b_kmajor=1, lda=512, ldb=512, ldc=512, bias=b)))
print("torch.empty                              host %.1f us/call" % t(lambda: torch.empty(256, 512, device="cuda", dtype=dt))[0])
xr = x.clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
print("CF.linear forward (autograd Function)    host %.1f us/call, incl. device %.1f us" % t(lambda: CF.linear(xr, wr, b)))
def fb():
    o = CF.linear(xr, wr, b); o.backward(o)
print("CF.linear fwd+bwd                        host %.1f us/call, incl. device %.1f us" % t(fb, 500))
g = torch.ones(512, device="cuda", dtype=dt); be = torch.zeros(512, device="cuda", dtype=dt)
print("CF.layer_norm forward                    host %.1f us/call, incl. device %.1f us" % t(lambda: CF.layer_norm(xr, g, be)))
print("torch.matmul (ATen baseline)             host %.1f us/call, incl. device %.1f us" % t(lambda: torch.matmul(x, w.t())))
if len(sys.argv) > 1:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): fb()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
