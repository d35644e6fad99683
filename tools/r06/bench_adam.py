#!/usr/bin/env python3
"""cst_adam_step over 166.8 M parameters (bf16 gradient / parameter, fp32 master + moments): launch time and GB/s (28 B per element)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
K = importlib.import_module("chimera-st_amd.kernels")
n = 166781440
g = torch.randn(n, device="cuda").to(torch.bfloat16)
p = torch.randn(n, device="cuda").to(torch.bfloat16)
ma, m, v = p.float(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
sc = torch.ones(1, device="cuda")
import inspect
print(inspect.signature(K.adam_step))
def run():
    K.adam_step(ma, m, v, g, p, 2e-4, 0.9, 0.98, 1e-8, 0.0, 3, sc)
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("adam %.3f ms  %.2f TB/s  checksum %r" % (ms, n * 28 / ms / 1e9, float(ma[:1000].sum())))
