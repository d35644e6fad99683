#!/usr/bin/env python3
"""What the epilogues of the N = 3072 / K = 768 launches of a wav2vec2 layer are made of (bf16, in-step row count): plain store, bias,
a second output, the GELU polynomial, an extra operand (residual add = the price of multiplying by a STORED act'), act' evaluated."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")
dt = torch.bfloat16


def timeit(fn, iters=30, warm=5):
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


M = int(sys.argv[1]) if len(sys.argv) > 1 else 31760
D, F = 768, 3072
x = torch.randn(M, D, device="cuda").to(dt); w1 = (torch.randn(F, D, device="cuda") * 0.03).to(dt); b1 = (torch.randn(F, device="cuda") * 0.1).to(dt)
h = torch.empty(M, F, device="cuda", dtype=dt); z = torch.randn(M, F, device="cuda").to(dt); r = torch.randn(M, F, device="cuda").to(dt)
o = torch.empty(M, F, device="cuda", dtype=dt)
G, R = L.ACT_GELU, L.ACT_RELU
g = lambda **kw: (lambda: K.gemm(x, w1, h, M, F, D, a_kmajor=1, b_kmajor=1, lda=D, ldb=D, ldc=F, **kw))
rows = [("plain", g()), ("bias", g(bias=b1)), ("bias+aux_out", g(bias=b1, aux_out=o, ld_aux_out=F)), ("bias+gelu", g(bias=b1, act=G)),
        ("bias+gelu+aux_out", g(bias=b1, act=G, aux_out=o, ld_aux_out=F)), ("bias+relu+aux_out", g(bias=b1, act=R, aux_out=o, ld_aux_out=F)),
        ("resid", g(resid=r, ld_resid=F)), ("dact gelu", g(dact=G, aux_in=z, ld_aux_in=F)), ("dact relu", g(dact=R, aux_in=z, ld_aux_in=F))]
fl = 2.0 * M * D * F
for rep in range(2):
    print(" | ".join("%s %.1f us %.0f TF" % (n, t, fl / t / 1e6) for n, t in ((n, timeit(f)) for n, f in rows)))
