#!/usr/bin/env python3
"""Does the implicit-GEMM operand layout of the conv stack cost the persistent GEMM anything?  Layer 1 of the wav2vec2 feature
extractor at the bench's size: per utterance A = [47 999 frames x (3 taps x 512 channels)] read as OVERLAPPING rows of the layer-0
output (row pitch 2 x 512 elements = 2048 B, a power of two) against the same product on a materialised copy (row pitch 1536
elements), plain store and with the GELU + pre-activation epilogue.  The operand is 3.1 GB per launch: HBM-cold in a timing loop too.
usage (GPU box): python tools/bench_conv_layout.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
K = importlib.import_module("chimera-st_amd.kernels"); L = importlib.import_module("chimera-st_amd.lib")
dt = torch.bfloat16
B, Lin, C, k, s = 32, 95999, 512, 3, 2
Lout = (Lin - k) // s + 1
x = (torch.randn(B, Lin, C, device="cuda") * 0.5).to(dt)
w = (torch.randn(C, k * C, device="cuda") * (k * C) ** -0.5).to(dt)
bias = torch.zeros(C, device="cuda", dtype=dt)
y = torch.empty(B, Lout, C, device="cuda", dtype=dt); z = torch.empty_like(y)
xm = torch.as_strided(x, (B, Lout, k * C), (Lin * C, s * C, 1)).contiguous()   # materialised [B, Lout, 1536]

def timeit(fn, iters=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

fl = 2.0 * B * Lout * C * k * C
for name, A, lda, sa in (("implicit (pitch 1024 el = 2048 B)", x, s * C, Lin * C), ("materialised (pitch 1536 el)", xm, k * C, Lout * k * C)):
    for epi, kw in (("plain", {}), ("bias + GELU + pre-activation out", dict(bias=bias, act=L.ACT_GELU, aux_out=z, ld_aux_out=C))):
        f = lambda: K.gemm(A, w, y, Lout, C, k * C, a_kmajor=1, b_kmajor=1, lda=lda, ldb=k * C, ldc=C, batch0=B, sa=(sa, 0), sb=(0, 0),
                           sc=(Lout * C, 0), split_k=1, **kw)
        ms = timeit(f)
        print("%-36s %-34s %7.3f ms  %6.0f TF/s" % (name, epi, ms, fl / ms / 1e9))
# the same FLOPs as ONE tall GEMM [B * Lout, 1536] x [512, 1536]^T on the materialised operand (no batching, N = 512)
f = lambda: K.gemm(xm, w, y, B * Lout, C, k * C, a_kmajor=1, b_kmajor=1, lda=k * C, ldb=k * C, ldc=C, split_k=1)
ms = timeit(f)
print("%-36s %-34s %7.3f ms  %6.0f TF/s" % ("materialised, one tall problem", "plain", ms, fl / ms / 1e9))
o = torch.empty(B * Lout, C, device="cuda", dtype=dt)
ms = timeit(lambda: torch.mm(xm.view(B * Lout, k * C), w.t(), out=o))
print("%-36s %-34s %7.3f ms  %6.0f TF/s" % ("vendor (torch.mm), materialised", "plain", ms, fl / ms / 1e9))
