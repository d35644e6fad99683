#!/usr/bin/env python3
"""Stress of the claimed-item walk of the persistent GEMM: random multi-round shapes launched back to back on three streams
(more than one lap of the 1024-slot counter ring), every result compared with an fp32 matmul of the same bf16 operands."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
dt = torch.bfloat16
g = torch.Generator().manual_seed(7)
streams = [torch.cuda.Stream() for _ in range(3)]
shapes = []
for _ in range(12):
    M = int(torch.randint(17, 40, (1,), generator=g)) * 256 - int(torch.randint(0, 32, (1,), generator=g)) * 8
    N = int(torch.randint(12, 24, (1,), generator=g)) * 256 - int(torch.randint(0, 32, (1,), generator=g)) * 8
    Kd = int(torch.randint(1, 9, (1,), generator=g)) * 64 + int(torch.randint(0, 8, (1,), generator=g)) * 8
    shapes.append((M, N, Kd))
ops = []
for i, (M, N, Kd) in enumerate(shapes):
    A = (torch.randn(M, Kd, generator=g) * 0.5).to(dt).cuda(); B = (torch.randn(N, Kd, generator=g) * 0.5).to(dt).cuda()
    ops.append((A, B, (A.float() @ B.float().t()), [torch.empty(M, N, dtype=dt, device="cuda") for _ in range(3)]))
torch.cuda.synchronize()
n = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for i, (A, B, ref, outs) in enumerate(ops):
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                M, Kd = A.shape; N = B.shape[0]
                K.gemm(A, B, outs[si], M, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N)
                n += 1
torch.cuda.synchronize()
bad = 0
for A, B, ref, outs in ops:
    tol = 0.02 * float(ref.abs().max())
    for o in outs:
        if not torch.isfinite(o.float()).all() or float((o.float() - ref).abs().max()) > tol or not torch.equal(o, outs[0]):
            bad += 1
print("%d launches on 3 streams, %d shapes, mismatching outputs: %d" % (n, len(ops), bad))
sys.exit(1 if bad else 0)
