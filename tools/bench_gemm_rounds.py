#!/usr/bin/env python3
"""How a persistent GEMM launch's time grows with the number of rounds of work items (fc1 shape, A panels L2-resident through the
batch-stride-0 trick): time = fixed + rounds x per-round.  Separates launch-level fixed cost from steady-state throughput."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
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


N, Kd = int(os.environ.get("N", 3072)), int(os.environ.get("K", 768))
A = (torch.randn(256, Kd, device="cuda") * 0.5).to(dt)
B = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
tiles_n = (N + 255) // 256
EPI = int(os.environ.get("EPI", 0))  # 1: bias + GELU + pre-activation output (the fc1 forward epilogue: two stored images per tile)
L = importlib.import_module("chimera-st_amd.lib")
bias = torch.zeros(N, device="cuda", dtype=dt)
for nb in (1, 2, 4, 8, 11, 16, 21, 22, 32, 43, 64, 86, 128, 188, 256, 376, 512):
    C = torch.empty(nb * 256, N, dtype=dt, device="cuda")
    if EPI:
        Z = torch.empty_like(C)
        C2, Z2 = C.view(-1, N), Z.view(-1, N)
        # aux_out has no batch stride of its own: run the epilogue variant as ONE problem of nb * 256 rows over a row-repeating A
        A2 = A.repeat(nb, 1)
        ms = timeit(lambda: K.gemm(A2, B, C2, 256 * nb, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, bias=bias, act=L.ACT_GELU,
                                   aux_out=Z2, ld_aux_out=N, split_k=1))
    else:
        ms = timeit(lambda: K.gemm(A, B, C, 256, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, batch0=nb, sa=(0, 0), sb=(0, 0), sc=(256 * N, 0), split_k=1))
    tiles = nb * tiles_n
    print("batches %4d tiles %5d rounds %6.2f : %.4f ms  %.0f TF/s  (%.2f us per round)" % (nb, tiles, tiles / 256.0, ms, 2.0 * 256 * nb * N * Kd / ms / 1e9, ms * 1e3 / max(tiles / 256.0, 1.0)))
