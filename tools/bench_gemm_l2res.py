import importlib, os, sys, torch
sys.path.insert(0, "/root/repo")
K = importlib.import_module("chimera-st_amd.kernels"); L = importlib.import_module("chimera-st_amd.lib")
dt = torch.bfloat16
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
mode = os.environ.get("CST_GEMM4W", "1")
for (N, Kd, nb) in ((3072, 768, 188), (768, 3072, 188), (3072, 768, 376)):
    A = (torch.randn(256, Kd, device="cuda") * 0.5).to(dt); B = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
    C = torch.empty(nb * 256, N, dtype=dt, device="cuda")
    f = lambda: K.gemm(A, B, C, 256, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, batch0=nb, sa=(0, 0), sb=(0, 0), sc=(256 * N, 0), split_k=1)
    ms = timeit(f)
    print("mode %s L2-resident A: N=%d K=%d batches=%d: %.3f ms %.0f TF/s" % (mode, N, Kd, nb, ms, 2.0 * 256 * nb * N * Kd / ms / 1e9))
    # and with the output also confined (every batch writes the same C tile rows): no HBM write stream
    f2 = lambda: K.gemm(A, B, C, 256, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, batch0=nb, sa=(0, 0), sb=(0, 0), sc=(0, 0), split_k=1)
    ms = timeit(f2)
    print("mode %s L2-resident A and C: N=%d K=%d batches=%d: %.3f ms %.0f TF/s" % (mode, N, Kd, nb, ms, 2.0 * 256 * nb * N * Kd / ms / 1e9))
