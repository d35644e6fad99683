import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
M, N, Kd = 47968, 3072, 768
dt = torch.bfloat16
x = torch.randn(M, Kd, device="cuda").to(dt); w = (torch.randn(N, Kd, device="cuda") * 0.05).to(dt)
y = torch.empty(M, N, device="cuda", dtype=dt); dy = torch.randn(M, N, device="cuda").to(dt); dx = torch.empty_like(x); dw = torch.empty_like(w)
for _ in range(3):
    K.gemm(x, w, y, M, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, split_k=1)
    K.gemm(dy, w, dx, M, Kd, N, a_kmajor=1, b_kmajor=0, lda=N, ldb=Kd, ldc=Kd, split_k=1)
    K.gemm(dy, x, dw, N, Kd, M, a_kmajor=0, b_kmajor=0, lda=N, ldb=Kd, ldc=Kd, split_k=-1)
torch.cuda.synchronize()
