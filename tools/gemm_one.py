#!/usr/bin/env python3
"""One bf16 GEMM launch set for counter collection:
   python tools/gemm_one.py <layout kk|km|mk|mm> M N K [iters] [fc1]
`fc1` adds the wav2vec2 fc1-forward epilogue (bias + GELU + pre-activation side output)."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")
l, M, N, Kd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
fc1 = len(sys.argv) > 6 and sys.argv[6] == "fc1"
dt = torch.bfloat16
ak, bk = l[0] == "k", l[1] == "k"
A = (torch.rand(M, Kd, device="cuda") * 2 - 1).to(dt) if ak else (torch.rand(Kd, M, device="cuda") * 2 - 1).to(dt)
B = (torch.rand(N, Kd, device="cuda") * 2 - 1).to(dt) if bk else (torch.rand(Kd, N, device="cuda") * 2 - 1).to(dt)
C = torch.empty(M, N, device="cuda", dtype=dt)
kw = {}
if fc1:
    kw = dict(bias=torch.zeros(N, device="cuda", dtype=dt), act=L.ACT_GELU, aux_out=torch.empty(M, N, device="cuda", dtype=dt), ld_aux_out=N, split_k=1)
for _ in range(iters):
    K.gemm(A, B, C, M, N, Kd, a_kmajor=int(ak), b_kmajor=int(bk), lda=A.shape[1], ldb=B.shape[1], ldc=N, **kw)
torch.cuda.synchronize()
