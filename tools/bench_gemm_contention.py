#!/usr/bin/env python3
"""The persistent GEMM under CU contention — what an RCCL all-reduce overlapped with backward does to it.  A long 32-tile GEMM
(32 workgroups, one item each) holds 32 CUs on a side stream while the fc1-forward launch (2 256 items) is timed on the main
stream.  With the static stride the 32 workgroups that cannot start run their whole share after everybody else (launch time
about doubles); with the claimed items (default) they take what is left.  Run twice: default and CST_GEMM8P_STATIC=1."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
dt = torch.bfloat16
M, D, F = 47968, 768, 3072
x = torch.randn(M, D, device="cuda").to(dt); w = (torch.randn(F, D, device="cuda") * 0.03).to(dt); h = torch.empty(M, F, device="cuda", dtype=dt)
hm, hn, hk = 1024, 2048, int(os.environ.get("HOG_K", "131072"))
ha = (torch.randn(hm, hk, device="cuda") * 0.01).to(dt); hb = (torch.randn(hn, hk, device="cuda") * 0.01).to(dt); hc = torch.empty(hm, hn, device="cuda", dtype=dt)
side = torch.cuda.Stream()


def fc1():
    K.gemm(x, w, h, M, F, D, a_kmajor=1, b_kmajor=1, lda=D, ldb=D, ldc=F)


def timed(n=10, hog=False):
    for _ in range(3):
        fc1()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if hog:
        with torch.cuda.stream(side):
            K.gemm(ha, hb, hc, hm, hn, hk, a_kmajor=1, b_kmajor=1, lda=hk, ldb=hk, ldc=hn, split_k=1)
        torch.cuda._sleep(200000)  # let the hog take its CUs first
    s.record()
    for _ in range(n):
        fc1()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


with torch.cuda.stream(side):
    hs, he = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K.gemm(ha, hb, hc, hm, hn, hk, a_kmajor=1, b_kmajor=1, lda=hk, ldb=hk, ldc=hn, split_k=1)
    hs.record(); K.gemm(ha, hb, hc, hm, hn, hk, a_kmajor=1, b_kmajor=1, lda=hk, ldb=hk, ldc=hn, split_k=1); he.record()
torch.cuda.synchronize()
print("mode %s | hog alone %.2f ms (32 CUs) | fc1 alone %.3f ms | fc1 with 32 CUs held %.3f ms"
      % ("static" if os.environ.get("CST_GEMM8P_STATIC") else "claimed", hs.elapsed_time(he), timed(), timed(hog=True)))
