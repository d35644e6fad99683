#!/usr/bin/env python3
"""Micro-benchmarks of the hot kernels at BASELINE shapes (B=32 x 30 s): prints achieved TFLOP/s / GB/s.
Usage (GPU box): python tools/bench_kernels.py [--dtype bf16|f32]"""
import argparse, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")


def timeit(fn, iters=10, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--B", type=int, default=32)
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda"
    B, T = a.B, 1499
    M = B * T
    print("device", torch.cuda.get_device_name(0), "dtype", a.dtype)
    for name, (N, Kd) in {"w2v fc1 768->3072": (3072, 768), "w2v fc2 3072->768": (768, 3072), "w2v qkv 768->768": (768, 768)}.items():
        x = torch.randn(M, Kd, device=dev).to(dt); w = (torch.randn(N, Kd, device=dev) * 0.05).to(dt)
        y = torch.empty(M, N, device=dev, dtype=dt); dy = torch.randn(M, N, device=dev).to(dt)
        dx = torch.empty(M, Kd, device=dev, dtype=dt); dw = torch.empty(N, Kd, device=dev, dtype=dt)
        fl = 2.0 * M * N * Kd
        t = timeit(lambda: K.gemm(x, w, y, M, N, Kd, a_kmajor=1, b_kmajor=1, lda=Kd, ldb=Kd, ldc=N, split_k=1))
        print("%-22s fwd  %8.3f ms %8.1f TF/s" % (name, t, fl / t / 1e9))
        t = timeit(lambda: K.gemm(dy, w, dx, M, Kd, N, a_kmajor=1, b_kmajor=0, lda=N, ldb=Kd, ldc=Kd, split_k=1))
        print("%-22s dX   %8.3f ms %8.1f TF/s" % (name, t, fl / t / 1e9))
        t = timeit(lambda: K.gemm(dy, x, dw, N, Kd, M, a_kmajor=0, b_kmajor=0, lda=N, ldb=Kd, ldc=Kd, split_k=-1))
        print("%-22s dW   %8.3f ms %8.1f TF/s" % (name, t, fl / t / 1e9))
        tt = timeit(lambda: torch.matmul(x, w.t()))
        print("%-22s torch.matmul fwd (hipBLASLt ref) %8.3f ms %8.1f TF/s" % (name, tt, fl / tt / 1e9))
    # conv layer 1: Cin=512,k=3,s=2, Lin=95999
    Lin, C = 95999, 512
    Lout = (Lin - 3) // 2 + 1
    x = torch.randn(B, Lin, C, device=dev).to(dt); w = (torch.randn(C, 3 * C, device=dev) * 0.03).to(dt)
    y = torch.empty(B, Lout, C, device=dev, dtype=dt)
    t = timeit(lambda: K.gemm(x, w, y, Lout, C, 3 * C, a_kmajor=1, b_kmajor=1, lda=2 * C, ldb=3 * C, ldc=C, batch0=B, sa=(Lin * C, 0), sc=(Lout * C, 0), act=L.ACT_GELU, split_k=1), iters=5)
    print("conv1 implicit gemm    fwd  %8.3f ms %8.1f TF/s" % (t, 2.0 * B * Lout * C * 3 * C / t / 1e9))
    del x, y
    # attention
    H, D = 12, 64
    q = torch.randn(B, T, H * D, device=dev).to(dt); k_ = torch.randn(B, T, H * D, device=dev).to(dt); v = torch.randn(B, T, H * D, device=dev).to(dt)
    fl = 4.0 * B * H * T * T * D
    t = timeit(lambda: K.attn_fwd(q, k_, v, H, D, None, False, 0.125))
    print("attn fwd T=1499        %8.3f ms %8.1f TF/s" % (t, fl / t / 1e9))
    o, lse = K.attn_fwd(q, k_, v, H, D, None, False, 0.125)
    do = torch.randn_like(o)
    t = timeit(lambda: K.attn_bwd(do, q, k_, v, o, lse, H, D, None, False, 0.125), iters=5)
    print("attn bwd T=1499        %8.3f ms %8.1f TF/s (5 gemm-equivalents)" % (t, 2.5 * fl / t / 1e9))
    # layernorm
    x = torch.randn(M, 768, device=dev).to(dt); g = torch.ones(768, device=dev, dtype=dt); b = torch.zeros(768, device=dev, dtype=dt)
    t = timeit(lambda: K.layernorm_fwd(x, None, g, b, 1e-5))
    print("layernorm fwd 768      %8.3f ms %8.1f GB/s" % (t, 2.0 * x.numel() * x.element_size() / t / 1e6))
    y, s, mean, rstd = K.layernorm_fwd(x, None, g, b, 1e-5)
    t = timeit(lambda: K.layernorm_bwd(x, x, g, mean, rstd))
    print("layernorm bwd 768      %8.3f ms %8.1f GB/s" % (t, 3.0 * x.numel() * x.element_size() / t / 1e6))
    # conv0
    S = 480000
    wav = torch.randn(B, S, device=dev) * 0.1
    w0 = (torch.randn(512, 10, device=dev) * 0.3).to(dt); g0 = torch.ones(512, device=dev, dtype=dt); b0 = torch.zeros(512, device=dev, dtype=dt)
    t = timeit(lambda: K.conv0_fwd(wav, w0, g0, b0, 10, 5), iters=3, warm=1)
    L0 = (S - 10) // 5 + 1
    print("conv0+GN+GELU fwd      %8.3f ms %8.1f GB/s (write-bound)" % (t, (B * L0 * 512 * w0.element_size() + 2 * B * S * 4) / t / 1e6))
    y0, mean, rstd, gram = K.conv0_fwd(wav, w0, g0, b0, 10, 5)
    t = timeit(lambda: K.conv0_bwd(y0, wav, w0, g0, b0, mean, rstd, gram, 10, 5), iters=3, warm=1)
    print("conv0+GN+GELU bwd      %8.3f ms %8.1f GB/s" % (t, (B * L0 * 512 * w0.element_size() + B * S * 4) / t / 1e6))


if __name__ == "__main__":
    main()
