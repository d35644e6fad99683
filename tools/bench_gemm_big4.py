#!/usr/bin/env python3
"""The hipBLASLt shape of a workgroup — 4 waves, each a 128 x 128 wave tile of a 256 x 256 x 64 macro-tile, accumulators in AGPRs — as an
instantiation of the generic DMA kernel (gemm_glds_kernel<Cfg<256, 256, 2, 2>, 2 stages>, CST_GEMM_FORCE_CFG=big4) next to the
persistent 8-wave kernel: is the lower LDS traffic per MFMA (0.5 vs 0.75 ds_read_b128) worth a new kernel?  
    CST_GEMM_EXPERIMENT=1 python tools/bench_gemm_big4.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("CST_GEMM_EXPERIMENT"), "run with CST_GEMM_EXPERIMENT=1"
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")


def t(fn, iters=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for (m, n, k) in [(47968, 768, 3072), (47968, 3072, 768), (47968, 2304, 768), (8192, 8192, 8192), (31760, 768, 3072), (31760, 768, 2304), (31760, 768, 768)]:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    row, ref = [], None
    for cfg in ("8p", "big4", "big4n", "big4r", "large"):
        os.environ["CST_GEMM_FORCE_CFG"] = cfg
        ms = t(lambda: K.gemm(a, w, c, m, n, k, a_kmajor=1, b_kmajor=1, lda=k, ldb=k, ldc=n, split_k=1))
        if ref is None:
            ref = c.clone()
        row.append("%s %.3f ms (%.0f TF/s)%s" % (cfg, ms, 2.0 * m * n * k / ms / 1e9, "" if torch.equal(ref, c) else " [bits differ]"))
    os.environ["CST_GEMM_FORCE_CFG"] = ""
    hb = t(lambda: torch.matmul(a, w.t(), out=c))
    print("%6d x %5d x %5d: %s | hipBLASLt %.3f ms (%.0f TF/s)" % (m, n, k, " | ".join(row), hb, 2.0 * m * n * k / hb / 1e9), flush=True)

print("with the step's epilogues (fc2 forward: bias + residual + dropout 0.1; dX: residual):")
for (m, n, k, epi) in [(31760, 768, 3072, "plain"), (31760, 768, 3072, "fc2"), (31760, 768, 3072, "resid"), (31760, 768, 2304, "resid"), (47968, 768, 3072, "fc2"), (47968, 768, 3072, "resid")]:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    kw = dict(resid=torch.randn(m, n, device="cuda").bfloat16(), ld_resid=n) if epi != "plain" else {}
    if epi == "fc2":
        kw.update(bias=torch.randn(n, device="cuda").bfloat16(), drop_p=0.1, drop_key=12345)
    row, ref = [], None
    for cfg in ("8p", "4w", "big4n"):
        os.environ["CST_GEMM_FORCE_CFG"] = cfg
        ms = t(lambda: K.gemm(a, w, c, m, n, k, a_kmajor=1, b_kmajor=1, lda=k, ldb=k, ldc=n, split_k=1, **kw))
        if ref is None:
            ref = c.clone()
        row.append("%s %.3f ms (%.0f TF/s)%s" % (cfg, ms, 2.0 * m * n * k / ms / 1e9, "" if torch.equal(ref, c) else " [bits differ]"))
    os.environ["CST_GEMM_FORCE_CFG"] = ""
    print("%6d x %5d x %5d %-5s: %s" % (m, n, k, epi, " | ".join(row)), flush=True)

print("weight gradients dW[n_out, k_in] = dY^T X over `tokens` rows (both operands mn-major), split-K as the dispatcher picks it for 256 x 256 tiles:")
for (n_out, k_in, tokens, split) in [(3072, 768, 31760, 7), (768, 3072, 31760, 7), (2304, 768, 31760, 9), (768, 768, 31760, 28), (512, 1536, 47999, 21)]:
    dy = (torch.rand(tokens, n_out, device="cuda") * 2 - 1).bfloat16()
    x = (torch.rand(tokens, k_in, device="cuda") * 2 - 1).bfloat16()
    dw = torch.empty(n_out, k_in, device="cuda", dtype=torch.bfloat16)
    row, ref = [], None
    for cfg in ("large", "big4r"):
        os.environ["CST_GEMM_FORCE_CFG"] = cfg
        ms = t(lambda: K.gemm(dy, x, dw, n_out, k_in, tokens, a_kmajor=0, b_kmajor=0, lda=n_out, ldb=k_in, ldc=k_in, split_k=split))
        if ref is None:
            ref = dw.clone()
        row.append("%s %.3f ms (%.0f TF/s)%s" % (cfg, ms, 2.0 * tokens * n_out * k_in / ms / 1e9, "" if torch.equal(ref, dw) else " [bits differ]"))
    os.environ["CST_GEMM_FORCE_CFG"] = ""
    hb = t(lambda: torch.matmul(dy.t(), x, out=dw))
    print("%5d x %5d x %6d split %2d: %s | hipBLASLt %.3f ms (%.0f TF/s)" % (n_out, k_in, tokens, split, " | ".join(row), hb, 2.0 * tokens * n_out * k_in / hb / 1e9), flush=True)
