#!/usr/bin/env python3
"""gemm4w.hip (four waves, 256 x 192 tiles, one tile per workgroup) against the persistent 8-wave kernel over M, with the residual
epilogue, at K = 3072 / 2304 / 1536 and N = 768: where whole rounds of 256 CUs pay (the dispatch rule of cst_gemm4w_supported).
Timing loop = operands warm in the Infinity Cache: the in-step comparison is tools/gemm_shapes_in_step.py with CST_GEMM_NO_4W=1.
    CST_GEMM_EXPERIMENT=1 CST_GEMM_4W_MAXTILES=100000 python tools/bench_gemm4w_sweep.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
def t(fn, iters=30):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for k in (3072, 2304, 1536):
  for m in (16384, 20000, 24000, 28000, 31760, 36000, 40000, 44000, 47968, 56000, 65536):
    n = 768
    a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16(); w = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16); r = torch.randn(m, n, device="cuda").bfloat16()
    row = []
    for cfg in ("8p", "4w"):
        os.environ["CST_GEMM_FORCE_CFG"] = cfg
        try:
            ms = t(lambda: K.gemm(a, w, c, m, n, k, a_kmajor=1, b_kmajor=1, lda=k, ldb=k, ldc=n, split_k=1, resid=r, ld_resid=n))
            row.append("%s %.3f" % (cfg, ms))
        except RuntimeError:
            row.append("%s n/a" % cfg)
    os.environ["CST_GEMM_FORCE_CFG"] = ""
    print("K=%d M=%6d tiles4w=%4d rounds %.2f : %s" % (k, m, -(-m // 256) * 4, -(-m // 256) * 4 / 256.0, " | ".join(row)), flush=True)
