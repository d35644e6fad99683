#!/usr/bin/env python3
"""Weight-gradient GEMMs of the 512-wide layers (both operands mn-major) through every register-staged tile configuration x split count
(CST_GEMM_EXPERIMENT=1; time includes the split-K reduce launch).  CAUTION: the timing loop re-launches one problem back to back, so its
operands (<= 12 MB) stay in L2 / Infinity Cache; inside an update they come from HBM.  What this tool ranks first for 512 x 512 outputs
(64 x 64 tiles, few K slices) was slower in the step (tools/gemm_shapes_in_step.py is the judge for small problems).
    CST_GEMM_EXPERIMENT=1 python tools/bench_gemm_cfg_dw.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("CST_GEMM_EXPERIMENT"), "run with CST_GEMM_EXPERIMENT=1"
K = importlib.import_module("chimera-st_amd.kernels")
SHAPES = [(512, 512, 2048), (512, 512, 4064), (512, 512, 7901), (512, 512, 12000), (1536, 512, 4064), (512, 2048, 4064), (1536, 512, 7901), (512, 2048, 7901), (512, 10000, 4064)]


def t(fn, iters=40):
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


for (m, n, k) in SHAPES:
    a = (torch.rand(k, m, device="cuda") * 2 - 1).bfloat16()
    b = (torch.rand(k, n, device="cuda") * 2 - 1).bfloat16()
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    os.environ["CST_GEMM_FORCE_CFG"] = ""
    auto = t(lambda: K.gemm(a, b, c, m, n, k, a_kmajor=0, b_kmajor=0, lda=m, ldb=n, ldc=n, split_k=-1))
    ref = (a.float().t() @ b.float())
    best = []
    for cfg in ("small", "narrowm", "narrown", "skinny", "large"):
        os.environ["CST_GEMM_FORCE_CFG"] = cfg
        res = []
        for sp in (1, 2, 3, 4, 6, 8, 12, 16, 24):
            if k // 64 // sp < 4:
                continue
            try:
                ms = t(lambda: K.gemm(a, b, c, m, n, k, a_kmajor=0, b_kmajor=0, lda=m, ldb=n, ldc=n, split_k=sp))
            except RuntimeError:
                continue
            err = float((c.float() - ref).abs().max() / ref.abs().max())
            res.append((ms, sp, err))
        if res:
            ms, sp, err = min(res)
            best.append("%s s=%d %.1f us%s" % (cfg, sp, ms * 1e3, "" if err < 2e-2 else " [WRONG %.1e]" % err))
    os.environ["CST_GEMM_FORCE_CFG"] = ""
    print("dW %5d x %5d x %6d: auto %.1f us | %s" % (m, n, k, auto * 1e3, " | ".join(best)), flush=True)
