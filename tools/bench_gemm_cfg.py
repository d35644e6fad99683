#!/usr/bin/env python3
"""k/k GEMMs of the 512-wide layers through every tile configuration of the library (CST_GEMM_EXPERIMENT=1 makes cst_gemm read
CST_GEMM_FORCE_CFG per call): which configuration should the dispatcher pick for mid-size M and short K?
    CST_GEMM_EXPERIMENT=1 python tools/bench_gemm_cfg.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
assert os.environ.get("CST_GEMM_EXPERIMENT"), "run with CST_GEMM_EXPERIMENT=1"
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")

SHAPES = [(7901, 512, 512, ""), (7901, 1536, 512, ""), (7901, 2048, 512, "act"), (7901, 512, 2048, ""), (7901, 512, 1536, ""), (12000, 512, 512, ""),
          (4064, 512, 512, ""), (4064, 2048, 512, "act"), (4064, 512, 2048, ""), (4064, 1536, 512, ""), (31760, 768, 768, ""), (2000, 512, 512, ""), (2000, 10000, 512, "")]


def t(fn, iters=50):
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


for (m, n, k, epi) in SHAPES:
    a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    b = torch.zeros(n, device="cuda", dtype=torch.bfloat16)
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    kw = dict(bias=b)
    if epi == "act":
        kw.update(act=L.ACT_RELU, aux_out=torch.empty(m, n, device="cuda", dtype=torch.bfloat16), ld_aux_out=n)
    row = []
    ref = None
    for cfg in ("", "skinny4", "skinny2", "small2", "small3", "8p"):
        os.environ["CST_GEMM_FORCE_CFG"] = cfg
        try:
            ms = t(lambda: K.gemm(a, w, c, m, n, k, a_kmajor=1, b_kmajor=1, lda=k, ldb=k, ldc=n, split_k=1, **kw))
        except RuntimeError as e:
            row.append("%s: n/a" % cfg)
            continue
        if ref is None:
            ref = c.clone()
        ok = torch.equal(ref, c)
        row.append("%s %.1f us (%.0f TF/s)%s" % (cfg or "auto", ms * 1e3, 2.0 * m * n * k / ms / 1e9, "" if ok else " [bits differ]"))
    os.environ["CST_GEMM_FORCE_CFG"] = ""
    print("%6d x %5d x %5d %-4s: %s" % (m, n, k, epi, " | ".join(row)), flush=True)
