#!/usr/bin/env python3
"""Which Tensile kernels hipBLASLt picks for the forward FFN shapes (a measurement of what wins on this chip, not a dependency).
Run under `rocprofv3 --kernel-trace --stats --output-format csv -- python3 tools/probes/hipblaslt_names.py`; also prints its own timings."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
K = importlib.import_module("chimera-st_amd.kernels")

def t(fn, iters=20):
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

for (m, n, k) in ((47968, 3072, 768), (47968, 768, 3072), (47968, 2304, 768), (47968, 768, 768), (8192, 8192, 8192), (33500, 3072, 768), (33500, 768, 3072)):
    a = (torch.rand(m, k, device="cuda") * 2 - 1).bfloat16()
    w = (torch.rand(n, k, device="cuda") * 2 - 1).bfloat16()
    ms = t(lambda: torch.matmul(a, w.t()))
    c = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    ours = t(lambda: K.gemm(a, w, c, m, n, k, a_kmajor=1, b_kmajor=1, lda=k, ldb=k, ldc=n, split_k=1))
    print("%6d x %5d x %5d (A k-major, W [N,K]): torch.matmul/hipBLASLt %.3f ms %.0f TF/s | this library %.3f ms %.0f TF/s | ratio %.2f" %
          (m, n, k, ms, 2.0 * m * n * k / ms / 1e9, ours, 2.0 * m * n * k / ours / 1e9, ms / ours), flush=True)
