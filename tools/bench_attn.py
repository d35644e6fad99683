#!/usr/bin/env python3
"""Attention kernels at the wav2vec2 encoder shape (B=32, T=1499, 12 heads x 64), with and without probability dropout,
interleaved rounds in one process.  Usage (GPU box): [ATT_B=.. ATT_T=.. ATT_SCALE=..] python tools/bench_attn.py [rounds]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("chimera-st_amd.kernels")
L = importlib.import_module("chimera-st_amd.lib")
B, T, H, D = int(os.environ.get("ATT_B", 32)), int(os.environ.get("ATT_T", 1499)), 12, 64
SC = float(os.environ.get("ATT_SCALE", 1.0))  # data scale: 0 = all-zero operands (clock / power probe)
dt = torch.bfloat16
q = (SC * torch.randn(B, T, H * D, device="cuda")).to(dt); k = (SC * torch.randn(B, T, H * D, device="cuda")).to(dt); v = (SC * torch.randn(B, T, H * D, device="cuda")).to(dt)
do = (SC * torch.randn(B, T, H * D, device="cuda")).to(dt)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
fl = 4.0 * B * H * T * T * D


def t(fn, iters=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


res = {}
for r in range(rounds):
    for p in (0.0, 0.1):
        o, lse = K.attn_fwd(q, k, v, H, D, None, False, 0.125, "bt", "bt", p, 99)
        res.setdefault(("fwd", p), []).append(t(lambda: K.attn_fwd(q, k, v, H, D, None, False, 0.125, "bt", "bt", p, 99)))
        res.setdefault(("bwd", p), []).append(t(lambda: K.attn_bwd(do, q, k, v, o, lse, H, D, None, False, 0.125, "bt", "bt", p, 99)))
for (what, p), v_ in sorted(res.items()):
    m = sorted(v_)[len(v_) // 2]
    print("%s dropout %.1f: median %.3f ms  min %.3f ms  (%.0f TF/s at %s gemm-equivalents)" % (what, p, m, min(v_), (fl if what == "fwd" else 2.5 * fl) / m / 1e9, "2" if what == "fwd" else "5"))
L.prof_enable(True)
